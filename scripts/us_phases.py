"""Phase breakdown of k_update_small (kernels_kalman.hip) on the GPU box, -DMSCKF_ABLATE build only:
    make -C msckf_mono_amd/csrc ablate && MSCKF_HIP_LIB=msckf_mono_amd/lib_ab/libmsckf_hip_ablate.so python scripts/us_phases.py
One double (US_DTYPE=f32: float) filter at the cfg2 window (10 cameras, 50 tracks), call by call; shader-clock cycles per launch and phase."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from msckf_mono_amd import capi, scenario as sc  # noqa: E402
import helpers as H  # noqa: E402

N, F, nf = 10, 50, 40
tr = sc.Trajectory(2, 3, N, F, nf)
bt = capi.Batch(1, N, F, N, capi.F32 if os.environ.get("US_DTYPE") == "f32" else capi.F64)
bt.initialize(0, tr.cfg, tr.imu0)
out = (C.c_ulonglong * 16)()
for k in range(nf):
    if k == 20:
        bt.sync(); bt.L.msckf_hip_debug_us_cycles(out, 1)
    H.device_frame(bt, 0, tr, k, N)
bt.sync()
bt.L.msckf_hip_debug_us_cycles(out, 1)
v = np.array(out, dtype=np.float64); n = max(v[15], 1)
names = ["A Gram", "P->LDS + B loads", "B chol(Lam^)", "C PHt", "D S", "E loads", "E chol(S)+W", "G dx", "H downdate", "inject"]
print("k_update_small launches", int(v[15]), {a: int(c / n) for a, c in zip(names, v[:10])}, "total cycles/launch", int(v[:10].sum() / n))
