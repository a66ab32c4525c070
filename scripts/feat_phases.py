"""Phase breakdown of k_feature_pair (kernels_feature.hip) on the GPU box, -DMSCKF_ABLATE build only:
    make -C msckf_mono_amd/csrc ablate && MSCKF_HIP_LIB=msckf_mono_amd/lib_ab/libmsckf_hip_ablate.so python scripts/feat_phases.py
Prints shader-clock cycles per wavefront and phase (mean over all wavefronts of the launches) at the cfg3 window."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from msckf_mono_amd import capi, scenario as sc  # noqa: E402

N, F, B, nf = 30, 200, int(os.environ.get("PHASES_B", "64")), 40
trajs = [sc.Trajectory(3, b, N, F, nf) for b in range(B)]
bt = capi.Batch(B, N, F, N, capi.F32)
for b, tr in enumerate(trajs):
    bt.initialize(b, tr.cfg, tr.imu0)
bt.scenario_alloc(nf, 10)
for k in range(nf):
    for b, tr in enumerate(trajs):
        fr = tr.frames[k]
        bt.scenario_set(k, b, tr.imu_for_frame(k), fr["M"], fr["slots"], fr["obs"], 1 if fr["Nw"] == N else 0)
bt.scenario_commit()
bt.run_frames(0, 32); bt.sync()
bt.L.msckf_hip_debug_set(200, 0x100000)
out = (C.c_ulonglong * 8)()
bt.L.msckf_hip_debug_featp_cycles(out, 1)
bt.run_frames(32, nf); bt.sync()
bt.L.msckf_hip_debug_featp_cycles(out, 1)
v = np.array(out, dtype=np.float64)
n = max(v[7], 1)
names = ["pairing", "loads+motion", "LM", "jacobian+B+publish", "G assembly", "factorizations", "tail"]
print("k_feature_pair wavefronts", int(v[7]), {a: int(c / n) for a, c in zip(names, v[:7])}, "total cycles / wavefront", int(v[:7].sum() / n))
