"""Phase breakdown of the blocked matrix-core Cholesky (kernels_chol.hip) on the GPU box, -DMSCKF_ABLATE build only:
    make -C msckf_mono_amd/csrc ablate && MSCKF_HIP_LIB=msckf_mono_amd/lib_ab/libmsckf_hip_ablate.so python scripts/chol_phases.py
Prints shader-clock cycles per launch and phase for the GRAM (f64) and GAIN (f32) instances at the cfg3 window."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from msckf_mono_amd import capi, scenario as sc  # noqa: E402

N, F, B, nf = 30, 200, int(os.environ.get("PHASES_B", "8")), 40
trajs = [sc.Trajectory(3, b, N, F, nf) for b in range(B)]
bt = capi.Batch(B, N, F, N, capi.F32)
bt.set_compression(3); bt.set_covariance_update(0)
for b, tr in enumerate(trajs):
    bt.initialize(b, tr.cfg, tr.imu0)
bt.scenario_alloc(nf, 10)
for k in range(nf):
    for b, tr in enumerate(trajs):
        fr = tr.frames[k]
        bt.scenario_set(k, b, tr.imu_for_frame(k), fr["M"], fr["slots"], fr["obs"], 1 if fr["Nw"] == N else 0)
bt.scenario_commit()
bt.run_frames(0, 32); bt.sync()
if os.environ.get("CHOL_DBG"):
    bt.L.msckf_hip_debug_set(400, int(os.environ["CHOL_DBG"]))   # ablation knob of kernels_chol.hip (timing only: results are wrong)
out = (C.c_ulonglong * 16)()
bt.L.msckf_hip_debug_chol_cycles(out, 1)
po8 = (C.c_ulonglong * 8)()
bt.L.msckf_hip_debug_prop_cycles(po8, 1)
g40 = (C.c_ulonglong * 40)()
bt.L.msckf_hip_debug_gram_cycles(g40, 1)
m16 = (C.c_ulonglong * 16)()
bt.L.msckf_hip_debug_gemm_cycles(m16, 1)
if hasattr(bt.L, "msckf_hip_debug_chol_sub"):
    bt.L.msckf_hip_debug_chol_sub((C.c_ulonglong * 32)(), 1)
bt.run_frames(32, nf); bt.sync()
bt.L.msckf_hip_debug_chol_cycles(out, 1)
names = ["load", "panel->LDS", "diag block", "L21", "outputs", "trailing"]
for m, nm in enumerate(("GRAM f64", "GAIN f32")):
    v = np.array(out[8 * m:8 * m + 8], dtype=np.float64)
    n = max(v[6], 1)
    print(nm, "launches", int(v[6]), {a: int(c / n) for a, c in zip(names, v[:6])}, "total cycles/launch", int(v[:6].sum() / n))
sub = (C.c_ulonglong * 32)()
if hasattr(bt.L, "msckf_hip_debug_chol_sub"):
    bt.L.msckf_hip_debug_chol_sub(sub, 1)
    sv = np.array(sub, dtype=np.float64).reshape(4, 8)
    for part in range(4):
        if sv[part, 5] > 0:
            print("GAIN load phase, part", part, {a: int(c / sv[part, 5]) for a, c in zip(["own S blocks", "publish", "rendezvous", "read siblings", "masks + rest"], sv[part, :5])})
bt.L.msckf_hip_debug_prop_cycles(po8, 1)
v = np.array(po8, dtype=np.float64); n = max(v[5], 1)
print("k_propagate launches", int(v[5]), {a: int(c / n) for a, c in zip(["load", "state chain", "Phi series", "P_II/Phi_tot chains", "write back + P_IC"], v[:5])}, "total", int(v[:5].sum() / n))
bt.L.msckf_hip_debug_gram_cycles(g40, 1)
g = np.array(g40, dtype=np.float64).reshape(8, 5)
for strip in range(8):
    if g[strip, 4] > 0:
        print("k_gram SYRK strip", strip, "launches", int(g[strip, 4]), {a: int(c / g[strip, 4]) for a, c in zip(["start-up + count", "K loop", "group sum", "epilogue"], g[strip, :4])}, "total", int(g[strip, :4].sum() / g[strip, 4]))
bt.L.msckf_hip_debug_gemm_cycles(m16, 1)
m = np.array(m16, dtype=np.float64).reshape(2, 8)
for row, nm in enumerate(("PHt", "downdate")):
    if m[row, 4] > 0:
        print("k_gemm_mfma", nm, "tile (1,1) launches", int(m[row, 4]), {a: int(c / m[row, 4]) for a, c in zip(["set-up", "first k-tile", "k loop", "epilogue"], m[row, :4])}, "total", int(m[row, :4].sum() / m[row, 4]))
