"""Where a GEMM launch's time goes (-DMSCKF_ABLATE build): wall-clock start / end of every workgroup of the last PHt and
downdate launches of a 64-trajectory cfg3 batch -- dispatch skew, per-workgroup duration, tail.
    make -C msckf_mono_amd/csrc ablate && MSCKF_HIP_LIB=msckf_mono_amd/lib_ab/libmsckf_hip_ablate.so python scripts/experiments/gemm_wg_trace.py"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from msckf_mono_amd import capi, scenario as sc

N, F, B, nf = 30, 200, 64, 36
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
bt.run_frames(0, nf); bt.sync()
buf = (C.c_ulonglong * (2 * 4096 * 2))()
bt.L.msckf_hip_debug_gemm_trace(buf)
t = np.array(buf, dtype=np.float64).reshape(2, 4096, 2)
for row, nm in enumerate(("PHt", "downdate")):
    a = t[row][: 16 * B]
    live = a[:, 1] > a[:, 0]           # workgroups that returned early recorded start == end
    s0 = a[:, 0].min()
    st, en = (a[:, 0] - s0) / 100.0, (a[:, 1] - s0) / 100.0   # us at 100 MHz
    dur = (en - st)[live]
    print(nm, "workgroups", len(a), "with work", int(live.sum()),
          "| start of last workgroup %.1f us | end of last %.1f us | duration of a working one: median %.1f, max %.1f us" % (st.max(), en.max(), np.median(dur), dur.max()))
    order = np.argsort(en)[-5:]
    print("   last to finish (wg index, tile x, y, trajectory, start, end):", [(int(i), int(i % 4), int(i // 4 % 4), int(i // 16), round(float(st[i]), 1), round(float(en[i]), 1)) for i in order])
    hist, edges = np.histogram(st, bins=8)
    print("   start-time histogram (us):", [(round(float(e), 1), int(h)) for e, h in zip(edges, hist)])
