"""Phase timers of k_literal (MSCKF_HIP_LITERAL_TIMERS=1) at BASELINE configs[3]'s geometry (30-camera window, 200 tracks,
anisotropic noise): wall time of single frames of a resident scenario at 8 and 128 trajectories, the phase table of two
trajectories on stderr."""
import sys
import time

sys.path.insert(0, "/root/repo")
import bench
from msckf_mono_amd import capi

for B in (8, 128):
    c = dict(bench.CONFIGS["cfg4"]); c["B"] = B
    nfr = 36
    trajs = bench.make_trajectories(c, 0, nfr)
    bt = capi.Batch(B, 30, 200, 30, capi.F32)
    bt.scenario_alloc(nfr, 10)
    for b, tr in enumerate(trajs):
        bt.initialize(b, tr.cfg, tr.imu0)
        for f in range(nfr):
            fr = tr.frames[f]
            bt.scenario_set(f, b, tr.imu_for_frame(f), fr["M"], fr["slots"], fr["obs"], 1 if fr["Nw"] == 30 else 0)
    bt.scenario_commit()
    bt.run_frames(0, 32); bt.sync()
    for f in range(32, 36):
        t0 = time.perf_counter(); bt.run_frames(f, f + 1); bt.sync(); dt = time.perf_counter() - t0
        print("B", B, "frame", f, "ms %.2f" % (dt * 1e3))
    print(bt.literal_info(0)); print(bt.literal_info(B - 1))
    bt.close()
