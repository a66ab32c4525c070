"""Phase timers of k_literal (MSCKF_HIP_LITERAL_TIMERS=1) at BASELINE configs[3]'s geometry (30-camera window, 200 tracks,
anisotropic noise): wall time of single frames of a resident scenario at 8 and 128 trajectories, the phase table of two
trajectories on stderr.  --all: 24 more frames at 128 trajectories, per frame the slowest trajectory of phase 2 (a launch lasts
as long as its slowest workgroup) and the step timers of every trajectory that has kept handed-through rows."""
import sys
import time

sys.path.insert(0, "/root/repo")
import bench
from msckf_mono_amd import capi

for B in (8, 128):
    c = dict(bench.CONFIGS["cfg4"]); c["B"] = B
    nfr = 60 if "--all" in sys.argv else 36
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
    if B == 128 and "--all" in sys.argv:      # a launch lasts as long as its slowest trajectory: per frame, the slowest one of every phase
        import os, re, tempfile
        import numpy as np
        pat = re.compile(r"\[k_literal b=(\d+)\] us: explicit rows (\d+) Gram (\d+) sweep (\d+) kept (\d+) handed-through rows (\d+) basis products (\d+) Z fill (\d+) eliminate (\d+) store (\d+) total (\d+)")
        for f in range(36, nfr):
            bt.run_frames(f, f + 1); bt.sync()
            tmp = tempfile.TemporaryFile(mode="w+"); sys.stderr.flush(); old = os.dup(2); os.dup2(tmp.fileno(), 2)
            infos = [bt.literal_info(b) for b in range(B)]
            os.dup2(old, 2); os.close(old); tmp.seek(0)
            text = tmp.read().splitlines()
            rows = np.array([list(map(int, m.groups())) for m in map(pat.match, text) if m])
            for l in [l for l in text if "kept handed-through rows =" in l][:6]:
                b_h = int(l.split("b=")[1].split("]")[0]); print("  ", l, "| kept .. Z fill of it", rows[b_h, 4:8].tolist())
            p2 = rows[:, 4:8].sum(1); w = int(p2.argmax())
            print("frame", f, "phase 2 (kept .. Z fill) median %d max %d at b=%d" % (np.median(p2), p2.max(), w), "its timers", rows[w, 1:].tolist(),
                  "handed-through kept", infos[w]["kept_handed_through_rows"], "| trajectories with such rows:", sum(1 for i in infos if i["kept_handed_through_rows"] > 0),
                  "| total median %d max %d" % (np.median(rows[:, 10]), rows[:, 10].max()))
    bt.close()
