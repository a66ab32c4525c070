"""What slows the first timed window after a pause (windows[1] = 0.40 x median after the 64 state reads of
bench.py)?  cfg3, 64 trajectories; windows of 20 frames each, median of a run of back-to-back windows as the yardstick, then
one window after each kind of pause, on the streamed path (uploader + enqueue threads) and on the resident path with one
stream (no host threads in the timed region at all: whatever is slow there is the device, not the threads)."""
import json
import os
import sys
import time

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")     # as bench.py: the slices' streams + the copy stream on hardware queues of their own
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
QUICK = "--quick" in sys.argv                        # tests/test_gpu_pause.py: the streamed path only, the state-read pause three times
import bench
from msckf_mono_amd import capi

K = 20
c = dict(bench.CONFIGS["cfg3"])
B, N = c["B"], c["N"]
NW = 14 if QUICK else 40
nfr = N + 5 + K * NW
trajs = bench.make_trajectories(c, 0, nfr)
bt = capi.Batch(B, N, c["F"], N, capi.F32)
bt.scenario_alloc(nfr, bench.K_IMU)
for b, tr in enumerate(trajs):
    bt.initialize(b, tr.cfg, tr.imu0)
    for f in range(nfr):
        fr = tr.frames[f]
        bt.scenario_set(f, b, tr.imu_for_frame(f), fr["M"], fr["slots"], fr["obs"], 1 if fr["Nw"] == N else 0)
bt.scenario_commit()
import torch


def smi():
    import subprocess
    try:
        o = subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True, timeout=20).stdout
        return " | ".join(l.strip() for l in o.splitlines() if "sclk" in l or "mclk" in l)[:200]
    except Exception as e:
        return repr(e)


def window(f, streamed):
    bt.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    (bt.run_frames_streamed if streamed else bt.run_frames)(f, f + K)
    bt.sync()
    return B * K / (time.perf_counter() - t0)


def pause_reads():
    for b in range(B):
        bt.last_stats(b, strict=False); bt.imu_state(b)


def pause_sleep(ms):
    time.sleep(ms * 1e-3)


summary = {}
for streams, streamed in (((4, True),) if QUICK else ((4, True), (1, False), (4, False))):
    bt.set_streams(streams)
    pins = bench.host_cpus_for_rank(0, streams + 1)
    if pins:
        bt.set_host_affinity(pins)
    bt.set_upload_ring(6, 0)
    f = 0
    run = bt.run_frames_streamed if streamed else bt.run_frames
    if streamed:
        bt.scenario_pin(0, nfr)
    if f == 0 and streams == 4 and streamed:
        run(0, N + 5); f = N + 5
    else:
        f = f_next
    base = []
    for _ in range(5):
        base.append(window(f, streamed)); f += K
    med = float(np.median(base))
    res = {}
    kinds = (("64 state reads", pause_reads), ("64 state reads (2)", pause_reads), ("64 state reads (3)", pause_reads)) if QUICK else \
            (("64 state reads", pause_reads), ("sleep 1 ms", lambda: pause_sleep(1)), ("sleep 5 ms", lambda: pause_sleep(5)), ("sleep 50 ms", lambda: pause_sleep(50)))
    for name, fn in kinds:
        fn()
        a = window(f, streamed); f += K
        b2 = window(f, streamed); f += K
        res[name] = (round(a / med, 3), round(b2 / med, 3))
    f_next = f
    print("streams", streams, "streamed", streamed, "median %.0f" % med, "first / second window after the pause (x median):", res, flush=True)
    summary["streams=%d,streamed=%d" % (streams, int(streamed))] = {"median": med, "after_pause": res}
if not QUICK:
    print(smi())
bt.close()
print(json.dumps(summary))
