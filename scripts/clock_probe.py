"""Do the occasional slow windows coincide with a drop of the device's clocks?  cfg3, 64 trajectories, streamed in
four slices as bench.py runs it: NW windows of 20 frames back to back while a thread samples the amdgpu hwmon files (shader
clock freq1_input, memory clock freq2_input, power1_average / power1_input) of the first card about every 0.3 ms; per window:
updates/s, the lowest and the mean shader clock seen inside it, the mean power.  Prints the windows below 0.9 x median in full."""
import glob
import json
import os
import sys
import threading
import time

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from msckf_mono_amd import capi

K, NW = 20, int(sys.argv[1]) if len(sys.argv) > 1 else 120


def hwmon_files():
    out = {}
    for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        for name in ("freq1_input", "freq2_input", "power1_average", "power1_input"):
            p = os.path.join(d, name)
            if os.path.exists(p) and name not in out:
                out[name] = p
        if "freq1_input" in out:
            break
    return out


files = hwmon_files()
print("sampling", files, flush=True)
samples = []          # (t, sclk_hz, mclk_hz, power_uW)
stop = False


def sampler():
    fds = {k: open(p, "rb", buffering=0) for k, p in files.items()}

    def rd(k):
        f = fds.get(k)
        if f is None:
            return 0
        try:
            f.seek(0)
            return int(f.read().split()[0])
        except Exception:
            return -1
    while not stop:
        samples.append((time.perf_counter(), rd("freq1_input"), rd("freq2_input"), rd("power1_average") if "power1_average" in fds else rd("power1_input")))
        time.sleep(0.0002)


c = dict(bench.CONFIGS["cfg3"])
B, N = c["B"], c["N"]
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
bt.set_streams(4)
pins = bench.host_cpus_for_rank(0, 5)
if pins:
    bt.set_host_affinity(pins)
bt.set_upload_ring(6, 0)
bt.scenario_pin(0, nfr)
bt.run_frames_streamed(0, N + 5); bt.sync()
th = threading.Thread(target=sampler, daemon=True); th.start()
time.sleep(0.05)
f = N + 5
win = []
for w in range(NW):
    t0 = time.perf_counter()
    bt.run_frames_streamed(f, f + K); bt.sync()
    t1 = time.perf_counter()
    win.append((t0, t1)); f += K
stop = True; th.join()
S = np.array(samples, dtype=np.float64)
rate = np.array([B * K / (t1 - t0) for t0, t1 in win])
med = float(np.median(rate))
rows = []
for w, (t0, t1) in enumerate(win):
    m = (S[:, 0] >= t0) & (S[:, 0] <= t1)
    s = S[m]
    rows.append({"window": w, "updates_per_s": float(rate[w]), "x_median": float(rate[w] / med), "samples": int(m.sum()),
                 "sclk_mhz_min": float(s[:, 1].min() / 1e6) if len(s) else None, "sclk_mhz_mean": float(s[:, 1].mean() / 1e6) if len(s) else None,
                 "mclk_mhz_min": float(s[:, 2].min() / 1e6) if len(s) else None, "power_w_mean": float(s[:, 3].mean() / 1e6) if len(s) else None})
slow = [r for r in rows if r["x_median"] < 0.9]
allr = np.array([[r["sclk_mhz_min"] or 0, r["sclk_mhz_mean"] or 0, r["power_w_mean"] or 0] for r in rows])
print(json.dumps({"windows": NW, "median": med, "min": float(rate.min()), "max": float(rate.max()), "samples_total": len(samples),
                  "sample_period_ms": float(np.median(np.diff(S[:, 0])) * 1e3) if len(S) > 1 else None,
                  "sclk_mhz_min_over_all_windows": float(allr[:, 0].min()), "sclk_mhz_mean_over_all_windows": float(allr[:, 1].mean()),
                  "power_w_mean_over_all_windows": float(allr[:, 2].mean()), "slow_windows": slow,
                  "five_slowest": sorted(rows, key=lambda r: r["x_median"])[:5]}))
bt.close()
