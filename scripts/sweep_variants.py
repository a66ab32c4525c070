#!/usr/bin/env python3
"""One process, one resident scenario (cfg3), many launch variants: streams x upload ring / hand-over, each
re-initialised and measured on the SAME frames (window fill, warm-up, then R windows of K steps; median updates/s).
Experiment helper for DESIGN.md's tables; the driver's number comes from bench.py.
Usage: sweep_variants.py [--steps 20] [--windows 5] [--hwq 8] "streams=3,streamed=1,ring=6,mode=0" ..."""
import argparse, json, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--windows", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--hwq", type=int, default=8)
    ap.add_argument("--config", default="cfg3")
    ap.add_argument("variants", nargs="+")
    a = ap.parse_args()
    os.environ["GPU_MAX_HW_QUEUES"] = str(a.hwq)
    import bench
    c = dict(bench.CONFIGS[a.config])
    N, F, B = c["N"], c["F"], c["B"]
    K, W, R = a.steps, a.warmup, a.windows
    nfr = N + W + K * R
    trajs = bench.make_trajectories(c, 0, nfr)
    import torch
    from msckf_mono_amd import capi
    bt = capi.Batch(B, N, F, N, capi.F16H if c["dtype"] == "f16h" else capi.F32, 0)
    bt.scenario_alloc(nfr, bench.K_IMU)
    for b, tr in enumerate(trajs):
        for f in range(nfr):
            fr = tr.frames[f]
            bt.scenario_set(f, b, tr.imu_for_frame(f), fr["M"], fr["slots"], fr["obs"], 1 if fr["Nw"] == N else 0)
    bt.scenario_commit()
    bt.scenario_pin(N, nfr)
    ref = None
    for v in a.variants:
        kv = dict(streams=3, streamed=1, ring=6, mode=0, early=0, overlap=0)
        kv.update({k: int(x) for k, x in (p.split("=") for p in v.split(",") if p)})
        for b, tr in enumerate(trajs):
            bt.initialize(b, tr.cfg, tr.imu0)
        bt.set_streams(kv["streams"]); bt.set_upload_ring(kv["ring"], kv["mode"])
        bt.set_gate_early_accept(bool(kv["early"]))
        bt.set_feature_overlap(bool(kv["overlap"]))
        run = bt.run_frames_streamed if kv["streamed"] else bt.run_frames
        bt.run_frames(0, N); run(N, N + W); bt.sync()
        vals = []
        f = N + W
        for _ in range(R):
            torch.cuda.synchronize(); bt.sync()
            t0 = time.perf_counter()
            run(f, f + K); bt.sync()
            vals.append(B * K / (time.perf_counter() - t0)); f += K
        x = np.concatenate([bt.imu_state(b) for b in (0, B // 2, B - 1)])
        same = True if ref is None else bool(np.array_equal(x, ref))
        if ref is None:
            ref = x
        print(json.dumps(dict(variant=kv, median=round(float(np.median(vals))), min=round(min(vals)), max=round(max(vals)),
                              ms_per_step=round(1e3 * B / float(np.median(vals)), 4), bit_identical_to_first=same)), flush=True)


if __name__ == "__main__":
    main()
