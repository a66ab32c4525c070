#!/usr/bin/env python3
"""bench.py -- filter updates/sec of the MI355X-native batched MSCKF core (BASELINE.json metric).

Workload (config.workload): BASELINE.json configs[2] -- synthetic 30-camera window, 200 ending feature
tracks per update, float, 64 batched trajectories per GPU (SURVEY.md section 8d "cfg3").  One *step* = one
filter update for every trajectory of the batch: 10 x propagate + augmentState + marginalize(200 tracks)
+ prune of the oldest camera state.  All inputs (IMU samples and per-frame track work-lists) are resident
in HBM before the timed region; the timed region launches kernels only (no host syncs, no H2D).

Launch: `python bench.py --gpus 1 --steps K --warmup W`, or one rank per GPU under torch.distributed.run
(trajectories are independent: rank r runs its own 64 trajectories, weak scaling, no data-path
collective; RCCL is used only for the timing reduction and the end-of-run ATE all-reduce).

Prints ONE JSON line on rank 0; `roofline` is for the dominant kernel (the longest single kernel of a step,
k_feature since the compression moved to the information form),
`cpu_baseline` is the CPU oracle (restatement of the reference; the reference itself cannot be built
here -- see BASELINE.md) timed on this box's host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_WIN, F_TRK, B_TRAJ, K_IMU, CONFIG_ID = 30, 200, 64, 10, 3
PEAK_F32_TFLOPS = 157.3   # MI355X dense f32 (vector = f32-input MFMA) peak, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def alg_flops_update(Ms, N, K_imu=K_IMU):
    """Algorithmic FLOP of one filter update (SURVEY.md section 8d formulas, isotropic noise) -> dict per stage."""
    Ms = np.asarray(Ms, dtype=np.float64)
    rho = 2 * Ms - 3
    D, n = 15 + 6 * N, 6 * N
    m, r = float(rho.sum()), float(15 + 6 * N)
    per_track = 600 * Ms + 150 * Ms + 12 * (2 * Ms) * (6 * Ms + 1) + (2 * rho * (6 * Ms) ** 2 + 2 * rho ** 2 * (6 * Ms) + rho ** 3 / 3 + 2 * rho ** 2)
    compress = 2 * m * n ** 2 - (2.0 / 3.0) * n ** 3 + 4 * m * n
    kalman = 2 * D * D * r + 2 * r * r * D + r ** 3 / 3 + 2 * r * r * D + 2 * D * r + 2 * D * D * r + 4 * D ** 3 + 2 * D * r * r + 2 * D * D * r
    propagate = K_imu * (4 * 15 ** 3 + 2 * 15 ** 2 * n + 600)
    augment = 72 * D + 432
    return dict(feature=float(per_track.sum()), compress=float(compress), kalman=float(kalman), propagate=float(propagate), augment=float(augment))


def alg_bytes_update(Ms, N, s=4, K_imu=K_IMU):
    D = 15 + 6 * N
    return 2 * D * D * s + float(np.sum(2 * np.asarray(Ms)) * s) + 7 * N * s + D * s + 7 * K_imu * s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--streams", type=int, default=2, help="HIP streams for the timed region (halves of the batch run concurrently)")
    ap.add_argument("--no-early-accept-pass", action="store_true", help="skip the extra measurement with the gate early accept (profiling runs)")
    ap.add_argument("--compression", type=int, default=-1, help="msckf_hip_set_compression route (A/B runs; -1 = library default)")
    ap.add_argument("--cov-form", type=int, default=0, help="msckf_hip_set_covariance_update form (A/B runs; 0 = library default)")
    ap.add_argument("--gate-early-accept", action="store_true",
                    help="exact early accept of the chi-square gate (msckf_hip_set_gate_early_accept); OFF for the headline number")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # test hooks (tests/test_bench_multirank.py): run the N>1 code path on a 1-GPU box with gloo
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    if os.environ.get("BENCH_DEVICE_OVERRIDE") is not None:
        local_rank = int(os.environ["BENCH_DEVICE_OVERRIDE"])
    torch.cuda.set_device(local_rank)
    red_dev = "cuda" if backend == "nccl" else "cpu"
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))   # RCCL over xGMI
        else:
            dist.init_process_group(backend)
    from msckf_mono_amd import capi, scenario as sc

    K, W = args.steps, args.warmup
    fill = N_WIN                     # frames needed to reach the steady-state window
    n_frames = fill + W + 3 * K      # [fill | warmup | timed | profiled | timed with the gate early accept]
    t_gen = time.time()
    trajs = [sc.Trajectory(CONFIG_ID, rank * B_TRAJ + b, N_WIN, F_TRK, n_frames) for b in range(B_TRAJ)]
    cfg = trajs[0].cfg
    bt = capi.Batch(B_TRAJ, N_WIN, F_TRK, N_WIN, capi.F32, local_rank)
    bt.scenario_alloc(n_frames, K_IMU)
    for b, tr in enumerate(trajs):
        bt.initialize(b, cfg, tr.imu0)
        for f in range(n_frames):
            fr = tr.frames[f]
            bt.scenario_set(f, b, tr.imu_for_frame(f), fr["M"], fr["slots"], fr["obs"], 1 if fr["Nw"] == N_WIN else 0)
    bt.scenario_commit()
    t_gen = time.time() - t_gen

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        bt.sync()

    bt.set_streams(args.streams)
    if args.compression >= 0:
        bt.set_compression(args.compression)
    bt.set_covariance_update(args.cov_form)
    bt.set_gate_early_accept(args.gate_early_accept)
    bt.run_frames(0, fill + W)       # window fill + warm-up (untimed)
    barrier()
    t0 = time.perf_counter()
    bt.run_frames(fill + W, fill + W + K)
    bt.sync()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if dist is not None:
        tt = torch.tensor([elapsed], device=red_dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        dist.barrier()

    # ---- profiled pass over K further frames: HIP events on the library's stream, per stage
    bt.profile_enable(True)
    bt.run_frames(fill + W + K, fill + W + 2 * K)
    prof = bt.profile_read()
    bt.profile_enable(False)

    # ---- gate pass-rate / algorithmic work on the frames that were timed
    stats = [bt.last_stats(b) for b in range(B_TRAJ)]
    pass_rate = float(np.mean([s["n_passed"] / max(s["n_tracks"], 1) for s in stats]))
    fl = dict(feature=0.0, compress=0.0, kalman=0.0, propagate=0.0, augment=0.0)
    by = 0.0
    for tr in trajs:
        for f in range(fill + W, fill + W + K):
            one = alg_flops_update(tr.frames[f]["M"], N_WIN)
            for k2 in fl:
                fl[k2] += one[k2] / (K * B_TRAJ)
            by += alg_bytes_update(tr.frames[f]["M"], N_WIN) / (K * B_TRAJ)
    f_update = sum(fl.values())

    # ---- the same K-frame measurement with the optional exact early accept of the chi-square gate (reported beside
    # the headline value, never as it: its gain depends on the ratio of residual noise to feature_cov)
    early_ms = None
    if not args.gate_early_accept and not args.no_early_accept_pass:
        bt.set_gate_early_accept(True)
        barrier()
        te0 = time.perf_counter()
        bt.run_frames(fill + W + 2 * K, fill + W + 3 * K)
        bt.sync()
        torch.cuda.synchronize()
        early_el = time.perf_counter() - te0
        bt.set_gate_early_accept(False)
        if dist is not None:
            tt = torch.tensor([early_el], device=red_dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            early_el = float(tt.item())
        early_ms = 1e3 * early_el / K

    # ---- ATE of the end-of-run position against ground truth (RCCL all-reduce of {sum e^2, n})
    last = (fill + W + 3 * K - 1) if early_ms is not None else (fill + W + 2 * K - 1)
    se = 0.0
    for b, tr in enumerate(trajs):
        e = bt.imu_state(b)[13:16] - tr.gt_frames["p"][last]
        se += float(e @ e)
    acc = torch.tensor([se, float(B_TRAJ)], device=red_dev, dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(acc, op=dist.ReduceOp.SUM)
    ate = float(np.sqrt(acc[0].item() / acc[1].item()))

    if rank == 0:
        updates = world * B_TRAJ * K
        value = updates / elapsed
        ms_per_step = 1e3 * elapsed / K
        stage_ms = {k2: v[0] / max(v[1], 1) for k2, v in prof.items()}
        # single kernels with their own HIP-event pair; "kalman" is a launch SET (6 GEMMs + gain + inject + symmetrize)
        # and is listed in stage_ms_per_step only.  Algorithmic FLOP = SURVEY.md section 8d per-unit figures of the
        # REFERENCE's algorithm (dense gate products, Householder compression), not the instructions executed.
        kern_ms = {"k_feature": stage_ms["feature"], "k_gram + k_chol_blk (compression: two kernels)": stage_ms["compress_stage1"] + stage_ms["compress_merge"],
                   "k_propagate": stage_ms["propagate"]}
        kern_fl = {"k_feature": fl["feature"], "k_gram + k_chol_blk (compression: two kernels)": fl["compress"], "k_propagate": fl["propagate"]}
        dom = max(kern_ms, key=kern_ms.get)
        dom_flops = kern_fl[dom] * B_TRAJ
        achieved = dom_flops / (kern_ms[dom] * 1e-3) / 1e12 if kern_ms[dom] > 0 else 0.0
        out = {
            "metric": "filter updates/sec (30-cam window, 200 feats)", "value": value, "unit": "updates/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[2]: synthetic 30-cam window / 200 feats, float, 64 batched trajectories per GPU",
                       "cam_window": N_WIN, "tracks_per_update": F_TRK, "trajectories_per_gpu": B_TRAJ, "imu_per_update": K_IMU,
                       "parallelism": "replicated trajectories, %d per rank" % B_TRAJ, "noise": "isotropic (f_u = f_v)",
                       "gate_early_accept": bool(args.gate_early_accept)},
            "roofline": {"bound": "mfma", "kernel": dom, "achieved": achieved, "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_F32_TFLOPS, "traffic": pmc_traffic(dom.split(" ")[0]),
                         "kernel_ms_per_step": kern_ms[dom], "alg_flops_per_launch": dom_flops,
                         "kalman_set": {"ms_per_step": stage_ms["kalman"], "alg_flops_per_step": fl["kalman"] * B_TRAJ,
                                        "tflops": fl["kalman"] * B_TRAJ / (stage_ms["kalman"] * 1e-3) / 1e12 if stage_ms["kalman"] > 0 else 0.0},
                         "alg_flops_per_update": f_update, "alg_bytes_per_update": by,
                         "whole_update_tflops": f_update * value / 1e12 / world,
                         "whole_update_frac": f_update * value / 1e12 / world / PEAK_F32_TFLOPS,
                         "hbm_frac_alg": by * value / 1e9 / world / PEAK_HBM_GBS,
                         "stage_ms_per_step": stage_ms},
            "gate_pass_rate": pass_rate, "ate_m": ate, "scenario_gen_s": t_gen,
            "with_gate_early_accept": None if early_ms is None else {
                "value": world * B_TRAJ * K / (early_ms * 1e-3 * K), "ms_per_step": early_ms,
                "note": "same K steps measured again with msckf_hip_set_gate_early_accept(1): exact bound gamma <= |r_o|^2/sigma^2, identical results; not the headline value"},
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(trajs[0], fill + W, args.cpu_seconds)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def pmc_traffic(kernel):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (separate --pmc
    FETCH_SIZE / WRITE_SIZE runs of this same command, profiles/pmc_traffic.json, written by
    scripts/rocpd_pmc.py); FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950.  None if absent."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None
    try:
        t = json.load(open(path)).get(kernel)
        return None if t is None else float(t["bytes_per_launch"])
    except Exception:
        return None


def cpu_baseline(tr, frame, budget_s):
    """Time the CPU oracle (oracle/, restatement of msckf.h) on this box's host cores: the FAITHFUL mode
    (reference's algorithmic steps, incl. full m x m Q) and the LEAN mode, one filter update per filter,
    one filter per thread at a time (the reference is single-threaded per trajectory)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    cores = os.cpu_count() or 1
    o = po.Oracle(po.F32, po.LEAN)
    o.initialize(tr.cfg, tr.imu0)
    for k in range(frame):           # bring one filter to the steady-state window (lean mode, same results)
        o.propagate(tr.imu_for_frame(k)); o.augmentState(k, 0.0)
        fr = tr.frames[k]
        if len(fr["M"]):
            o.setTracks(fr["M"], fr["slots"], fr["obs"]); o.marginalize()
        if o.getNumCamStates() == N_WIN:
            o.dropOldest(1)
    fr = tr.frames[frame]
    rd = tr.imu_for_frame(frame)
    # lean: several updates per core
    per_core = 4
    lean = [o.clone() for _ in range(cores * per_core)]
    t_lean = po.time_updates(lean, cores, 1, rd, frame, fr["M"], fr["slots"], fr["obs"], 1)
    lean_rate = len(lean) / t_lean
    # faithful: one update per core (each takes seconds: full Q of a ~5 800-row stack)
    n_f = min(cores, 32) if budget_s >= 10 else max(1, min(cores, 32) // 4)   # ~0.45 GB of dense Q/R_o per filter
    faithful = [o.clone() for _ in range(n_f)]
    for f in faithful:
        f.setMode(po.FAITHFUL)
    t_f = po.time_updates(faithful, min(cores, n_f), 1, rd, frame, fr["M"], fr["slots"], fr["obs"], 1)
    return {"value": n_f / t_f, "unit": "updates/s", "cores": min(cores, n_f), "kind": "port",
            "sample": "%d filters x 1 filter update (30-cam window, 200 tracks, f32), oracle FAITHFUL mode, %.1f s wall" % (n_f, t_f),
            "lean_value": lean_rate, "lean_sample": "%d filters x 1 update, oracle LEAN mode (thin QR, no dense R_o), %.2f s wall" % (len(lean), t_lean)}


if __name__ == "__main__":
    main()
