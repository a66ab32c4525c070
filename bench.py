#!/usr/bin/env python3
"""bench.py -- filter updates/sec of the MI355X-native batched MSCKF core (BASELINE.json metric).

Workload (config.workload), default `--config cfg3`: BASELINE.json configs[2] -- synthetic 30-camera window, 200 ending
feature tracks per update, float, 64 batched trajectories per GPU (SURVEY.md section 8d "cfg3").  One *step* = one filter
update for every trajectory of the batch: 10 x propagate + augmentState + marginalize(200 tracks) + prune of the oldest
camera state.  `value` is SURVEY.md 8d's metric: every frame's inputs (IMU samples + the frame's track work-lists) are
copied from page-locked host memory to the device INSIDE the timed region, pipelined with the kernels
(msckf_hip_run_frames_streamed); `resident_inputs` holds the same windows with all inputs already in HBM (upper bound).

`--config cfg4`: BASELINE.json configs[3] stand-in (EuRoC MH_01..05 are not on disk): 5 synthetic sequences x noise seeds,
EuRoC cam0 intrinsics (f_u != f_v), float, 128 trajectories per GPU, per-sequence ATE all-reduced over the ranks.

`--config cfg5`: BASELINE.json configs[4] -- 60-camera window, 500 tracks, fp16 measurement Jacobian / fp32 covariance
(dtype MSCKF_HIP_F16H_F32P), 512 trajectories per GPU (32 distinct scenarios, each run by 16 filters; `--trajectories`
overrides); no CPU leg at this size (one update of the reference's algorithm builds a 28 000 x 28 000 Q).

Launch: `python bench.py --gpus N --steps K --warmup W`.  N > 1 without a torch.distributed.run environment re-executes
itself as `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ...` (one rank per GPU, RCCL over xGMI; it fails
if the node has fewer than N devices); under torch.distributed.run it uses the ranks it is given.  Trajectories are
independent: rank r runs its own block of trajectories, weak scaling, no data-path collective; RCCL is used only for the
timing reduction and the end-of-run ATE all-reduce.

Prints ONE JSON line on rank 0.  `value` is the first timed K-step window (the driver's contract); `repeats` holds the
further windows (median / min / max).  `roofline` is for the dominant kernel (the longest single kernel of a step by HIP
events on the library's stream).  `cpu_baseline` (rank 0, N = 1): the REFERENCE'S OWN SOURCE (oracle/_ref/lib_ref.so =
/root/reference/include/msckf_mono/msckf.h compiled against oracle/ref_shim; `kind: "reference"`) timed on this box's
host cores on a bounded sample of the same workload, the restatement's LEAN mode beside it; `ate_vs_ref_m` = RMS position
difference between the HIP path and the CPU oracle on sampled trajectories at the end of the timed window.
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

K_IMU = 10
PEAK_F32_TFLOPS = 157.3   # MI355X dense f32 (vector = f32-input MFMA) peak, MI355X_MICROARCH.md
PEAK_F64_TFLOPS = 78.6
PEAK_HBM_GBS = 8000.0
CONFIGS = {
    # cfg2 has its own driver (run_cfg2): one double-precision filter through the per-call API
    "cfg2": dict(cid=2, N=10, F=50, B=1, iso=True, nseq=1, dtype="f64",
                 workload="BASELINE.json configs[1]: synthetic IMU + features, 10-cam window / 50 feats, double, 1 GPU single trajectory, "
                          "driven call by call through the drop-in shim as datasets/asl_msckf.cpp:227-296 drives the reference"),
    # name: config id (seed family), window, tracks, trajectories per GPU, isotropic noise, sequences
    "cfg3": dict(cid=3, N=30, F=200, B=64, iso=True, nseq=1, dtype="f32",
                 workload="BASELINE.json configs[2]: synthetic 30-cam window / 200 feats, float, 64 batched trajectories per GPU"),
    "cfg4": dict(cid=4, N=30, F=200, B=128, iso=False, nseq=5, dtype="f32",
                 workload="BASELINE.json configs[3] stand-in: 5 synthetic sequences x noise seeds (EuRoC MH_01..05 not on disk), "
                          "EuRoC cam0 intrinsics f_u != f_v, float, 30-cam window / 200 feats, 128 trajectories per GPU"),
    "cfg5": dict(cid=5, N=60, F=500, B=512, iso=True, nseq=1, dtype="f16h", uniq=32,
                 workload="BASELINE.json configs[4]: synthetic 60-cam window / 500 feats, fp16 measurement Jacobian / fp32 covariance, "
                          "512 batched trajectories per GPU (32 distinct scenarios, each run by 16 filters)"),
}


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
    gram = float(np.sum(3.0 * (n + 1) ** 2 + 54.0 * Ms))   # information form as built: 3 rows of B^ per track against the upper triangle + block diagonal
    return dict(feature=float(per_track.sum()), compress=float(compress), kalman=float(kalman), propagate=float(propagate), augment=float(augment), gram=gram)


def executed_flops_update(Ms, N, K_imu=K_IMU):
    """FLOP of one filter update AS BUILT (the algorithm the kernels execute, 2 per FMA, useful work counted once) -> dict
    per stage.  k_feature: triangulation + Jacobian (750 M), f64 reflectors + B^ (300 M), block-sparse gate
    G = H_x P_cc H_x^T from 6 x 6 blocks (192 per pair, M (M + 1) / 2 pairs), register Cholesky of N = G + sigma^2 I with
    the four rows [r ; H_f] riding along ((2M)^3 / 3 + 4 (2M)^2).  Compression in information form: SYRK 3 (n + 1)^2 per track + block diagonal 54 M, Cholesky
    (n + 1)^3 / 3.  Kalman in square-root gain form: P[:,15:] T^T (triangular T: D n^2), S = T PHt (n^3), S = L L^T with
    [PHt ; r_n^T] riding along (n^3 / 3 + D n^2), symmetric rank-n downdate (D^2 n)."""
    Ms = np.asarray(Ms, dtype=np.float64)
    rho = 2 * Ms - 3
    n, D = 6.0 * N, 15.0 + 6 * N
    feature = float(np.sum(1050 * Ms + 96 * Ms * (Ms + 1) + (2 * Ms) ** 3 / 3 + 4 * (2 * Ms) ** 2))
    gram = float(np.sum(3.0 * (n + 1) ** 2 + 54.0 * Ms))
    chol_gram = (n + 1) ** 3 / 3
    kalman = D * n * n + n ** 3 + (n ** 3 / 3 + D * n * n) + D * D * n
    propagate = K_imu * (4 * 15 ** 3 + 2 * 15 ** 2 * n + 600)
    augment = 72 * D + 432
    return dict(feature=feature, compress_stage1=gram, compress_merge=float(chol_gram), kalman=float(kalman),
                propagate=float(propagate), augment=float(augment))


def literal_flops_update(Ms, N):
    """FLOP (f64, 2 per FMA) of the literal anisotropic compression AS BUILT (kernels_literal.hip) for one update -> dict per launch.
    k_lit_pre: pivoted QR of H_f (2M x 3) + six rows of 6M entries per track; k_lit_gamma: 12 contraction rows per pair of
    tracks against the lower triangle of an n x n matrix (6 n^2 per track); k_literal: explicit rows (e x 12 M), the sweep on
    [explicit rows ; Gram matrix] (per step a rank-1 update of each: 2 (e - p)(n + 1 - k) + (n + 1 - k)^2), the basis products
    (~90 per entry of an r x r matrix) and the elimination of the r pivots of the (r + n + 1)-square Z (sum (nz - k)^2)."""
    Ms = np.asarray(Ms, dtype=np.float64)
    n = 6.0 * N; n1 = n + 1; e = 15 + n; r = e - 13
    k = np.arange(int(n)); p = 15 + k
    sweep = float(np.sum(2 * (e - p) * (n1 - k) + (n1 - k) ** 2))
    nz = r + n1
    elim = float(np.sum((nz - np.arange(int(r))) ** 2))
    return dict(lit_pre=float(np.sum(2 * (2 * Ms) * 9 * 2 + 6 * 6 * Ms * 14)), lit_gamma=float(len(Ms) * 6 * n * n),
                literal=float(e * 12 * np.mean(Ms) * 2 + sweep + 90 * r * r + elim))


def alg_bytes_update(Ms, N, s=4, K_imu=K_IMU):
    D = 15 + 6 * N
    return 2 * D * D * s + float(np.sum(2 * np.asarray(Ms)) * s) + 7 * N * s + D * s + 7 * K_imu * s


def _make_traj(a):
    from msckf_mono_amd import scenario as sc
    cid, g, N, F, nfr, iso, seq = a
    cfg = sc.filter_config(N, isotropic=iso)
    tr = sc.Trajectory(cid, g, N, F, nfr, cfg=cfg, path_id=seq)
    tr.landmarks = None   # not needed by the bench; keeps the pickles small
    return tr


def make_trajectories(c, rank, n_frames):
    """Scenario generation (host, numpy) for this rank's trajectories, spread over a process pool; runs BEFORE torch / HIP
    are initialised in this process (fork)."""
    from concurrent.futures import ProcessPoolExecutor
    nu = min(c.get("uniq", c["B"]), c["B"])            # distinct scenarios (cfg5: the batch repeats them)
    if os.environ.get("BENCH_SAME_SEEDS"):              # test hook (tests/test_bench_multirank.py): every rank runs rank 0's trajectories
        rank = 0
    jobs = [(c["cid"], rank * nu + b, c["N"], c["F"], n_frames, c["iso"], (rank * nu + b) % c["nseq"]) for b in range(nu)]
    nproc = max(1, min(32, (os.cpu_count() or 2) // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1"))), len(jobs)))
    if nproc == 1:
        out = [_make_traj(j) for j in jobs]
    else:
        with ProcessPoolExecutor(nproc) as ex:
            out = list(ex.map(_make_traj, jobs, chunksize=max(1, len(jobs) // (4 * nproc))))
    return [out[b % nu] for b in range(c["B"])]


def csrc_hash():
    """identity of the kernel sources a measurement belongs to: sha256 over msckf_mono_amd/csrc/*.hip, *.h (sorted by name)"""
    import glob
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "msckf_mono_amd", "csrc")
    for fn in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h"))):
        h.update(os.path.basename(fn).encode()); h.update(open(fn, "rb").read())
    return h.hexdigest()[:16]


def _parse_cpulist(txt):
    out = []
    for part in txt.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.extend(range(int(a), int(b or a) + 1))
    return out


def host_cpus_for_rank(local_rank, n):
    """n host cores for this rank's uploading + enqueue threads: on the GPU's NUMA node when sysfs tells, one disjoint block
    per local rank, never the node's first two cores (interrupt and runtime helper threads land there).  [] = do not pin."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
        cpus = allowed
        try:
            import torch
            pr = torch.cuda.get_device_properties(local_rank)
            bdf = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
            node = int(open("/sys/bus/pci/devices/%s/numa_node" % bdf).read())
            if node >= 0:
                on_node = [c for c in _parse_cpulist(open("/sys/devices/system/node/node%d/cpulist" % node).read()) if c in set(allowed)]
                if len(on_node) >= n + 2:
                    cpus = on_node
        except Exception:
            pass
        lws = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
        pool = cpus[2:] if len(cpus) >= lws * n + 2 else cpus
        if len(pool) < lws * n:
            return []
        per = len(pool) // lws
        blk = pool[(local_rank % lws) * per:(local_rank % lws) * per + n]
        return blk if len(blk) == n else []
    except Exception:
        return []


def device_clocks():
    """What rocm-smi reports for the shader / memory clocks right after the timed windows (off the timed path; best effort): leases
    of this pool have run the same binaries up to 1.6x apart, and the line should say on which kind it ran."""
    import subprocess
    try:
        o = subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True, timeout=20).stdout
        return [ln.strip() for ln in o.splitlines() if ("sclk" in ln or "mclk" in ln or "fclk" in ln)][:8] or None
    except Exception:
        return None


def self_launch_if_needed(args):
    """`python bench.py --gpus N` (N > 1) outside torch.distributed.run: re-execute as N ranks, one per GPU, on this node."""
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is not None:
        if args.gpus > 1 and int(env_world) != args.gpus:
            raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%s" % (args.gpus, env_world))
        return
    if args.gpus <= 1:
        return
    if os.environ.get("BENCH_DIST_BACKEND", "nccl") == "nccl":     # (the gloo test hook runs several ranks on one device)
        import torch
        have = torch.cuda.device_count()
        if have < args.gpus:
            raise SystemExit("bench.py: --gpus %d requested, %d HIP device(s) visible on this node" % (args.gpus, have))
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush(); sys.stderr.flush()
    os.execvpe(sys.executable, cmd, dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)     # the driver's invocation: --steps 20 --warmup 5
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="cfg3")
    ap.add_argument("--trajectories", type=int, default=0, help="trajectories per GPU (default: the configuration's)")
    ap.add_argument("--repeats", type=int, default=-1, help="timed K-step windows in all (default: until >= 0.5 s of timed region, 3..12)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="default run only: skip the short passes of BASELINE's other configurations (other_configs in the JSON line)")
    ap.add_argument("--parity-samples", type=int, default=0, help="with --no-cpu-baseline: still run the CPU oracle on this many sampled trajectories for ate_vs_ref (0 = none)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--streams", type=int, default=4, help="HIP streams for the timed region (slices of the batch run concurrently)")
    ap.add_argument("--no-early-accept-pass", action="store_true", help="skip the extra measurement with the gate early accept (profiling runs)")
    ap.add_argument("--no-upload-pass", action="store_true",
                    help="time the windows with resident inputs only (profiling runs: `value` is then the resident-input rate and says so)")
    ap.add_argument("--ring", type=int, default=6, help="device staging sets of the streamed-input run (2..8)")
    ap.add_argument("--upload-mode", type=int, default=0, help="0 host hand-over of uploaded frames (default), 1 device-side event waits")
    ap.add_argument("--compression", type=int, default=-1, help="msckf_hip_set_compression route (A/B runs; -1 = library default, 0 TSQR, 3 information form with the blocked matrix-core Cholesky)")
    ap.add_argument("--cov-form", type=int, default=0, help="msckf_hip_set_covariance_update form (A/B runs; 0 = square-root gain, blocked solve (default), 1 Joseph, 2 square-root gain, register-resident solve)")
    ap.add_argument("--no-pin", action="store_true", help="leave the uploading / enqueue threads where the scheduler puts them (default: one core each on the GPU's NUMA node)")
    ap.add_argument("--aniso-mode", type=int, default=0, help="u_var' != v_var' (cfg4): 0 the reference's literal R_n = Q_1^T R_o Q_1 on the device (default), 1 rows pre-whitened (GLS)")
    ap.add_argument("--prune-redundant", action="store_true",
                    help="cfg4: run the ASL runner's per-image cycle (asl_msckf.cpp:269-294, pruneRedundantStates :289 included) for the batch in "
                         "lockstep through msckf_hip_image_cycle_range (host bookkeeping per trajectory, batched device stages) instead of the resident scenario")
    ap.add_argument("--gate-early-accept", action="store_true",
                    help="exact early accept of the chi-square gate (msckf_hip_set_gate_early_accept); OFF for the headline number")
    args = ap.parse_args()
    self_launch_if_needed(args)
    if args.config == "cfg2":
        return run_cfg2(args)
    if args.prune_redundant:
        return run_cycle(args)
    c = dict(CONFIGS[args.config])
    if args.trajectories > 0:
        c["workload"] = c["workload"].replace("%d trajectories per GPU" % c["B"], "%d trajectories per GPU" % args.trajectories).replace(
            "%d batched trajectories per GPU" % c["B"], "%d batched trajectories per GPU" % args.trajectories)
        c["B"] = args.trajectories
    if args.config == "cfg5":          # big windows: one repeat window by default, no CPU legs that take minutes
        args.no_early_accept_pass = True
        if args.repeats <= 0:
            args.repeats = 2
    N_WIN, F_TRK, B_TRAJ = c["N"], c["F"], c["B"]

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    K, W = args.steps, args.warmup
    fill = N_WIN                     # frames needed to reach the steady-state window
    est_ms = 0.7 * B_TRAJ / 64.0 * (N_WIN / 30.0) ** 2 * (F_TRK / 200.0)    # rough step time, only used to size the number of repeat windows
    R = args.repeats if args.repeats > 0 else int(min(48, max(3, np.ceil(250.0 / (K * est_ms)))))
    streamed = not args.no_upload_pass
    R2 = min(R, 3) if streamed else 0                       # resident-input windows beside the streamed ones
    extra = R2 + 1 + (0 if (args.gate_early_accept or args.no_early_accept_pass) else 1)
    W2 = max(W, 3)                           # untimed frames after the state reads that follow the first window (see below)
    n_frames = fill + W + W2 + K * (R + 1 + extra)   # [fill | warmup | window 0 | one window straight after the state reads | re-warm | R - 1 timed windows | R2 resident windows | profiled | early-accept window]
    rendezvous_only = bool(os.environ.get("BENCH_RENDEZVOUS_ONLY"))   # test hook, see below
    t_gen = time.time()
    trajs = [] if rendezvous_only else make_trajectories(c, rank, n_frames)
    t_gen = time.time() - t_gen

    # more hardware queues than the runtime's default 4: the slices' streams + the copy stream must not share one
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import torch
    if not torch.cuda.is_available() and not (rendezvous_only and os.environ.get("BENCH_DIST_BACKEND") == "gloo"):
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # test hooks (tests/test_bench_multirank.py): run the N>1 code path on a 1-GPU box with gloo
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    if os.environ.get("BENCH_DEVICE_OVERRIDE") is not None:
        local_rank = int(os.environ["BENCH_DEVICE_OVERRIDE"])
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
    red_dev = "cuda" if backend == "nccl" else "cpu"
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))   # RCCL over xGMI
        else:
            dist.init_process_group(backend)
    if rendezvous_only:                               # test hook: prove the N-rank launch + collective, then stop (no filter run)
        ranks = torch.zeros(world, dtype=torch.float64, device=red_dev)
        ranks[rank] = rank + 1
        if dist is not None:
            dist.all_reduce(ranks, op=dist.ReduceOp.SUM)
        if rank == 0:
            print(json.dumps({"n_gpus": world, "ranks_seen": [int(x) - 1 for x in ranks.tolist()], "backend": backend if dist is not None else None}))
        if dist is not None:
            dist.barrier(); dist.destroy_process_group()
        return
    from msckf_mono_amd import capi, shard

    t_up = time.time()
    bt = capi.Batch(B_TRAJ, N_WIN, F_TRK, N_WIN, capi.F16H if c["dtype"] == "f16h" else capi.F32, local_rank)
    bt.scenario_alloc(n_frames, K_IMU)
    bt.set_anisotropic_noise(args.aniso_mode)
    for b, tr in enumerate(trajs):
        bt.initialize(b, tr.cfg, tr.imu0)
        for f in range(n_frames):
            fr = tr.frames[f]
            bt.scenario_set(f, b, tr.imu_for_frame(f), fr["M"], fr["slots"], fr["obs"], 1 if fr["Nw"] == N_WIN else 0)
    bt.scenario_commit()
    t_up = time.time() - t_up

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        bt.sync()

    def timed(f0, f1, streamed=False):
        """K frames bracketed by barrier + synchronize on both sides; max over ranks"""
        barrier()
        t0 = time.perf_counter()
        (bt.run_frames_streamed if streamed else bt.run_frames)(f0, f1)
        bt.sync()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([el], device=red_dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
            dist.barrier()
        return el

    bt.set_streams(args.streams)
    pin_cpus = [] if args.no_pin else host_cpus_for_rank(local_rank, args.streams + 1)
    if pin_cpus:
        bt.set_host_affinity(pin_cpus)
    bt.set_upload_ring(args.ring, args.upload_mode)
    if args.compression >= 0:
        bt.set_compression(args.compression)
    bt.set_covariance_update(args.cov_form)
    bt.set_gate_early_accept(args.gate_early_accept)
    f = fill + W
    run_timed_path = bt.run_frames_streamed if streamed else bt.run_frames
    if streamed:                     # page-lock the frames that will be streamed (set-up, like every other allocation)
        bt.scenario_pin(0, f + K * (R + 1) + W2)
    run_timed_path(0, fill)          # window fill (untimed), on the path that is timed: the staging ring wraps several times
    run_timed_path(fill, f)          # W warm-up steps (untimed)
    bt.sync()
    bt.imu_state(0); bt.last_stats(0, strict=False)   # first device-to-host reads of the process (the runtime sets its read path up lazily): not inside a timed window
    elapsed = timed(f, f + K, streamed=streamed)   # ---- THE timed region: exactly K steps -> `value`
    f_end_timed = f + K
    sample = sorted(set(int(x) for x in np.linspace(0, B_TRAJ - 1, 8)))
    p_dev_sample = {b: bt.imu_state(b)[13:16].copy() for b in sample}   # positions at the end of the timed window
    stats = [bt.last_stats(b, strict=False) for b in range(B_TRAJ)]
    f += K
    # The 2 x B small synchronous reads above leave the device idle for milliseconds; on some leases the windows right after
    # such a pause ran at 0.4 - 0.8 of the median (BENCH_r04: windows[1], [2]; not reproducible on others: scripts/pause_probe.py
    # reads 0.98 - 0.99 after state reads or sleeps of 1 .. 50 ms).  The repeat windows measure the steady state, so they start
    # after W2 untimed frames, like the first window after its warm-up; `value` (window 0) is not affected either way.
    # (rounds 1-4 timed their repeat windows straight after the reads; ONE such window is still recorded, outside the median, so
    # that the two conventions stay comparable: repeats.first_after_reads)
    el_after_reads = timed(f, f + K, streamed=streamed); f += K
    run_timed_path(f, f + W2); bt.sync(); f += W2
    rep = [elapsed]
    for _ in range(R - 1):           # further windows of the same size: spread of the measurement
        rep.append(timed(f, f + K, streamed=streamed)); f += K
    res_rep = []
    for _ in range(R2):              # the same kind of window with every input already resident in HBM (upper bound)
        res_rep.append(timed(f, f + K)); f += K
    h2d_gbs = None
    if streamed:
        try:                         # what the box's host-to-device path delivers from page-locked memory (64 MB copies)
            hbuf = torch.empty(64 << 20, dtype=torch.uint8).pin_memory()
            dbuf = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
            dbuf.copy_(hbuf, non_blocking=True); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(4):
                dbuf.copy_(hbuf, non_blocking=True)
            torch.cuda.synchronize()
            h2d_gbs = 4 * (64 << 20) / (time.perf_counter() - t0) / 1e9
            del hbuf, dbuf
        except Exception:
            h2d_gbs = None

    clocks = device_clocks() if (rank == 0 and not os.environ.get("BENCH_NO_SMI")) else None
    # ---- profiled pass over K further frames: HIP events on the library's stream, per stage (single stream)
    bt.profile_enable(True)
    bt.run_frames(f, f + K)
    prof = bt.profile_read()
    bt.profile_enable(False)
    ev_pair_ms = bt.profile_event_overhead()   # what an event pair with nothing between its records reads on that stream
    f += K

    # ---- the same K-frame measurement with the optional exact early accept of the chi-square gate (reported beside
    # the headline value, never as it: its gain depends on the ratio of residual noise to feature_cov)
    early_ms = None
    if not args.gate_early_accept and not args.no_early_accept_pass:
        bt.set_gate_early_accept(True)
        early_ms = 1e3 * timed(f, f + K) / K
        bt.set_gate_early_accept(False)
        f += K
    last = f - 1

    # ---- gate pass-rate / algorithmic work on the frames that were timed
    pass_rate = float(np.mean([s["n_passed"] / max(s["n_tracks"], 1) for s in stats]))
    fl = dict(feature=0.0, compress=0.0, kalman=0.0, propagate=0.0, augment=0.0, gram=0.0)
    ex = dict(feature=0.0, compress_stage1=0.0, compress_merge=0.0, kalman=0.0, propagate=0.0, augment=0.0)
    lit_on = (not c["iso"]) and args.aniso_mode == 0
    if lit_on:
        ex.update(lit_pre=0.0, lit_gamma=0.0, literal=0.0)
    by = 0.0
    up_bytes = 0.0
    for tr in trajs:
        for ff in range(fill + W, fill + W + K):
            one = alg_flops_update(tr.frames[ff]["M"], N_WIN)
            for k2 in fl:
                fl[k2] += one[k2] / (K * B_TRAJ)
            one = executed_flops_update(tr.frames[ff]["M"], N_WIN)
            if lit_on:
                one.update(literal_flops_update(tr.frames[ff]["M"], N_WIN))
            for k2 in ex:
                ex[k2] += one[k2] / (K * B_TRAJ)
            by += alg_bytes_update(tr.frames[ff]["M"], N_WIN) / (K * B_TRAJ)
            # one frame's streamed block: IMU samples, track count + drop count, lengths + offsets, (slot, u, v) per observation
            up_bytes += (K_IMU * 7 * 4 + 8 + 2 * 4 * F_TRK + 12.0 * float(np.sum(tr.frames[ff]["M"]))) / K
    f_update = sum(v for k2, v in fl.items() if k2 != "gram")   # the reference's algorithm (gram = the same stage as built, not additive)

    # ---- end-of-run ATE of the position against ground truth, per sequence: one all-reduce(sum) of {sum |e|^2, n}
    # per sequence over the ranks (RCCL), msckf_mono_amd/shard.py
    nseq = c["nseq"]
    p_est = [bt.imu_state(b)[13:16] for b in range(B_TRAJ)]
    p_gt = [tr.gt_frames["p"][last] for tr in trajs]
    acc = shard.ate_local(p_est, p_gt, [tr.path_id for tr in trajs], nseq)
    ate_seq = shard.ate_allreduce(acc, device=red_dev if dist is not None else None)
    acc_all = acc.sum(0)
    if dist is not None:
        t_all = torch.tensor(acc_all, device=red_dev, dtype=torch.float64)
        dist.all_reduce(t_all, op=dist.ReduceOp.SUM)
        acc_all = t_all.cpu().numpy()
    ate = float(np.sqrt(acc_all[0] / max(acc_all[1], 1.0)))
    # which ranks took part, and (test hook BENCH_SAME_SEEDS: every rank ran the same trajectories) whether they all ended in
    # the same bits -- one all-gather of the final positions, off the timed path
    ranks_seen, ranks_equal = [rank], None
    if dist is not None:
        mine = torch.tensor(np.concatenate([[float(rank)], np.concatenate(p_est)]), device=red_dev, dtype=torch.float64)
        got = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(got, mine)
        got = [g.cpu().numpy() for g in got]
        ranks_seen = sorted(int(g[0]) for g in got)
        if os.environ.get("BENCH_SAME_SEEDS"):
            ranks_equal = bool(all(np.array_equal(g[1:], got[0][1:]) for g in got))

    if rank == 0:
        updates = world * B_TRAJ * K
        value = updates / elapsed
        ms_per_step = 1e3 * elapsed / K
        rep_vals = [updates / e for e in rep]
        res_vals = [updates / e for e in res_rep]
        # a stage timer is an event pair around the stage's launches: the pair's own reading (marker packets) is measured
        # separately and taken off, so that a single-kernel stage reads what rocprofv3's kernel trace reads for that kernel
        stage_raw = {k2: v[0] / max(v[1], 1) for k2, v in prof.items()}
        stage_ms = {k2: (max(v - ev_pair_ms, 0.0) if prof[k2][1] > 0 else 0.0) for k2, v in stage_raw.items()}
        # utilisation on EXECUTED work: the FLOP of the algorithm as built per stage / the stage's HIP-event time / the peak
        # of the arithmetic the stage runs in (f64 matrix cores for the information-form compression, f32 elsewhere)
        ex_peak = dict(feature=PEAK_F32_TFLOPS, compress_stage1=PEAK_F64_TFLOPS, compress_merge=PEAK_F64_TFLOPS, kalman=PEAK_F32_TFLOPS,
                       propagate=PEAK_F32_TFLOPS, augment=PEAK_F32_TFLOPS, lit_pre=PEAK_F64_TFLOPS, lit_gamma=PEAK_F64_TFLOPS, literal=PEAK_F64_TFLOPS)
        executed_model = {}
        for k2, flop in ex.items():
            t_ms = stage_ms.get(k2, 0.0)
            tf = flop * B_TRAJ / (t_ms * 1e-3) / 1e12 if t_ms > 0 else 0.0
            executed_model[k2] = {"flop_per_step": flop * B_TRAJ, "ms_per_step": t_ms, "tflops": tf, "peak": ex_peak[k2], "frac": tf / ex_peak[k2]}
        ex_update = sum(ex.values())
        # single kernels with their own HIP-event pair; "kalman" is a launch SET (2 GEMMs + blocked gain solve + inject +
        # downdate) and is listed in stage_ms_per_step only.  Algorithmic FLOP = SURVEY.md section 8d per-unit figures of
        # the REFERENCE's algorithm (dense gate products, Householder compression), not the instructions executed.
        kernels = {
            "k_feature": dict(ms=stage_ms["feature"], flops=fl["feature"], bound="valu", peak=PEAK_F32_TFLOPS,
                              why="k_feature_pair: two tracks per wavefront (shortest with longest), lane = observation within a half; vector ALU + LDS "
                                  "crossbar (the gate's register-resident Cholesky exchanges its pivot column through ds_bpermute), 0 MFMA, f32 vector peak"),
            "k_gram": dict(ms=stage_ms["compress_stage1"], flops=fl["gram"], bound="mfma", peak=PEAK_F64_TFLOPS,
                           why="SYRK of the projected blocks on v_mfma_f64_16x16x4 (information form: its own FLOP, ~20x fewer than the "
                               "reference's Householder QR of the stack): bound by load latency, not by the matrix cores"),
            "k_chol_mfma": dict(ms=stage_ms["compress_merge"], flops=ex.get("compress_merge", 0.0), bound="mfma", peak=PEAK_F64_TFLOPS,
                                why="blocked f64 Cholesky of the Gram matrix ((n+1)^3/3 FLOP: the information form's own, the reference has no "
                                    "such stage), one workgroup per trajectory: bound by its 16-pivot diagonal-block chains"),
            "k_propagate": dict(ms=stage_ms["propagate"], flops=fl["propagate"], bound="valu", peak=PEAK_F32_TFLOPS,
                                why="sequential 15x15 chain per trajectory: latency bound"),
        }
        if lit_on:      # the literal anisotropic compression's six launches (k_lit_pre, k_lit_gamma, four k_lit_phase) as three stages, each with its own event pair in the profiled pass
            kernels["k_literal"] = dict(ms=stage_ms["literal"], flops=ex["literal"], bound="valu", peak=PEAK_F64_TFLOPS,
                                        why="one workgroup per trajectory: explicit rows, the Householder sweep for its decisions in LDS panels of 16, "
                                            "the basis products and the blocked elimination of R_n -- f64 vector arithmetic on 128 of 256 compute units, "
                                            "bound by the panels' dependent phases (barriers, LDS and L2 round trips), not by arithmetic")
            kernels["k_lit_gamma"] = dict(ms=stage_ms["lit_gamma"], flops=ex["lit_gamma"], bound="mfma", peak=PEAK_F64_TFLOPS,
                                          why="H_u^T H_u as a sum of rank-6 terms per track on v_mfma_f64_16x16x4, operands straight from L2: load-latency bound")
            kernels["k_lit_pre"] = dict(ms=stage_ms["lit_pre"], flops=ex["lit_pre"], bound="valu", peak=PEAK_F64_TFLOPS,
                                        why="a wavefront per track: pivoted QR of a 2M x 3 block in registers + six rows of output: latency / store bound")
        dom = max(kernels, key=lambda k2: kernels[k2]["ms"])
        kd = kernels[dom]
        dom_flops = kd["flops"] * B_TRAJ
        achieved = dom_flops / (kd["ms"] * 1e-3) / 1e12 if kd["ms"] > 0 else 0.0
        pmc, pmc_meta = pmc_block(dom)
        dom_stage = {"k_feature": "feature", "k_gram": "compress_stage1", "k_chol_mfma": "compress_merge", "k_propagate": "propagate",
                     "k_literal": "literal", "k_lit_gamma": "lit_gamma", "k_lit_pre": "lit_pre"}[dom]
        frac = achieved / kd["peak"]
        whole_frac = f_update * value / 1e12 / world / PEAK_F32_TFLOPS
        out = {
            "metric": "filter updates/sec (%d-cam window, %d feats)" % (N_WIN, F_TRK), "value": value, "unit": "updates/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32" if c["dtype"] == "f32" else "f32 (measurement Jacobian stored as f16)", "data": "synthetic",
            "config": {"workload": c["workload"], "name": args.config,
                       "cam_window": N_WIN, "tracks_per_update": F_TRK, "trajectories_per_gpu": B_TRAJ, "imu_per_update": K_IMU,
                       "parallelism": "replicated trajectories, %d per rank" % B_TRAJ,
                       "noise": "isotropic (f_u = f_v)" if c["iso"] else ("anisotropic (EuRoC f_u != f_v): " + ("the reference's R_o_j = A_j^T R_j A_j / HouseholderQR / R_n = Q_1^T R_o Q_1 on the device (kernels_literal.hip), zero-tail tolerance 8e-4 "
                                                                                                      "(the exact-arithmetic limit of Eigen's makeHouseholder rule; the reference's letter rule = tolerance 0 is selectable: msckf_hip_set_anisotropic_noise(h, 0, 0))" if args.aniso_mode == 0 else "rows pre-whitened by 1/sigma (GLS)")),
                       "sequences": nseq, "gate_early_accept": bool(args.gate_early_accept), "streams": args.streams,
                       "host_affinity": pin_cpus or None},
            "inputs": ("uploaded per frame inside the timed region (SURVEY.md 8d): page-locked host memory -> staging ring on a copy stream, "
                       "compact work-lists, %d sets" % args.ring) if streamed else "resident in HBM before the timed region (--no-upload-pass)",
            "upload": None if not streamed else {"bytes_per_step_per_gpu": int(up_bytes), "ring": args.ring,
                                                 "hand_over": "host" if args.upload_mode == 0 else "device events", "h2d_GBps_pinned_64MB": h2d_gbs},
            # the spread of the measurement where the driver's record shows it (nested keys end up under extra_keys there):
            # value_suspect = window 0 fell below 0.93 x the median of the run's windows (a slow window, DESIGN.md 9, reached `value`)
            "repeats_median": float(np.median(rep_vals)), "repeats_min_over_median": float(np.min(rep_vals) / np.median(rep_vals)),
            "value_over_repeats_median": float(rep_vals[0] / np.median(rep_vals)), "value_suspect": bool(rep_vals[0] < 0.93 * np.median(rep_vals)),
            "repeats": {"windows": len(rep_vals), "steps_each": K, "values": rep_vals, "median": float(np.median(rep_vals)),
                        "min": float(np.min(rep_vals)), "max": float(np.max(rep_vals)),
                        "first_after_reads": updates / el_after_reads,
                        "note": "value = windows[0] (the contract's K timed steps); the others are the same measurement on later frames, after the "
                                "state reads that follow window 0 and %d untimed frames (a pause of the device is not part of a steady-state window); "
                                "first_after_reads = one more window timed straight after those reads, without the untimed frames (rounds 1-4's convention), not in the median" % W2},
            "resident_inputs": None if not res_vals else {
                "values": res_vals, "median": float(np.median(res_vals)), "ms_per_step": 1e3 * float(np.median(res_rep)) / K,
                "streamed_over_resident": float(np.median(rep_vals) / np.median(res_vals)),
                "note": "the same K-step windows with every frame's inputs already in HBM (msckf_hip_run_frames): an upper bound, never `value`"},
            "roofline": {"bound": kd["bound"], "kernel": dom, "achieved": executed_model[dom_stage]["tflops"], "peak": kd["peak"], "unit": "TFLOP/s",
                         "frac": executed_model[dom_stage]["frac"],
                         "executed_frac": executed_model[dom_stage]["frac"], "executed_tflops": executed_model[dom_stage]["tflops"],
                         "executed_flops_per_launch": executed_model[dom_stage]["flop_per_step"],
                         "alg_equivalent_tflops": achieved,
                         "traffic": None if pmc is None else pmc.get("bytes_per_launch"),
                         "why": kd["why"], "kernel_ms_per_step": kd["ms"], "kernel_ms_note": "HIP-event pair around the launch on the library's stream, minus the reading of an empty pair (event_pair_overhead_ms): comparable with rocprofv3's kernel-trace duration", "alg_flops_per_launch": dom_flops,
                         "note": "achieved / frac = FLOP of the algorithm the kernel executes (executed_model: block-sparse gate, one Cholesky of "
                                 "G + sigma^2 I) per launch / measured kernel time / peak -- a utilisation, <= 1 by construction; alg_equivalent_* = the "
                                 "REFERENCE algorithm's FLOP for this stage (SURVEY.md 8d: dense gate products) / the same time: what the run time "
                                 "would buy of the reference's arithmetic at peak, not a utilisation (it may exceed 1)",
                         "alg_equivalent_ratio": frac,
                         "executed": None if pmc is None else pmc.get("executed"),
                         "traffic_source": pmc_meta,
                         "executed_model": executed_model,
                         "executed_flop_per_update": ex_update,
                         "whole_update_executed_tflops": ex_update * value / 1e12 / world,
                         "whole_update_executed_frac": ex_update * value / 1e12 / world / PEAK_F32_TFLOPS,
                         "kalman_set": {"ms_per_step": stage_ms["kalman"], "alg_flops_per_step": fl["kalman"] * B_TRAJ,
                                        "tflops_alg": fl["kalman"] * B_TRAJ / (stage_ms["kalman"] * 1e-3) / 1e12 if stage_ms["kalman"] > 0 else 0.0,
                                        "note": "square-root gain form: ~4.4 D^3 FLOP executed instead of the Joseph sequence's 16.3 D^3"},
                         "alg_flops_per_update": f_update, "alg_bytes_per_update": by,
                         "whole_update_alg_equivalent_tflops": f_update * value / 1e12 / world,
                         "whole_update_alg_equivalent_ratio": whole_frac,
                         "hbm_frac_alg": by * value / 1e9 / world / PEAK_HBM_GBS,
                         "stage_ms_per_step": stage_ms, "stage_ms_per_step_raw_event_pairs": stage_raw, "event_pair_overhead_ms": ev_pair_ms},
            "ranks_seen": ranks_seen, "ranks_bit_identical_for_equal_seeds": ranks_equal,
            "gate_pass_rate": pass_rate, "ate_m": ate, "ate_per_sequence_m": [float(x) for x in ate_seq],
            "scenario_gen_s": t_gen, "scenario_upload_s": t_up, "device_clocks_after_timed_windows": clocks,
            "with_gate_early_accept": None if early_ms is None else {
                "value": world * B_TRAJ * K / (early_ms * 1e-3 * K), "ms_per_step": early_ms,
                "note": "same K steps measured again with msckf_hip_set_gate_early_accept(1): exact bound gamma <= |r_o|^2/sigma^2, identical results; not the headline value"},
        }
        if world == 1 and args.config != "cfg5":
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(trajs[0], fill + W, args.cpu_seconds, N_WIN)
            if not args.no_cpu_baseline or args.parity_samples > 0:
                smp = sample if args.no_cpu_baseline is False else sample[:max(1, args.parity_samples)]
                out.update(ate_vs_reference(trajs, smp, p_dev_sample, f_end_timed, N_WIN, anisotropic=not c["iso"], literal=args.aniso_mode == 0))
        elif args.config == "cfg5":
            out["cpu_baseline"] = None
            out["cpu_baseline_note"] = ("not run at this size: one update of the reference's algorithm on a 60-camera / 500-track window builds a "
                                        "~28 000 x 28 000 Q (minutes and > 3 GB per filter); parity at this geometry is held by the -m gpu tests")
        # BASELINE.json's other configurations, shortened, in the same line (the default run only: cfg3, one GPU)
        if args.config == "cfg3" and world == 1 and streamed and not args.no_other_configs and args.trajectories <= 0:
            bt.close()
            out["other_configs"] = run_other_configs()
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def run_other_configs():
    """Short passes of BASELINE.json's other configurations after the default (cfg3) measurement, each as its own process of this
    script: cfg2 (one double filter through the shim), cfg4 on the literal anisotropic route and pre-whitened, cfg5 (60-camera
    window, fp16 Jacobian) -> {name: {value, ms_per_step, dtype, workload, roofline: {kernel, frac, peak, bound}, parity, wall_s}}.
    A pass that fails or runs over its limit is reported as such; it never takes the headline line with it."""
    import subprocess
    lit = ["--config", "cfg4", "--steps", "10", "--warmup", "6", "--repeats", "3", "--no-cpu-baseline", "--no-early-accept-pass", "--parity-samples", "2"]
    # cfg4_literal_640: BASELINE configs[3]'s own share per GPU (5 sequences x 1 024 seeds over 8 GPUs = 640 trajectories): the
    # literal route's kernels are one workgroup per trajectory, so 128 trajectories leave half of the 256 compute units idle
    specs = [("cfg2", ["--config", "cfg2", "--steps", "20", "--warmup", "5", "--repeats", "3", "--no-cpu-baseline"]),
             ("cfg4_literal", lit), ("cfg4_whitened", lit + ["--aniso-mode", "1"]),
             ("cfg4_literal_640", ["--config", "cfg4", "--trajectories", "640", "--steps", "6", "--warmup", "4", "--repeats", "3", "--no-cpu-baseline", "--no-early-accept-pass"]),
             ("cfg5", ["--config", "cfg5", "--steps", "5", "--warmup", "2", "--repeats", "2", "--no-cpu-baseline", "--no-early-accept-pass"]),
             # the runner's cycle with pruneRedundantStates for the batch in lockstep (host bookkeeping + batched device stages)
             ("cfg4_runner_cycle_prune_redundant", ["--config", "cfg4", "--prune-redundant", "--steps", "6", "--warmup", "2"])]
    res = {}
    for name, extra in specs:
        t0 = time.time()
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--no-other-configs"] + extra, capture_output=True, text=True, timeout=300)
            j = json.loads(r.stdout.strip().splitlines()[-1])
            rf = j.get("roofline") or {}
            if name == "cfg2":
                parity = j.get("parity")
            elif name == "cfg4_literal_640":
                parity = "as cfg4_literal (the same trajectories' first 128 and 512 more seeds)"
            elif name == "cfg4_runner_cycle_prune_redundant":
                parity = j.get("parity")
            elif name == "cfg5":
                parity = "no CPU leg at this size (cpu_baseline_note); held by tests/test_gpu_configs.py and tests/test_gpu_vs_reference.py (-m gpu)"
            else:
                parity = {k2: j.get(k2) for k2 in ("ate_vs_ref_m", "ate_vs_ref_is", "ate_vs_ref_literal_m", "ate_vs_ref_whitened_m", "ate_vs_ref_note")}
            res[name] = {"value": j["value"], "unit": j["unit"], "ms_per_step": j["ms_per_step"], "steps": j["steps"], "dtype": j["dtype"],
                         "workload": j["config"]["workload"], "noise": j["config"].get("noise"), "trajectories_per_gpu": j["config"].get("trajectories_per_gpu"),
                         "repeats_median": (j.get("repeats") or {}).get("median"),
                         "roofline": {"kernel": rf.get("kernel"), "frac": rf.get("frac"), "peak": rf.get("peak"), "bound": rf.get("bound"),
                                      "kernel_ms": rf.get("kernel_ms_per_step", rf.get("kernel_ms_per_update")), "stage_ms_per_step": rf.get("stage_ms_per_step", rf.get("stage_ms_per_update"))},
                         "latency_us": (j.get("latency_us") or {}).get("per_update_mean"),
                         "parity": parity, "wall_s": time.time() - t0}
        except Exception as ex_:
            res[name] = {"error": repr(ex_)[:300], "wall_s": time.time() - t0}
    return res


def run_cycle(args):
    """BASELINE.json configs[3] as the ASL runner runs it: per image augmentState -> update -> addFeatures -> marginalize ->
    pruneRedundantStates (asl_msckf.cpp:289) -> pruneEmptyStates, for ALL trajectories of the batch in lockstep through
    msckf_hip_image_cycle_range -- the feature bookkeeping of update() / addFeatures() / pruneRedundantStates on the host per
    trajectory (spread over host threads), every device stage one launch sequence for the batch, every read-back (poses for
    findRedundantCamStates, pruned states' poses) one copy for the batch.  Inputs are feature ids + normalized coordinates per
    image, as the front-end hands them to the filter (msckf_mono_amd/scenario.py: Trajectory.stream()).  `value` = filter
    updates per second of that loop, inputs converted beforehand (the timed region is propagate_range + image_cycle_range per
    image).  Parity: two sampled trajectories run again filter by filter through the per-filter entries -- bit-identical."""
    c = dict(CONFIGS[args.config])
    if args.trajectories > 0:
        c["B"] = args.trajectories
    N, F, B = c["N"], c["F"], c["B"]
    K, W = args.steps, args.warmup
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 or args.gpus > 1:
        raise SystemExit("bench.py --prune-redundant: 1 GPU")
    t0 = time.time()
    nf = N + W + K
    trajs = make_trajectories(c, 0, nf)
    streams = [tr.stream() for tr in trajs]
    t_gen = time.time() - t0
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    from msckf_mono_amd import capi

    def frame_inputs(k, idx):
        rd = np.ascontiguousarray(np.stack([trajs[b].imu_for_frame(k) for b in idx]), dtype=np.float64)
        return rd, capi.Batch.pack_image_inputs(len(idx), [k] * len(idx), [trajs[b].frame_times[k] for b in idx],
                                                [streams[b][k]["cur"] for b in idx], [streams[b][k]["new"] for b in idx])

    def fresh(idx):
        bt = capi.Batch(len(idx), N + 2, max(F + 32, 64), N + 2, capi.F32)
        bt.set_anisotropic_noise(args.aniso_mode)
        for i, b in enumerate(idx):
            bt.initialize(i, trajs[b].cfg, trajs[b].imu0)
        return bt

    allb = list(range(B))
    bt = fresh(allb)
    pre = [frame_inputs(k, allb) for k in range(nf)]     # conversions outside the timed region
    n_pruned0 = 0
    for k in range(N + W):
        rd, packed = pre[k]
        bt.propagate_range(0, B, rd); bt.image_cycle_range_packed(0, B, packed)
    bt.sync(); torch.cuda.synchronize()
    n_pruned0 = sum(len(bt.pruned_state_ids(b)) for b in (0, B // 2, B - 1))
    win = []
    t_all = time.perf_counter()
    for k in range(N + W, nf):
        t1 = time.perf_counter()
        rd, packed = pre[k]
        bt.propagate_range(0, B, rd); bt.image_cycle_range_packed(0, B, packed)
        win.append(time.perf_counter() - t1)
    bt.sync(); torch.cuda.synchronize()
    elapsed = time.perf_counter() - t_all
    # the same K images once more with the library's stage timers (HIP events around the device stages): what the device part costs
    bt.profile_enable(True)
    ncam = [bt.num_cam_states(b) for b in allb]
    state = {b: (bt.imu_state(b).copy(), bt.covariance(b).copy()) for b in (0, B - 1)}
    prof = bt.profile_read(); bt.profile_enable(False)
    n_pruned = sum(len(bt.pruned_state_ids(b)) for b in (0, B // 2, B - 1))
    bt.close()
    # ---- parity: the first and the last trajectory again, filter by filter through the per-filter entries
    same = True
    for b in (0, B - 1):
        one = fresh([b])
        for k in range(nf):
            one.propagate_range(0, 1, trajs[b].imu_for_frame(k))
            one.augment_state(0, k, trajs[b].frame_times[k])
            one.update(0, streams[b][k]["cur"][0], streams[b][k]["cur"][1]); one.add_features(0, streams[b][k]["new"][0], streams[b][k]["new"][1])
            one.marginalize(0); one.prune_redundant_states(0); one.prune_empty_states(0)
        same = same and bool(np.array_equal(one.imu_state(0), state[b][0]) and np.array_equal(one.covariance(0), state[b][1]))
        one.close()
    value = B * K / elapsed
    out = {
        "metric": "filter updates/sec (%d-cam window, %d feats)" % (N, F), "value": value, "unit": "updates/s", "n_gpus": 1, "steps": K, "warmup": W,
        "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": c["workload"] + "; the ASL runner's per-image cycle WITH pruneRedundantStates (asl_msckf.cpp:269-294) for the batch in lockstep "
                                               "(msckf_hip_image_cycle_range: update() / addFeatures() bookkeeping on the host per trajectory, device stages batched)",
                   "name": args.config, "cam_window": N, "tracks_per_update": F, "trajectories_per_gpu": B, "prune_redundant": True,
                   "noise": "anisotropic (EuRoC f_u != f_v), literal route" if args.aniso_mode == 0 else "anisotropic, rows pre-whitened",
                   "host_threads": min(os.cpu_count() or 1, 32)},
        "window_size_at_end": {"min": int(min(ncam)), "max": int(max(ncam))}, "pruned_states_of_3_sampled_trajectories": [int(n_pruned0), int(n_pruned)],
        "per_image_ms": {"median": 1e3 * float(np.median(win)), "min": 1e3 * float(np.min(win)), "max": 1e3 * float(np.max(win))},
        "note": "host-bound: the list surgery of update() for ~3 000 live features per trajectory and image runs on the host (as in the reference), "
                "the device waits for it; the resident-scenario path (bench.py --config cfg4) measures the device alone",
        "parity": {"per_filter_calls_bit_identical": same, "trajectories": [0, B - 1]},
        "roofline": None, "cpu_baseline": None, "scenario_gen_s": t_gen,
    }
    print(json.dumps(out))


def run_cfg2(args):
    """BASELINE.json configs[1]: ONE double-precision filter (10-camera window, 50 tracks per update), driven exactly as the
    reference's callers drive theirs: bench_src/cfg2_driver.cpp is compiled against the drop-in shim (include/msckf_mono/msckf.h)
    and libmsckf_hip.so and makes one propagate() call per IMU sample with the by-value getImuState() after it, then
    augmentState / update / addFeatures / marginalize / pruneEmptyStates per image (asl_msckf.cpp:227-296), timing every stage
    with the host wall clock as the reference's StageTiming does.  `value` = filter updates per second of that loop (latency
    bound: one trajectory cannot fill the chip); beside it the reference's own source in double on one host core
    (`cpu_baseline.kind` "reference"), the per-stage microseconds of both, the HIP-event stage timers of the same frames, and
    the state against the CPU oracle at the end of the run."""
    import subprocess
    import tempfile
    c = CONFIGS["cfg2"]
    N, F = c["N"], c["F"]
    K, W = args.steps, args.warmup
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 or args.gpus > 1:
        raise SystemExit("bench.py --config cfg2 is the single-trajectory latency configuration: 1 GPU")
    from msckf_mono_amd import scenario as sc
    nf = N + W + K
    tr = sc.Trajectory(2, 0, N, F, nf, cfg=sc.filter_config(N, isotropic=True))
    st = tr.stream()
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    from msckf_mono_amd import capi
    # ---- the timed loop: C++ caller over the shim
    build = tempfile.mkdtemp(prefix="cfg2_")
    exe = os.path.join(build, "cfg2_driver")
    libdir = os.path.dirname(capi.LIB_PATH)
    cc = subprocess.run(["g++", "-std=c++17", "-O2", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "bench_src", "cfg2_driver.cpp"), "-o", exe,
                         "-L" + libdir, "-lmsckf_hip", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"],
                        capture_output=True, text=True)
    if cc.returncode:
        raise SystemExit("cfg2 driver did not compile: " + cc.stderr[-2000:])
    cam, noise, prm = capi.pack_config(tr.cfg)
    lines = [" ".join(repr(float(x)) for x in np.concatenate([cam, noise, prm, tr.imu0])), "%d %d" % (nf, N + W)]
    for k in range(nf):
        rd = tr.imu_for_frame(k)
        lines.append(str(len(rd)) + " " + " ".join(repr(float(x)) for x in rd.ravel()))
        for kind in ("cur", "new"):
            obs, ids = st[k][kind]
            lines.append(str(len(ids)) + " " + " ".join("%r %r %d" % (float(z[0]), float(z[1]), i) for z, i in zip(obs, ids)))
    text = "\n".join(lines) + "\n"
    runs = []
    for _ in range(max(1, args.repeats if args.repeats > 0 else 5)):       # every run replays the whole sequence: fill + warm-up untimed, K frames timed
        r = subprocess.run([exe], input=text, capture_output=True, text=True, timeout=600)
        if r.returncode:
            raise SystemExit("cfg2 driver failed (%d): %s" % (r.returncode, r.stderr[-2000:]))
        runs.append(json.loads(r.stdout.strip().splitlines()[-1]))
    first = runs[0]
    frame_us = np.array(first["frame_us"])
    value = len(frame_us) / (frame_us.sum() * 1e-6)
    rep_vals = [len(x["frame_us"]) / (sum(x["frame_us"]) * 1e-6) for x in runs]

    # ---- the same frames through the C-ABI with the library's HIP-event stage timers (which kernel dominates, executed FLOP)
    f = capi.MSCKF(capi.F64, n_cap=N + 3, f_cap=max(F, 64), m_cap=N + 3)
    f.initialize(tr.cfg, tr.imu0)
    sid = 0
    Ms_timed = []
    for k in range(nf):
        if k == N + W:
            f.batch.sync(); f.batch.profile_enable(True)
        for r7 in tr.imu_for_frame(k):
            sid += 1; f.propagate(r7)
        f.augmentState(sid, float(k)); f.update(*st[k]["cur"]); f.addFeatures(*st[k]["new"]); f.marginalize(); f.pruneEmptyStates()
        if k >= N + W:
            Ms_timed.append(tr.frames[k]["M"])
    prof = f.batch.profile_read(); f.batch.profile_enable(False)
    ev_pair_ms = f.batch.profile_event_overhead()
    imu_c = f.getImuState(); P_c = f.getCovariance()
    stage_ms = {k2: max(v[0] / max(v[1], 1) - ev_pair_ms, 0.0) if v[1] else 0.0 for k2, v in prof.items()}
    ex = dict(feature=0.0, compress_stage1=0.0, compress_merge=0.0, kalman=0.0)
    for Ms in Ms_timed:
        one = executed_flops_update(Ms, N)
        for k2 in ex:
            ex[k2] += one[k2] / len(Ms_timed)
    ex_peak = dict(feature=PEAK_F64_TFLOPS, compress_stage1=PEAK_F64_TFLOPS, compress_merge=PEAK_F64_TFLOPS, kalman=PEAK_F64_TFLOPS)
    model = {k2: {"flop_per_update": ex[k2], "ms": stage_ms.get(k2, 0.0), "tflops": (ex[k2] / (stage_ms[k2] * 1e-3) / 1e12 if stage_ms.get(k2, 0) > 0 else 0.0),
                  "peak": ex_peak[k2]} for k2 in ex}
    for m in model.values():
        m["frac"] = m["tflops"] / m["peak"]
    dom_stage = max(model, key=lambda k2: model[k2]["ms"])
    dom_kernel = {"feature": "k_feature<double>", "compress_stage1": "k_gram", "compress_merge": "k_chol_mfma<double> (Gram)", "kalman": "kalman launch set (k_gemm_mfma<double>, k_gain_w)"}[dom_stage]

    # ---- CPU: the reference's own source in double on ONE core (the reference is single-threaded), and the oracle for parity
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    def cpu_run(o, timed_from):
        o.initialize(tr.cfg, tr.imu0)
        sid2 = 0; t_acc = 0.0; stages = dict(imu_prop=0.0, msckf_augment_state=0.0, msckf_update=0.0, msckf_add_features=0.0, msckf_marginalize=0.0, msckf_prune_empty_states=0.0)
        for k in range(nf):
            t0 = time.perf_counter()
            rd = tr.imu_for_frame(k)
            o.propagate(rd); sid2 += len(rd); t1 = time.perf_counter()
            o.augmentState(sid2, float(k)); t2 = time.perf_counter()
            o.update(*st[k]["cur"]); t3 = time.perf_counter()
            o.addFeatures(*st[k]["new"]); t4 = time.perf_counter()
            o.marginalize(); t5 = time.perf_counter()
            o.pruneEmptyStates(); t6 = time.perf_counter()
            if k >= timed_from:
                t_acc += t6 - t0
                for nm, dt in zip(stages, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5)):
                    stages[nm] += dt
        n_t = nf - timed_from
        return n_t / t_acc, {k2: 1e6 * v / n_t for k2, v in stages.items()}
    lean = po.Oracle(po.F64, po.LEAN)
    lean_rate, lean_stage = cpu_run(lean, N + W)
    cpu = None
    if not args.no_cpu_baseline:
        if po.ref_available():
            ref_rate, ref_stage = cpu_run(po.Oracle(po.F64, impl="ref"), N + W)
            cpu = {"value": ref_rate, "unit": "updates/s", "cores": 1, "kind": "reference",
                   "sample": "%d filter updates of one double-precision filter (10-cam window, 50 tracks), the reference's own msckf.h over oracle/ref_shim "
                             "(not Eigen: neither Eigen nor Boost is installed), one thread" % K,
                   "stage_us": ref_stage, "lean_value": lean_rate, "lean_stage_us": lean_stage}
        else:
            cpu = {"value": lean_rate, "unit": "updates/s", "cores": 1, "kind": "port", "sample": "%d filter updates, oracle LEAN mode (lib_ref.so not present)" % K, "stage_us": lean_stage}
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers as Hh
    err_state = Hh.rel(np.array(first["imu"]), lean.getImuState()[:16])
    err_capi = Hh.state_errors(imu_c, lean.getImuState(), f.getCamStates()[0], lean.getCamStates()[0], P_c, lean.getCovariance())
    trP = float(np.trace(lean.getCovariance()))
    out = {
        "metric": "filter updates/sec (%d-cam window, %d feats)" % (N, F), "value": value, "unit": "updates/s", "n_gpus": 1, "steps": K, "warmup": W,
        "ms_per_step": 1e3 / value, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": c["workload"], "name": "cfg2", "cam_window": N, "tracks_per_update": F, "trajectories_per_gpu": 1, "imu_per_update": K_IMU,
                   "parallelism": "one trajectory, one stream: every call of the reference's API is a call into libmsckf_hip.so",
                   "noise": "isotropic (f_u = f_v)"},
        "latency_us": {"per_update_mean": float(frame_us.mean()), "per_update_median": float(np.median(frame_us)), "per_update_max": float(frame_us.max()),
                       "stage_mean": first["stage_us"],
                       "note": "host wall clock around each call of the shim, as the reference's StageTiming (asl_msckf.cpp:229-296); imu_prop = 10 x "
                               "(propagate + by-value getImuState, answered from the library's host copy of the IMU state: no device round trip); read_state = getImuState + getNumCamStates after the image"},
        "repeats": {"runs": len(rep_vals), "values": rep_vals, "median": float(np.median(rep_vals)), "min": float(np.min(rep_vals)), "max": float(np.max(rep_vals)),
                    "note": "value = the first run; every run replays fill + warm-up untimed and times the same K frames"},
        "roofline": {"bound": "mfma" if dom_stage != "feature" else "valu", "kernel": dom_kernel, "achieved": model[dom_stage]["tflops"], "peak": PEAK_F64_TFLOPS, "unit": "TFLOP/s",
                     "frac": model[dom_stage]["frac"], "traffic": None,
                     "traffic_note": "no PMC pass for this configuration: a single trajectory's kernels are one workgroup (or one wavefront per track) each, "
                                     "bound by launch and dependency latency -- see latency_us",
                     "kernel_ms_per_update": model[dom_stage]["ms"], "executed_model": model, "stage_ms_per_update": stage_ms, "event_pair_overhead_ms": ev_pair_ms,
                     "note": "frac = FLOP of the algorithm as built for the stage (executed_flops_update) / HIP-event time of the stage / f64 peak"},
        "cpu_baseline": cpu,
        "parity": {"shim_run_vs_oracle_state_rel": err_state, "capi_run_vs_oracle": err_capi,
                   "trace_P_rel": abs(first["trace_P"] - trP) / trP, "bar": 1e-6},
    }
    print(json.dumps(out))


def pmc_block(kernel):
    """HBM bytes per launch and executed-instruction figures of a kernel from the committed rocprofv3 PMC passes
    (separate --pmc runs of this same command, profiles/pmc_traffic.json written by scripts/rocpd_pmc.py; FETCH_SIZE is
    doubled as MI355X_MICROARCH.md prescribes for gfx950) and where they came from (profile tag, commit and kernel-source
    hash of the passes: they are NOT collected in the run that prints them).  The counters are only reported when the
    passes ran on the kernel sources this run was built from (csrc_hash); otherwise (None, meta with the reason)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None, {"stale": True, "reason": "profiles/pmc_traffic.json absent"}
    try:
        t = json.load(open(path))
        meta = dict(t.get("_meta", {}), file="profiles/pmc_traffic.json", collected="separate rocprofv3 --pmc passes, not this run")
        now = csrc_hash()
        if meta.get("csrc_hash") != now:
            meta.update(stale=True, csrc_hash_now=now,
                        reason="the PMC passes ran on other kernel sources (csrc_hash %s) than this run's (%s): traffic / executed withheld" % (meta.get("csrc_hash"), now))
            return None, meta
        meta["stale"] = False
        for k2, v in t.items():
            if k2 == kernel or k2.startswith(kernel):
                return v, meta
    except Exception as e:
        return None, {"stale": True, "reason": "profiles/pmc_traffic.json unreadable: %r" % (e,)}
    return None, meta


def _oracle_window(o, tr, k, N):
    o.propagate(tr.imu_for_frame(k)); o.augmentState(k, 0.0)
    fr = tr.frames[k]
    if len(fr["M"]):
        o.setTracks(fr["M"], fr["slots"], fr["obs"]); o.marginalize()
    if o.getNumCamStates() == N:
        o.dropOldest(1)


def ate_vs_reference(trajs, sample, p_dev, n_run, N, anisotropic=False, literal=True):
    """'ATE vs ref' of BASELINE.json's metric: the CPU oracle (float, LEAN = same results as the reference's steps) runs
    the sampled trajectories free from frame 0 to the end of the timed window on host threads; reported: its ATE against
    ground truth, the HIP path's ATE on the same trajectories, and the RMS position difference between the two.  With
    anisotropic pixel noise (f_u != f_v) the oracle restates the reference's R_o_j = A_j^T R_j A_j / R_n = Q_1^T R_o Q_1 with
    the zero-tail tolerance the device's literal route uses (`ate_vs_ref_m`); the distance to the oracle's pre-whitened mode
    (the GLS update, msckf_hip_set_anisotropic_noise(h, 1)) is reported beside it."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    res, resw = {}, {}

    def run(b, whiten, out):
        tr = trajs[b]
        o = po.Oracle(po.F32, po.LEAN)
        if whiten:
            o.setWhiten(True)
        elif anisotropic:
            o.setTinyRowTol(8e-4)
        o.initialize(tr.cfg, tr.imu0)
        for k in range(n_run):
            _oracle_window(o, tr, k, N)
        out[b] = o.getImuState()[13:16]
    th = [threading.Thread(target=run, args=(b, False, res)) for b in sample]
    if anisotropic:
        th += [threading.Thread(target=run, args=(b, True, resw)) for b in sample]
    t0 = time.time()
    for t in th:
        t.start()
    for t in th:
        t.join()
    gt = {b: trajs[b].gt_frames["p"][n_run - 1] for b in sample}
    rms = lambda d: float(np.sqrt(np.mean([float(np.sum(np.square(x))) for x in d])))
    main_ref = res if (literal or not anisotropic) else resw
    out = {"ate_ref_m": rms([main_ref[b] - gt[b] for b in sample]), "ate_hip_sample_m": rms([p_dev[b] - gt[b] for b in sample]),
           "ate_vs_ref_m": rms([p_dev[b] - main_ref[b] for b in sample]),
           "ate_vs_ref_note": "%d sampled trajectories, free-running from frame 0 to the end of the timed window (%d frames), float CPU oracle "
                              "vs HIP path, %.1f s wall" % (len(sample), n_run, time.time() - t0)}
    if anisotropic:
        out["ate_vs_ref_note"] += ("; anisotropic noise: the reference's own update is defined only to its rounding envelope there (two roundings of its "
                                   "source differ by 1e-4 .. 4e-4 on the gyro bias per update, DESIGN.md 3.3) -- the restatement compared here takes the "
                                   "exact-arithmetic limit of its zero-tail rule (tolerance 8e-4 in float), as the device's literal route does")
        out["ate_vs_ref_literal_m"] = rms([p_dev[b] - res[b] for b in sample])
        out["ate_vs_ref_whitened_m"] = rms([p_dev[b] - resw[b] for b in sample])
        out["ate_vs_ref_is"] = "literal restatement (zero-tail tolerance 8e-4)" if literal else "pre-whitened restatement"
    return out


def cpu_baseline(tr, frame, budget_s, N):
    """The reference's own source (oracle/_ref/lib_ref.so: msckf.h unmodified over oracle/ref_shim -- Eigen/Boost are
    not installed; its full m x m Q and dense R_o included) timed on this box's host cores: one filter update per filter,
    one filter per thread at a time (the reference is single-threaded per trajectory).  Beside it the restatement's LEAN
    mode (thin QR, no dense R_o: the same results without the reference's O(m^2) waste)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    cores = os.cpu_count() or 1
    o = po.Oracle(po.F32, po.LEAN)
    o.initialize(tr.cfg, tr.imu0)
    for k in range(frame):           # bring one filter to the steady-state window (lean mode, same results)
        _oracle_window(o, tr, k, N)
    fr = tr.frames[frame]
    rd = tr.imu_for_frame(frame)
    # lean: several updates per core
    per_core = 4
    lean = [o.clone() for _ in range(cores * per_core)]
    t_lean = po.time_updates(lean, cores, 1, rd, frame, fr["M"], fr["slots"], fr["obs"], 1)
    lean_rate = len(lean) / t_lean
    # reference source: one update per filter (each takes seconds: full Q of a ~5 800-row stack, dense R_o)
    n_f = min(cores, 32) if budget_s >= 10 else max(1, min(cores, 32) // 4)   # ~0.45 GB of dense Q / R_o per filter
    if po.ref_available():
        kind, what = "reference", "reference source (msckf.h over oracle/ref_shim)"
        filters = []
        for _ in range(n_f):
            r = po.Oracle(po.F32, impl="ref")
            r.initialize(tr.cfg, tr.imu0)
            while r.getNumCamStates() < o.getNumCamStates():
                r.augmentState(r.getNumCamStates(), 0.0)
            cams, _ids = o.getCamStates()
            r.setCovariance(o.getCovariance()); r.setImuState(o.getImuState())
            for i, cpose in enumerate(cams):
                r.setCamPose(i, cpose)
            r.setNumResidualized(o.numResidualized())
            filters.append(r)
    else:
        kind, what = "port", "oracle FAITHFUL mode (lib_ref.so not present)"
        filters = [o.clone() for _ in range(n_f)]
        for f in filters:
            f.setMode(po.FAITHFUL)
    t_f = po.time_updates(filters, min(cores, n_f), 1, rd, frame, fr["M"], fr["slots"], fr["obs"], 1)
    return {"value": n_f / t_f, "unit": "updates/s", "cores": min(cores, n_f), "kind": kind,
            "implementation": "the reference's own msckf.h / types.h / matrix_utils.h, unmodified, compiled against oracle/ref_shim (this repository's "
                              "Eigen / Boost stand-in with naive GEMMs: Eigen is not installed), NOT Eigen" if kind == "reference" else "oracle restatement, FAITHFUL mode",
            "sample": "%d filters x 1 filter update (30-cam window, 200 tracks, f32), %s, %.1f s wall" % (n_f, what, t_f),
            "lean_value": lean_rate, "lean_cores": cores,
            "lean_sample": "%d filters x 1 update, oracle LEAN mode (thin QR, no dense R_o), %.2f s wall" % (len(lean), t_lean)}


if __name__ == "__main__":
    main()
