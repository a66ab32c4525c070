"""Multi-GPU sharding of independent trajectories.

Trajectories / Monte-Carlo runs never exchange data during a run, so the path shards with NO data-path
collective: global trajectory g lives on rank g // per_rank (weak scaling: per-rank batch fixed).  The only
collective is the end-of-run ATE reduction: one all-reduce(sum) of {sum |e|^2, n} per sequence, after
which ATE = sqrt(sum / n).  Backend "nccl" is RCCL over xGMI on MI355X; the CPU tests use gloo.
"""
import numpy as np


def trajectory_ids(rank, world, per_rank):
    """Global trajectory indices owned by `rank` (contiguous block, weak scaling)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return list(range(rank * per_rank, (rank + 1) * per_rank))


def owner_of(global_traj, per_rank):
    return global_traj // per_rank


def ate_local(p_est, p_gt, seq_ids, n_seq):
    """Per-sequence partial sums [n_seq, 2] = {sum |p_est - p_gt|^2, count} for this rank's trajectories."""
    acc = np.zeros((n_seq, 2), dtype=np.float64)
    for pe, pg, s in zip(p_est, p_gt, seq_ids):
        d = np.asarray(pe, dtype=np.float64) - np.asarray(pg, dtype=np.float64)
        acc[s, 0] += float(np.sum(d * d))
        acc[s, 1] += d.shape[0] if d.ndim > 1 else 1
    return acc


def ate_allreduce(acc, device=None):
    """All-reduce the partial sums over the default process group (if initialised) and return the
    per-sequence ATE (RMSE of position, no alignment: the runner initialises from ground truth,
    datasets/asl_msckf.cpp:151-159)."""
    import torch
    import torch.distributed as dist
    t = torch.as_tensor(np.asarray(acc, dtype=np.float64))
    if device is not None:
        t = t.to(device)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    t = t.cpu().numpy()
    return np.sqrt(t[:, 0] / np.maximum(t[:, 1], 1.0))
