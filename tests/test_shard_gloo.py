"""N>1 path on CPU: world_size-2 gloo run of the trajectory sharding + end-of-run ATE all-reduce."""
import os
import subprocess
import sys
import textwrap

import numpy as np

from msckf_mono_amd import shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_partition_is_disjoint_and_complete():
    world, per = 8, 64
    seen = []
    for r in range(world):
        ids = shard.trajectory_ids(r, world, per)
        assert len(ids) == per and all(shard.owner_of(g, per) == r for g in ids)
        seen += ids
    assert sorted(seen) == list(range(world * per))


def test_ate_local_matches_definition():
    rng = np.random.default_rng(0)
    est, gt = rng.normal(size=(5, 3)), rng.normal(size=(5, 3))
    acc = shard.ate_local(est, gt, [0, 1, 0, 1, 1], 2)
    for s in (0, 1):
        idx = [i for i, q in enumerate([0, 1, 0, 1, 1]) if q == s]
        assert np.isclose(acc[s, 0], sum(np.sum((est[i] - gt[i]) ** 2) for i in idx))
        assert acc[s, 1] == len(idx)
    assert np.allclose(shard.ate_allreduce(acc), np.sqrt(acc[:, 0] / acc[:, 1]))


WORKER = textwrap.dedent("""
    import os, sys, json
    import numpy as np
    sys.path.insert(0, %r)
    import torch.distributed as dist
    from msckf_mono_amd import shard
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    per = 3
    ids = shard.trajectory_ids(rank, world, per)
    rng = np.random.default_rng(1234)
    est_all, gt_all = rng.normal(size=(world * per, 3)), rng.normal(size=(world * per, 3))
    seq_all = [g %% 2 for g in range(world * per)]
    acc = shard.ate_local(est_all[ids], gt_all[ids], [seq_all[g] for g in ids], 2)
    ate = shard.ate_allreduce(acc)
    if rank == 0:
        full = shard.ate_local(est_all, gt_all, seq_all, 2)
        ref = np.sqrt(full[:, 0] / full[:, 1])
        print(json.dumps({"ate": ate.tolist(), "ref": ref.tolist(), "ids": ids}))
    dist.barrier()
    dist.destroy_process_group()
""")


def test_world_size_2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29617", str(script)],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert np.allclose(res["ate"], res["ref"], rtol=1e-12)
    assert res["ids"] == [0, 1, 2]
