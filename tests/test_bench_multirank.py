"""bench.py's N>1 code path (one process per rank, barrier + max-over-ranks timing, whole-job aggregate, ATE
all-reduce) exercised with 2 ranks on ONE GPU through gloo -- the driver's real multi-GPU run uses RCCL."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_two_ranks_on_one_gpu():
    env = dict(os.environ, BENCH_DIST_BACKEND="gloo", BENCH_DEVICE_OVERRIDE="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29641", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                       # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 4
    assert abs(d["value"] - 2 * 64 * 4 / (d["ms_per_step"] * 4e-3)) / d["value"] < 1e-6
    assert d["roofline"]["bound"] == "mfma" and 0 < d["roofline"]["frac"] < 1
    assert d["ate_m"] < 0.05
