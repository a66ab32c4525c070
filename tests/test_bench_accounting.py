"""bench.py's algorithmic-work formulas against the expectation table of SURVEY.md section 8d."""
import numpy as np

import bench


def _expected_Ms(N, F):
    # M_j cycling 3..N-1
    return np.array([3 + (j % (N - 3)) for j in range(F)])


def test_cfg3_flops_and_bytes_match_survey():
    fl = bench.alg_flops_update(_expected_Ms(30, 200), 30)
    total = sum(v for k, v in fl.items() if k != "gram")   # gram = the compression stage as built, not additive
    assert abs(total - 7.36e8) / 7.36e8 < 0.03, total
    assert abs(fl["compress"] - 3.65e8) / 3.65e8 < 0.03
    assert abs(fl["kalman"] - 1.21e8) / 1.21e8 < 0.03
    assert abs(fl["feature"] - 2.49e8) / 2.49e8 < 0.03
    by = bench.alg_bytes_update(_expected_Ms(30, 200), 30)
    assert abs(by - 3.31e5) / 3.31e5 < 0.03


def test_cfg2_flops():
    fl = bench.alg_flops_update(_expected_Ms(10, 50), 10)
    assert abs(sum(v for k, v in fl.items() if k != "gram") - 1.31e7) / 1.31e7 < 0.05


def test_executed_model_is_far_below_the_reference_algorithm():
    """The as-built FLOP model (block-sparse gate, information-form compression, square-root gain) that bench.py prices
    `executed_frac` with: ~4 D^3 for the Kalman set, an order of magnitude below the dense gate for the per-track stage."""
    Ms = _expected_Ms(30, 200)
    ex, fl = bench.executed_flops_update(Ms, 30), bench.alg_flops_update(Ms, 30)
    D = 195.0
    assert 3.5 * D ** 3 < ex["kalman"] < 4.5 * D ** 3 and ex["kalman"] < 0.3 * fl["kalman"]
    assert ex["feature"] < 0.15 * fl["feature"]
    assert abs(ex["compress_stage1"] - fl["gram"]) < 1e-6 * fl["gram"]
    assert sum(ex.values()) < 0.12 * sum(v for k, v in fl.items() if k != "gram")


def test_gpus_flag_self_launches_ranks_without_torchrun():
    """`python bench.py --gpus 2` outside torch.distributed.run must come up as TWO ranks (re-exec under
    torch.distributed.run, rank -> device, process group) and print n_gpus 2; here with gloo and the rendezvous-only hook
    (no GPU in this container).  With the default backend it must refuse when the node has fewer devices."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                         capture_output=True, text=True, timeout=300, env=dict(env, BENCH_RENDEZVOUS_ONLY="1", BENCH_DIST_BACKEND="gloo"))
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == [0, 1]
    import torch
    if torch.cuda.device_count() < 2:
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300, env=env)
        assert out.returncode != 0 and "--gpus 2 requested" in (out.stderr + out.stdout)
