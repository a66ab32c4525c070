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
