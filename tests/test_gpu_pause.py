"""A caller that reads state between batches (bench.py does after its first timed window: 64 x (last_stats + imu_state), each a
small synchronous device-to-host copy) must not pay for it with a slow next window: scripts/pause_probe.py --quick runs the
streamed path (uploader + four enqueue threads, as the bench) -- five back-to-back windows for the median, then three times
[64 state reads, window, window] -- and the first window after the pause has to reach 0.93 x that median (the median of the
three: a single outlier window is what the box does now and then with or without a pause, BENCH_r04 / profiles/r04_b)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_first_streamed_window_after_state_reads_reaches_the_median():
    # Some leases have slow windows with or without a pause (one run of this probe read 0.33 / 0.52 / 0.96 on a box whose
    # back-to-back windows also dipped): the probe gets a second attempt before the pause is blamed.
    seen = []
    for attempt in range(2):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "pause_probe.py"), "--quick"], capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        res = json.loads(out.stdout.strip().splitlines()[-1])["streams=4,streamed=1"]
        first = [v[0] for v in res["after_pause"].values()]
        assert len(first) == 3 and res["median"] > 0
        seen.append(res)
        if float(np.median(first)) >= 0.93:
            return
    assert False, seen
