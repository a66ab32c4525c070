cd /root/repo
O=gpurun_out/r04h; mkdir -p $O
MSCKF_HIP_LITERAL_TIMERS=1 python - <<PY 2>&1 | tail -4
import sys, numpy as np
sys.path.insert(0, "/root/repo")
from msckf_mono_amd import capi, scenario as sc
import bench
c = dict(bench.CONFIGS["cfg4"]); c["B"] = 8
nfr = 33
trajs = bench.make_trajectories(c, 0, nfr)
bt = capi.Batch(8, 30, 200, 30, capi.F32)
bt.scenario_alloc(nfr, 10)
for b, tr in enumerate(trajs):
    bt.initialize(b, tr.cfg, tr.imu0)
    for f in range(nfr):
        fr = tr.frames[f]
        bt.scenario_set(f, b, tr.imu_for_frame(f), fr["M"], fr["slots"], fr["obs"], 1 if fr["Nw"] == 30 else 0)
bt.scenario_commit()
bt.run_frames(0, 32); bt.sync()
bt.run_frames(32, 33); bt.sync()
for b in (0, 2):
    print(bt.literal_info(b))
PY
timeout 1500 python -m pytest tests/test_gpu_vs_reference.py tests/test_gpu_literal.py tests/test_gpu_parity.py tests/test_cpp_shim.py tests/test_bench_multirank.py -m gpu -x -q --deselect tests/test_bench_multirank.py::test_cfg4_monte_carlo_mode_reports_per_sequence_ate 2>&1 | tail -15 > $O/pytest.txt
tail -5 $O/pytest.txt
timeout 600 python bench.py --config cfg2 --steps 20 --warmup 5 > $O/bench_cfg2.json 2> $O/bench_cfg2.err
timeout 600 python bench.py --config cfg4 --steps 10 --warmup 3 --repeats 3 --no-cpu-baseline --no-early-accept-pass > $O/bench_cfg4.json 2> $O/bench_cfg4.err
python - <<PY
import json
j = json.loads(open("$O/bench_cfg2.json").read().strip().splitlines()[-1])
print("cfg2", round(j["value"]), j["latency_us"]["stage_mean"], j["repeats"]["values"], "cpu", j["cpu_baseline"]["value"])
j = json.loads(open("$O/bench_cfg4.json").read().strip().splitlines()[-1])
print("cfg4", round(j["value"]), round(j["ms_per_step"], 4), j["repeats"]["values"], {k: round(x, 4) for k, x in j["roofline"]["stage_ms_per_step"].items()}, j["ate_m"])
PY
