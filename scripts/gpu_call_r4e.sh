cd /root/repo
O=gpurun_out/r04e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_literal.py "tests/test_bench_multirank.py::test_cfg2_single_trajectory_latency_config_runs" -m gpu -x -q 2>&1 | tail -25 > $O/pytest.txt
tail -6 $O/pytest.txt
timeout 600 python bench.py --config cfg2 --steps 20 --warmup 5 > $O/bench_cfg2.json 2> $O/bench_cfg2.err
timeout 600 python bench.py --config cfg4 --steps 10 --warmup 3 --repeats 3 --no-cpu-baseline --no-early-accept-pass > $O/bench_cfg4.json 2> $O/bench_cfg4.err
python - <<PY
import json
j = json.loads(open("$O/bench_cfg2.json").read().strip().splitlines()[-1])
print("cfg2", round(j["value"]), j["latency_us"], j["roofline"]["kernel"], j["roofline"]["frac"], j["roofline"]["stage_ms_per_update"], "cpu", j["cpu_baseline"]["value"], j["cpu_baseline"].get("lean_value"), j["parity"])
j = json.loads(open("$O/bench_cfg4.json").read().strip().splitlines()[-1])
print("cfg4", round(j["value"]), round(j["ms_per_step"], 4), j["repeats"]["values"], j["resident_inputs"], {k: round(x, 4) for k, x in j["roofline"]["stage_ms_per_step"].items()}, j["ate_m"])
PY
python - <<PY
# routes taken by a cfg4 batch in steady state
import sys, numpy as np
sys.path.insert(0, "/root/repo")
from msckf_mono_amd import capi, scenario as sc
import bench
c = dict(bench.CONFIGS["cfg4"]); c["B"] = 32
nfr = 36
trajs = bench.make_trajectories(c, 0, nfr)
bt = capi.Batch(32, 30, 200, 30, capi.F32)
bt.scenario_alloc(nfr, 10)
for b, tr in enumerate(trajs):
    bt.initialize(b, tr.cfg, tr.imu0)
    for f in range(nfr):
        fr = tr.frames[f]
        bt.scenario_set(f, b, tr.imu_for_frame(f), fr["M"], fr["slots"], fr["obs"], 1 if fr["Nw"] == 30 else 0)
bt.scenario_commit()
for f in range(30, nfr):
    bt.run_frames(f, f + 1); bt.sync()
    infos = [bt.literal_info(b) for b in range(32)]
    print(f, "routes", [i["route"] for i in infos].count(1), "fast of 32; min_indep_mlog max", max(i["min_indep_mlog"] for i in infos), "max_dep_mlog min", min(i["max_dep_mlog"] for i in infos))
PY
