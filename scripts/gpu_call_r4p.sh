cd /root/repo
O=gpurun_out/r04p; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_literal.py tests/test_gpu_vs_reference.py -m gpu -x -q 2>&1 | tail -8 > $O/pytest.txt
tail -4 $O/pytest.txt
MSCKF_HIP_LITERAL_TIMERS=1 python - <<PY 2>&1 | tail -12
import sys, numpy as np, time
sys.path.insert(0, "/root/repo")
from msckf_mono_amd import capi, scenario as sc
import bench
for B in (8, 128):
    c = dict(bench.CONFIGS["cfg4"]); c["B"] = B
    nfr = 36
    trajs = bench.make_trajectories(c, 0, nfr)
    bt = capi.Batch(B, 30, 200, 30, capi.F32)
    bt.scenario_alloc(nfr, 10)
    for b, tr in enumerate(trajs):
        bt.initialize(b, tr.cfg, tr.imu0)
        for f in range(nfr):
            fr = tr.frames[f]
            bt.scenario_set(f, b, tr.imu_for_frame(f), fr["M"], fr["slots"], fr["obs"], 1 if fr["Nw"] == 30 else 0)
    bt.scenario_commit()
    bt.run_frames(0, 32); bt.sync()
    for f in range(32, 36):
        t0 = time.perf_counter(); bt.run_frames(f, f + 1); bt.sync(); dt = time.perf_counter() - t0
        print("B", B, "frame", f, "ms %.2f" % (dt * 1e3))
    print(bt.literal_info(0)); print(bt.literal_info(B - 1))
    bt.close()
PY
timeout 900 python bench.py --config cfg4 --steps 10 --warmup 3 --repeats 3 --no-cpu-baseline --no-early-accept-pass > $O/bench_cfg4.json 2> $O/bench_cfg4.err
python - <<PY
import json
j = json.loads(open("$O/bench_cfg4.json").read().strip().splitlines()[-1])
print("cfg4", round(j["value"]), round(j["ms_per_step"], 4), j["repeats"]["values"], {k: round(x, 4) for k, x in j["roofline"]["stage_ms_per_step"].items()}, j["ate_m"])
PY
