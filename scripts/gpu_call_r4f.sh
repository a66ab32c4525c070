cd /root/repo
O=gpurun_out/r04f; mkdir -p $O
for r in 2 1; do
MSCKF_HIP_LITERAL_ROUTE=$r timeout 600 python bench.py --config cfg4 --trajectories 8 --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline --no-early-accept-pass --no-upload-pass > $O/bench_cfg4_b8_route$r.json 2> $O/bench_cfg4_b8_route$r.err
done
python - <<PY
import json
for r in (2, 1):
    j = json.loads(open("$O/bench_cfg4_b8_route%d.json" % r).read().strip().splitlines()[-1])
    print("route", r, round(j["value"]), {k: round(x, 4) for k, x in j["roofline"]["stage_ms_per_step"].items()}, j["ate_m"])
PY
python - <<PY
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
bt.run_frames(0, 30); bt.sync()
for f in range(30, nfr):
    bt.run_frames(f, f + 1); bt.sync()
    infos = [bt.literal_info(b) for b in range(32)]
    print(f, "fast", [i["route"] for i in infos].count(1), "of 32; general on", [b for b, i in enumerate(infos) if i["route"] != 1], "min_indep_mlog max", max(i["min_indep_mlog"] for i in infos), "max_dep_mlog min", min(i["max_dep_mlog"] for i in infos), infos[0])
PY
