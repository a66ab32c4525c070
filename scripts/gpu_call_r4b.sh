# round 4, call B: A/B of the round-3 library (.ab_old, commit 160938b) against this tree on the same box
cd /root/repo
O=gpurun_out/r04b; mkdir -p $O
rocm-smi --showclocks --showperflevel --showpower 2>/dev/null | head -30 > $O/smi.txt
(cd .ab_old && timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-early-accept-pass > ../$O/bench_old.json 2> ../$O/bench_old.err)
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-early-accept-pass > $O/bench_new.json 2> $O/bench_new.err
(cd .ab_old && timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-early-accept-pass > ../$O/bench_old2.json 2> ../$O/bench_old2.err)
timeout 600 python bench.py --config cfg4 --trajectories 8 --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline --no-early-accept-pass --no-upload-pass > $O/bench_cfg4_b8.json 2> $O/bench_cfg4_b8.err
python - <<PY
import json
for f in ["bench_old", "bench_new", "bench_old2", "bench_cfg4_b8"]:
    try:
        j = json.loads(open("$O/%s.json" % f).read().strip().splitlines()[-1])
        print(f, round(j["value"]), round(j["ms_per_step"], 4), {k: round(v, 4) for k, v in j["roofline"]["stage_ms_per_step"].items()})
    except Exception as e:
        print(f, "failed", e)
PY
