cd /root/repo
O=gpurun_out/r04i; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q --deselect tests/test_bench_multirank.py::test_cfg4_monte_carlo_mode_reports_per_sequence_ate 2>&1 | tail -15 > $O/pytest.txt
tail -5 $O/pytest.txt
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-early-accept-pass > $O/bench.json 2> $O/bench.err
(cd .ab_old && timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-early-accept-pass > ../$O/bench_old.json 2> ../$O/bench_old.err)
MSCKF_HIP_LITERAL_TIMERS=1 timeout 600 python bench.py --config cfg4 --steps 10 --warmup 3 --repeats 3 --no-cpu-baseline --no-early-accept-pass > $O/bench_cfg4.json 2> $O/bench_cfg4.err
python - <<PY
import json, numpy as np
for f in ["bench", "bench_old", "bench_cfg4"]:
    j = json.loads(open("$O/%s.json" % f).read().strip().splitlines()[-1])
    v = np.array(j["repeats"]["values"])
    print(f, round(j["value"]), round(j["ms_per_step"], 4), "median", round(float(np.median(v))), "min/med %.3f" % (v.min() / np.median(v)), {k: round(x, 4) for k, x in j["roofline"]["stage_ms_per_step"].items()}, "frac", j["roofline"].get("frac"))
PY
