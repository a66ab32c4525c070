# round 4, call A: the literal anisotropic route + the rendezvous-free prune on the GPU
cd /root/repo
O=gpurun_out/r04a; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_bench_multirank.py::test_cfg4_monte_carlo_mode_reports_per_sequence_ate 2>&1 | tail -25 > $O/pytest.txt
tail -8 $O/pytest.txt
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
for m in 0 1; do
  timeout 600 python bench.py --config cfg4 --trajectories 8 --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline --no-early-accept-pass --no-upload-pass --aniso-mode $m > $O/bench_cfg4_b8_mode$m.json 2> $O/bench_cfg4_b8_mode$m.err
done
python - <<PY
import json
for f in ["bench", "bench_cfg4_b8_mode0", "bench_cfg4_b8_mode1"]:
    try:
        j = json.loads(open("$O/%s.json" % f).read().strip().splitlines()[-1])
        print(f, round(j["value"]), j["ms_per_step"], {k: round(v, 4) for k, v in j["roofline"]["stage_ms_per_step"].items()}, j.get("repeats", {}).get("values"))
    except Exception as e:
        print(f, "failed", e)
PY
