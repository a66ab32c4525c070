# round 4, call C: the literal route's fast path on the device, the shim executables, cfg4 numbers
cd /root/repo
O=gpurun_out/r04c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_literal.py tests/test_cpp_shim.py tests/test_gpu_vs_reference.py -m gpu -x -q 2>&1 | tail -25 > $O/pytest.txt
tail -6 $O/pytest.txt
for m in 0 1; do
  timeout 600 python bench.py --config cfg4 --steps 4 --warmup 2 --repeats 2 --no-cpu-baseline --no-early-accept-pass --aniso-mode $m > $O/bench_cfg4_mode$m.json 2> $O/bench_cfg4_mode$m.err
done
MSCKF_HIP_LITERAL_ROUTE=1 timeout 900 python bench.py --config cfg4 --trajectories 16 --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline --no-early-accept-pass --no-upload-pass > $O/bench_cfg4_b16_general.json 2> $O/bench_cfg4_b16_general.err
python - <<PY
import json
for f in ["bench_cfg4_mode0", "bench_cfg4_mode1", "bench_cfg4_b16_general"]:
    try:
        j = json.loads(open("$O/%s.json" % f).read().strip().splitlines()[-1])
        print(f, round(j["value"]), round(j["ms_per_step"], 4), {k: round(v, 4) for k, v in j["roofline"]["stage_ms_per_step"].items()}, "ate", j["ate_m"])
    except Exception as e:
        print(f, "failed", e)
PY
