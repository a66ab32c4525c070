# round 4, call D: literal fast path at the 30-camera window, cfg4 numbers, thread pinning A/B on the headline
cd /root/repo
O=gpurun_out/r04d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_literal.py tests/test_cpp_shim.py tests/test_gpu_vs_reference.py -m gpu -x -q 2>&1 | tail -25 > $O/pytest.txt
tail -6 $O/pytest.txt
timeout 600 python bench.py --config cfg4 --steps 4 --warmup 2 --repeats 2 --no-cpu-baseline --no-early-accept-pass > $O/bench_cfg4_mode0.json 2> $O/bench_cfg4_mode0.err
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-early-accept-pass > $O/bench_pin.json 2> $O/bench_pin.err
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-early-accept-pass --no-pin > $O/bench_nopin.json 2> $O/bench_nopin.err
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-early-accept-pass > $O/bench_pin2.json 2> $O/bench_pin2.err
lscpu | head -25 > $O/lscpu.txt; cat /sys/class/drm/card*/device/numa_node > $O/numa.txt 2>&1
python - <<PY
import json, numpy as np
for f in ["bench_cfg4_mode0", "bench_pin", "bench_nopin", "bench_pin2"]:
    try:
        j = json.loads(open("$O/%s.json" % f).read().strip().splitlines()[-1])
        v = np.array(j["repeats"]["values"])
        print(f, round(j["value"]), round(j["ms_per_step"], 4), "windows", len(v), "min/med %.3f" % (v.min() / np.median(v)), "med", round(float(np.median(v))), {k: round(x, 4) for k, x in j["roofline"]["stage_ms_per_step"].items()}, j["config"].get("host_affinity"))
    except Exception as e:
        print(f, "failed", e)
PY
