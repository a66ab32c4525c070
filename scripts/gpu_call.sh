cd /root/repo
O=gpurun_out/r03_x; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest.txt
tail -4 $O/pytest.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
j=json.loads(open("/root/repo/gpurun_out/r03_x/bench.json").read().strip().splitlines()[-1])
print(round(j["value"]), round(j["repeats"]["median"]), round(j["resident_inputs"]["median"]), j["resident_inputs"]["streamed_over_resident"])
r=j["roofline"]; print(r["kernel"], r["frac"], r["executed_frac"], r["kernel_ms_per_step"], r["event_pair_overhead_ms"]); print(j["cpu_baseline"])
PY
