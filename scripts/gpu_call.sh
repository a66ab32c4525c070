cd /root/repo
O=gpurun_out/r03ar; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest.txt
tail -4 $O/pytest.txt
python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
j=json.loads(open("/root/repo/gpurun_out/r03ar/bench.json").read().strip().splitlines()[-1])
print(round(j["value"]), round(j["repeats"]["median"]), round(j["resident_inputs"]["median"]))
r=j["roofline"]; print(r["kernel"], r["kernel_ms_per_step"], r["event_pair_overhead_ms"], r["stage_ms_per_step"]); print(r["stage_ms_per_step_raw_event_pairs"])
PY
python scripts/sweep_variants.py --steps 20 --windows 5 "streams=4,streamed=0" "streams=4,streamed=1" "streams=1,streamed=0" > $O/sweep.txt 2>&1
cat $O/sweep.txt | cut -c1-250
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/p1 -o r -- python /root/repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-early-accept-pass --no-upload-pass --repeats 1 --streams 1 > /tmp/b1.log 2>&1
DB=$(find /tmp/p1 -name "*.db" | head -1)
ROCPD_TAIL=20 python /root/repo/scripts/rocpd_summary.py $DB /root/repo/$O/kernel_stats_tail20.md > /dev/null
head -11 /root/repo/$O/kernel_stats_tail20.md
