cd /root/repo
O=gpurun_out/r03_w; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest.txt
tail -4 $O/pytest.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
PHASES_B=8 MSCKF_HIP_LIB=/root/repo/msckf_mono_amd/lib_ab/libmsckf_hip_ablate.so timeout 120 python scripts/chol_phases.py > $O/phases.txt 2>&1
grep -E "GRAM|GAIN|k_propagate" $O/phases.txt | cut -c1-300
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --config cfg4 --steps 20 --warmup 5 --no-cpu-baseline --no-early-accept-pass > $O/bench_cfg4.json 2>/dev/null
python bench.py --config cfg5 --steps 10 --warmup 3 --no-cpu-baseline --no-early-accept-pass > $O/bench_cfg5.json 2>/dev/null
python - <<'PY'
import json
for f in ["bench","bench_cfg4","bench_cfg5"]:
    j=json.loads(open(f"/root/repo/gpurun_out/r03_w/{f}.json").read().strip().splitlines()[-1])
    print(f, round(j["value"]), round(j.get("repeats",{}).get("median",0)), round(j.get("resident_inputs",{}).get("median",0)), j.get("resident_inputs",{}).get("streamed_over_resident"))
PY
python scripts/sweep_variants.py --steps 20 --windows 5 "streams=4,streamed=0" "streams=4,streamed=1" "streams=1,streamed=0" > $O/sweep.txt 2>&1
cat $O/sweep.txt | cut -c1-250
TAG=r03_w PMC_COMMIT=6696d04+ bash scripts/profile_round.sh > $O/profile_round.log 2>&1
head -12 $O/kernel_stats_tail20.md
