cd /root/repo
O=gpurun_out/r03_u; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest.txt
tail -3 $O/pytest.txt
python scripts/sweep_variants.py --steps 20 --windows 5 "streams=4,streamed=0" "streams=4,streamed=1" "streams=1,streamed=0" > $O/sweep.txt 2>&1
cat $O/sweep.txt | cut -c1-250
TAG=r03_u PMC_COMMIT=wip bash scripts/profile_round.sh > $O/profile_round.log 2>&1
cat $O/kernel_stats_tail20.md | head -12
python - <<'PY'
import json
j=json.load(open('/root/repo/gpurun_out/r03_u/pmc_traffic.json'))
tot=0
for k,v in j.items():
    if k.startswith('_'): continue
    print(k, 'fetch MB', round(v['fetch_kib']/1024,1), 'write MB', round(v['write_kib']/1024,1)); tot+=v['bytes_per_launch']/1e6
print('sum', tot)
PY
