cd /root/repo
O=gpurun_out/r03ag; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest.txt
tail -3 $O/pytest.txt
for i in 1 2; do python bench.py --config cfg5 --steps 10 --warmup 3 --no-cpu-baseline --no-early-accept-pass > $O/bench_cfg5_$i.json 2>$O/cfg5.err
python -c "
import json;j=json.loads(open('$O/bench_cfg5_$i.json').read().strip().splitlines()[-1]);print('cfg5', j['value'], j.get('repeats',{}).get('median'), j.get('resident_inputs',{}).get('median'))"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/p1 -o r -- python /root/repo/bench.py --config cfg5 --steps 6 --warmup 2 --no-cpu-baseline --no-early-accept-pass --no-upload-pass --repeats 1 --streams 1 > /tmp/b1.log 2>&1
DB=$(find /tmp/p1 -name "*.db" | head -1)
ROCPD_TAIL=12 python /root/repo/scripts/rocpd_summary.py $DB /root/repo/$O/kernel_stats_cfg5.md > /dev/null
head -14 /root/repo/$O/kernel_stats_cfg5.md
