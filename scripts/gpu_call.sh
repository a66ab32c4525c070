cd /root/repo
O=gpurun_out/r03u; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest.txt
tail -5 $O/pytest.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-early-accept-pass > $O/bench4.json 2> $O/bench4.err
python -c "
import json;j=json.loads(open('$O/bench4.json').read().strip().splitlines()[-1]);print(j['value'], j.get('repeats'), j.get('resident_inputs'))"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/p1 -o r -- python /root/repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-early-accept-pass --no-upload-pass --repeats 1 --streams 1 > /tmp/b1.log 2>&1
DB=$(find /tmp/p1 -name "*.db" | head -1)
ROCPD_TAIL=20 python /root/repo/scripts/rocpd_summary.py $DB /root/repo/$O/kernel_stats_tail20.md > /dev/null
cat /root/repo/$O/kernel_stats_tail20.md; tail -1 /tmp/b1.log | cut -c1-200
