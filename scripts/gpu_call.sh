cd /root/repo
O=gpurun_out/r03as; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest.txt
tail -3 $O/pytest.txt
python scripts/sweep_variants.py --steps 20 --windows 5 "streams=4,streamed=0" "streams=4,streamed=1" "streams=1,streamed=0" > $O/sweep.txt 2>&1
cat $O/sweep.txt | cut -c1-250
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/p1 -o r -- python /root/repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-early-accept-pass --no-upload-pass --repeats 1 --streams 1 > /tmp/b1.log 2>&1
DB=$(find /tmp/p1 -name "*.db" | head -1)
ROCPD_TAIL=20 python /root/repo/scripts/rocpd_summary.py $DB /root/repo/$O/kernel_stats_tail20.md > /dev/null
grep gemm /root/repo/$O/kernel_stats_tail20.md
