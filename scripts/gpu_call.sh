cd /root/repo
O=gpurun_out/r03m; mkdir -p $O
MSCKF_HIP_GRAM_PARTS=4 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x 2>&1 | tail -5 > $O/pytest_p4.txt
timeout 300 python scripts/sweep_variants.py --steps 20 --windows 5 "streams=4,streamed=0" "streams=1,streamed=0" > $O/sweep_p3.txt 2>&1
MSCKF_HIP_GRAM_PARTS=4 timeout 300 python scripts/sweep_variants.py --steps 20 --windows 5 "streams=4,streamed=0" "streams=1,streamed=0" > $O/sweep_p4.txt 2>&1
cd /tmp && export TMPDIR=/tmp
COMMON="--steps 10 --warmup 3 --no-cpu-baseline --no-early-accept-pass --no-upload-pass --repeats 1 --streams 1"
MSCKF_HIP_GRAM_PARTS=4 rocprofv3 --kernel-trace --stats -d /tmp/p1 -o r -- python /root/repo/bench.py $COMMON > /tmp/b1.log 2>&1
DB=$(find /tmp/p1 -name "*.db" | head -1)
ROCPD_TAIL=20 python /root/repo/scripts/rocpd_summary.py $DB /root/repo/$O/kernel_stats_tail20_p4.md > /dev/null
cd /root/repo
tail -3 $O/pytest_p4.txt; cat $O/sweep_p3.txt $O/sweep_p4.txt | cut -c1-200; head -8 $O/kernel_stats_tail20_p4.md
