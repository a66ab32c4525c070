cd /root/repo
O=gpurun_out/r03q; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -30 > $O/pytest.txt
MSCKF_HIP_LIB=/root/repo/msckf_mono_amd/lib_ab/libmsckf_hip_ablate.so timeout 300 python scripts/chol_phases.py 2>&1 | grep -v "^k_gram" > $O/phases.txt
timeout 700 python scripts/sweep_variants.py --steps 20 --windows 5 "streams=4,streamed=0" "streams=4" "streams=1,streamed=0" > $O/sweep.txt 2>&1
MSCKF_HIP_FUSED_S=0 timeout 300 python scripts/sweep_variants.py --steps 20 --windows 5 "streams=4,streamed=0" "streams=1,streamed=0" > $O/sweep_unfused.txt 2>&1
cd /tmp && export TMPDIR=/tmp
COMMON="--steps 10 --warmup 3 --no-cpu-baseline --no-early-accept-pass --no-upload-pass --repeats 1 --streams 1"
rocprofv3 --kernel-trace --stats -d /tmp/p1 -o r -- python /root/repo/bench.py $COMMON > /tmp/b1.log 2>&1
DB=$(find /tmp/p1 -name "*.db" | head -1)
ROCPD_TAIL=20 python /root/repo/scripts/rocpd_summary.py $DB /root/repo/$O/kernel_stats_tail20.md > /dev/null
cd /root/repo
tail -4 $O/pytest.txt; cat $O/phases.txt | cut -c1-300; cat $O/sweep.txt $O/sweep_unfused.txt | cut -c1-200; head -12 $O/kernel_stats_tail20.md
