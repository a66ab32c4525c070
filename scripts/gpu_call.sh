cd /root/repo
O=gpurun_out/r03ap; mkdir -p $O
PHASES_B=8 MSCKF_HIP_LIB=/root/repo/msckf_mono_amd/lib_ab/libmsckf_hip_ablate.so timeout 120 python scripts/chol_phases.py > $O/phases.txt 2>&1
grep -E "GRAM|GAIN" $O/phases.txt | cut -c1-300
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest.txt
tail -4 $O/pytest.txt
timeout 300 python scripts/sweep_variants.py --steps 20 --windows 5 "streams=4,streamed=0" "streams=4,streamed=1" "streams=1,streamed=0" > $O/sweep.txt 2>&1
cat $O/sweep.txt | cut -c1-250
MSCKF_HIP_FUSED_S=1 timeout 300 python scripts/sweep_variants.py --steps 20 --windows 5 "streams=4,streamed=0" "streams=1,streamed=0" > $O/sweep_unsplit.txt 2>&1
cat $O/sweep_unsplit.txt | cut -c1-250
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p1 -o r -- python /root/repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-early-accept-pass --no-upload-pass --repeats 1 --streams 1 > /tmp/b1.log 2>&1
DB=$(find /tmp/p1 -name "*.db" | head -1)
ROCPD_TAIL=20 python /root/repo/scripts/rocpd_summary.py $DB /root/repo/$O/kernel_stats_tail20.md > /dev/null
head -6 /root/repo/$O/kernel_stats_tail20.md
