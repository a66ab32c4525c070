# round-4: k_select_diag with the first load level of four rounds batched -- full GPU suite + A/B sweep against the previous library (p3)
cd /root/repo
O=gpurun_out/r04q; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt; grep -q "smoke ok" $O/smoke.txt || exit 1
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
V='streams=4,streamed=0 streams=4,streamed=1 streams=1,streamed=0'
run() { tag=$1; shift; env "$@" timeout 600 python scripts/sweep_variants.py --steps 20 --windows 7 $V > $O/sweep_$tag.txt 2>&1; echo "== $tag"; cut -c1-200 $O/sweep_$tag.txt | grep median; }
run p3 MSCKF_HIP_LIB=msckf_mono_amd/lib_ab/libmsckf_hip_p3.so
run new X=1
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /tmp/pq -o r -- python /root/repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-early-accept-pass --no-upload-pass --repeats 1 --streams 1 > /tmp/bq.log 2>&1; DB=$(find /tmp/pq -name "*.db" | head -1); ROCPD_TAIL=20 python /root/repo/scripts/rocpd_summary.py $DB /root/repo/$O/kernel_stats.md > /dev/null; grep "select_diag\|k_feature" /root/repo/$O/kernel_stats.md
