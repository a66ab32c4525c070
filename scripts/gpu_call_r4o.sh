cd /root/repo
O=gpurun_out/r04o; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt; grep -q "smoke ok" $O/smoke.txt || exit 1
echo "== phases"; MSCKF_HIP_LIB=msckf_mono_amd/lib_ab/libmsckf_hip_ablate.so timeout 300 python scripts/chol_phases.py 2>&1 | grep "GRAM\|GAIN" | tee $O/phases_bal.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
V='streams=4,streamed=0 streams=4,streamed=1 streams=1,streamed=0'
run() { tag=$1; shift; env "$@" timeout 600 python scripts/sweep_variants.py --steps 20 --windows 7 $V > $O/sweep_$tag.txt 2>&1; echo "== $tag"; cut -c1-200 $O/sweep_$tag.txt | grep median; }
run p3 MSCKF_HIP_LIB=msckf_mono_amd/lib_ab/libmsckf_hip_p3.so
run new X=1
