# round-4 experiment call: smoke + GPU suite on the tree, then one sweep process per environment variant (the knobs are read at create time)
cd /root/repo
O=gpurun_out/r04m; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt
grep -q "smoke ok" $O/smoke.txt || exit 1
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $O/pytest.txt; tail -3 $O/pytest.txt
grep -q " passed" $O/pytest.txt && ! grep -q failed $O/pytest.txt || exit 1
V='streams=4,streamed=0 streams=4,streamed=1 streams=1,streamed=0'
run() { tag=$1; shift; env "$@" timeout 600 python scripts/sweep_variants.py --steps 20 --windows 7 $V > $O/sweep_$tag.txt 2>&1; echo "== $tag"; cut -c1-200 $O/sweep_$tag.txt | grep median; }
run base X=1
run split16 MSCKF_HIP_FEATURE_SPLIT=16
run pad MSCKF_HIP_CHOL_LDS_PAD=200000
run base2 X=1
