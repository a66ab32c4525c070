cd /root/repo
O=gpurun_out/r03af; mkdir -p $O
PHASES_B=8 MSCKF_HIP_LIB=/root/repo/msckf_mono_amd/lib_ab/libmsckf_hip_ablate.so python scripts/chol_phases.py > $O/phases.txt 2>&1
grep -E "GRAM|GAIN" $O/phases.txt | cut -c1-300
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest.txt
tail -5 $O/pytest.txt
python scripts/sweep_variants.py --steps 20 --windows 5 "streams=4,streamed=0" "streams=4,streamed=1" "streams=1,streamed=0" > $O/sweep.txt 2>&1
cat $O/sweep.txt | cut -c1-250
