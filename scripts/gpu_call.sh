cd /root/repo
O=gpurun_out/r03ak; mkdir -p $O
for dbg in 0 1; do CHOL_DBG=$dbg PHASES_B=8 MSCKF_HIP_LIB=/root/repo/msckf_mono_amd/lib_ab/libmsckf_hip_ablate.so python scripts/chol_phases.py > $O/phases_$dbg.txt 2>&1; echo dbg=$dbg; grep -E "GRAM|GAIN" $O/phases_$dbg.txt | cut -c1-300; done
