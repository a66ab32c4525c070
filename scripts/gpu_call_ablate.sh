cd /root/repo
O=gpurun_out/r03j; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
MSCKF_HIP_LIB=/root/repo/msckf_mono_amd/lib_ab/libmsckf_hip_ablate.so rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d /tmp/pf -o r -- python /root/repo/scripts/feat_insts.py > /root/repo/$O/feat_times.txt 2>&1
DB=$(find /tmp/pf -name "*.db" | head -1)
python /root/repo/scripts/feat_insts.py --dump $DB > /root/repo/$O/feat_ablation.md 2>&1
cd /root/repo
cat $O/feat_ablation.md; grep -v "^W2026" $O/feat_times.txt | tail -14
