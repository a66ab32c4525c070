# Collect the round's evidence on the GPU box (run through gpurun from the repo root):
#   bench.py JSON, rocprofv3 kernel trace of bench.py (single stream, resident inputs), PMC passes (traffic, instruction mix,
#   issue / wait breakdown).  Outputs land in gpurun_out/$TAG; the summaries to be judged are then copied into profiles/ by hand.
TAG=${TAG:-r03}
OUT=/root/repo/gpurun_out/$TAG
set -x
mkdir -p $OUT
export PMC_TAG=$TAG PMC_COMMIT=${PMC_COMMIT:-unknown}
export PMC_CSRC_HASH=$(python -c 'import bench; print(bench.csrc_hash())')
cd /tmp && export TMPDIR=/tmp
COMMON="--steps 10 --warmup 3 --no-cpu-baseline --no-early-accept-pass --no-upload-pass --repeats 1 --streams 1"
rocprofv3 --kernel-trace --stats -d /tmp/p1 -o r -- python /root/repo/bench.py $COMMON > /tmp/b1.log 2>&1
DB=$(find /tmp/p1 -name "*.db" | head -1); tail -1 /tmp/b1.log > $OUT/bench_profiled.json
ROCPD_TAIL=20 python /root/repo/scripts/rocpd_summary.py $DB $OUT/kernel_stats_tail20.md > /dev/null
python /root/repo/scripts/rocpd_summary.py $DB $OUT/kernel_stats_all.md > /dev/null
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p2 -o r -- python /root/repo/bench.py $COMMON > /tmp/b2.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/p3 -o r -- python /root/repo/bench.py $COMMON > /tmp/b3.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD --kernel-trace -d /tmp/p4 -o r -- python /root/repo/bench.py $COMMON > /tmp/b4.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_ACTIVE_INST_LDS --kernel-trace -d /tmp/p5 -o r -- python /root/repo/bench.py $COMMON > /tmp/b5.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE --kernel-trace -d /tmp/p6 -o r -- python /root/repo/bench.py $COMMON > /tmp/b6.log 2>&1
tail -2 /tmp/b5.log | cut -c1-300
python /root/repo/scripts/rocpd_pmc.py $OUT/pmc.md $(find /tmp/p2 /tmp/p3 /tmp/p4 /tmp/p5 /tmp/p6 -name "*.db") > /dev/null
ls -la $OUT
