set -x
mkdir -p gpurun_out/r1f
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py > gpurun_out/r1f/bench.json 2> gpurun_out/r1f/bench.err; tail -c 600 gpurun_out/r1f/bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/p1 -o r -- python /root/repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-early-accept-pass --streams 1 > /tmp/b1.log 2>&1
DB=$(find /tmp/p1 -name "*.db" | head -1); tail -1 /tmp/b1.log > /root/repo/gpurun_out/r1f/bench_profiled.json
ROCPD_TAIL=10 python /root/repo/scripts/rocpd_summary.py $DB /root/repo/gpurun_out/r1f/kernel_stats_tail10.md > /dev/null
python /root/repo/scripts/rocpd_summary.py $DB /root/repo/gpurun_out/r1f/kernel_stats_all.md > /dev/null
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p2 -o r -- python /root/repo/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-early-accept-pass --streams 1 > /tmp/b2.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/p3 -o r -- python /root/repo/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-early-accept-pass --streams 1 > /tmp/b3.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace -d /tmp/p4 -o r -- python /root/repo/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-early-accept-pass --streams 1 > /tmp/b4.log 2>&1
tail -2 /tmp/b4.log | cut -c1-300
python /root/repo/scripts/rocpd_pmc.py /root/repo/gpurun_out/r1f/pmc.md $(find /tmp/p2 /tmp/p3 /tmp/p4 -name "*.db") > /dev/null
ls -la /root/repo/gpurun_out/r1f
