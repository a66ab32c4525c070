# round-4: kernel stats + PMC passes of the other configurations (cfg2: one double filter call by call; cfg4: literal anisotropic route)
cd /root/repo
O=/root/repo/gpurun_out/r04r; mkdir -p $O; rm -f $O/*.md
cd /tmp && export TMPDIR=/tmp
C2="--config cfg2 --steps 20 --warmup 5 --repeats 1"
rocprofv3 --kernel-trace --stats -d /tmp/pc2 -o r -- python /root/repo/bench.py $C2 > /tmp/c2.log 2>&1
for db in $(find /tmp/pc2 -name "*.db"); do echo "== $db" >> $O/kernel_stats_cfg2.md; ROCPD_TAIL=20 python /root/repo/scripts/rocpd_summary.py $db $O/kernel_stats_cfg2.md > /dev/null; done
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pc2a -o r -- python /root/repo/bench.py $C2 > /tmp/c2a.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pc2b -o r -- python /root/repo/bench.py $C2 > /tmp/c2b.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU --kernel-trace -d /tmp/pc2c -o r -- python /root/repo/bench.py $C2 > /tmp/c2c.log 2>&1
mkdir -p /tmp/pmcout; PMC_TAG=r04r_cfg2 python /root/repo/scripts/rocpd_pmc.py /tmp/pmcout/pmc_cfg2.md $(find /tmp/pc2a /tmp/pc2b /tmp/pc2c -name "*.db") > /dev/null 2>&1; cp /tmp/pmcout/pmc_cfg2.md $O/ 2>/dev/null
C4="--config cfg4 --steps 6 --warmup 2 --repeats 1 --no-cpu-baseline --no-early-accept-pass --no-upload-pass --streams 1"
rocprofv3 --kernel-trace --stats -d /tmp/pc4 -o r -- python /root/repo/bench.py $C4 > /tmp/c4.log 2>&1
DB=$(find /tmp/pc4 -name "*.db" | head -1); ROCPD_TAIL=6 python /root/repo/scripts/rocpd_summary.py $DB $O/kernel_stats_cfg4_literal.md > /dev/null
head -14 $O/kernel_stats_cfg2.md; head -8 $O/kernel_stats_cfg4_literal.md; ls -la $O; head -12 $O/pmc_cfg2.md
