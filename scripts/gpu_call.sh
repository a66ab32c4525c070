# One gpurun call, parametrised (from the repo root on the GPU box):
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'TAG=r05_a STEPS="lit_tests lit_timers cfg4" bash scripts/gpu_call.sh'
# STEPS (any subset, run in this order):
#   tests        the whole GPU suite (pytest -m gpu)            lit_tests   tests/test_gpu_literal.py + test_gpu_vs_reference.py only
#   pytest:<args> pytest with the given arguments (one token: use commas for spaces)
#   smoke        __graft_entry__.smoke()
#   bench        default bench.py (cfg3, the driver's line)     cfgs        cfg2, cfg4 (literal + whitened), cfg5 as their own lines
#   cfg4         bench.py --config cfg4 (literal route) only    lit_timers  phase timers of k_literal at B = 8 and 128 (cfg4 geometry), the
#                previous library (msckf_mono_amd/lib_ab/libmsckf_hip_prev.so, when there) first; with lit_timers_all also per frame the
#                slowest trajectory of every phase and the step timers of every trajectory with kept handed-through rows
#   sweep        scripts/sweep_variants.py (streams / streamed) profile     scripts/profile_round.sh (kernel stats + PMC passes of cfg3)
#   lit_profile  rocprofv3 kernel stats + PMC passes of bench.py --config cfg4 (literal route)    lit_stats  the kernel stats alone (all launches + last six)
#   cfg2_profile rocprofv3 kernel stats of bench.py --config cfg2
#   ab_env       bench.py (cfg3, short) with AB_ENV set against the default, alternating on the same lease (AB_ARGS: extra bench.py flags)
#   pause        scripts/pause_probe.py: the first window after a pause (state reads, sleeps) against the median, streamed / resident
# Outputs land in gpurun_out/$TAG; copy what is to be judged into profiles/ by hand (profiles/README.md lists them).
cd /root/repo
TAG=${TAG:-round}
STEPS=${STEPS:-"tests smoke bench cfgs sweep profile"}
O=gpurun_out/$TAG; mkdir -p $O
has() { case " $STEPS " in *" $1 "*) return 0;; esac; return 1; }
summ() { python - "$@" <<PY
import json, sys
for f in sys.argv[1:]:
    try:
        j = json.loads(open("$O/%s.json" % f).read().strip().splitlines()[-1])
        r = j.get("roofline", {})
        print(f, round(j["value"]), "ms/step", round(j["ms_per_step"], 4), "median", round(j.get("repeats", {}).get("median", 0)), "min", round(j.get("repeats", {}).get("min", 0)), "kernel", r.get("kernel"), "frac", r.get("frac"), {k: round(x, 4) for k, x in (r.get("stage_ms_per_step") or {}).items()})
    except Exception as e:
        print(f, "failed", e)
PY
}
for s in $STEPS; do case $s in pytest:*) a=${s#pytest:}; timeout 1800 python -m pytest ${a//,/ } 2>&1 | tail -15 > $O/pytest_sel.txt; tail -6 $O/pytest_sel.txt;; esac; done
if has tests; then timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest.txt; tail -4 $O/pytest.txt; fi
if has lit_tests; then timeout 1500 python -m pytest tests/test_gpu_literal.py tests/test_gpu_vs_reference.py -m gpu -x -q 2>&1 | tail -12 > $O/pytest_lit.txt; tail -6 $O/pytest_lit.txt; fi
if has smoke; then python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1; fi
if has lit_timers; then
  # (with msckf_mono_amd/lib_ab/libmsckf_hip_prev.so present: the previous library first, on the same lease -- leases differ by up to 1.6x)
  P=msckf_mono_amd/lib_ab/libmsckf_hip_prev.so
  if [ -f $P ]; then MSCKF_HIP_LIB=$P MSCKF_HIP_LITERAL_TIMERS=1 python scripts/lit_timers.py 2>&1 | grep "k_literal b=0" | sed 's/^/prev /' | tee $O/lit_timers_prev.txt; fi
  MSCKF_HIP_LITERAL_TIMERS=1 python scripts/lit_timers.py 2>&1 | tail -14 | tee $O/lit_timers.txt
  if has lit_timers_all; then MSCKF_HIP_LITERAL_TIMERS=1 python scripts/lit_timers.py --all > $O/lit_timers_all.txt 2>&1; fi
fi
if has pause; then python scripts/pause_probe.py 2>&1 | tail -6 | tee $O/pause_probe.txt; fi
if has ab_env; then   # A/B inside one lease: AB_ENV="VAR=val" (the alternative) against the default, alternating, twice each
  BA="--steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-early-accept-pass --repeats 6 ${AB_ARGS:-}"
  for i in 1 2; do
    env $AB_ENV python bench.py $BA > $O/bench_alt$i.json 2> $O/bench_alt$i.err; summ bench_alt$i
    python bench.py $BA > $O/bench_def$i.json 2> $O/bench_def$i.err; summ bench_def$i
  done
fi
if has feat_pmc; then   # instruction / issue counters of the per-track kernel only (two PMC passes), default library and with AB_ENV set
  cd /tmp && export TMPDIR=/tmp
  CM="--steps 6 --warmup 2 --no-cpu-baseline --no-early-accept-pass --no-upload-pass --no-other-configs --repeats 1 --streams 1"
  for v in def alt; do
    E=""; if [ $v = alt ]; then E="$AB_ENV"; fi
    env $E rocprofv3 --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES --kernel-trace -d /tmp/f1$v -o r -- python /root/repo/bench.py $CM > /tmp/f1$v.log 2>&1
    env $E rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE --kernel-trace -d /tmp/f2$v -o r -- python /root/repo/bench.py $CM > /tmp/f2$v.log 2>&1
    rm -f /root/repo/$O/feat_pmc_$v.md
    python /root/repo/scripts/rocpd_pmc.py /root/repo/$O/feat_pmc_$v.md $(find /tmp/f1$v /tmp/f2$v -name "*.db") | grep -i "k_feature" | cut -c1-200
  done
  cd /root/repo
fi
if has bench; then python bench.py > $O/bench.json 2> $O/bench.err; summ bench; fi
if has cfg4; then python bench.py --config cfg4 --steps 10 --warmup 3 --no-cpu-baseline --no-early-accept-pass --no-other-configs > $O/bench_cfg4.json 2> $O/bench_cfg4.err; summ bench_cfg4; fi
if has cfgs; then
  python bench.py --config cfg2 --steps 20 --warmup 5 --no-other-configs > $O/bench_cfg2.json 2>/dev/null
  python bench.py --config cfg4 --steps 10 --warmup 3 --no-cpu-baseline --no-early-accept-pass --no-other-configs > $O/bench_cfg4.json 2>/dev/null
  python bench.py --config cfg4 --steps 20 --warmup 5 --no-cpu-baseline --no-early-accept-pass --no-other-configs --aniso-mode 1 > $O/bench_cfg4_whitened.json 2>/dev/null
  python bench.py --config cfg5 --steps 10 --warmup 3 --no-cpu-baseline --no-early-accept-pass --no-other-configs > $O/bench_cfg5.json 2>/dev/null
  summ bench_cfg2 bench_cfg4 bench_cfg4_whitened bench_cfg5
fi
if has sweep; then python scripts/sweep_variants.py --steps 20 --windows 5 "streams=4,streamed=0" "streams=4,streamed=1" "streams=1,streamed=0" > $O/sweep.txt 2>&1; cut -c1-250 $O/sweep.txt; fi
if has profile; then TAG=$TAG PMC_COMMIT=${PMC_COMMIT:-unknown} bash scripts/profile_round.sh > $O/profile_round.log 2>&1; head -12 $O/kernel_stats_tail20.md; fi
if has lit_stats; then    # kernel trace of cfg4 on the literal route only (no PMC passes): all launches and the last six (steady state) per kernel
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats -d /tmp/q0 -o r -- python /root/repo/bench.py --config cfg4 --steps 6 --warmup 2 --no-cpu-baseline --no-early-accept-pass --no-upload-pass --no-other-configs --repeats 1 --streams 1 > /tmp/c0.log 2>&1
  python /root/repo/scripts/rocpd_summary.py $(find /tmp/q0 -name "*.db" | head -1) /root/repo/$O/kernel_stats_cfg4_literal.md > /dev/null
  ROCPD_TAIL=6 python /root/repo/scripts/rocpd_summary.py $(find /tmp/q0 -name "*.db" | head -1) /root/repo/$O/kernel_stats_cfg4_literal_tail6.md > /dev/null
  head -14 /root/repo/$O/kernel_stats_cfg4_literal_tail6.md
  cd /root/repo
fi
if has lit_profile || has cfg2_profile; then
  cd /tmp && export TMPDIR=/tmp
  if has lit_profile; then
    C4="--config cfg4 --steps 6 --warmup 2 --no-cpu-baseline --no-early-accept-pass --no-upload-pass --no-other-configs --repeats 1 --streams 1"
    rocprofv3 --kernel-trace --stats -d /tmp/q1 -o r -- python /root/repo/bench.py $C4 > /tmp/c1.log 2>&1
    python /root/repo/scripts/rocpd_summary.py $(find /tmp/q1 -name "*.db" | head -1) /root/repo/$O/kernel_stats_cfg4_literal.md > /dev/null
    ROCPD_TAIL=6 python /root/repo/scripts/rocpd_summary.py $(find /tmp/q1 -name "*.db" | head -1) /root/repo/$O/kernel_stats_cfg4_literal_tail6.md > /dev/null
    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/q2 -o r -- python /root/repo/bench.py $C4 > /tmp/c2.log 2>&1
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/q3 -o r -- python /root/repo/bench.py $C4 > /tmp/c3.log 2>&1
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD --kernel-trace -d /tmp/q4 -o r -- python /root/repo/bench.py $C4 > /tmp/c4.log 2>&1
    rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_ACTIVE_INST_LDS --kernel-trace -d /tmp/q5 -o r -- python /root/repo/bench.py $C4 > /tmp/c5.log 2>&1
    mkdir -p /root/repo/$O/lit
    PMC_TAG=$TAG PMC_COMMIT=${PMC_COMMIT:-unknown} PMC_CSRC_HASH=$(cd /root/repo && python -c 'import bench; print(bench.csrc_hash())') python /root/repo/scripts/rocpd_pmc.py /root/repo/$O/lit/pmc_cfg4_literal.md $(find /tmp/q2 /tmp/q3 /tmp/q4 /tmp/q5 -name "*.db") > /dev/null
    cp /root/repo/$O/lit/pmc_cfg4_literal.md /root/repo/$O/pmc_cfg4_literal.md
    if [ -f /root/repo/$O/pmc_traffic.json ]; then python /root/repo/scripts/pmc_merge.py /root/repo/$O/pmc_traffic.json /root/repo/$O/lit/pmc_traffic.json; fi
    head -16 /root/repo/$O/kernel_stats_cfg4_literal.md
  fi
  if has cfg2_profile; then
    rocprofv3 --kernel-trace --stats -d /tmp/q6 -o r -- python /root/repo/bench.py --config cfg2 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > /tmp/c6.log 2>&1
    python /root/repo/scripts/rocpd_summary.py $(find /tmp/q6 -name "*.db" | head -1) /root/repo/$O/kernel_stats_cfg2_f64.md > /dev/null
    head -14 /root/repo/$O/kernel_stats_cfg2_f64.md
  fi
  cd /root/repo
fi
