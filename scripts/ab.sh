#!/bin/bash
# A/B: previous library vs current, alternating, 2 streams then 1
P=msckf_mono_amd/lib_ab/libmsckf_hip_prev.so
for s in 2 2 1; do
  echo -n "prev "; MSCKF_HIP_LIB=$P python scripts/bench_sweep.py $s
  echo -n "cur  "; python scripts/bench_sweep.py $s
  echo -n "curT "; MSCKF_GRAM_DBG=16 python scripts/bench_sweep.py $s
done
