# A/B on one lease: msckf_mono_amd/lib_ab/libmsckf_hip_prev.so against the current library, alternating, twice each
#   TAG=r06_x [AB_ARGS="--config cfg5 ..."] bash scripts/ab_lib.sh
cd /root/repo
O=gpurun_out/${TAG:-ab}; mkdir -p $O
P=msckf_mono_amd/lib_ab/libmsckf_hip_prev.so
BA="--steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-early-accept-pass --repeats 6 ${AB_ARGS:-}"
for i in 1 2; do
  for v in prev cur; do
    if [ $v = prev ]; then export MSCKF_HIP_LIB=$P; else unset MSCKF_HIP_LIB; fi
    python bench.py $BA > $O/bench_$v$i.json 2> $O/bench_$v$i.err
    python - $O/bench_$v$i.json $v$i <<PY
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = j.get("roofline", {})
    print(sys.argv[2], round(j["value"]), "ms/step", round(j["ms_per_step"], 4), "median", round(j.get("repeats_median") or 0), "kernel", r.get("kernel"), {k: round(x, 4) for k, x in (r.get("stage_ms_per_step") or {}).items()})
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
  done
done
