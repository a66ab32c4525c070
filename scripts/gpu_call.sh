# One gpurun call that validates a tree and collects a round's evidence (from the repo root on the GPU box):
#   TAG=r04_a PMC_COMMIT=$(git rev-parse --short HEAD) /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash scripts/gpu_call.sh'
# ~6 GPU-minutes: GPU parity tests, smoke, default bench + cfg2 + cfg4 + cfg5, variant sweep, kernel stats + PMC passes.
# Outputs land in gpurun_out/$TAG; copy what is to be judged into profiles/ by hand (profiles/README.md lists them).
cd /root/repo
TAG=${TAG:-round}
O=gpurun_out/$TAG; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest.txt
tail -4 $O/pytest.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --config cfg2 --steps 20 --warmup 5 > $O/bench_cfg2.json 2>/dev/null
python bench.py --config cfg4 --steps 10 --warmup 3 --no-cpu-baseline --no-early-accept-pass > $O/bench_cfg4.json 2>/dev/null
python bench.py --config cfg4 --steps 20 --warmup 5 --no-cpu-baseline --no-early-accept-pass --aniso-mode 1 > $O/bench_cfg4_whitened.json 2>/dev/null
python bench.py --config cfg5 --steps 10 --warmup 3 --no-cpu-baseline --no-early-accept-pass > $O/bench_cfg5.json 2>/dev/null
python - <<PY
import json
for f in ["bench", "bench_cfg2", "bench_cfg4", "bench_cfg4_whitened", "bench_cfg5"]:
    try:
        j = json.loads(open("$O/%s.json" % f).read().strip().splitlines()[-1])
        print(f, round(j["value"]), round(j.get("repeats", {}).get("median", 0)), round((j.get("resident_inputs") or {}).get("median", 0)), "frac", j["roofline"].get("frac"))
    except Exception as e:
        print(f, "failed", e)
PY
python scripts/sweep_variants.py --steps 20 --windows 5 "streams=4,streamed=0" "streams=4,streamed=1" "streams=1,streamed=0" > $O/sweep.txt 2>&1
cut -c1-250 $O/sweep.txt
TAG=$TAG PMC_COMMIT=${PMC_COMMIT:-unknown} bash scripts/profile_round.sh > $O/profile_round.log 2>&1
head -12 $O/kernel_stats_tail20.md
