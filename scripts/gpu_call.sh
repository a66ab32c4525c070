cd /root/repo
O=gpurun_out/r03l; mkdir -p $O
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
TAG=r03l PMC_COMMIT=$(cat .commit_id 2>/dev/null) timeout 900 bash scripts/profile_round.sh > $O/profile.log 2>&1
timeout 600 python bench.py --config cfg5 --steps 10 --warmup 3 > $O/bench_cfg5.json 2> $O/bench_cfg5.err
timeout 600 python bench.py --config cfg4 --steps 20 --warmup 5 --cpu-seconds 5 > $O/bench_cfg4.json 2> $O/bench_cfg4.err
tail -c 300 $O/bench.json; echo; tail -c 400 $O/bench_cfg5.json; tail -3 $O/bench_cfg5.err; head -16 $O/kernel_stats_tail20.md
