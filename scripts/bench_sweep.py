#!/usr/bin/env python3
"""Run bench.py with several --streams values and print value / ms_per_step (experiment helper).
Usage: bench_sweep.py 2 4 8 [--steps 40]"""
import json, subprocess, sys, os
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
steps = "40"
args = [a for a in sys.argv[1:]]
if "--steps" in args:
    i = args.index("--steps"); steps = args[i + 1]; del args[i:i + 2]
for s in args:
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--steps", steps, "--warmup", "5", "--streams", s],
                         capture_output=True, text=True).stdout.strip().splitlines()[-1]
    d = json.loads(out)
    print("streams", s, "updates/s", round(d["value"]), "ms/step", round(d["ms_per_step"], 4), flush=True)
