#!/usr/bin/env python3
"""Experiment: time the feature stage of the bench workload with ablation knobs.  Each knob is measured
from a freshly warmed-up filter so that earlier (garbage-producing) ablations cannot change the workload."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from msckf_mono_amd import capi, scenario as sc
N, F, B, K = 30, 200, int(os.environ.get("FEAT_B", "64")), 10
nf = N + 8
trs = [sc.Trajectory(3, b, N, F, nf) for b in range(B)]
knobs = [("full", 0), ("noLM", 1), ("noG", 2), ("noChol", 4), ("noF64", 8), ("noG+noChol", 6), ("all", 15), ("full", 0)]
if len(sys.argv) > 1:
    knobs = [("k%s" % a, int(a)) for a in sys.argv[1:]]
for name, knob in knobs:
    bt = capi.Batch(B, N, F, N, capi.F32)
    bt.scenario_alloc(nf, K)
    for b, tr in enumerate(trs):
        bt.initialize(b, tr.cfg, tr.imu0)
        for f in range(nf):
            fr = tr.frames[f]
            bt.scenario_set(f, b, tr.imu_for_frame(f), fr["M"], fr["slots"], fr["obs"], 1 if fr["Nw"] == N else 0)
    bt.scenario_commit()
    bt.L.msckf_hip_debug_set(200, 0)
    bt.run_frames(0, N + 2); bt.sync()
    bt.L.msckf_hip_debug_set(200, knob)
    bt.profile_enable(True)
    bt.run_frames(N + 2, N + 3)          # ONE frame: the workload is identical for every knob
    p = bt.profile_read()
    bt.L.msckf_hip_debug_set(200, 0)
    print(name, {k: round(v[0] / max(v[1], 1), 3) for k, v in p.items() if k in ("feature",)}, bt.last_stats(0)["m_rows"])
    bt.close()
