#!/usr/bin/env python3
"""Experiment: executed-instruction breakdown of k_feature by ablation knob (ablate build only).
Run under `rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace`;
every knob gets a freshly warmed-up filter and ONE knobbed frame (the last k_feature dispatch of its group), then
`feat_insts.py --dump <db>` lists the counters of those dispatches.
  MSCKF_HIP_LIB=msckf_mono_amd/lib_ab/libmsckf_hip_ablate.so python scripts/feat_insts.py"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
KNOBS = [("full", 0), ("noLM", 1), ("noPload", 2), ("noG", 32), ("noChol", 4), ("noF64", 8),
         ("noPublish", 64), ("noMotion", 256), ("gate_off(G,chol)", 32 | 4), ("all", 1 | 32 | 4 | 8 | 64 | 256), ("full", 0)]
N, F, B = 30, 200, 64
NF = N + 3


def dump(dbp):
    import re
    import sqlite3
    from collections import defaultdict
    db = sqlite3.connect(dbp)
    per = defaultdict(lambda: defaultdict(float))
    for name, disp, cname, val in db.execute("select kernel_name, dispatch_id, counter_name, value from counters_collection"):
        if "k_feature" in name:
            per[disp][cname] += val
    disps = sorted(per)
    groups = [disps[i:i + NF] for i in range(0, len(disps), NF)]
    cn = sorted({c for d in per.values() for c in d})
    print("| knob | " + " | ".join(cn) + " |")
    print("|---|" + "---|" * len(cn))
    base = None
    for (name, _k), g in zip(KNOBS, groups):
        v = per[g[-1]]
        if base is None:
            base = dict(v)
        print("| %s | " % name + " | ".join("%.4g (%+.1f%%)" % (v[c], 100.0 * (v[c] - base[c]) / max(base[c], 1.0)) for c in cn) + " |")


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--dump":
        return dump(sys.argv[2])
    from msckf_mono_amd import capi, scenario as sc
    trs = [sc.Trajectory(3, b, N, F, NF) for b in range(B)]
    for name, knob in KNOBS:
        bt = capi.Batch(B, N, F, N, capi.F32)
        bt.scenario_alloc(NF, 10)
        for b, tr in enumerate(trs):
            bt.initialize(b, tr.cfg, tr.imu0)
            for f in range(NF):
                fr = tr.frames[f]
                bt.scenario_set(f, b, tr.imu_for_frame(f), fr["M"], fr["slots"], fr["obs"], 1 if fr["Nw"] == N else 0)
        bt.scenario_commit()
        bt.L.msckf_hip_debug_set(200, 0)
        bt.run_frames(0, NF - 1); bt.sync()
        bt.L.msckf_hip_debug_set(200, knob)
        bt.profile_enable(True)
        bt.run_frames(NF - 1, NF)
        p = bt.profile_read()
        bt.L.msckf_hip_debug_set(200, 0)
        print(name, knob, {k: round(v[0] / max(v[1], 1), 4) for k, v in p.items() if k == "feature"}, flush=True)
        bt.close()


if __name__ == "__main__":
    main()
