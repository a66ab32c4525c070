#!/usr/bin/env python3
"""Ad-hoc GPU bring-up check: HIP path vs CPU oracle, stage by stage (run through gpurun)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import helpers as H
import numpy as np
from msckf_mono_amd import scenario as sc, capi
import pyoracle as po

def run(dtype, N, F, nf, teacher=True, mcap=None):
    tr = sc.Trajectory(2, 0, N, F, nf)
    odt = po.F64 if dtype == capi.F64 else po.F32
    o = po.Oracle(odt, po.LEAN); o.initialize(tr.cfg, tr.imu0)
    bt = capi.Batch(1, N, max(F, 1), mcap or max(N, 4), dtype)
    bt.initialize(0, tr.cfg, tr.imu0)
    worst = {}
    for k in range(nf):
        if teacher and k > 0:
            H.copy_oracle_to_device(o, bt, 0)
        # stage-wise
        o.propagate(tr.imu_for_frame(k)); bt.propagate_range(0, 1, tr.imu_for_frame(k))
        e1 = H.state_errors(bt.imu_state(0), o.getImuState(), *[x[0] for x in (bt.cam_states(0), o.getCamStates())], bt.covariance(0), o.getCovariance())
        o.augmentState(k, 0); bt.augment_range(0, 1)
        e2 = H.state_errors(bt.imu_state(0), o.getImuState(), bt.cam_states(0)[0], o.getCamStates()[0], bt.covariance(0), o.getCovariance())
        fr = tr.frames[k]
        e3 = {}
        if len(fr["M"]):
            o.setTracks(fr["M"], fr["slots"], fr["obs"]); o.marginalize()
            bt.set_tracks(0, fr["M"], fr["slots"], fr["obs"]); bt.marginalize_range(0, 1)
            e3 = H.state_errors(bt.imu_state(0), o.getImuState(), bt.cam_states(0)[0], o.getCamStates()[0], bt.covariance(0), o.getCovariance())
            so, sd = o.lastStats(), bt.last_stats(0)
            to, td = o.lastTracks(), bt.last_tracks(0)
            g_err = np.max(np.abs(to[:, 4] - td[:, 4]) / np.maximum(np.abs(to[:, 4]), 1e-12)) if len(to) else 0
            pf_err = np.max(np.abs(to[:, 5:8] - td[:, 5:8])) if len(to) else 0
            dx_o, dx_d = o.lastDeltaX(), bt.last_deltax(0)
            dxe = H.rel(dx_d, dx_o, 1e-12) if len(dx_o) == len(dx_d) and len(dx_o) else -1
            if k % 5 == 0 or k == nf - 1 or H.worst(e3) > 1e-3:
                print("  frame", k, "stats o/d", [so[x] for x in ("n_passed", "n_motion_rejected", "n_gate_rejected", "m_rows")], [sd[x] for x in ("n_passed", "n_motion_rejected", "n_gate_rejected", "m_rows")],
                      "gamma_rel %.2e pf %.2e dx_rel %.2e" % (g_err, pf_err, dxe))
        if o.getNumCamStates() == N:
            o.dropOldest(1); bt.drop_oldest_range(0, 1, 1)
        e4 = H.state_errors(bt.imu_state(0), o.getImuState(), bt.cam_states(0)[0], o.getCamStates()[0], bt.covariance(0), o.getCovariance())
        for name, e in (("prop", e1), ("aug", e2), ("upd", e3), ("prune", e4)):
            for kk, vv in e.items():
                worst[(name, kk)] = max(worst.get((name, kk), 0), vv)
    print("dtype", dtype, "N", N, "F", F, "teacher", teacher)
    for name in ("prop", "aug", "upd", "prune"):
        print("   ", name, {kk[1]: "%.2e" % vv for kk, vv in worst.items() if kk[0] == name})

if __name__ == "__main__":
    t = time.time()
    run(capi.F64, 6, 12, 14)
    run(capi.F64, 10, 50, 30)
    run(capi.F64, 10, 50, 30, teacher=False)
    run(capi.F32, 10, 50, 30)
    run(capi.F32, 30, 200, 45, mcap=32)
    print("total s", time.time() - t)
