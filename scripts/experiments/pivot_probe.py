#!/usr/bin/env python3
"""Where does the factorization of S meet a non-positive pivot on a long float run?  (diagnostic for the sticky
STAT_ERR_PIVOT flag; prints the frames at which it trips, with the size of P's gauge variance at that moment)"""
import sys, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from msckf_mono_amd import capi, scenario as sc

N, F, nf = 10, 40, int(sys.argv[1]) if len(sys.argv) > 1 else 4000
step = int(sys.argv[2]) if len(sys.argv) > 2 else 100
tr = sc.Trajectory(2, 91, N, F, nf)
for form in (0, 1):
    bt = capi.Batch(1, N, F, N, capi.F32)
    bt.set_covariance_update(form)
    bt.scenario_alloc(nf, sc.IMU_PER_FRAME)
    bt.initialize(0, tr.cfg, tr.imu0)
    for k in range(nf):
        fr = tr.frames[k]
        bt.scenario_set(k, 0, tr.imu_for_frame(k), fr["M"], fr["slots"], fr["obs"], 1 if fr["Nw"] == N else 0)
    bt.scenario_commit()
    trips = []
    for k0 in range(0, nf, step):
        bt.run_frames(k0, min(nf, k0 + step)); bt.sync()
        try:
            st = bt.last_stats(0)
        except capi.HipError as e:
            P = bt.covariance(0)
            w = np.linalg.eigvalsh(P)
            trips.append(k0)
            if len(trips) <= 8:
                print("form", form, "trip in frames [%d, %d)" % (k0, k0 + step), "eig(P) min %.3e max %.3e" % (w.min(), w.max()),
                      "diag p %.3e" % P[12:15, 12:15].trace(), "err to gt %.3f m" % np.linalg.norm(bt.imu_state(0)[13:16] - tr.gt_frames["p"][min(nf, k0 + step) - 1]), flush=True)
            bt.clear_error_flags(0)
    P = bt.covariance(0); w = np.linalg.eigvalsh(P)
    print("form", form, "trips:", len(trips), trips[:20], "final eig(P) min %.3e max %.3e" % (w.min(), w.max()),
          "final err %.3f m" % np.linalg.norm(bt.imu_state(0)[13:16] - tr.gt_frames["p"][nf - 1]), flush=True)
    bt.close()
