"""Capture per-update inputs of the literal anisotropic compression (what k_feature leaves: M, slots, H_x, r per track) from
the CPU oracle, for prototyping (scripts/experiments/planb.py).  Writes /tmp/planb_cases.pkl."""
import ctypes as C
import os
import pickle
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import helpers as H
import pyoracle as po
from msckf_mono_amd import scenario as sc

_dp = C.POINTER(C.c_double); _ip = C.POINTER(C.c_int)


def _matrix(o, which):
    cols = C.c_int(0)
    buf = np.zeros(1 << 22)
    o.L.oracle_last_matrix.restype = C.c_int
    n = o.L.oracle_last_matrix(o.h, which, buf.ctypes.data_as(_dp), C.c_long(buf.size), C.byref(cols))
    assert n >= 0
    return buf[:n * cols.value].reshape((cols.value, n)).T.copy()


def _track_inputs(o, cap_f=1024, cap_m=64):
    M = np.zeros(cap_f, dtype=np.int32); ps = np.zeros(cap_f, dtype=np.int32); sl = np.zeros((cap_f, cap_m), dtype=np.int32)
    hx = np.zeros((cap_f, cap_m, 12)); r = np.zeros((cap_f, 2 * cap_m))
    F = o.L.oracle_last_track_inputs(o.h, M.ctypes.data_as(_ip), ps.ctypes.data_as(_ip), sl.ctypes.data_as(_ip),
                                     hx.ctypes.data_as(_dp), r.ctypes.data_as(_dp), cap_f, cap_m)
    return F, M[:F].copy(), ps[:F].copy(), sl[:F].copy(), hx[:F].copy(), r[:F].copy()


def main():
    cases = []
    for (N, F, nf, traj, tol) in [(8, 24, 14, 5, 2e-7), (8, 24, 14, 6, 2e-7), (10, 50, 14, 7, 2e-7), (6, 6, 12, 3, 2e-7), (12, 80, 18, 9, 2e-7)]:
        cfg = sc.filter_config(N, isotropic=False); cfg["translation_threshold"] = 0.01
        tr = sc.Trajectory(2, traj, N, F, nf, cfg=cfg)
        o = po.Oracle(po.F64, po.LEAN)
        o.L.oracle_set_tiny_row_tol(o.h, C.c_double(tol)); o.L.oracle_set_capture(o.h, 1)
        o.initialize(tr.cfg, tr.imu0)
        u, v = tr.cfg["u_var_prime"], tr.cfg["v_var_prime"]
        for k in range(nf):
            H.oracle_frame(o, tr, k, N)
            st = o.lastStats()
            if st["m_rows"] == 0:
                continue
            Fk, M, ps, sl, hx, r = _track_inputs(o)
            T = _matrix(o, 2); ncam = (T.shape[1] - 15) // 6
            T = T[:, 15:15 + 6 * ncam]; rn = _matrix(o, 3); Rn = _matrix(o, 4)
            Y = np.hstack([T, rn])
            L_or = Y.T @ np.linalg.solve(Rn, Y)
            cases.append(dict(name=f"N{N}F{F}t{traj}k{k}", N=ncam, M=M, inc=ps, slots=sl, hx=hx, r=r, u=u, v=v, tol=tol, L_or=L_or, nr=T.shape[0]))
    # 30-camera window (cfg4 geometry)
    N, F, nf = 30, 200, 32
    for g in (0, 7, 3):
        cfg = sc.filter_config(N, isotropic=False)
        tr = sc.Trajectory(4, g, N, F, nf, cfg=cfg, path_id=g % 5)
        o = po.Oracle(po.F64, po.GRAM); o.setWhiten(True); o.setCapture(True); o.initialize(tr.cfg, tr.imu0)
        u, v = tr.cfg["u_var_prime"], tr.cfg["v_var_prime"]
        for k in range(nf):
            H.oracle_frame(o, tr, k, N)
            if k < 29:
                continue
            Fk, M, ps, sl, hx, r = _track_inputs(o, cap_m=64)
            su, sv = np.sqrt(u), np.sqrt(v)
            hx = hx.copy(); r = r.copy()
            hx[:, :, 0:6] *= su; hx[:, :, 6:12] *= sv; r[:, 0::2] *= su; r[:, 1::2] *= sv
            cases.append(dict(name=f"N30g{g}k{k}", N=N, M=M, inc=ps, slots=sl, hx=hx, r=r, u=u, v=v, tol=1e-10, L_or=None, nr=None))
    with open("/tmp/planb_cases.pkl", "wb") as f:
        pickle.dump(cases, f)
    print(len(cases), "cases")


if __name__ == "__main__":
    main()
