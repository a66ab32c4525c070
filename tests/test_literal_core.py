"""The literal anisotropic compression of the HIP library (msckf_mono_amd/csrc/literal_core.h: R_o_j = A_j^T R_j A_j,
HouseholderQR of the stack in column order with the zero-tail rule, R_n = Q_1^T R_o Q_1; msckf.h:423-431, 1343-1366) is
written for two compilers.  Here its host build (tests/cpp/literal_host.cpp, g++ -DLIT_HOST) is held against the oracle's
restatement of the same lines on the CPU: same kept rows, same information matrix [T_H | r_n]^T R_n^-1 [T_H | r_n]."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import helpers as H
from msckf_mono_amd import scenario as sc

ROOT = H.ROOT
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


@pytest.fixture(scope="module")
def lit(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("lit") / "liblit_host.so")
    out = subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unknown-pragmas", "-Wno-unused-variable", "-Wno-unused-but-set-variable", "-o", so, os.path.join(ROOT, "tests", "cpp", "literal_host.cpp")],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    return C.CDLL(so)


@pytest.fixture(scope="module")
def po(oracle_lib):
    return oracle_lib


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
    assert F >= 0
    return F, M[:F].copy(), ps[:F].copy(), sl[:F].copy(), hx[:F].copy(), r[:F].copy()


def gram_of_the_stack(N, M, inc, slots, hx, r):
    """[H_o | r_o]^T [H_o | r_o] = sum over the stacked tracks of [H_x | r]^T (I - Q_f Q_f^T) [H_x | r] -- what k_gram
    accumulates on the device (independent of the null-space basis); numpy, independent of the code under test"""
    n = 6 * N
    L = np.zeros((n + 1, n + 1))
    for t in range(len(M)):
        if not inc[t]:
            continue
        m2 = 2 * M[t]
        X = np.zeros((m2, n + 1)); Hf = np.zeros((m2, 3))
        for o in range(M[t]):
            for i in range(2):
                X[2 * o + i, 6 * slots[t, o]:6 * slots[t, o] + 6] = hx[t, o, 6 * i:6 * i + 6]
                Hf[2 * o + i] = -hx[t, o, 6 * i + 3:6 * i + 6]
        X[:, n] = r[t, :m2]
        Q = np.linalg.qr(Hf)[0]
        L += X.T @ X - (Q.T @ X).T @ (Q.T @ X)
    return L


def host_compress(lit, N, M, inc, slots, hx, r, u, v, tol, route=1):
    """route 1: the sweep over the dense stack; 0: the compact route (needs H_o^T H_o, as k_gram accumulates it on the device)"""
    F, m_cap = hx.shape[0], hx.shape[1]
    n = 6 * N
    Lam = np.zeros((n + 1, n + 1)); info = np.zeros(8, dtype=np.int32)
    TH = np.zeros((n + 1, n + 15))
    Lin = np.ascontiguousarray(gram_of_the_stack(N, M, inc, slots, hx, r)) if route != 1 else None
    rc = lit.lit_host_compress(F, m_cap, N, np.ascontiguousarray(inc, dtype=np.int32).ctypes.data_as(_ip),
                               np.ascontiguousarray(M, dtype=np.int32).ctypes.data_as(_ip),
                               np.ascontiguousarray(slots, dtype=np.int32).ctypes.data_as(_ip),
                               np.ascontiguousarray(hx).ctypes.data_as(_dp), np.ascontiguousarray(r).ctypes.data_as(_dp),
                               C.c_double(u), C.c_double(v), C.c_double(tol), int(route),
                               Lin.ctypes.data_as(_dp) if Lin is not None else None, Lam.ctypes.data_as(_dp), info.ctypes.data_as(_ip),
                               TH.ctypes.data_as(_dp))
    assert rc == 0
    L = np.tril(Lam); L = L + np.tril(L, -1).T
    return L, info, TH.T[:info[1]]


def oracle_information(o, N):
    """[T_H | r_n]^T R_n^-1 [T_H | r_n] over the camera columns, from the restatement's captured compression"""
    T = _matrix(o, 2)[:, 15:15 + 6 * N]; rn = _matrix(o, 3); Rn = _matrix(o, 4)
    Y = np.hstack([T, rn])
    return Y.T @ np.linalg.solve(Rn, Y), T.shape[0]


@pytest.mark.parametrize("N,F,nf,traj,tol", [(8, 24, 14, 5, 2e-7), (8, 24, 14, 6, 2e-7), (10, 50, 14, 7, 2e-7), (6, 6, 12, 3, 2e-7), (12, 80, 18, 9, 2e-7), (8, 24, 10, 5, 0.0)])
def test_host_build_of_the_literal_core_matches_the_restatement(lit, po, N, F, nf, traj, tol):
    cfg = sc.filter_config(N, isotropic=False); cfg["translation_threshold"] = 0.01
    tr = sc.Trajectory(2, traj, N, F, nf, cfg=cfg)
    o = po.Oracle(po.F64, po.LEAN)
    o.L.oracle_set_tiny_row_tol(o.h, C.c_double(tol)); o.L.oracle_set_capture(o.h, 1)
    o.initialize(tr.cfg, tr.imu0)
    u, v = tr.cfg["u_var_prime"], tr.cfg["v_var_prime"]
    assert u != v
    compared = 0
    middle = []
    for k in range(nf):
        H.oracle_frame(o, tr, k, N)      # the oracle's window at update time is read back below
        st = o.lastStats()
        if st["m_rows"] == 0:
            continue
        Fk, M, ps, sl, hx, r = _track_inputs(o)
        ncam = (_matrix(o, 2).shape[1] - 15) // 6
        L_or, nr_or = oracle_information(o, ncam)
        L_host, info, TH = host_compress(lit, ncam, M, ps, sl, hx, r, u, v, tol)
        assert info[0] == st["m_rows"] and info[4] == 2
        if tol > 0:
            assert info[1] == nr_or == st["r_rows"], (k, info, nr_or)
            err = np.linalg.norm(L_host - L_or) / np.linalg.norm(L_or)
            assert err < 1e-9, (k, err)
            # the compact route (default on the device): the same Householder steps on [explicit first 15 + 6N rows ; Gram matrix
            # of the rest] -- no stack -- must reflect and skip at the same steps and give the same information matrix, whatever
            # the shape of the stack (few rows, a dependent column in the middle of the sweep: frame 13 of trajectory 6)
            L_c, info_c, _ = host_compress(lit, ncam, M, ps, sl, hx, r, u, v, tol, route=0)
            assert info_c[4] == 3 and info_c[1] == nr_or and info_c[2] == info[2], (k, info_c, info)      # (info[3] counts skips with a non-zero tail only)
            assert np.linalg.norm(L_c - L_or) / np.linalg.norm(L_or) < 1e-9, k
            middle.append(int(info[2]) + int(info[3]) < min(st["m_rows"], 15 + 6 * ncam) - 15 or bool(info[3] > 7))
        else:
            # the reference's rule to the letter keeps rounding-level rows whose Q columns are rounding noise: the two
            # builds agree on everything but those (same count of kept rows, information matrix to ~1e-3)
            assert info[1] == nr_or
            assert np.linalg.norm(L_host - L_or) / np.linalg.norm(L_or) < 2e-2
        compared += 1
    assert compared >= nf - 4
    if tol > 0 and (N, traj) == (8, 6):
        assert middle[-1]              # frame 13: 15 + 129 rows, rank 31 of 42 observed columns in the middle of the sweep (11 skipped steps)


def test_compact_route_at_the_30_camera_window_equals_the_sweep_over_the_stack(lit, po):
    """BASELINE configs[3] geometry (30-camera window, 200 tracks, EuRoC intrinsics): the compact route against the sweep over
    the dense ~5 800 x 181 stack, with f64 Jacobians and with float-rounded ones (the float filter's), in a good and in the
    weakest of the benchmark's sequences: same steps reflected / skipped, information matrix to 1e-8."""
    N, F, nf = 30, 200, 32
    f32 = lambda x: x.astype(np.float32).astype(np.float64)
    for g in (0, 7):
        cfg = sc.filter_config(N, isotropic=False)
        tr = sc.Trajectory(4, g, N, F, nf, cfg=cfg, path_id=g % 5)
        o = po.Oracle(po.F64, po.GRAM); o.setWhiten(True); o.setCapture(True); o.initialize(tr.cfg, tr.imu0)     # (cheap way to a steady-state window)
        u, v = tr.cfg["u_var_prime"], tr.cfg["v_var_prime"]
        for k in range(nf):
            H.oracle_frame(o, tr, k, N)
        Fk, M, ps, sl, hx, r = _track_inputs(o, cap_m=64)
        su, sv = np.sqrt(u), np.sqrt(v)
        hx[:, :, 0:6] *= su; hx[:, :, 6:12] *= sv; r[:, 0::2] *= su; r[:, 1::2] *= sv       # undo the whitening of the captured rows
        for hh, rr, tol in ((hx, r, 1e-10), (f32(hx), f32(r), 8e-4)):
            L_c, ic, _ = host_compress(lit, N, M, ps, sl, hh, rr, u, v, tol, route=0)
            L_g, ig, _ = host_compress(lit, N, M, ps, sl, hh, rr, u, v, tol, route=1)
            assert ic[0] == ig[0] > 5000 and ic[2] == ig[2] and abs(int(ic[1]) - int(ig[1])) <= 2, (g, ic, ig)
            assert np.linalg.norm(L_c - L_g) / np.linalg.norm(L_g) < 1e-8, g


def test_blocked_elimination_is_independent_of_the_panel_width(lit, po, monkeypatch):
    """The elimination of R_n (literal_core.h: information_from_rn) and the sweep (sweep_gram_blocked) run in panels staged in
    the device's LDS; the panel width follows the size of that area (16 at 94 KB, fewer when it is smaller, step by step when
    nothing fits).  Every width applies the same eliminations (products grouped differently): the information matrix of a
    30-camera window moves by rounding only."""
    N, F, nf = 30, 200, 32
    cfg = sc.filter_config(N, isotropic=False)
    tr = sc.Trajectory(4, 0, N, F, nf, cfg=cfg, path_id=0)
    o = po.Oracle(po.F64, po.GRAM); o.setWhiten(True); o.setCapture(True); o.initialize(tr.cfg, tr.imu0)
    u, v = tr.cfg["u_var_prime"], tr.cfg["v_var_prime"]
    for k in range(nf):
        H.oracle_frame(o, tr, k, N)
    Fk, M, ps, sl, hx, r = _track_inputs(o, cap_m=64)
    su, sv = np.sqrt(u), np.sqrt(v)
    hx[:, :, 0:6] *= su; hx[:, :, 6:12] *= sv; r[:, 0::2] *= su; r[:, 1::2] *= sv
    ref = None
    for stage in (12000, 8000, 3000, 400, 100):
        monkeypatch.setenv("LIT_HOST_STAGE", str(stage))
        L_c, ic, _ = host_compress(lit, N, M, ps, sl, hx, r, u, v, 1e-10, route=0)
        if ref is None:
            ref = L_c
        else:
            assert np.linalg.norm(L_c - ref) <= 1e-11 * np.linalg.norm(ref), stage


def _without_slots(M, inc, slots, hx, r, drop):
    """the same tracks without their observations in the camera slots `drop` (tracks left with fewer than 3 leave the stack)"""
    M = M.copy(); slots = slots.copy(); hx = hx.copy(); r = r.copy(); inc = np.array(inc).copy()
    for t in range(len(M)):
        keep = [o for o in range(M[t]) if slots[t, o] not in drop]
        k = len(keep)
        slots[t, :k] = slots[t, keep]; hx[t, :k] = hx[t, keep]
        rr = r[t].copy()
        for q, o in enumerate(keep):
            r[t, 2 * q:2 * q + 2] = rr[2 * o:2 * o + 2]
        M[t] = k
        if k < 3:
            inc[t] = 0
    return M, inc, slots, hx, r


@pytest.mark.parametrize("drop", [(20,), (5, 27), (1,)])
def test_handed_through_rows_deep_in_the_sweep(lit, po, monkeypatch, drop):
    """A camera state that no stacked track observes has six zero columns: their steps reflect nothing, their rows are handed
    through and kept (the later columns have entries there), and their columns of Q are the unit vectors under all the
    reflectors before them -- 6 x slot steps deep.  The compact route builds those columns from the column operations of the
    sweep in blocks of sixteen (literal_core.h: compact_basis, par_gemm4, reflector_chain, extras_products); the sweep over the
    dense stack is the definition.  Also with the staging area too small for anything (every stand-in path)."""
    N, F, nf = 30, 200, 32
    cfg = sc.filter_config(N, isotropic=False)
    tr = sc.Trajectory(4, 0, N, F, nf, cfg=cfg, path_id=0)
    o = po.Oracle(po.F64, po.GRAM); o.setWhiten(True); o.setCapture(True); o.initialize(tr.cfg, tr.imu0)
    u, v = tr.cfg["u_var_prime"], tr.cfg["v_var_prime"]
    for k in range(nf):
        H.oracle_frame(o, tr, k, N)
    Fk, M, ps, sl, hx, r = _track_inputs(o, cap_m=64)
    su, sv = np.sqrt(u), np.sqrt(v)
    hx[:, :, 0:6] *= su; hx[:, :, 6:12] *= sv; r[:, 0::2] *= su; r[:, 1::2] *= sv
    M2, inc2, sl2, hx2, r2 = _without_slots(M, ps, sl, hx, r, set(drop))
    L_g, ig, _ = host_compress(lit, N, M2, inc2, sl2, hx2, r2, u, v, 1e-10, route=1)
    for stage in (None, "100"):
        if stage:
            monkeypatch.setenv("LIT_HOST_STAGE", stage)
        L_c, ic, _ = host_compress(lit, N, M2, inc2, sl2, hx2, r2, u, v, 1e-10, route=0)
        assert ic[6] >= 6 and ic[1] == ig[1] and ic[2] == ig[2], (drop, ic, ig)       # six (or more) kept handed-through rows
        assert np.linalg.norm(L_c - L_g) / np.linalg.norm(L_g) < 1e-9, (drop, stage)
