"""Numerical experiment (CPU, numpy): is an information-form (Gram) compression of the stacked Jacobian as
accurate in float32 as the Householder-QR compression?  Captures (Hx_j, Hf_j, r_j, P) of cfg3 frames from the
numpy twin and replays the update in float32 both ways against the float64 result."""
import sys, time, numpy as np, scipy.linalg as sla
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle")
from msckf_mono_amd import scenario
import np_oracle

N, F = int(sys.argv[1]) if len(sys.argv) > 1 else 30, int(sys.argv[2]) if len(sys.argv) > 2 else 200
nfr = int(sys.argv[3]) if len(sys.argv) > 3 else N + 6
cfg = scenario.filter_config(N)
tr = scenario.Trajectory(3, 0, N, F, nfr, cfg)
flt = np_oracle.NpMSCKF(cfg, tr.imu0, nullspace="householder")
cap = {}
orig_jac = flt.jac
def jac(p_f, slots, obs):
    M, D = len(slots), flt.P.shape[0]
    Hf, Hx, r = np.zeros((2 * M, 3)), np.zeros((2 * M, D)), np.zeros(2 * M)
    for c, s in enumerate(slots):
        cam = flt.cams[s]
        C = np_oracle.q2R(cam["q"]); pc = C @ (p_f - cam["p"]); X, Y, Z = pc
        Ji = np.array([[1, 0, -X / Z], [0, 1, -Y / Z]]) / Z
        A = np.hstack([Ji @ np_oracle.skew(pc), -Ji @ C])
        u = np.concatenate([C @ flt.g, np_oracle.skew(p_f - cam["p"]) @ flt.g])
        H = A - np.outer(A @ u, u) / (u @ u)
        Hf[2 * c:2 * c + 2] = -H[:, 3:6]; Hx[2 * c:2 * c + 2, 15 + 6 * s:21 + 6 * s] = H
        r[2 * c:2 * c + 2] = obs[c] - pc[:2] / Z
    U = sla.qr(Hf, mode="full")[0]
    A_j = U[:, 3:]
    cap.setdefault("cur", []).append((Hx, Hf, r))
    return A_j.T @ Hx, A_j.T @ r, A_j
flt.jac = jac

def kalman(T, rn, P, sig2, dt):
    T = T.astype(dt); rn = rn.astype(dt); P = P.astype(dt)
    D = P.shape[0]
    TH = np.zeros((T.shape[0], D), dt); TH[:, 15:] = T
    S = TH @ P @ TH.T + dt(sig2) * np.eye(T.shape[0], dtype=dt)
    L = np.linalg.cholesky(S.astype(dt))
    PHt = P @ TH.T
    W = sla.solve_triangular(L, PHt.T, lower=True).astype(dt)       # L^-1 (PHt)^T
    K = sla.solve_triangular(L.T, W, lower=False).astype(dt).T
    dx = K @ rn
    A = np.eye(D, dtype=dt) - K @ TH
    Pn = A @ P @ A.T + dt(sig2) * (K @ K.T)
    return dx, (Pn + Pn.T) / 2

def chol_skip(Lam, dt, tol):
    """upper T with T^T T = Lam; a pivot below tol*original diagonal zeroes the row (semidefinite skip)"""
    A = Lam.astype(dt).copy(); n = A.shape[0]; T = np.zeros((n, n), dt); d0 = np.diag(A).copy(); skipped = 0
    for k in range(n):
        p = A[k, k]
        if p <= tol * d0[k] or p <= 0:
            skipped += 1; continue
        s = dt(1) / np.sqrt(p)
        row = A[k, k:] * s
        T[k, k:] = row
        A[k:, k:] -= np.outer(row, row).astype(dt)
    return T, skipped

def rel(a, b): return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)

sig2 = cfg["u_var_prime"]
orig_mu = flt.measurement_update
results = []
def mu(H, r, R):
    P = flt.P.copy(); trk = cap.pop("cur")
    # keep only the tracks that passed the gate: match by count from H rows (gate rejects are rare; verify)
    n = H.shape[1] - 15
    rows = sum(2 * hx.shape[0] // 2 * 1 for hx, _, _ in trk)
    # float64 truth via economic QR
    Rq = np.linalg.qr(np.hstack([H[:, 15:], r[:, None]]), mode="r")
    T64, rn64 = np.triu(Rq[:n, :n]), Rq[:n, n]
    dx64, P64 = kalman(T64, rn64, P, sig2, np.float64)
    # (1) float32 QR
    H32 = np.hstack([H[:, 15:], r[:, None]]).astype(np.float32)
    Rq = np.linalg.qr(H32, mode="r")
    dx1, P1 = kalman(np.triu(Rq[:n, :n]), Rq[:n, n], P, sig2, np.float32)
    # (2) float32 Gram, subtractive per-track form
    dt = np.float32
    Lam = np.zeros((n, n), dt); y = np.zeros(n, dt); m_chk = 0
    passed = H.shape[0]
    for Hx, Hf, rr in trk:
        Hx32, Hf32, r32 = Hx[:, 15:].astype(dt), Hf.astype(dt), rr.astype(dt)
        Qf = np.linalg.qr(Hf32, mode="reduced")[0].astype(dt)
        B = (Qf.T @ Hx32).astype(dt); c = (Qf.T @ r32).astype(dt)
        Lam += (Hx32.T @ Hx32 - B.T @ B).astype(dt); y += (Hx32.T @ r32 - B.T @ c).astype(dt)
        m_chk += 2 * Hx.shape[0] // 2 * 1 - 3 + (Hx.shape[0] - 2 * Hx.shape[0] // 2)
    gate_all = (sum(hx.shape[0] - 3 for hx, _, _ in trk) == passed)
    out = dict(m=passed, gate_all=gate_all, qr_dx=rel(dx1, dx64), qr_P=rel(P1, P64), qr_PII=rel(P1[:15, :15], P64[:15, :15]))
    if gate_all:
        for tol in ():
            T2, sk = chol_skip(Lam, dt, tol)
            # r_n = T^-T y on the non-skipped rows
            nzr = np.diag(T2) > 0
            Tn = T2[nzr][:, :]
            rn = np.zeros(n, dt)
            # solve T^T rn = y restricted: use lstsq on the kept rows (small)
            rn_k = np.linalg.lstsq(Tn.T.astype(np.float64), y.astype(np.float64), rcond=None)[0].astype(dt)
            dx2, P2 = kalman(Tn, rn_k, P, sig2, dt)
            out[f"gram{tol:g}"] = (sk, rel(dx2, dx64), rel(P2, P64), rel(P2[:15, :15], P64[:15, :15]))
        # (3) float32-rounded inputs, Gram accumulated in float64; reflectors/B in f32 (a) or f64 (b)
        for name, fdt in (("a_f32B", np.float32), ("b_f64B", np.float64)):
            Lam = np.zeros((n, n)); y = np.zeros(n)
            for Hx, Hf, rr in trk:
                Hx32, Hf32, r32 = Hx[:, 15:].astype(np.float32), Hf.astype(np.float32), rr.astype(np.float32)
                Qf = np.linalg.qr(Hf32.astype(fdt), mode="reduced")[0].astype(fdt)
                B = (Qf.T @ Hx32.astype(fdt)).astype(fdt).astype(np.float64); c = (Qf.T @ r32.astype(fdt)).astype(np.float64)
                Hx64 = Hx32.astype(np.float64)
                Lam += Hx64.T @ Hx64 - B.T @ B; y += Hx64.T @ r32.astype(np.float64) - B.T @ c
            ev = np.linalg.eigvalsh(Lam)
            for tol in (1e-7, 1e-10, 1e-13):
                A = Lam.copy(); T = np.zeros((n, n)); d0 = np.diag(A).copy(); yy = y.copy(); rn = np.zeros(n); cl = 0
                for k in range(n):
                    pv = A[k, k]
                    if pv <= tol * d0[k]: cl += 1; continue
                    sc = 1 / np.sqrt(pv); row = A[k, k:] * sc; row[0] = np.sqrt(pv)
                    T[k, k:] = row; rn[k] = yy[k] * sc
                    A[k:, k:] -= np.outer(row, row); yy[k:] -= row * rn[k]
                dx2, P2 = kalman(T, rn, P, sig2, np.float32)
                out[f"{name}_{tol:g}"] = (cl, f"{rel(dx2, dx64):.1e}", f"{rel(P2, P64):.1e}", f"{rel(P2[:15, :15], P64[:15, :15]):.1e}")
            out[name + "_evmin"] = f"{ev[0]:.1e}/{ev[-1]:.1e}"
    results.append(out); print({k: (f'{v:.1e}' if isinstance(v, float) else v) for k, v in out.items()}, flush=True)
    orig_mu(H, r, R)
flt.measurement_update = mu

t0 = time.time()
for k in range(nfr):
    fr = tr.frames[k]
    for rd in tr.imu_for_frame(k): flt.propagate(rd)
    flt.augment(k)
    cap.pop("cur", None)
    if len(fr["M"]):
        flt.set_tracks(fr["M"], fr["slots"], fr["obs"])
        flt.marginalize()
    if len(flt.cams) == N: flt.drop_oldest(1)
    print("frame", k, "ncam", len(flt.cams), "t", round(time.time() - t0, 1), flush=True)
