"""Prototype (numpy, CPU): the literal anisotropic compression as a basis-free projection.

The reference keeps Q_1 = the columns of HouseholderQR(H_o)'s Q whose rows of R are non-zero and uses
   r_n = Q_1^T r_o,  R_n = Q_1^T R_o Q_1,  T_H = Q_1^T H_o                     (msckf.h:1343-1366)
so what the filter sees, Lam^ = [T_H | r_n]^T R_n^-1 [T_H | r_n], depends on Q_1 only through S = range(Q_1):
   Lam^ = A^T Bs (Bs^T R_o Bs)^-1 Bs^T A          for ANY basis Bs of S, A = [H_o | r_o].
S = span(e_i, i < 15 kept) + span(columns of H_o below row 15 that REFLECTED) + span(Q e_h, h a handed-through row that is kept).
The sweep is needed only for its decisions (which steps reflect, which rows are kept); it runs on
   E   : the rows that can become pivot rows, explicitly
   Ghat: the Gram matrix of all rows from the pivot row down (one rank-1 downdate per step, no inner products over rows).
"""
import pickle
import sys

import numpy as np
import scipy.linalg as sla

TINY = 2.2250738585072014e-308


def build_stack(c):
    N, M, inc, slots, hx, r = c["N"], c["M"], c["inc"], c["slots"], c["hx"], c["r"]
    n = 6 * N
    rows, Acal, tracks = [], [], []
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
        Q, _, _ = sla.qr(Hf, pivoting=True)
        A = Q[:, 3:]
        rows.append(A.T @ X); Acal.append(A); tracks.append((t, X, Q[:, :3]))
    Xs = np.vstack(rows)
    return Xs, Acal, tracks


def dense_literal(c, Xs, Acal):
    """the reference's sequence on the dense stack (= literal_core.h: literal_general)"""
    n = 6 * c["N"]; u, v, tol = c["u"], c["v"], c["tol"]
    X = Xs.copy(); m = X.shape[0]; D = 15 + n
    steps_total = min(m, D); msteps = max(steps_total - 15, 0)
    refl = []; V = []; taus = []
    for k in range(msteps):
        p = 15 + k
        tail2 = float(X[p + 1:, k] @ X[p + 1:, k]); head2 = float(X[:p + 1, k] @ X[:p + 1, k])
        zero2 = max(TINY, tol * tol * (head2 + tail2))
        if tail2 <= zero2:
            X[p + 1:, k] = 0; refl.append(False); V.append(None); taus.append(0.0); continue
        c0 = X[p, k]; beta = np.sqrt(c0 * c0 + tail2); beta = -beta if c0 >= 0 else beta
        vv = np.zeros(m); vv[p] = 1; vv[p + 1:] = X[p + 1:, k] / (c0 - beta); tk = (beta - c0) / beta
        X[:, k + 1:] -= tk * np.outer(vv, vv @ X[:, k + 1:])
        X[p, k] = beta; X[p + 1:, k] = 0
        refl.append(True); V.append(vv); taus.append(tk)
    R = X[:steps_total]
    mask = np.zeros_like(R[:, :n], dtype=bool)
    for i in range(steps_total):
        mask[i, max(i - 15, 0):] = True
    rmax = np.abs(R[:, :n][mask]).max()
    kept = [i for i in range(steps_total) if (np.abs(R[i, max(i - 15, 0):n]) > tol * rmax).any()] if tol > 0 else \
           [i for i in range(steps_total) if (R[i, max(i - 15, 0):n] != 0).any()]
    Q1 = np.zeros((m, len(kept)))
    for a, i in enumerate(kept):
        Q1[i, a] = 1
    for k in range(msteps - 1, -1, -1):
        if refl[k]:
            Q1 -= taus[k] * np.outer(V[k], V[k] @ Q1)
    Ro = sla.block_diag(*[A.T @ np.diag(np.tile([u, v], A.shape[0] // 2)) @ A for A in Acal])
    TH = Q1.T @ Xs
    Rn = Q1.T @ Ro @ Q1
    L = TH.T @ np.linalg.solve(Rn, TH)
    return dict(L=L, kept=kept, refl=refl, Q1=Q1, Ro=Ro, R=R)


def planb_dense_basis(c, Xs, ref):
    """basis-free form with the basis built from dense objects (checks the subspace claim)"""
    n = 6 * c["N"]; m = Xs.shape[0]
    kept, refl = ref["kept"], ref["refl"]
    K15 = [i for i in kept if i < 15]
    C = [k for k, f in enumerate(refl) if f]
    piv = set(15 + k for k in C)
    Kh = [i for i in kept if i >= 15 and i not in piv]
    cols = []
    for i in K15:
        e = np.zeros(m); e[i] = 1; cols.append(e)
    for k in C:
        x = Xs[:, k].copy(); x[:15] = 0; cols.append(x)
    for h in Kh:
        cols.append(ref["Q1"][:, kept.index(h)])
    Bs = np.array(cols).T
    Ms = Bs.T @ ref["Ro"] @ Bs
    Nm = Bs.T @ Xs
    return Nm.T @ np.linalg.solve(Ms, Nm), len(Kh)


def main():
    cases = pickle.load(open("/tmp/planb_cases.pkl", "rb"))
    sel = sys.argv[1] if len(sys.argv) > 1 else ""
    for c in cases:
        if sel and sel not in c["name"]:
            continue
        Xs, Acal, tracks = build_stack(c)
        ref = dense_literal(c, Xs, Acal)
        nL = np.linalg.norm(ref["L"])
        e_or = np.linalg.norm(ref["L"] - c["L_or"]) / nL if c["L_or"] is not None else float("nan")
        Lb, nh = planb_dense_basis(c, Xs, ref)
        print(f'{c["name"]:14s} m={Xs.shape[0]:5d} kept={len(ref["kept"]):3d} refl={sum(ref["refl"]):3d} steps={len(ref["refl"]):3d} extras={nh:2d}  vs oracle {e_or:.1e}  basis-free {np.linalg.norm(Lb - ref["L"]) / nL:.1e}')


if __name__ == "__main__" and (len(sys.argv) < 2 or sys.argv[1] != "compact"):
    main()


def show_extras(name):
    cases = pickle.load(open("/tmp/planb_cases.pkl", "rb"))
    for c in cases:
        if c["name"] != name:
            continue
        Xs, Acal, tracks = build_stack(c)
        ref = dense_literal(c, Xs, Acal)
        refl, kept = ref["refl"], ref["kept"]
        print("skipped steps:", [k for k, f in enumerate(refl) if not f])
        piv = set(15 + k for k, f in enumerate(refl) if f)
        print("extras rows:", [i for i in kept if i >= 15 and i not in piv])
        rows0 = np.cumsum([0] + [A.shape[1] for A in Acal])
        print("row0 of tracks:", rows0[:12], " M:", [A.shape[0] // 2 for A in Acal][:12])


# ----------------------------------------------------------------------------------------------------------------------
# Step 2: the same thing from compressed quantities only (what the device will hold): no m x n stack.
def planb_compact(c, Xs_dense=None):
    N, M, inc, slots, hx, r = c["N"], c["M"], c["inc"], c["slots"], c["hx"], c["r"]
    u, v, tol = c["u"], c["v"], c["tol"]
    n = 6 * N; n1 = n + 1; D = 15 + n; dlt = u - v
    trk = [t for t in range(len(M)) if inc[t]]
    rho = [2 * M[t] - 3 for t in trk]
    row0 = np.concatenate([[0], np.cumsum(rho)]).astype(int)
    m = int(row0[-1]); e = min(m, D); msteps = max(e - 15, 0)
    # per track: Q_f (any orthonormal basis for Gamma), A_j (pivoted Householder: the reference's basis) for explicit rows
    Lam = np.zeros((n1, n1)); Du = np.zeros((n, n)); BD = np.zeros((n, n))
    E0 = np.zeros((e, n1)); See = np.zeros((e, e)); Xe = np.zeros((e, n))
    for a, t in enumerate(trk):
        m2 = 2 * M[t]
        X = np.zeros((m2, n1)); Hf = np.zeros((m2, 3))
        for o in range(M[t]):
            for i in range(2):
                X[2 * o + i, 6 * slots[t, o]:6 * slots[t, o] + 6] = hx[t, o, 6 * i:6 * i + 6]
                Hf[2 * o + i] = -hx[t, o, 6 * i + 3:6 * i + 6]
        X[:, n] = r[t, :m2]
        Q, _, _ = sla.qr(Hf, pivoting=True)
        Qf = Q[:, :3]
        Bh = Qf.T @ X                                    # 3 x n1   (k_gram's B^)
        Lam += X.T @ X - Bh.T @ Bh                       # H_o^T H_o (k_gram)
        Xu = X[0::2, :n]; Qu = Qf[0::2]                  # u rows
        Ch = Qu.T @ Xu; W = Qu.T @ Qu
        Dh = Ch - 0.5 * W @ Bh[:, :n]
        Du += Xu.T @ Xu
        BD += Bh[:, :n].T @ Dh + Dh.T @ Bh[:, :n]
        if row0[a] < e:                                  # explicit rows of this track
            A = Q[:, 3:]
            for q in range(min(rho[a], e - row0[a])):
                i = row0[a] + q
                ai = A[:, q]
                E0[i] = ai @ X
                du = np.zeros(m2); du[0::2] = ai[0::2]
                wi = du - Qf @ (Qf.T @ du)
                Xe[i] = wi @ X[:, :n]
                for q2 in range(min(rho[a], e - row0[a])):
                    See[i, row0[a] + q2] = ai[0::2] @ A[0::2, q2]
    Gam = Du - BD
    gram = m > e
    # ---- the sweep: decisions, R rows, reflectors' explicit parts
    E = E0.copy()
    Gh = Lam - E0[:15].T @ E0[:15] if gram else None
    refl = np.zeros(msteps, dtype=bool); tau = np.zeros(msteps); dn = np.zeros(msteps); Acoef = np.zeros((msteps, n1))
    n_skip_tol = 0
    for k in range(msteps):
        p = 15 + k
        c0 = E[p, k]
        if gram:
            g = max(Gh[k, k], 0.0)
            tail2 = max(g - c0 * c0, 0.0)
            col2 = Lam[k, k]
            t2 = max(tol * tol, 1e-7)
        else:
            tail2 = float(E[p + 1:, k] @ E[p + 1:, k]); col2 = float(E[:, k] @ E[:, k]); t2 = tol * tol
        zero2 = max(TINY, t2 * col2)
        if tail2 <= zero2:
            if tail2 > TINY: n_skip_tol += 1
            E[p + 1:, k] = 0
            if gram: Gh[k:, k:] -= np.outer(E[p, k:], E[p, k:])
            continue
        refl[k] = True
        beta = np.sqrt(c0 * c0 + tail2); beta = -beta if c0 >= 0 else beta
        dn[k] = 1.0 / (c0 - beta); tau[k] = (beta - c0) / beta
        vv = E[p + 1:, k] * dn[k]
        if gram:
            Rrow = Gh[k, k:] / beta
            s = E[p, k:] - Rrow
        else:
            s = tau[k] * (E[p, k:] + vv @ E[p + 1:, k:])
            Rrow = E[p, k:] - s
        E[p + 1:, k + 1:] -= np.outer(vv, s[1:])
        E[p, k:] = Rrow; E[p, k] = beta
        E[p + 1:, k] = vv                                 # reflector's explicit part, in place
        Acoef[k, k + 1:] = s[1:] * dn[k]
        if gram: Gh[k:, k:] -= np.outer(Rrow, Rrow)
    # ---- kept rows
    R = E.copy()
    for k in range(msteps):
        R[15 + k + 1:, k] = 0
    rmax = 0.0
    for i in range(e):
        rmax = max(rmax, np.abs(R[i, max(i - 15, 0):n]).max())
    kept = [i for i in range(e) if ((np.abs(R[i, max(i - 15, 0):n]) > tol * rmax).any() if tol > 0 else (R[i, max(i - 15, 0):n] != 0).any())]
    K15 = [i for i in kept if i < 15]
    C = [k for k in range(msteps) if refl[k]]
    piv = set(15 + k for k in C)
    Kh = [i for i in kept if i >= 15 and i not in piv]
    # ---- extras: q_h = H_.. e_h in coordinates [t ; B0 y]
    Th = np.zeros((e, len(Kh))); Yh = np.zeros((n, len(Kh)))
    if Kh:
        kmax = max(Kh) - 15
        Y = np.zeros((n, kmax)); Gb0 = (Lam[:n, :n] - E0[:, :n].T @ E0[:, :n]) if gram else np.zeros((n, n))
        for j in range(kmax):
            if not refl[j]: continue
            Y[j, j] = 1.0
            for i in range(j):
                if refl[i]: Y[:, j] -= Y[:, i] * Acoef[i, j]
        for a, h in enumerate(Kh):
            t = np.zeros(e); t[h] = 1.0; y = np.zeros(n)
            for j in range(h - 15 - 1, -1, -1):
                if not refl[j]: continue
                pj = 15 + j
                ve = np.zeros(e); ve[pj] = 1.0; ve[pj + 1:] = E[pj + 1:, j]
                yv = Y[:, j] * dn[j]
                al = tau[j] * (ve @ t + yv @ (Gb0 @ y))
                t -= al * ve; y -= al * yv
            Th[:, a] = t; Yh[:, a] = y
    # ---- assembly of M_S = Bs^T R_o Bs and N = Bs^T A for Bs = [e_i | x'_c | q_h]
    na, nb_, nh = len(K15), len(C), len(Kh); nr = na + nb_ + nh
    Ttil = np.zeros((e, nr)); Yt = np.zeros((n, nr))
    for a, i in enumerate(K15): Ttil[i, a] = 1.0
    for a, cc in enumerate(C): Ttil[:15, na + a] = -E0[:15, cc]; Yt[cc, na + a] = 1.0
    for a in range(nh): Ttil[:, na + nb_ + a] = Th[:, a] - E0[:, :n] @ Yh[:, a]; Yt[:, na + nb_ + a] = Yh[:, a]
    Om = Ttil.T @ See @ Ttil + Ttil.T @ Xe @ Yt + (Ttil.T @ Xe @ Yt).T + Yt.T @ Gam @ Yt
    G0 = Lam - E0[:15].T @ E0[:15]
    GB = np.zeros((nr, nr)); Nm = np.zeros((nr, n1))
    GB[:na, :na] = np.eye(na)
    GB[na:na + nb_, na:na + nb_] = G0[np.ix_(C, C)]
    for a, h in enumerate(Kh):
        GB[na + nb_ + a, na + nb_ + a] = 1.0
        GB[na + nb_ + a, na:na + nb_] = R[h, C]; GB[na:na + nb_, na + nb_ + a] = R[h, C]
        Nm[na + nb_ + a] = R[h]
    Nm[:na] = E0[K15]; Nm[na:na + nb_] = G0[C]
    Ms = v * GB + dlt * Om
    L = Nm.T @ np.linalg.solve(Ms, Nm)
    return dict(L=L, kept=kept, refl=refl, nh=nh, skip_tol=n_skip_tol)


def main2():
    cases = pickle.load(open("/tmp/planb_cases.pkl", "rb"))
    sel = sys.argv[2] if len(sys.argv) > 2 else ""
    for c in cases:
        if sel and sel not in c["name"]:
            continue
        Xs, Acal, tracks = build_stack(c)
        ref = dense_literal(c, Xs, Acal)
        pb = planb_compact(c)
        nL = np.linalg.norm(ref["L"])
        same = (pb["kept"] == ref["kept"]) and (list(pb["refl"]) == list(ref["refl"]))
        print(f'{c["name"]:14s} m={Xs.shape[0]:5d} kept={len(ref["kept"]):3d} extras={pb["nh"]:2d} same decisions={same}  compact-vs-dense {np.linalg.norm(pb["L"] - ref["L"]) / nL:.1e}')


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "compact":
    main2()
