"""Independent numpy/scipy restatement of the reference filter -- TEST INFRASTRUCTURE ONLY.

Second, deliberately different implementation of /root/reference/include/msckf_mono/msckf.h used to pin
the C++ oracle (oracle/msckf_oracle.hpp) a second time: the reference itself ships no tests or golden vectors
(SURVEY.md section 8c); the first pin is the reference's own msckf.h compiled against oracle/ref_shim
(oracle/_ref/lib_ref.so, tests/test_ref_vs_oracle.py).  Where the C++ oracle uses
hand-written Householder/LDLT/LU/Pade, this file uses scipy.linalg.expm / qr / svd / numpy solve, dense
matrices everywhere, and scipy.stats.chi2.ppf for the gate table (msckf.h:91-95).  It also generates the
golden fixtures in tests/golden (scripts/gen_golden.py).

Null space (msckf.h:954-955): `nullspace="svd"` takes the last 2M-3 columns of the full U of an SVD, as
the reference does with JacobiSVD; `nullspace="householder"` takes them from scipy's full QR of H_f_j (the
C++ oracle's documented choice).  For u_var' == v_var' both give the same update.
"""
import numpy as np
import scipy.linalg as sla
from scipy.stats import chi2


def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]], dtype=float)


def q2R(q):  # Eigen toRotationMatrix, q = (w,x,y,z)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def qmul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                     aw * by + ay * bw + az * bx - ax * bz, aw * bz + az * bw + ax * by - ay * bx])


def qinv(q):
    return np.array([q[0], -q[1], -q[2], -q[3]]) / np.dot(q, q)


def qrot(q, v):  # Eigen _transformVector
    u = np.asarray(q[1:])
    uv = 2 * np.cross(u, v)
    return v + q[0] * uv + np.cross(u, uv)


def update_quat(dtheta):  # msckf.h:851-872
    dq = 0.5 * np.asarray(dtheta)
    cs = dq @ dq
    q = np.array([1.0 if cs > 1 else np.sqrt(1 - cs), -dq[0], -dq[1], -dq[2]])
    return q / np.linalg.norm(q)


class NpMSCKF:
    def __init__(self, cfg, imu29, nullspace="svd"):
        self.cfg = cfg
        self.nullspace = nullspace
        x = np.asarray(imu29, dtype=float)
        self.q, self.bg, self.v, self.ba, self.p, self.g = x[0:4].copy(), x[4:7].copy(), x[7:10].copy(), x[10:13].copy(), x[13:16].copy(), x[16:19].copy()
        self.q_null, self.v_null, self.p_null = self.q.copy(), self.v.copy(), self.p.copy()
        self.P = np.diag(np.asarray(cfg["P0_diag"], dtype=float))
        self.Q = np.diag(np.asarray(cfg["Q_imu_diag"], dtype=float))
        self.cams = []      # dicts: q, p, id, feats(list)
        self.tracks = []    # dicts: id, obs(list), cam_ids(list), initialized, p_f_G
        self.tracked_ids = []
        self.to_resid = []
        self.n_resid = 0
        self.chi = [chi2.ppf(0.05, i) for i in range(1, 100)]
        self.q_CI, self.p_C_I = np.asarray(cfg["q_CI"], float), np.asarray(cfg["p_C_I"], float)
        self.last = {}
        self.map = []

    # ---- msckf.h:101-145
    def propagate(self, rd):
        om, a, dT = np.asarray(rd[0:3], float), np.asarray(rd[3:6], float), float(rd[6])
        C = q2R(self.q)
        wh, ah = om - self.bg, a - self.ba
        F = np.zeros((15, 15))
        F[0:3, 0:3] = -skew(wh); F[0:3, 3:6] = -np.eye(3)
        F[6:9, 0:3] = -C.T @ skew(ah); F[6:9, 9:12] = -C.T; F[12:15, 6:9] = np.eye(3)
        G = np.zeros((15, 12))
        G[0:3, 0:3] = -np.eye(3); G[3:6, 3:6] = np.eye(3); G[6:9, 6:9] = -C.T; G[9:12, 9:12] = np.eye(3)
        # RK (msckf.h:1425-1467)
        Om = np.zeros((4, 4)); Om[0:3, 0:3] = -skew(wh); Om[0:3, 3] = wh; Om[3, 0:3] = -wh
        Om *= 0.5
        y0 = np.array([-self.q[1], -self.q[2], -self.q[3], self.q[0]])
        k0 = Om @ y0
        k1 = Om @ (y0 + (k0 / 4.) * dT)
        k2 = Om @ (y0 + (k0 / 8. + k1 / 8.) * dT)
        k3 = Om @ (y0 + (-k1 / 2. + k2) * dT)
        k4 = Om @ (y0 + (k0 * 3. / 16. + k3 * 9. / 16.) * dT)
        k5 = Om @ (y0 + (-k0 * 3. / 7. + k1 * 2. / 7. + k2 * 12. / 7. - k3 * 12. / 7. + k4 * 8. / 7.) * dT)
        yt = y0 + (7. * k0 + 32. * k2 + 12. * k3 + 32. * k4 + 7. * k5) * dT / 90.
        qn = np.array([yt[3], -yt[0], -yt[1], -yt[2]]); qn /= np.linalg.norm(qn)
        vn = self.v + (C.T @ ah + self.g) * dT
        pn = self.p + self.v * dT
        Phi = sla.expm(F * dT)
        Rk = q2R(self.q_null)
        Phi[0:3, 0:3] = q2R(qn) @ Rk.T
        u = Rk @ self.g
        s = u / (u @ u)
        A1 = Phi[6:9, 0:3].copy()
        w1 = skew(self.v_null - vn) @ self.g
        Phi[6:9, 0:3] = A1 - np.outer(A1 @ u - w1, s)
        A2 = Phi[12:15, 0:3].copy()
        w2 = skew(dT * self.v_null + self.p_null - pn) @ self.g
        Phi[12:15, 0:3] = A2 - np.outer(A2 @ u - w2, s)
        Pii = Phi @ (self.P[:15, :15] + G @ self.Q @ G.T * dT) @ Phi.T
        self.q, self.v, self.p = qn, vn, pn
        self.q_null, self.v_null, self.p_null = qn.copy(), vn.copy(), pn.copy()
        self.P[:15, :15] = (Pii + Pii.T) / 2
        if self.P.shape[0] > 15:
            self.P[:15, 15:] = Phi @ self.P[:15, 15:]
            self.P[15:, :15] = self.P[:15, 15:].T

    # ---- msckf.h:148-212
    def augment(self, state_id):
        self.map = []
        qc = qmul(self.q_CI, self.q); qc /= np.linalg.norm(qc)
        pc = self.p + qrot(qinv(self.q), self.p_C_I)
        D = self.P.shape[0]
        J = np.zeros((6, D))
        J[0:3, 0:3] = q2R(self.q_CI)
        J[3:6, 0:3] = skew(qrot(qinv(self.q), self.p_C_I))
        J[3:6, 12:15] = np.eye(3)
        T = np.vstack([np.eye(D), J])
        Pa = T @ self.P @ T.T
        self.P = (Pa + Pa.T) / 2
        self.cams.append(dict(q=qc, p=pc, id=state_id, feats=[]))

    # ---- msckf.h:215-299 / 302-332 / 1469-1485
    def update(self, meas, ids):
        ids = list(ids)
        self.to_resid = []
        remove = []
        for n, fid in enumerate(list(self.tracked_ids)):
            tr = self.tracks[n]
            valid = fid in ids
            if valid:
                tr["obs"].append(np.asarray(meas[ids.index(fid)], float))
                self.cams[-1]["feats"].append(fid)
                tr["cam_ids"].append(self.cams[-1]["id"])
            if (not valid) or len(tr["obs"]) >= self.cfg["max_track_length"]:
                slots = []
                for ci, c in enumerate(self.cams):
                    if fid in c["feats"]:
                        c["feats"].remove(fid); slots.append(ci)
                if len(slots) >= self.cfg["min_track_length"]:
                    self.to_resid.append(dict(id=fid, obs=[o.copy() for o in tr["obs"]], slots=slots))
                remove.append(fid)
        for fid in remove:
            i = self.tracked_ids.index(fid)
            del self.tracks[i]; del self.tracked_ids[i]

    def add_features(self, meas, ids):
        for m, fid in zip(meas, ids):
            if fid in self.tracked_ids:
                return
            self.tracks.append(dict(id=fid, obs=[np.asarray(m, float)], cam_ids=[self.cams[-1]["id"]]))
            self.cams[-1]["feats"].append(fid)
            self.tracked_ids.append(fid)

    def set_tracks(self, M, slots, obs):
        self.to_resid = []
        o = 0
        for t, m in enumerate(M):
            self.to_resid.append(dict(id=t, obs=[np.asarray(obs[o + i], float) for i in range(m)], slots=[int(s) for s in slots[o:o + m]]))
            o += m

    # ---- msckf.h:980-1025
    def check_motion(self, z0, cams):
        if len(cams) < 2:
            return False
        d = np.array([z0[0], z0[1], 1.0]); d /= np.linalg.norm(d)
        d = q2R(cams[0]["q"]).T @ d
        best = 0.0
        for c in cams[1:]:
            t = c["p"] - cams[0]["p"]
            best = max(best, np.linalg.norm(t - (t @ d) * d))
        return best > self.cfg["translation_threshold"]

    # ---- msckf.h:1147-1285
    def triangulate(self, cams, obs):
        C0, p0 = q2R(cams[0]["q"]), cams[0]["p"]
        Rs = [q2R(c["q"]) @ C0.T for c in cams]
        ts = [q2R(c["q"]) @ (p0 - c["p"]) for c in cams]

        def h(x, i):
            return Rs[i] @ np.array([x[0], x[1], 1.0]) + x[2] * ts[i]

        def cost(x):
            e = 0.0
            for i in range(len(cams)):
                hh = h(x, i)
                e += np.sum((hh[:2] / hh[2] - obs[i]) ** 2)
            return e
        m = Rs[-1] @ np.array([obs[0][0], obs[0][1], 1.0])
        A = np.array([m[0] - obs[-1][0] * m[2], m[1] - obs[-1][1] * m[2]])
        b = np.array([obs[-1][0] * ts[-1][2] - ts[-1][0], obs[-1][1] * ts[-1][2] - ts[-1][1]])
        depth = (A @ b) / (A @ A)
        x = np.array([obs[0][0], obs[0][1], 1.0 / depth])
        lam, total = 1e-3, cost(x)
        outer = 0
        while True:
            AA, bb = np.zeros((3, 3)), np.zeros(3)
            for i in range(len(cams)):
                hh = h(x, i)
                W = np.column_stack([Rs[i][:, 0], Rs[i][:, 1], ts[i]])
                J = np.vstack([W[0] / hh[2] - hh[0] / hh[2] ** 2 * W[2], W[1] / hh[2] - hh[1] / hh[2] ** 2 * W[2]])
                r = hh[:2] / hh[2] - obs[i]
                e = np.linalg.norm(r)
                w = 1.0 if e <= 0.01 else 0.01 / (2 * e)
                AA += w * w * J.T @ J
                bb += w * w * J.T @ r
            inner = 0
            while True:
                delta = np.linalg.solve(AA + lam * np.eye(3), bb)
                xn = x - delta
                dn = np.linalg.norm(delta)
                nc = cost(xn)
                if nc < total:
                    reduced, x, total = True, xn, nc
                    lam = max(lam / 10, 1e-10)
                else:
                    reduced = False
                    lam = min(lam * 10, 1e12)
                go = inner < 10 and not reduced
                inner += 1
                if not go:
                    break
            go = outer < 10 and dn > 5e-7
            outer += 1
            if not go:
                break
        fin = np.array([x[0] / x[2], x[1] / x[2], 1 / x[2]])
        valid = all((Rs[i] @ fin + ts[i])[2] > 0 for i in range(len(cams)))
        if total / (2 * len(cams) ** 2) > self.cfg["max_gn_cost_norm"]:
            valid = False
        return C0.T @ fin + p0, valid

    # ---- msckf.h:905-978
    def jac(self, p_f, slots, obs):
        M, D = len(slots), self.P.shape[0]
        Hf, Hx, r = np.zeros((2 * M, 3)), np.zeros((2 * M, D)), np.zeros(2 * M)
        for c, s in enumerate(slots):
            cam = self.cams[s]
            C = q2R(cam["q"])
            pc = C @ (p_f - cam["p"])
            X, Y, Z = pc
            Ji = np.array([[1, 0, -X / Z], [0, 1, -Y / Z]]) / Z
            A = np.hstack([Ji @ skew(pc), -Ji @ C])
            u = np.concatenate([C @ self.g, skew(p_f - cam["p"]) @ self.g])
            H = A - np.outer(A @ u, u) / (u @ u)
            Hf[2 * c:2 * c + 2] = -H[:, 3:6]
            Hx[2 * c:2 * c + 2, 15 + 6 * s:21 + 6 * s] = H
            r[2 * c:2 * c + 2] = obs[c] - pc[:2] / Z
        if self.nullspace == "svd":
            U = np.linalg.svd(Hf, full_matrices=True)[0]
        else:
            U = sla.qr(Hf, mode="full")[0]
        A_j = U[:, 3:]
        return A_j.T @ Hx, A_j.T @ r, A_j

    # ---- msckf.h:336-449, 1103-1124
    def marginalize(self):
        self.last = dict(n_tracks=len(self.to_resid), tracks=[])
        if not self.to_resid:
            return
        Hs, rs, Rs = [], [], []
        good = []
        for tr in self.to_resid:
            cams = [self.cams[s] for s in tr["slots"]]
            info = dict(motion_ok=1, tri_valid=0, gate_pass=0, gamma=0.0, p_f_G=np.zeros(3))
            self.last["tracks"].append(info)
            if self.n_resid > 3 and not self.check_motion(tr["obs"][0], cams):
                info["motion_ok"] = 0
                continue
            p_f, valid = self.triangulate(cams, tr["obs"])
            info["p_f_G"], info["tri_valid"] = p_f, int(valid)
            if valid:
                self.map.append(p_f)
                self.n_resid += 1
                good.append((tr, p_f, info))
        uvar, vvar = self.cfg["u_var_prime"], self.cfg["v_var_prime"]
        for tr, p_f, info in good:
            Ho, ro, A_j = self.jac(p_f, tr["slots"], tr["obs"])
            Rj = np.diag(np.tile([uvar, vvar], len(tr["slots"])))
            S = Ho @ self.P @ Ho.T + uvar * np.eye(Ho.shape[0])
            gamma = ro @ np.linalg.solve(S, ro)
            info["gamma"] = gamma
            if gamma < self.chi[len(tr["slots"])]:      # table[dof+1], dof = M-1  (msckf.h:433,1117)
                info["gate_pass"] = 1
                Hs.append(Ho); rs.append(ro); Rs.append(A_j.T @ Rj @ A_j)
        if not Hs:
            return
        self.measurement_update(np.vstack(Hs), np.concatenate(rs), sla.block_diag(*Rs))

    # ---- msckf.h:1325-1423
    def measurement_update(self, H, r, R):
        m, D = H.shape
        self.last["m_rows"] = m
        if m == 0:
            return
        Q, Rq = sla.qr(H, mode="full")
        Rq = np.triu(Rq)
        nz = np.any(Rq != 0, axis=1)
        T_H, Q1 = Rq[nz], Q[:, :Rq.shape[0]][:, nz]
        self.last["r_rows"] = int(nz.sum())
        r_n, R_n = Q1.T @ r, Q1.T @ R @ Q1
        P = self.P
        K = P @ T_H.T @ np.linalg.inv(T_H @ P @ T_H.T + R_n)
        dx = K @ r_n
        self.last["dx"] = dx
        self.q = qmul(update_quat(dx[0:3]), self.q)
        self.bg += dx[3:6]; self.v += dx[6:9]; self.ba += dx[9:12]; self.p += dx[12:15]
        for i, c in enumerate(self.cams):
            q = qmul(update_quat(dx[15 + 6 * i:18 + 6 * i]), c["q"])
            c["q"] = q / np.linalg.norm(q)
            c["p"] = c["p"] + dx[18 + 6 * i:21 + 6 * i]
        A = np.eye(D) - K @ T_H
        Pc = A @ P @ A.T + K @ R_n @ K.T
        self.P = (Pc + Pc.T) / 2

    # ---- msckf.h:685-761
    def prune_empty(self):
        mx, num = self.cfg["max_cam_states"], len(self.cams)
        if num < mx or self.cams[0]["feats"]:
            return
        last = num - mx - 1
        for i in range(1, num - mx):
            if self.cams[i]["feats"]:
                last = i - 1
                break
        if last >= 0:
            self.drop_oldest(last + 1)

    def drop_oldest(self, n):
        keep = np.r_[0:15, 15 + 6 * n:self.P.shape[0]]
        self.P = self.P[np.ix_(keep, keep)]
        self.cams = self.cams[n:]

    def imu29(self):
        return np.concatenate([self.q, self.bg, self.v, self.ba, self.p, self.g, self.q_null, self.v_null, self.p_null])

    def cam_array(self):
        return np.array([np.concatenate([c["q"], c["p"]]) for c in self.cams]).reshape(-1, 7)
