"""Seeded synthetic IMU + feature-track scenarios for the batched MSCKF core.

The reference consumes (a) IMU samples `imuReading{omega, a, dT}` (types.h:78-84, produced at 200 Hz by
datasets/asl_readers.cpp:181-206) and (b) per-image lists of undistorted *normalized* feature
coordinates with ids (corner_detector.cpp:320-439 -> MSCKF::update/addFeatures, msckf.h:215,302).  No
EuRoC data exists in this image, so this module generates both from an analytic trajectory:

  p(t) = [3 cos 0.4t, 3 sin 0.4t, 0.5 sin 0.8t] m, camera looking radially outwards (good parallax),
  small roll/pitch wobble, EuRoC cam0 extrinsics (euroc/MH_03_kalibr.yaml:5-9), 200 Hz IMU, 20 Hz camera.

Everything is driven by a counter-based splitmix64 stream so that a (config, trajectory) pair always
yields the same data on any machine: seed = 0x5EED0000 + 1000*config + trajectory.

Steady-state window (what BASELINE.json's configs are quoted on): at update time the filter holds
exactly N camera states; each frame F tracks end (last observation in the previous frame, slot N-2) and
span slots [N-1-M_j, N-2] with M_j = 3 + (rng mod (N-3)); after the update the oldest state is pruned.
"""
import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)
_GAMMA = np.uint64(0x9E3779B97F4A7C15)

# Kalibr T_cam_imu for EuRoC cam0 (reference euroc/MH_03_kalibr.yaml:5-9)
T_CAM_IMU = np.array([
    [0.0148655429818, -0.999880929698, 0.00414029679422, -0.021640145497],
    [0.999557249008, 0.0149672133247, 0.025715529948, -0.064676986768],
    [-0.0257744366974, 0.00375618835797, 0.999660727178, 0.009810730590],
    [0.0, 0.0, 0.0, 1.0]])
EUROC_INTRINSICS = (458.654, 457.296, 367.215, 248.375)  # f_u f_v c_u c_v (MH_03_kalibr.yaml:13)
GRAVITY = np.array([0.0, 0.0, -9.81])                    # datasets/asl_msckf.cpp:154
IMU_RATE, CAM_RATE = 200, 20
IMU_PER_FRAME = IMU_RATE // CAM_RATE


class SplitMix64:
    """Counter-based splitmix64: value i of the stream is mix(seed + (i+1)*gamma)."""

    def __init__(self, seed):
        self.seed = np.uint64(int(seed) & 0xFFFFFFFFFFFFFFFF)
        self.ctr = 0

    def u64(self, n):
        with np.errstate(over="ignore"):
            i = np.arange(self.ctr + 1, self.ctr + n + 1, dtype=np.uint64)
            z = self.seed + i * _GAMMA
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            z = z ^ (z >> np.uint64(31))
        self.ctr += n
        return z

    def uniform(self, n):
        return (self.u64(n) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)

    def normal(self, n):
        m = (n + 1) // 2
        u1 = 1.0 - self.uniform(m)
        u2 = self.uniform(m)
        r = np.sqrt(-2.0 * np.log(u1))
        return np.concatenate([r * np.cos(2 * np.pi * u2), r * np.sin(2 * np.pi * u2)])[:n]

    def integers(self, n, mod):
        return (self.u64(n) % np.uint64(mod)).astype(np.int64)


def rot_to_quat(R):
    """Rotation matrix -> (w,x,y,z) such that Eigen's toRotationMatrix() gives R back."""
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
        q = np.zeros(4)
        q[0] = (R[k, j] - R[j, k]) / s
        q[1 + i] = 0.25 * s
        q[1 + j] = (R[j, i] + R[i, j]) / s
        q[1 + k] = (R[k, i] + R[i, k]) / s
    if q[0] < 0:
        q = -q
    return q / np.linalg.norm(q)


def quat_to_rot(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def camera_extrinsics():
    """(q_CI, p_C_I): C_CI rotates IMU-frame vectors into the camera frame; p_C_I = camera origin in IMU frame."""
    R = T_CAM_IMU[:3, :3]
    # re-orthonormalise the 12-digit Kalibr matrix
    U, _, Vt = np.linalg.svd(R)
    R = U @ Vt
    return rot_to_quat(R), -R.T @ T_CAM_IMU[:3, 3]


def filter_config(N, isotropic=True, feature_px=7.0, gn_px=7.0):
    """Filter parameters: the effective EuRoC set of SURVEY.md section 5 with max_cam_states = N-1."""
    f_u, f_v, c_u, c_v = EUROC_INTRINSICS
    if isotropic:
        f_v = f_u
    q_CI, p_C_I = camera_extrinsics()
    w_var, dbg_var, a_var, dba_var = 1e-4, 3.6733e-5, 1e-2, 7e-2
    return dict(
        c_u=c_u, c_v=c_v, f_u=f_u, f_v=f_v, b=0.0, q_CI=q_CI, p_C_I=p_C_I,
        u_var_prime=(feature_px / f_u) ** 2, v_var_prime=(feature_px / f_v) ** 2,
        Q_imu_diag=[w_var] * 3 + [dbg_var] * 3 + [a_var] * 3 + [dba_var] * 3,
        P0_diag=[1e-5] * 3 + [1e-2] * 3 + [1e-2] * 3 + [1e-2] * 3 + [1e-12] * 3,
        max_gn_cost_norm=(gn_px / f_u) ** 2, min_rcond=3e-12, translation_threshold=0.1,
        redundancy_angle_thresh=0.005, redundancy_distance_thresh=0.05,
        min_track_length=3, max_track_length=1000, max_cam_states=N - 1)


# ------------------------------------------------------------------ analytic trajectory
# path_id -> (radius [m], turn rate [rad/s], vertical amplitude [m], vertical rate [rad/s], roll/pitch wobble [rad], wobble rate)
# 0 is the SURVEY 8d path every config uses; 1..4 are the other "sequences" of the cfg4 Monte-Carlo stand-in
# (BASELINE.json configs[3] names EuRoC MH_01..MH_05, which are not on disk).
PATHS = [(3.0, 0.4, 0.5, 0.8, 0.10, 0.7), (4.0, 0.3, 0.8, 0.5, 0.08, 0.5), (2.5, 0.5, 0.3, 1.0, 0.12, 0.9),
         (3.5, 0.35, 0.6, 0.7, 0.05, 0.6), (5.0, 0.25, 1.0, 0.4, 0.10, 0.8)]


def _path(t, path_id=0):
    R, w, Az, wz, _, _ = PATHS[path_id]
    t = np.asarray(t, dtype=np.float64)
    p = np.stack([R * np.cos(w * t), R * np.sin(w * t), Az * np.sin(wz * t)], -1)
    v = np.stack([-R * w * np.sin(w * t), R * w * np.cos(w * t), Az * wz * np.cos(wz * t)], -1)
    a = np.stack([-R * w * w * np.cos(w * t), -R * w * w * np.sin(w * t), -Az * wz * wz * np.sin(wz * t)], -1)
    return p, v, a


def _attitude(t, path_id=0):
    """R_GI(t) [..,3,3] (IMU axes in world) and body angular rate omega(t)."""
    _, w0, _, _, wob, wr = PATHS[path_id]
    t = np.asarray(t, dtype=np.float64)
    th = w0 * t
    o = np.stack([np.cos(th), np.sin(th), np.zeros_like(th)], -1)       # outward  -> z_I (camera axis)
    up = np.broadcast_to(np.array([0.0, 0.0, 1.0]), o.shape)             # up       -> x_I
    tg = np.stack([-np.sin(th), np.cos(th), np.zeros_like(th)], -1)     # tangent  -> -y_I
    Rn = np.stack([up, -tg, o], -1)
    al, dal = wob * np.sin(wr * t), wob * wr * np.cos(wr * t)
    be, dbe = wob * np.sin(wr * t + 1.0), wob * wr * np.cos(wr * t + 1.0)
    ca, sa, cb, sb = np.cos(al), np.sin(al), np.cos(be), np.sin(be)
    z, one = np.zeros_like(t), np.ones_like(t)
    Rx = np.stack([np.stack([one, z, z], -1), np.stack([z, ca, -sa], -1), np.stack([z, sa, ca], -1)], -2)
    Ry = np.stack([np.stack([cb, z, sb], -1), np.stack([z, one, z], -1), np.stack([-sb, z, cb], -1)], -2)
    R = Rn @ Rx @ Ry
    w = w0 + dal
    omega = np.stack([cb * w, dbe, sb * w], -1)
    return R, omega


def ground_truth(t, path_id=0):
    """dict of p, v, q_IG (w,x,y,z), R_GI at times t."""
    p, v, a = _path(t, path_id)
    R, om = _attitude(t, path_id)
    R2 = R.reshape(-1, 3, 3)
    q = np.stack([rot_to_quat(Ri.T) for Ri in R2]).reshape(np.shape(t) + (4,))
    return dict(p=p, v=v, a=a, R_GI=R, omega=om, q_IG=q)


def imu29_from_gt(gt_at_t, b_g, b_a):
    """Pack the 29-double IMU state layout used across the C-ABIs: q_IG b_g v b_a p g q_null v_null p_null."""
    q, v, p = gt_at_t["q_IG"], gt_at_t["v"], gt_at_t["p"]
    return np.concatenate([q, b_g, v, b_a, p, GRAVITY, q, v, p])


class Trajectory:
    """One seeded trajectory: IMU stream + per-frame ending-track work-lists."""

    def __init__(self, config_id, traj_idx, N, F, n_frames, cfg=None, t0=0.0, imu_noise_scale=0.05,
                 obs_noise_px=0.5, dense_tracks=False, first_timed_window_only=False, depth_range=(2.0, 10.0), path_id=0):
        self.N, self.F, self.n_frames = N, F, n_frames
        self.depth_range = depth_range   # landmark depth in the mid-track camera; (2, 10) m = SURVEY 8d, larger = low parallax
        self.t0 = t0
        self.cfg = cfg if cfg is not None else filter_config(N)
        self.seed = 0x5EED0000 + 1000 * config_id + traj_idx
        rng = SplitMix64(self.seed)
        self.dT = 1.0 / IMU_RATE
        self.b_g = np.full(3, 0.002)
        self.b_a = np.full(3, 0.02)
        # IMU sample i covers [t0 + i dT, t0 + (i+1) dT); frame k is taken after IMU samples [10(k-1)+... ]
        n_imu = n_frames * IMU_PER_FRAME
        ti = t0 + np.arange(n_imu) * self.dT
        self.path_id = path_id
        gt = ground_truth(ti + 0.5 * self.dT, path_id)            # mid-interval sampling
        q = self.cfg["Q_imu_diag"]
        sg = imu_noise_scale * np.sqrt(q[0] / self.dT)
        sa = imu_noise_scale * np.sqrt(q[6] / self.dT)
        C_IG = np.swapaxes(gt["R_GI"], -1, -2)
        a_body = np.einsum("nij,nj->ni", C_IG, gt["a"] - GRAVITY)
        om = gt["omega"] + self.b_g + sg * rng.normal(3 * n_imu).reshape(n_imu, 3)
        ac = a_body + self.b_a + sa * rng.normal(3 * n_imu).reshape(n_imu, 3)
        self.readings = np.concatenate([om, ac, np.full((n_imu, 1), self.dT)], 1)   # [n_imu, 7]
        # frame k happens at time t0 + (k+1)*10*dT, i.e. after IMU samples [10k, 10k+10)
        self.frame_times = t0 + (np.arange(n_frames) + 1) * IMU_PER_FRAME * self.dT
        self.gt_frames = ground_truth(self.frame_times, path_id)
        self.gt0 = ground_truth(np.array(t0), path_id)
        self.imu0 = imu29_from_gt(self.gt0, self.b_g, self.b_a)
        q_CI, p_C_I = self.cfg["q_CI"], self.cfg["p_C_I"]
        C_CI = quat_to_rot(q_CI)
        R_GI = self.gt_frames["R_GI"]
        self.C_CG = np.einsum("ij,nkj->nik", C_CI, R_GI)                 # C_CI * R_GI^T
        self.p_C = self.gt_frames["p"] + np.einsum("nij,j->ni", R_GI, p_C_I)
        sig_obs = obs_noise_px / self.cfg["f_u"]
        self.frames = []
        self.landmarks = []
        for k in range(n_frames):
            Nw = min(k + 1, N)
            if Nw < 4 or F == 0:
                self.frames.append(dict(Nw=Nw, M=np.zeros(0, np.int32), slots=np.zeros(0, np.int32), obs=np.zeros((0, 2))))
                self.landmarks.append(np.zeros((0, 3)))
                continue
            if dense_tracks:
                M = np.full(F, Nw - 1, dtype=np.int64)
                rng.u64(F)
            else:
                M = 3 + rng.integers(F, Nw - 3)
            pts = self._sample_landmarks(rng, k, M)
            Mmax = Nw - 1
            ar = np.arange(Mmax)[None, :]
            mask = ar < M[:, None]
            fr = np.clip(k - M[:, None] + ar, 0, k - 1)               # frames k-M .. k-1 (padded)
            pc = np.einsum("fmij,fmj->fmi", self.C_CG[fr], pts[:, None, :] - self.p_C[fr])
            z = pc[..., :2] / pc[..., 2:3] + sig_obs * rng.normal(2 * F * Mmax).reshape(F, Mmax, 2)
            sl = (Nw - 1 - M[:, None]) + ar
            self.frames.append(dict(Nw=Nw, M=M.astype(np.int32), slots=sl[mask].astype(np.int32), obs=z[mask]))
            self.landmarks.append(pts)

    def _sample_landmarks(self, rng, k, M):
        """One landmark per track, visible (90 deg FOV, in front) in frames k-M_j..k-1, depth_range (default 2-10 m) deep."""
        F = len(M)
        pts = np.zeros((F, 3))
        todo = np.arange(F)
        spread = 0.6
        for attempt in range(400):
            n = len(todo)
            u = rng.uniform(3 * n).reshape(n, 3)
            mid = k - 1 - (M[todo] // 2)
            ax, ay = spread * (2 * u[:, 0] - 1), spread * (2 * u[:, 1] - 1)
            d = self.depth_range[0] + (self.depth_range[1] - self.depth_range[0]) * u[:, 2]
            pc = np.stack([d * np.tan(ax), d * np.tan(ay), d], -1)
            pw = np.einsum("nji,nj->ni", self.C_CG[mid], pc) + self.p_C[mid]
            Mt = M[todo]
            ar = np.arange(int(Mt.max()))[None, :]
            mask = ar < Mt[:, None]
            fr = np.clip(k - Mt[:, None] + ar, 0, k - 1)
            q = np.einsum("fmij,fmj->fmi", self.C_CG[fr], pw[:, None, :] - self.p_C[fr])
            vis = (q[..., 2] > 0.5) & (np.abs(q[..., 0]) < q[..., 2]) & (np.abs(q[..., 1]) < q[..., 2])
            ok = np.all(vis | ~mask, axis=1)
            pts[todo[ok]] = pw[ok]
            todo = todo[~ok]
            if len(todo) == 0:
                break
            if attempt % 8 == 7:
                spread = max(0.7 * spread, 0.05)
            if attempt % 40 == 39:
                # the camera sweeps ~23 deg/s: a landmark cannot stay inside a 90 deg field of view for much
                # more than ~2 s, so over-long tracks (windows > 40 frames) are shortened (in place)
                M[todo] = np.maximum(3, (3 * M[todo]) // 4)
                spread = 0.6
        if len(todo):
            raise RuntimeError("landmark sampling failed")
        return pts

    # ---- views
    def imu_for_frame(self, k):
        return self.readings[k * IMU_PER_FRAME:(k + 1) * IMU_PER_FRAME]

    def stream(self):
        """Per-frame (cur_obs, cur_ids, new_obs, new_ids) as the front-end would hand them to
        MSCKF::update / addFeatures (asl_msckf.cpp:252-279).  Track t of frame k' gets id 10000*k' + t."""
        per_frame = [dict(cur=([], []), new=([], [])) for _ in range(self.n_frames)]
        for kp, fr in enumerate(self.frames):
            o = 0
            for t, M in enumerate(fr["M"]):
                fid = 10000 * kp + t
                for i in range(M):
                    k = kp - M + i
                    kind = "new" if i == 0 else "cur"
                    per_frame[k][kind][0].append(fr["obs"][o + i])
                    per_frame[k][kind][1].append(fid)
                o += M
        return per_frame
