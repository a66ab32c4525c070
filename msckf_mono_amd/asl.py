"""ASL (EuRoC `mav0`) dataset layout without ROS / OpenCV: reader, writer, deterministic runner, ATE.

Reference counterparts:
  * CSV / sensor.yaml formats     datasets/asl_readers.cpp:12-75 (cam0), :141-206 (imu0: t,wx,wy,wz,ax,ay,az),
                                  :244-342 (ground truth: t,p(3),q(w,x,y,z),v(3),b_g(3),b_a(3); q_IG = q^-1, v <- q*v)
  * driver loop                   datasets/asl_msckf.cpp:141-159 (initialise from ground truth), :206-296 (per sample:
                                  propagate; per image: augmentState, update, addFeatures, marginalize,
                                  pruneRedundantStates, pruneEmptyStates)
  * parameters                    datasets/asl_msckf.cpp:59-117
The reference's runner is wall-clock dependent (it drops frames when processing lags, :245-263, and sleeps to
camera rate, :472 -- SURVEY.md D5); this runner never skips, so a run is reproducible.

The image front-end (corner_detector.cpp) is out of scope; its OUTPUT is consumed instead from a "track dump"
`mav0/tracks0/data.csv` with lines `timestamp_ns,kind,feature_id,x,y` (kind 0 = tracked feature handed to
MSCKF::update, 1 = new feature handed to MSCKF::addFeatures; x,y undistorted normalized coordinates, as
produced by TrackHandler::tracked_features / new_features, corner_detector.cpp:320-439).
"""
import os

import numpy as np
import yaml

from . import scenario as sc


# ------------------------------------------------------------------------------------------------ writer
def _yaml_matrix(T):
    return {"cols": 4, "rows": 4, "data": [float(x) for x in np.asarray(T).ravel()]}


def write_dataset(root, traj, t0_ns=1403636579758555392):
    """Emit trajectory `traj` (msckf_mono_amd.scenario.Trajectory) as an ASL directory under `root`/mav0."""
    mav = os.path.join(root, "mav0")
    for sub in ("imu0", "cam0", "state_groundtruth_estimate0", "tracks0"):
        os.makedirs(os.path.join(mav, sub), exist_ok=True)
    dT = traj.dT
    n_imu = traj.readings.shape[0]
    t_imu = t0_ns + np.round(np.arange(n_imu) * dT * 1e9).astype(np.int64)
    with open(os.path.join(mav, "imu0", "data.csv"), "w") as f:
        f.write("#timestamp [ns],w_RS_S_x [rad s^-1],w_RS_S_y [rad s^-1],w_RS_S_z [rad s^-1],a_RS_S_x [m s^-2],a_RS_S_y [m s^-2],a_RS_S_z [m s^-2]\n")
        for t, r in zip(t_imu, traj.readings):
            f.write("%d,%s\n" % (t, ",".join("%.17g" % x for x in r[:6])))
    with open(os.path.join(mav, "imu0", "sensor.yaml"), "w") as f:
        yaml.safe_dump({"sensor_type": "imu", "T_BS": _yaml_matrix(np.eye(4)), "rate_hz": int(round(1.0 / dT)),
                        "gyroscope_noise_density": 1.6968e-04, "gyroscope_random_walk": 1.9393e-05,
                        "accelerometer_noise_density": 2.0e-3, "accelerometer_random_walk": 3.0e-3}, f)
    # cam0: T_BS maps sensor (camera) coordinates to the body (IMU) frame; the reader uses q_CI = quat(R_BS)^-1, p_C_I = p_BS
    cfg = traj.cfg
    C_CI = sc.quat_to_rot(cfg["q_CI"])
    T_BS = np.eye(4); T_BS[:3, :3] = C_CI.T; T_BS[:3, 3] = cfg["p_C_I"]
    with open(os.path.join(mav, "cam0", "sensor.yaml"), "w") as f:
        yaml.safe_dump({"sensor_type": "camera", "T_BS": _yaml_matrix(T_BS), "rate_hz": sc.CAM_RATE, "resolution": [752, 480],
                        "camera_model": "pinhole", "intrinsics": [cfg["f_u"], cfg["f_v"], cfg["c_u"], cfg["c_v"]],
                        "distortion_model": "radial-tangential", "distortion_coefficients": [0.0, 0.0, 0.0, 0.0]}, f)
    # image k is stamped with the LAST IMU sample of its group: the reference propagates the sample and then
    # processes the image carrying the same timestamp (asl_msckf.cpp:227-247)
    t_cam = t_imu[(np.arange(traj.n_frames) + 1) * sc.IMU_PER_FRAME - 1]
    with open(os.path.join(mav, "cam0", "data.csv"), "w") as f:
        f.write("#timestamp [ns],filename\n")
        for t in t_cam:
            f.write("%d,%d.png\n" % (t, t))
    # ground truth at IMU rate (file convention: q = q_IG^-1, v_file = q^-1 * v  <=>  reader's v <- q * v)
    gt = sc.ground_truth(traj.t0 + np.arange(n_imu) * dT)
    with open(os.path.join(mav, "state_groundtruth_estimate0", "data.csv"), "w") as f:
        f.write("#timestamp, p_RS_R_x [m], p_RS_R_y [m], p_RS_R_z [m], q_RS_w [], q_RS_x [], q_RS_y [], q_RS_z [], v_RS_R_x [m s^-1], v_RS_R_y [m s^-1], v_RS_R_z [m s^-1], b_w_RS_S_x [rad s^-1], b_w_RS_S_y [rad s^-1], b_w_RS_S_z [rad s^-1], b_a_RS_S_x [m s^-2], b_a_RS_S_y [m s^-2], b_a_RS_S_z [m s^-2]\n")
        for i in range(n_imu):
            q_IG = gt["q_IG"][i]
            q_file = np.array([q_IG[0], -q_IG[1], -q_IG[2], -q_IG[3]])
            v_file = sc.quat_to_rot(q_IG) @ gt["v"][i]
            row = np.concatenate([gt["p"][i], q_file, v_file, traj.b_g, traj.b_a])
            f.write("%d,%s\n" % (t_imu[i], ",".join("%.17g" % x for x in row)))
    with open(os.path.join(mav, "state_groundtruth_estimate0", "sensor.yaml"), "w") as f:
        yaml.safe_dump({"sensor_type": "visual-inertial", "T_BS": _yaml_matrix(np.eye(4))}, f)
    # front-end track dump
    st = traj.stream()
    with open(os.path.join(mav, "tracks0", "data.csv"), "w") as f:
        f.write("#timestamp [ns],kind (0 tracked / 1 new),feature_id,x,y\n")
        for k in range(traj.n_frames):
            for kind, key in ((0, "cur"), (1, "new")):
                for z, fid in zip(*st[k][key]):
                    f.write("%d,%d,%d,%.17g,%.17g\n" % (t_cam[k], kind, fid, z[0], z[1]))
    return mav


# ------------------------------------------------------------------------------------------------ reader
def _read_csv(path, ncol):
    rows = []
    with open(path) as f:
        for line in f:
            if not line.strip() or line.startswith("#"):
                continue
            rows.append(line.strip().split(",")[:ncol])
    return rows


def _rot_to_quat(R):
    return sc.rot_to_quat(np.asarray(R, dtype=np.float64))


def _load_sensor_yaml(path):
    """sensor.yaml as the reference reads it: through cv::FileStorage (asl_readers.h:54-63), which insists on the OpenCV header
    line `%YAML:1.0` (the reference's README has the user add it to every EuRoC sensor.yaml).  That line is not a YAML
    directive (`%YAML 1.0` would be), so it is dropped before parsing; files without it parse as well."""
    with open(path) as f:
        text = f.read()
    lines = text.splitlines()
    if lines and lines[0].strip().startswith("%YAML"):
        lines = lines[1:]
    return yaml.safe_load("\n".join(lines))


def read_dataset(mav):
    """Parse an ASL `mav0` directory -> dict(imu_t, readings[n,7], cam (q_CI, p_C_I, intrinsics), cam_t, gt, tracks)."""
    imu_cfg = _load_sensor_yaml(os.path.join(mav, "imu0", "sensor.yaml"))
    dT = 1.0 / float(imu_cfg["rate_hz"])                       # Q6: the reader sets dT = 1/rate_hz (asl_readers.cpp:170-171,202)
    rows = _read_csv(os.path.join(mav, "imu0", "data.csv"), 7)
    imu_t = np.array([int(r[0]) for r in rows], dtype=np.int64)
    readings = np.array([[float(x) for x in r[1:7]] + [dT] for r in rows])
    cam_cfg = _load_sensor_yaml(os.path.join(mav, "cam0", "sensor.yaml"))
    T = np.array(cam_cfg["T_BS"]["data"], dtype=np.float64).reshape(4, 4)
    q_bs = _rot_to_quat(T[:3, :3])
    q_CI = np.array([q_bs[0], -q_bs[1], -q_bs[2], -q_bs[3]])   # Quaternion(R_BS).inverse()   asl_readers.cpp:31
    cam = dict(q_CI=q_CI, p_C_I=T[:3, 3].copy(), intrinsics=[float(x) for x in cam_cfg["intrinsics"]], rate_hz=float(cam_cfg["rate_hz"]))
    cam_t = np.array([int(r[0]) for r in _read_csv(os.path.join(mav, "cam0", "data.csv"), 2)], dtype=np.int64)
    gt = None
    gpath = os.path.join(mav, "state_groundtruth_estimate0", "data.csv")
    if os.path.exists(gpath):
        rows = _read_csv(gpath, 17)
        g = np.array([[float(x) for x in r[1:17]] for r in rows])
        t = np.array([int(r[0]) for r in rows], dtype=np.int64)
        q_file = g[:, 3:7]
        q_IG = q_file * np.array([1, -1, -1, -1]) / np.sum(q_file ** 2, axis=1, keepdims=True)     # q^-1   :339
        v = np.stack([sc.quat_to_rot(q) @ vv for q, vv in zip(q_file, g[:, 7:10])])              # v <- q * v   :338
        gt = dict(t=t, p=g[:, 0:3], q_IG=q_IG, v=v, b_g=g[:, 10:13], b_a=g[:, 13:16])
    tracks = {}
    tpath = os.path.join(mav, "tracks0", "data.csv")
    if os.path.exists(tpath):
        for r in _read_csv(tpath, 5):
            e = tracks.setdefault(int(r[0]), {"cur": ([], []), "new": ([], [])})
            key = "cur" if int(r[1]) == 0 else "new"
            e[key][0].append([float(r[3]), float(r[4])]); e[key][1].append(int(r[2]))
    return dict(imu_t=imu_t, readings=readings, dT=dT, cam=cam, cam_t=cam_t, gt=gt, tracks=tracks)


# ------------------------------------------------------------------------------------------------ runner
def filter_config_from_dataset(ds, max_cam_states=20, feature_px=7.0, gn_px=11.0, min_track_length=3,
                               max_track_length=1000, translation_threshold=0.05):
    """The defaults of datasets/asl_msckf.cpp:73-117 (feature_covariance 7 px, max_gn_cost_norm 11 px, ...)."""
    f_u, f_v, c_u, c_v = ds["cam"]["intrinsics"]
    w_var, dbg_var, a_var, dba_var = 1e-5, 3.6733e-5, 1e-3, 7e-4
    return dict(c_u=c_u, c_v=c_v, f_u=f_u, f_v=f_v, b=0.0, q_CI=ds["cam"]["q_CI"], p_C_I=ds["cam"]["p_C_I"],
                u_var_prime=(feature_px / f_u) ** 2, v_var_prime=(feature_px / f_v) ** 2,
                Q_imu_diag=[w_var] * 3 + [dbg_var] * 3 + [a_var] * 3 + [dba_var] * 3,
                P0_diag=[1e-5] * 3 + [1e-2] * 3 + [1e-2] * 3 + [1e-2] * 3 + [1e-12] * 3,
                max_gn_cost_norm=(gn_px / f_u) ** 2, min_rcond=3e-12, translation_threshold=translation_threshold,
                redundancy_angle_thresh=0.005, redundancy_distance_thresh=0.05,
                min_track_length=min_track_length, max_track_length=max_track_length, max_cam_states=max_cam_states)


def initial_state(ds, start_ns=None):
    """First IMU state from the closest ground-truth entry at or before `start_ns` (asl_msckf.cpp:141-159)."""
    gt = ds["gt"]
    if gt is None:
        raise ValueError("dataset has no ground truth: use a stand-still initialisation instead")
    t0 = ds["imu_t"][0] if start_ns is None else start_ns
    i = max(int(np.searchsorted(gt["t"], t0, side="right")) - 1, 0)
    q, v, p = gt["q_IG"][i], gt["v"][i], gt["p"][i]
    return np.concatenate([q, gt["b_g"][i], v, gt["b_a"][i], p, sc.GRAVITY, q, v, p])


def _from_two_vectors(a, b):
    """Eigen::Quaternion::FromTwoVectors(a, b): the rotation that sends the direction of a onto the direction of b, (w,x,y,z)"""
    v0 = np.asarray(a, float) / np.linalg.norm(a); v1 = np.asarray(b, float) / np.linalg.norm(b)
    c = float(v1 @ v0)
    if c < -1.0 + 1e-12:                       # opposite vectors: any axis orthogonal to v0 (Eigen takes it from an SVD)
        axis = np.cross(v0, [1.0, 0.0, 0.0]) if abs(v0[0]) < 0.9 else np.cross(v0, [0.0, 1.0, 0.0])
        axis /= np.linalg.norm(axis)
        return np.array([0.0, axis[0], axis[1], axis[2]])
    axis = np.cross(v0, v1)
    sq = np.sqrt((1.0 + c) * 2.0)
    return np.concatenate([[sq * 0.5], axis / sq])


def standstill_initial_state(ds, calib_start_ns, calib_end_ns):
    """First IMU state of the no-ground-truth runner (datasets/asl_msckf_no_ground_truth.cpp:136-173): gyro bias = mean
    gyro reading over the stand-still interval, attitude from the mean accelerometer reading against gravity
    (q_IG = FromTwoVectors(-g, a_mean)), accelerometer bias = q_IG g + a_mean, zero position and velocity."""
    m = (ds["imu_t"] >= calib_start_ns) & (ds["imu_t"] < calib_end_ns)
    if not np.any(m):
        raise ValueError("no IMU samples inside the stand-still interval")
    rd = np.asarray(ds["readings"])[m]
    gyro_mean, accel_mean = rd[:, 0:3].mean(0), rd[:, 3:6].mean(0)
    g = np.array(sc.GRAVITY, dtype=float)
    q = _from_two_vectors(-g, accel_mean)
    w, x, y, z = q
    u = np.array([x, y, z]); uv = np.cross(u, g); uv = uv + uv
    qg = g + w * uv + np.cross(u, uv)          # q_IG * g (Eigen _transformVector)
    b_a = qg + accel_mean
    zero = np.zeros(3)
    return np.concatenate([q, gyro_mean, zero, b_a, zero, g, q, zero, zero])


# stage names of the reference's StageTiming message (msg/StageTiming.msg: string[] stages, float64[] times) as the
# ASL runner records them, datasets/asl_msckf.cpp:207-212 (TSTART/TEND/TRECORD) and :229-296
STAGES = ("imu_prop", "msckf_augment_state", "msckf_update", "msckf_add_features", "msckf_marginalize",
          "msckf_prune_redundant", "msckf_prune_empty_states")


def run(ds, flt, cfg, start_ns=None, prune_redundant=False, on_frame=None, stage_timing=None, standstill=None):
    """Drive `flt` (any object with the reference's member names: msckf_mono_amd.capi.MSCKF or the oracle) through the
    dataset in the reference's call order.  Returns the list of (timestamp_ns, imu_state29) after every image.
    `stage_timing`: a list that receives one StageTiming record per image, {"stamp": ns, "stages": [...], "times":
    [...]} with the reference's stage names and wall-clock seconds; `standstill` = (calib_start_ns, calib_end_ns) starts
    from the stand-still initialisation of asl_msckf_no_ground_truth.cpp instead of ground truth, at calib_end_ns; device work is synchronised at the end of every
    stage (`flt.sync()` when the filter has one) so that the numbers compare with the reference's synchronous CPU
    calls.  Timing changes nothing in the results."""
    import time
    sync = getattr(flt, "sync", None) if stage_timing is not None else None

    def timed(rec, name, fn, *a):
        if rec is None:
            return fn(*a)
        t0 = time.perf_counter()
        r = fn(*a)
        if sync is not None:
            sync()
        rec["stages"].append(name); rec["times"].append(time.perf_counter() - t0)
        return r

    if standstill is not None:
        flt.initialize(cfg, standstill_initial_state(ds, standstill[0], standstill[1]))
        start_ns = standstill[1] if start_ns is None else start_ns
    else:
        flt.initialize(cfg, initial_state(ds, start_ns))
    cam_set = set(int(t) for t in ds["cam_t"])
    out = []
    state_k = 0
    t_begin = ds["imu_t"][0] if start_ns is None else start_ns
    pend = []   # consecutive IMU samples between images are handed over in one call (fused on the device)
    for t, rd in zip(ds["imu_t"], ds["readings"]):
        if t < t_begin:
            continue
        state_k += 1                                                     # asl_msckf.cpp:227
        pend.append(rd)
        if int(t) in cam_set:
            rec = {"stamp": int(t), "stages": [], "times": []} if stage_timing is not None else None
            timed(rec, "imu_prop", flt.propagate, np.array(pend)); pend = []        # :229-238
            tr = ds["tracks"].get(int(t), {"cur": ([], []), "new": ([], [])})
            timed(rec, "msckf_augment_state", flt.augmentState, state_k, t / 1e9)                         # :268-271
            timed(rec, "msckf_update", flt.update, np.array(tr["cur"][0]).reshape(-1, 2), tr["cur"][1])   # :273-276
            timed(rec, "msckf_add_features", flt.addFeatures, np.array(tr["new"][0]).reshape(-1, 2), tr["new"][1])   # :278-281
            timed(rec, "msckf_marginalize", flt.marginalize)                                              # :283-286
            if prune_redundant:
                timed(rec, "msckf_prune_redundant", flt.pruneRedundantStates)                             # :288-291
            timed(rec, "msckf_prune_empty_states", flt.pruneEmptyStates)                                  # :293-296
            if rec is not None:
                stage_timing.append(rec)
            s = np.array(flt.getImuState())
            out.append((int(t), s))
            if on_frame is not None:
                on_frame(int(t), s)
    if pend:
        flt.propagate(np.array(pend))
    return out


def write_stage_timing(path, records):
    """StageTiming records as CSV `stamp_ns,stage,seconds` -- one line per (image, stage), the content of the
    reference's `stage_timing` topic (asl_msckf.cpp:193, :470)."""
    with open(path, "w") as f:
        f.write("#stamp [ns],stage,seconds\n")
        for r in records:
            for st, tm in zip(r["stages"], r["times"]):
                f.write("%d,%s,%.9f\n" % (r["stamp"], st, tm))


def ate(traj_out, ds):
    """RMSE of the estimated IMU position against ground truth at the image times, no alignment (the runner starts
    from ground truth): returns (ate_m, sum_sq, n) so that ranks can all-reduce the partial sums."""
    gt = ds["gt"]
    se, n = 0.0, 0
    for t, s in traj_out:
        i = max(int(np.searchsorted(gt["t"], t, side="right")) - 1, 0)
        # the state after processing image t has been propagated THROUGH the sample stamped t, i.e. to t + dT
        j = min(i + 1, len(gt["t"]) - 1)
        e = s[13:16] - gt["p"][j]
        se += float(e @ e); n += 1
    return (np.sqrt(se / max(n, 1)), se, n)
