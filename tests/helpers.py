"""Shared helpers for the parity tests: drive the CPU oracle and the HIP path with the same seeded inputs
and measure the parity metric of SURVEY.md section 8c."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def quat_angle(q, qref):
    """rotation angle of q (x) qref^-1 for (w,x,y,z) quaternions (atan2 form: accurate near zero)"""
    q = np.asarray(q, dtype=np.float64) / np.linalg.norm(q)
    r = np.asarray(qref, dtype=np.float64) / np.linalg.norm(qref)
    rc = np.array([r[0], -r[1], -r[2], -r[3]])
    w = q[0] * rc[0] - q[1:] @ rc[1:]
    v = q[0] * rc[1:] + rc[0] * q[1:] + np.cross(q[1:], rc[1:])
    return float(2 * np.arctan2(np.linalg.norm(v), abs(w)))


def rel(x, ref, floor=1e-3):
    return float(np.linalg.norm(np.asarray(x) - np.asarray(ref)) / max(np.linalg.norm(ref), floor))


def state_errors(imu, imu_ref, cams, cams_ref, P, P_ref):
    """dict of the parity metrics: relative L2 per vector field, rotation angle for attitudes, relative
    Frobenius norm for the covariance (full and IMU block)."""
    e = dict(
        q=quat_angle(imu[0:4], imu_ref[0:4]), bg=rel(imu[4:7], imu_ref[4:7]), v=rel(imu[7:10], imu_ref[7:10]),
        ba=rel(imu[10:13], imu_ref[10:13]), p=rel(imu[13:16], imu_ref[13:16]),
        P=rel(P, P_ref, 1e-30), Pii=rel(P[:15, :15], P_ref[:15, :15], 1e-30))
    if len(cams_ref):
        e["cam_q"] = max(quat_angle(a[:4], b[:4]) for a, b in zip(cams, cams_ref))
        e["cam_p"] = max(rel(a[4:7], b[4:7]) for a, b in zip(cams, cams_ref))
    return e


def worst(e):
    return max(e.values())


def oracle_frame(o, tr, k, N):
    o.propagate(tr.imu_for_frame(k))
    o.augmentState(k, tr.frame_times[k])
    fr = tr.frames[k]
    if len(fr["M"]):
        o.setTracks(fr["M"], fr["slots"], fr["obs"])
        o.marginalize()
    if o.getNumCamStates() == N:
        o.dropOldest(1)


def device_frame(batch, b, tr, k, N):
    batch.propagate_range(b, 1, tr.imu_for_frame(k))
    batch.augment_range(b, 1)
    fr = tr.frames[k]
    batch.set_tracks(b, fr["M"], fr["slots"], fr["obs"])
    if len(fr["M"]):
        batch.marginalize_range(b, 1)
    if batch.num_cam_states(b) == N:
        batch.drop_oldest_range(b, 1, 1)


def copy_oracle_to_device(o, batch, b):
    """teacher forcing: device state + covariance <- oracle"""
    cams, _ = o.getCamStates()
    batch.set_covariance(b, o.getCovariance())
    batch.set_imu_state(b, o.getImuState())
    for i, c in enumerate(cams):
        batch.set_cam_pose(b, i, c)
    batch.set_num_residualized(b, o.numResidualized())


def check_tracks(td, to, ok, prec, tr, fr, o, k):
    """Per-track parity of the triangulated point and of the gate statistic (rows of last_tracks: motion_ok tri_valid
    gate_pass included gamma p_f_G).  Double: 1e-6 on both.  Float: the point is the result of an iterative solve whose
    depth error scales with depth^2 / baseline, so it is held (i) through what the filter uses it for -- its
    reprojection in every camera of the track agrees with the oracle's point to 1e-4 normalized units (0.05 px) -- and
    (ii) directly to 2e-3 of its depth; gamma to 1e-3 relative + 1e-3 absolute."""
    if prec == "f64":
        assert np.allclose(td[ok, 5:8], to[ok, 5:8], rtol=0, atol=1e-6), k
        assert np.allclose(td[ok, 4], to[ok, 4], rtol=1e-6, atol=1e-9), k
        return
    cams, _ = o.getCamStates()
    off = np.concatenate([[0], np.cumsum(fr["M"])])
    for t in np.nonzero(ok)[0]:
        sl = fr["slots"][off[t]:off[t + 1]]
        pd, pr = td[t, 5:8], to[t, 5:8]
        worst_rp, depth = 0.0, 1e9
        for s in sl:
            R = q_to_rot(cams[s, :4])
            a, b = R @ (pd - cams[s, 4:7]), R @ (pr - cams[s, 4:7])
            worst_rp = max(worst_rp, float(np.abs(a[:2] / a[2] - b[:2] / b[2]).max()))
            depth = min(depth, float(b[2]))
        assert worst_rp < 1e-4, (k, t, worst_rp)
        assert np.linalg.norm(pd - pr) < 2e-3 * max(depth, 1.0), (k, t, pd, pr, depth)
    assert np.allclose(td[ok, 4], to[ok, 4], rtol=1e-3, atol=1e-3), (k, np.abs(td[ok, 4] - to[ok, 4]).max())


def q_to_rot(q):
    """Eigen toRotationMatrix of (w,x,y,z)"""
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def copy_device_to_oracle(batch, b, o):
    """teacher forcing the other way: oracle state + covariance <- device trajectory b (the oracle's window is built by
    replaying augmentState, then overwritten)"""
    cams, ids = batch.cam_states(b)
    while o.getNumCamStates() < len(cams):
        o.augmentState(int(ids[o.getNumCamStates()]) if len(ids) else 0, 0.0)
    o.setImuState(batch.imu_state(b))
    for i, c in enumerate(cams):
        o.setCamPose(i, c)
    o.setCovariance(batch.covariance(b))
    o.setNumResidualized(batch.num_residualized(b))
