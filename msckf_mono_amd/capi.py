"""ctypes binding of libmsckf_hip.so (the C-ABI declared in include/msckf_hip.h).

Product path: this module never imports anything from oracle/ and has no CPU fallback -- if the HIP
library cannot be loaded, or no GPU is visible, it raises.  `torch` (when installed) is imported first so
that the process ends up with ONE HIP runtime (torch bundles its own libamdhip64.so with the same soname).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MSCKF_HIP_LIB", os.path.join(_HERE, "libmsckf_hip.so"))   # override: A/B experiments only
_LIB = None

F32, F64, F16H = 0, 1, 2   # F16H: fp16 measurement Jacobian / f32 state + covariance (MSCKF_HIP_F16H_F32P)
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_up = C.POINTER(C.c_uint64)

# every symbol include/msckf_hip.h declares (checked by tests/test_capi_symbols.py)
SYMBOLS = [
    "msckf_hip_create", "msckf_hip_destroy", "msckf_hip_last_error", "msckf_hip_initialize", "msckf_hip_propagate",
    "msckf_hip_augment_state", "msckf_hip_update", "msckf_hip_add_features", "msckf_hip_marginalize",
    "msckf_hip_prune_empty_states", "msckf_hip_prune_redundant_states", "msckf_hip_finish",
    "msckf_hip_get_num_cam_states", "msckf_hip_get_imu_state", "msckf_hip_get_cam_states", "msckf_hip_get_map",
    "msckf_hip_get_pruned_state_ids", "msckf_hip_get_covariance", "msckf_hip_set_covariance", "msckf_hip_set_imu_state",
    "msckf_hip_set_cam_pose", "msckf_hip_get_num_residualized", "msckf_hip_set_num_residualized", "msckf_hip_last_stats",
    "msckf_hip_last_tracks", "msckf_hip_last_deltax", "msckf_hip_set_tracks", "msckf_hip_propagate_range",
    "msckf_hip_augment_range", "msckf_hip_marginalize_range", "msckf_hip_drop_oldest_range", "msckf_hip_scenario_alloc",
    "msckf_hip_scenario_set", "msckf_hip_scenario_commit", "msckf_hip_run_frames", "msckf_hip_run_frames_streamed", "msckf_hip_sync",
    "msckf_hip_profile_enable", "msckf_hip_profile_read", "msckf_hip_profile_read_ex", "msckf_hip_profile_event_overhead", "msckf_hip_set_streams", "msckf_hip_set_gate_early_accept",
    "msckf_hip_scenario_pin", "msckf_hip_set_upload_ring", "msckf_hip_clear_error_flags",
    "msckf_hip_set_compression", "msckf_hip_set_covariance_update", "msckf_hip_set_feature_overlap", "msckf_hip_get_pruned_states", "msckf_hip_get_cam_meta", "msckf_hip_get_tracked_feature_ids",
    "msckf_hip_set_anisotropic_noise", "msckf_hip_literal_info", "msckf_hip_get_error_flags", "msckf_hip_copy_state", "msckf_hip_set_host_affinity",
    "msckf_hip_image_cycle_range",
]


def build():
    subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(_HERE, "csrc")])


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libmsckf_hip.so is not built (run __graft_entry__.build()); there is no CPU fallback")
        try:
            import torch  # noqa: F401  (one HIP runtime per process)
        except Exception:
            pass
        L = C.CDLL(LIB_PATH)
        L.msckf_hip_last_error.restype = C.c_char_p
        _LIB = L
    return _LIB


class HipError(RuntimeError):
    pass


def _chk(rc):
    if rc < 0:
        raise HipError("msckf_hip call failed (%d): %s" % (rc, lib().msckf_hip_last_error().decode()))
    return rc


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_ip)


def pack_config(cfg):
    cam = np.array([cfg["c_u"], cfg["c_v"], cfg["f_u"], cfg["f_v"], cfg.get("b", 0.0)] + list(cfg["q_CI"]) + list(cfg["p_C_I"]), dtype=np.float64)
    noise = np.array([cfg["u_var_prime"], cfg["v_var_prime"]] + list(cfg["Q_imu_diag"]) + list(cfg["P0_diag"]), dtype=np.float64)
    params = np.array([cfg["max_gn_cost_norm"], cfg.get("min_rcond", 3e-12), cfg["translation_threshold"],
                       cfg.get("redundancy_angle_thresh", 0.005), cfg.get("redundancy_distance_thresh", 0.05),
                       cfg["min_track_length"], cfg["max_track_length"], cfg["max_cam_states"]], dtype=np.float64)
    return cam, noise, params


class Batch:
    """A batch of B independent filters resident on one GPU."""

    def __init__(self, B, n_cap, f_cap, m_cap, dtype=F32, device=0):
        self.L = lib()
        self.h = C.c_void_p()
        self.B, self.n_cap, self.f_cap, self.m_cap, self.dtype = B, n_cap, f_cap, m_cap, dtype
        _chk(self.L.msckf_hip_create(B, n_cap, f_cap, m_cap, dtype, device, C.byref(self.h)))

    def close(self):
        if self.h:
            self.L.msckf_hip_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- reference API, per trajectory
    def initialize(self, b, cfg, imu29):
        cam, noise, params = pack_config(cfg)
        a, pa = _d(cam); n, pn = _d(noise); p, pp = _d(params); s, ps = _d(imu29)
        _chk(self.L.msckf_hip_initialize(self.h, b, pa, pn, pp, ps))

    def propagate(self, b, readings):
        r, p = _d(np.asarray(readings).reshape(-1, 7))
        _chk(self.L.msckf_hip_propagate(self.h, b, p, r.shape[0]))

    def augment_state(self, b, state_id, time=0.0):
        _chk(self.L.msckf_hip_augment_state(self.h, b, int(state_id), C.c_double(time)))

    def update(self, b, meas, ids):
        m, pm = _d(np.asarray(meas).reshape(-1, 2)); i = np.ascontiguousarray(ids, dtype=np.uint64)
        _chk(self.L.msckf_hip_update(self.h, b, pm, i.ctypes.data_as(_up), len(i)))

    def add_features(self, b, meas, ids):
        m, pm = _d(np.asarray(meas).reshape(-1, 2)); i = np.ascontiguousarray(ids, dtype=np.uint64)
        _chk(self.L.msckf_hip_add_features(self.h, b, pm, i.ctypes.data_as(_up), len(i)))

    def marginalize(self, b):
        _chk(self.L.msckf_hip_marginalize(self.h, b))

    def prune_empty_states(self, b):
        _chk(self.L.msckf_hip_prune_empty_states(self.h, b))

    def prune_redundant_states(self, b):
        _chk(self.L.msckf_hip_prune_redundant_states(self.h, b))

    def finish(self, b):
        _chk(self.L.msckf_hip_finish(self.h, b))

    def num_cam_states(self, b):
        return _chk(self.L.msckf_hip_get_num_cam_states(self.h, b))

    def imu_state(self, b):
        o = np.zeros(29); _chk(self.L.msckf_hip_get_imu_state(self.h, b, o.ctypes.data_as(_dp))); return o

    def cam_states(self, b):
        o = np.zeros((self.n_cap, 7)); ids = np.zeros(self.n_cap, dtype=np.int32)
        n = _chk(self.L.msckf_hip_get_cam_states(self.h, b, o.ctypes.data_as(_dp), ids.ctypes.data_as(_ip), self.n_cap))
        return o[:n], ids[:n]

    def get_map(self, b):
        o = np.zeros((self.f_cap, 3)); n = _chk(self.L.msckf_hip_get_map(self.h, b, o.ctypes.data_as(_dp), self.f_cap)); return o[:n]

    def pruned_state_ids(self, b, cap=65536):
        o = np.zeros(cap, dtype=np.int32); n = _chk(self.L.msckf_hip_get_pruned_state_ids(self.h, b, o.ctypes.data_as(_ip), cap)); return o[:n]

    def pruned_states(self, b):
        """getPrunedStates() in full: rows of q_CG(4) p_C_G(3) time state_id last_correlated_id, sorted by state_id"""
        n = _chk(self.L.msckf_hip_get_pruned_states(self.h, b, None, None, None, None, 1 << 30))
        c = np.zeros((max(n, 1), 7)); t = np.zeros(max(n, 1)); ids = np.zeros(max(n, 1), dtype=np.int32); lc = np.zeros(max(n, 1), dtype=np.int32)
        n = _chk(self.L.msckf_hip_get_pruned_states(self.h, b, c.ctypes.data_as(_dp), t.ctypes.data_as(_dp), ids.ctypes.data_as(_ip), lc.ctypes.data_as(_ip), n))
        return np.concatenate([c[:n], t[:n, None], ids[:n, None].astype(np.float64), lc[:n, None].astype(np.float64)], 1)

    def cam_meta(self, b):
        """(time, len(tracked_feature_ids), last_correlated_id) per entry of getCamStates() (types.h:57-67)"""
        t = np.zeros(self.n_cap); k = np.zeros(self.n_cap, dtype=np.int32); lc = np.zeros(self.n_cap, dtype=np.int32)
        n = _chk(self.L.msckf_hip_get_cam_meta(self.h, b, t.ctypes.data_as(_dp), k.ctypes.data_as(_ip), lc.ctypes.data_as(_ip), self.n_cap))
        return t[:n], k[:n], lc[:n]

    def tracked_feature_ids(self, b, cam_index, cap=4096):
        o = np.zeros(cap, dtype=np.uint64)
        n = _chk(self.L.msckf_hip_get_tracked_feature_ids(self.h, b, int(cam_index), o.ctypes.data_as(_up), cap))
        return o[:n]

    # ---- additive accessors
    def covariance(self, b):
        D = 15 + 6 * self.num_cam_states(b)
        P = np.zeros((D, D), order="F")
        _chk(self.L.msckf_hip_get_covariance(self.h, b, P.ctypes.data_as(_dp), D))
        return np.array(P)

    def set_covariance(self, b, P):
        P = np.asfortranarray(P, dtype=np.float64)
        _chk(self.L.msckf_hip_set_covariance(self.h, b, P.ctypes.data_as(_dp), P.shape[0]))

    def set_imu_state(self, b, s):
        a, p = _d(s); _chk(self.L.msckf_hip_set_imu_state(self.h, b, p))

    def set_cam_pose(self, b, slot, qp):
        a, p = _d(qp); _chk(self.L.msckf_hip_set_cam_pose(self.h, b, int(slot), p))

    def num_residualized(self, b):
        n = C.c_longlong(0); _chk(self.L.msckf_hip_get_num_residualized(self.h, b, C.byref(n))); return n.value

    def set_num_residualized(self, b, n):
        _chk(self.L.msckf_hip_set_num_residualized(self.h, b, C.c_longlong(int(n))))

    def last_stats(self, b, strict=True):
        """strict: a sticky device-side error flag of the trajectory raises (msckf_hip_last_stats returns its code); otherwise the
        statistics come back either way with the flag bits under "error_flags" (a poll over many trajectories must not lose
        the numbers because one of them once clamped a pivot)"""
        o = np.zeros(7, dtype=np.int32)
        rc = self.L.msckf_hip_last_stats(self.h, b, o.ctypes.data_as(_ip))
        d = dict(zip(["n_tracks", "n_motion_rejected", "n_tri_rejected", "n_gate_rejected", "n_passed", "m_rows", "r_rows"], o.tolist()))
        if strict:
            _chk(rc)
            return d
        d["error_flags"] = self.error_flags(b)
        if rc < 0 and not d["error_flags"]:
            _chk(rc)
        return d

    def copy_state_from(self, other):
        _chk(self.L.msckf_hip_copy_state(self.h, other.h))

    def error_flags(self, b):
        f = C.c_int(0)
        _chk(self.L.msckf_hip_get_error_flags(self.h, int(b), C.byref(f)))
        return int(f.value)

    def last_tracks(self, b):
        o = np.zeros((self.f_cap, 8)); n = _chk(self.L.msckf_hip_last_tracks(self.h, b, o.ctypes.data_as(_dp), self.f_cap)); return o[:n]

    def last_deltax(self, b):
        cap = 15 + 6 * self.n_cap
        o = np.zeros(cap); n = _chk(self.L.msckf_hip_last_deltax(self.h, b, o.ctypes.data_as(_dp), cap)); return o[:n]

    # ---- batched path
    def set_tracks(self, b, M, slots, obs):
        Ma, pM = _i(M); s, ps = _i(slots); o, po = _d(obs)
        _chk(self.L.msckf_hip_set_tracks(self.h, b, len(Ma), pM, ps, po))

    def propagate_range(self, b0, nb, readings):
        r, p = _d(np.asarray(readings).reshape(nb, -1, 7))
        _chk(self.L.msckf_hip_propagate_range(self.h, b0, nb, p, r.shape[1]))

    def augment_range(self, b0, nb):
        _chk(self.L.msckf_hip_augment_range(self.h, b0, nb))

    def marginalize_range(self, b0, nb):
        _chk(self.L.msckf_hip_marginalize_range(self.h, b0, nb))

    def drop_oldest_range(self, b0, nb, n):
        _chk(self.L.msckf_hip_drop_oldest_range(self.h, b0, nb, int(n)))

    def image_cycle_range(self, b0, nb, state_ids, times, cur, new, prune_redundant=True, prune_empty=True):
        """The ASL runner's per-image cycle (asl_msckf.cpp:269-294) for trajectories b0 .. b0 + nb - 1 in lockstep:
        augmentState, update, addFeatures, marginalize, [pruneRedundantStates], [pruneEmptyStates].
        cur[i] / new[i] = (measurements [n][2], feature ids [n]) of trajectory b0 + i, as update() / addFeatures() take them."""
        self.image_cycle_range_packed(b0, nb, self.pack_image_inputs(nb, state_ids, times, cur, new), prune_redundant, prune_empty)

    @staticmethod
    def pack_image_inputs(nb, state_ids, times, cur, new):
        """the arguments of msckf_hip_image_cycle_range as contiguous arrays (a caller that replays images converts once, outside its loop)"""
        def cat(parts):
            n = np.array([len(p[1]) for p in parts], dtype=np.int32)
            m = np.concatenate([np.asarray(p[0], dtype=np.float64).reshape(-1, 2) for p in parts]) if n.sum() else np.zeros((0, 2))
            i = np.concatenate([np.asarray(p[1], dtype=np.uint64).reshape(-1) for p in parts]) if n.sum() else np.zeros(0, dtype=np.uint64)
            return np.ascontiguousarray(m, dtype=np.float64), np.ascontiguousarray(i, dtype=np.uint64), n
        return (np.ascontiguousarray(np.asarray(state_ids).reshape(nb), dtype=np.int32), np.ascontiguousarray(np.asarray(times, dtype=np.float64).reshape(nb))) + cat(cur) + cat(new)

    def image_cycle_range_packed(self, b0, nb, packed, prune_redundant=True, prune_empty=True):
        sid, tt, um, ui, un, nm, ni, nn = packed
        _chk(self.L.msckf_hip_image_cycle_range(self.h, b0, nb, sid.ctypes.data_as(_ip), tt.ctypes.data_as(_dp), um.ctypes.data_as(_dp), ui.ctypes.data_as(_up), un.ctypes.data_as(_ip),
                                                nm.ctypes.data_as(_dp), ni.ctypes.data_as(_up), nn.ctypes.data_as(_ip),
                                                (1 if prune_redundant else 0) | (2 if prune_empty else 0)))

    def scenario_alloc(self, n_frames, K):
        _chk(self.L.msckf_hip_scenario_alloc(self.h, n_frames, K))

    def scenario_set(self, frame, b, readings, M, slots, obs, n_drop):
        r, pr = _d(np.asarray(readings).reshape(-1, 7)); Ma, pM = _i(M); s, ps = _i(slots); o, po = _d(obs)
        _chk(self.L.msckf_hip_scenario_set(self.h, frame, b, pr, len(Ma), pM, ps, po, int(n_drop)))

    def scenario_commit(self):
        _chk(self.L.msckf_hip_scenario_commit(self.h))

    def run_frames(self, f0, f1):
        _chk(self.L.msckf_hip_run_frames(self.h, f0, f1))

    def run_frames_streamed(self, f0, f1):
        """run_frames with per-frame asynchronous H2D of the inputs (copy stream, ring of staging sets, compact work-lists)"""
        _chk(self.L.msckf_hip_run_frames_streamed(self.h, f0, f1))

    def scenario_pin(self, f0, f1):
        """page-lock the per-frame blocks of frames [f0, f1) ahead of a streamed run (outside any timed region)"""
        _chk(self.L.msckf_hip_scenario_pin(self.h, int(f0), int(f1)))

    def set_upload_ring(self, depth=6, mode=0):
        """staging sets of run_frames_streamed (2..8); mode 0 host hand-over (default), 1 device-side event waits"""
        _chk(self.L.msckf_hip_set_upload_ring(self.h, int(depth), int(mode)))

    def clear_error_flags(self, b):
        _chk(self.L.msckf_hip_clear_error_flags(self.h, int(b)))

    def sync(self):
        _chk(self.L.msckf_hip_sync(self.h))

    def set_host_affinity(self, cpus):
        """cpus[0]: the calling thread while it uploads frames; cpus[1 + i]: enqueue thread of slice i"""
        a = np.ascontiguousarray(list(cpus), dtype=np.int32)
        _chk(self.L.msckf_hip_set_host_affinity(self.h, a.ctypes.data_as(_ip), int(a.size)))

    def set_streams(self, n):
        _chk(self.L.msckf_hip_set_streams(self.h, int(n)))

    def set_compression(self, route):
        """-1 default for the window size; 0 Householder TSQR; 3 information form chol(H_o^T H_o) with the blocked matrix-core
        Cholesky k_chol_mfma (the default where it fits; 1 and 2 named retired factorizations and mean 3)"""
        _chk(self.L.msckf_hip_set_compression(self.h, int(route)))

    def set_covariance_update(self, form):
        """0 square-root gain form P - W W^T with the blocked matrix-core solve (default), 1 the reference's Joseph sequence,
        2 square-root gain form with the register-resident solve"""
        _chk(self.L.msckf_hip_set_covariance_update(self.h, int(form)))

    def set_feature_overlap(self, on):
        _chk(self.L.msckf_hip_set_feature_overlap(self.h, 1 if on else 0))

    def set_gate_early_accept(self, on):
        _chk(self.L.msckf_hip_set_gate_early_accept(self.h, 1 if on else 0))

    def set_anisotropic_noise(self, mode, tail_tol=-1.0):
        """u_var' != v_var': 0 (default) the reference's R_o_j = A_j^T R_j A_j / HouseholderQR / R_n = Q_1^T R_o Q_1 on the
        device (msckf.h:423-431, 1343-1366), 1 rows pre-whitened by 1/sigma.  tail_tol < 0: default, 0: the reference's
        zero-tail rule to the letter"""
        _chk(self.L.msckf_hip_set_anisotropic_noise(self.h, int(mode), C.c_double(float(tail_tol))))

    def literal_info(self, b):
        o = np.zeros(8, dtype=np.int32)
        _chk(self.L.msckf_hip_literal_info(self.h, int(b), o.ctypes.data_as(_ip)))
        return dict(zip(["m_rows", "kept_rows", "reflected", "skipped_by_tolerance", "route", "leading_rows_handed_through", "kept_handed_through_rows", "spare"], o.tolist()))

    def profile_enable(self, on=True):
        _chk(self.L.msckf_hip_profile_enable(self.h, 1 if on else 0))

    def profile_event_overhead(self):
        ms = C.c_double(0.0)
        _chk(self.L.msckf_hip_profile_event_overhead(self.h, C.byref(ms)))
        return float(ms.value)

    def profile_read(self):
        ms = np.zeros(16); cnt = np.zeros(16, dtype=np.int32)
        _chk(self.L.msckf_hip_profile_read_ex(self.h, ms.ctypes.data_as(_dp), cnt.ctypes.data_as(_ip), 16))
        names = ["propagate", "augment", "feature", "compress_stage1", "compress_merge", "kalman", "prune", "select", "lit_pre", "lit_gamma", "literal"]
        return {n: (float(m), int(c)) for n, m, c in zip(names, ms, cnt)}


class MSCKF:
    """Host-side mirror of `msckf_mono::MSCKF<_S>` (msckf.h:66-848) over the C-ABI: same member names and
    call order as the reference so that the parity tests read like code written against the reference."""

    def __init__(self, dtype=F32, n_cap=32, f_cap=256, m_cap=32, device=0):
        self.batch = Batch(1, n_cap, f_cap, m_cap, dtype, device)

    def initialize(self, cfg, imu29):
        self.batch.initialize(0, cfg, imu29)

    def propagate(self, reading):
        self.batch.propagate(0, reading)

    def augmentState(self, state_id, time=0.0):
        self.batch.augment_state(0, state_id, time)

    def update(self, measurements, feature_ids):
        self.batch.update(0, measurements, feature_ids)

    def addFeatures(self, features, feature_ids):
        self.batch.add_features(0, features, feature_ids)

    def marginalize(self):
        self.batch.marginalize(0)

    def pruneRedundantStates(self):
        self.batch.prune_redundant_states(0)

    def sync(self):
        self.batch.sync()

    def pruneEmptyStates(self):
        self.batch.prune_empty_states(0)

    def finish(self):
        self.batch.finish(0)

    def getNumCamStates(self):
        return self.batch.num_cam_states(0)

    def getImuState(self):
        return self.batch.imu_state(0)

    def getCamStates(self):
        return self.batch.cam_states(0)

    def getMap(self):
        return self.batch.get_map(0)

    def getPrunedStates(self):
        return self.batch.pruned_state_ids(0)

    def getPrunedStatesFull(self):
        return self.batch.pruned_states(0)

    def getCamMeta(self):
        return self.batch.cam_meta(0)

    def getCovariance(self):
        return self.batch.covariance(0)
