"""ctypes front for oracle/liboracle.so and oracle/_ref/lib_ref.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product path (msckf_mono_amd) must never do so.  Two checkers share one C-ABI and this one Python class:
  impl="oracle": oracle/liboracle.so, the restatement of /root/reference/include/msckf_mono/msckf.h
                 (oracle/msckf_oracle.hpp);
  impl="ref" ("ref_alt": same, second rounding): oracle/_ref/lib_ref.so, the reference's OWN unmodified sources compiled against the minimal
                 Eigen/Boost surface of oracle/ref_shim (oracle/ref_capi.cpp).  Built only where /root/reference
                 exists; the prebuilt library travels to the GPU box with the snapshot.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}

F32, F64 = 0, 1
FAITHFUL, LEAN, GRAM = 0, 1, 2   # GRAM: compression route of the HIP library restated on the CPU (msckf_oracle.hpp)

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_up = C.POINTER(C.c_uint64)


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def ref_available():
    return os.path.exists(os.path.join(_HERE, "_ref", "lib_ref.so"))


def lib(impl="oracle"):
    if impl not in _LIBS:
        path = {"oracle": os.path.join(_HERE, "liboracle.so"), "ref": os.path.join(_HERE, "_ref", "lib_ref.so"),
                "ref_alt": os.path.join(_HERE, "_ref", "lib_ref_alt.so")}[impl]
        if impl == "oracle" and os.environ.get("ORACLE_SANITIZED"):     # make -C oracle asan: AddressSanitizer + UBSan build of the restatement
            path = os.path.join(_HERE, "liboracle_asan.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.oracle_create.restype = C.c_void_p
        L.oracle_create.argtypes = [C.c_int, C.c_int]
        L.oracle_clone.restype = C.c_void_p
        L.oracle_clone.argtypes = [C.c_void_p]
        L.oracle_time_updates.restype = C.c_double
        L.oracle_num_residualized.restype = C.c_long
        for name in ("oracle_num_cam_states", "oracle_get_tracks", "oracle_last_tracks", "oracle_last_deltax",
                     "oracle_map_points", "oracle_pruned_ids", "oracle_get_cam_meta", "oracle_pruned_states", "oracle_is_reference"):
            getattr(L, name).restype = C.c_int
        assert L.oracle_is_reference() == (0 if impl == "oracle" else 1)
        _LIBS[impl] = L
    return _LIBS[impl]


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_ip)


def pack_config(cfg):
    """cfg: dict from msckf_mono_amd.scenario.filter_config -> (cam[12], noise[29], params[8])."""
    cam = np.array([cfg["c_u"], cfg["c_v"], cfg["f_u"], cfg["f_v"], cfg.get("b", 0.0)] + list(cfg["q_CI"]) + list(cfg["p_C_I"]), dtype=np.float64)
    noise = np.array([cfg["u_var_prime"], cfg["v_var_prime"]] + list(cfg["Q_imu_diag"]) + list(cfg["P0_diag"]), dtype=np.float64)
    params = np.array([cfg["max_gn_cost_norm"], cfg.get("min_rcond", 3e-12), cfg["translation_threshold"],
                       cfg.get("redundancy_angle_thresh", 0.005), cfg.get("redundancy_distance_thresh", 0.05),
                       cfg["min_track_length"], cfg["max_track_length"], cfg["max_cam_states"]], dtype=np.float64)
    return cam, noise, params


class Oracle:
    """One CPU filter (float or double).  Method names follow the reference's public API (msckf.h:72-848)."""

    def __init__(self, dtype=F64, mode=LEAN, impl="oracle"):
        self.L = lib(impl)
        self.impl = impl
        self.h = C.c_void_p(self.L.oracle_create(dtype, mode))
        self.dtype = dtype

    def __del__(self):
        try:
            self.L.oracle_destroy(self.h)
        except Exception:
            pass

    def initialize(self, cfg, imu29):
        cam, noise, params = pack_config(cfg)
        a, pa = _d(cam); b, pb = _d(noise); c, pc = _d(params); d, pd = _d(imu29)
        self.L.oracle_initialize(self.h, pa, pb, pc, pd)

    def propagate(self, readings):
        r, p = _d(np.asarray(readings).reshape(-1, 7))
        self.L.oracle_propagate(self.h, p, r.shape[0])

    def augmentState(self, state_id, time=0.0):
        self.L.oracle_augment(self.h, int(state_id), C.c_double(time))

    def update(self, meas, ids):
        m, pm = _d(np.asarray(meas).reshape(-1, 2)); i = np.ascontiguousarray(ids, dtype=np.uint64)
        self.L.oracle_update(self.h, pm, i.ctypes.data_as(_up), len(i))

    def addFeatures(self, meas, ids):
        m, pm = _d(np.asarray(meas).reshape(-1, 2)); i = np.ascontiguousarray(ids, dtype=np.uint64)
        self.L.oracle_add_features(self.h, pm, i.ctypes.data_as(_up), len(i))

    def marginalize(self):
        self.L.oracle_marginalize(self.h)

    def pruneRedundantStates(self):
        self.L.oracle_prune_redundant(self.h)

    def pruneEmptyStates(self):
        self.L.oracle_prune_empty(self.h)

    def finish(self):
        self.L.oracle_finish(self.h)

    def getNumCamStates(self):
        return self.L.oracle_num_cam_states(self.h)

    def getImuState(self):
        o = np.zeros(29); self.L.oracle_get_imu_state(self.h, o.ctypes.data_as(_dp)); return o

    def setImuState(self, s):
        a, p = _d(s); self.L.oracle_set_imu_state(self.h, p)

    def getCamStates(self):
        n = self.getNumCamStates()
        o = np.zeros((n, 7)); ids = np.zeros(n, dtype=np.int32)
        if n:
            self.L.oracle_get_cam_states(self.h, o.ctypes.data_as(_dp), ids.ctypes.data_as(_ip))
        return o, ids

    def getCamMeta(self):
        """(time[n], len(tracked_feature_ids)[n], last_correlated_id[n]) of getCamStates() (types.h:57-67)"""
        n = self.getNumCamStates()
        t = np.zeros(max(n, 1)); k = np.zeros(max(n, 1), dtype=np.int32); lc = np.zeros(max(n, 1), dtype=np.int32)
        self.L.oracle_get_cam_meta(self.h, t.ctypes.data_as(_dp), k.ctypes.data_as(_ip), lc.ctypes.data_as(_ip), n)
        return t[:n], k[:n], lc[:n]

    def getPrunedStates(self, cap=65536):
        """rows of q_CG(4) p_C_G(3) time state_id, sorted by state_id (getPrunedStates, msckf.h:840-848)"""
        o = np.zeros((cap, 9)); n = self.L.oracle_pruned_states(self.h, o.ctypes.data_as(_dp), cap); return o[:n]

    def chi2Table(self):
        """impl="ref" only: chi_squared_test_table as msckf.h:91-95 built it"""
        o = np.zeros(128); n = self.L.oracle_chi2_table(self.h, o.ctypes.data_as(_dp), 128); return o[:n]

    def setCamPose(self, i, qp):
        a, p = _d(qp); self.L.oracle_set_cam_pose(self.h, int(i), p)

    def getCovariance(self):
        D = 15 + 6 * self.getNumCamStates()
        P = np.zeros((D, D), order="F"); self.L.oracle_get_covariance(self.h, P.ctypes.data_as(_dp)); return np.array(P)

    def setCovariance(self, P):
        P = np.asfortranarray(P, dtype=np.float64)
        self.L.oracle_set_covariance(self.h, P.ctypes.data_as(_dp), P.shape[0])

    def setTracks(self, M, slots, obs):
        """M[F]; slots / obs flattened over tracks (sum(M) ints, sum(M) x 2 doubles)."""
        Ma, pM = _i(M); s, ps = _i(slots); o, po = _d(obs)
        self.L.oracle_set_tracks(self.h, len(Ma), pM, ps, po)

    def getTracks(self, cap_f=4096, cap_m=128):
        M = np.zeros(cap_f, dtype=np.int32); sl = np.zeros((cap_f, cap_m), dtype=np.int32); ob = np.zeros((cap_f, cap_m, 2))
        F = self.L.oracle_get_tracks(self.h, M.ctypes.data_as(_ip), sl.ctypes.data_as(_ip), ob.ctypes.data_as(_dp), cap_f, cap_m)
        assert F >= 0
        return M[:F], sl[:F], ob[:F]

    def dropOldest(self, n):
        self.L.oracle_drop_oldest(self.h, int(n))

    def lastStats(self):
        o = np.zeros(7, dtype=np.int32); self.L.oracle_last_stats(self.h, o.ctypes.data_as(_ip))
        return dict(zip(["n_tracks", "n_motion_rejected", "n_tri_rejected", "n_gate_rejected", "n_passed", "m_rows", "r_rows"], o.tolist()))

    def lastTracks(self, cap=8192):
        o = np.zeros((cap, 8)); n = self.L.oracle_last_tracks(self.h, o.ctypes.data_as(_dp), cap); return o[:n]

    def lastDeltaX(self, cap=4096):
        o = np.zeros(cap); n = self.L.oracle_last_deltax(self.h, o.ctypes.data_as(_dp), cap); return o[:n]

    def getMap(self, cap=8192):
        o = np.zeros((cap, 3)); n = self.L.oracle_map_points(self.h, o.ctypes.data_as(_dp), cap); return o[:n]

    def getPrunedIds(self, cap=65536):
        o = np.zeros(cap, dtype=np.int32); n = self.L.oracle_pruned_ids(self.h, o.ctypes.data_as(_ip), cap); return o[:n]

    def numResidualized(self):
        return self.L.oracle_num_residualized(self.h)

    def clone(self):
        c = Oracle.__new__(Oracle)
        c.L, c.dtype, c.impl = self.L, self.dtype, self.impl
        c.h = C.c_void_p(self.L.oracle_clone(self.h))
        return c

    def setWhiten(self, on=True):
        self.L.oracle_set_whiten(self.h, 1 if on else 0)

    def setColPivNull(self, on=True):
        """null-space basis of H_f_j: column-pivoted Householder Q (= JacobiSVD's trailing U columns, the reference;
        default) or unpivoted reflectors"""
        self.L.oracle_set_colpiv_null(self.h, 1 if on else 0)

    def setTinyRowTol(self, tol):
        """restatement only: zero-tail rule of HouseholderQR(H_o) with a rounding threshold (a tail below tol * |column| is the
        zero it stands for; rows of R with all entries below tol * max|R| are dropped) -- the reference's algorithm in its
        exact-arithmetic limit; 0 (default) = msckf.h:1343-1348 to the letter"""
        self.L.oracle_set_tiny_row_tol(self.h, C.c_double(float(tol)))

    def setCapture(self, on=True):
        self.L.oracle_set_capture(self.h, 1 if on else 0)

    def setMode(self, mode):
        self.L.oracle_set_mode(self.h, int(mode))

    def setNumResidualized(self, n):
        self.L.oracle_set_num_residualized(self.h, C.c_long(int(n)))


def time_updates(oracles, n_threads, reps, readings, state_id0, M, slots, obs, n_drop):
    """Wall seconds for every filter in `oracles` to run `reps` filter updates (oracle_time_updates)."""
    L = oracles[0].L
    hs = (C.c_void_p * len(oracles))(*[o.h for o in oracles])
    r, pr = _d(np.asarray(readings).reshape(-1, 7)); Ma, pM = _i(M); s, ps = _i(slots); o, po = _d(obs)
    return L.oracle_time_updates(hs, len(oracles), int(n_threads), int(reps), pr, r.shape[0], int(state_id0), len(Ma), pM, ps, po, int(n_drop))
