"""Property tests (hypothesis) for invariants the reference holds by construction:
  * the covariance is symmetric after every public call                         msckf.h:143, :197, :1401-1403
  * A_j^T H_f_j = 0 and A_j orthonormal (left null space)                        msckf.h:954-957
  * the gate statistic gamma, the gate decisions and -- with isotropic pixel noise -- the whole update do not depend
    on which orthonormal null-space basis is used                                SURVEY 8a Q1b
  * square_slice of a symmetric PSD matrix stays symmetric PSD                   matrix_utils.h:58-68 (pruning)
CPU: on the oracle and on the reference's own source; GPU (-m gpu): the same invariants on the HIP path."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

import helpers as H
from msckf_mono_amd import scenario as sc

SET = dict(max_examples=8, deadline=None, derandomize=True)


def _frames(f, tr, N, nf, after=None):
    for k in range(nf):
        f.propagate(tr.imu_for_frame(k))
        if after: after(f, "propagate", k)
        f.augmentState(k, tr.frame_times[k])
        if after: after(f, "augment", k)
        fr = tr.frames[k]
        if len(fr["M"]):
            f.setTracks(fr["M"], fr["slots"], fr["obs"]); f.marginalize()
            if after: after(f, "marginalize", k)
        if f.getNumCamStates() == N:
            f.dropOldest(1)
            if after: after(f, "prune", k)


@settings(**SET)
@given(traj=st.integers(0, 10_000), N=st.integers(5, 9), F=st.integers(4, 20), f32=st.booleans())
def test_covariance_symmetric_psd_after_every_public_call(oracle_lib, traj, N, F, f32):
    po = oracle_lib
    tr = sc.Trajectory(2, traj, N, F, N + 6)
    impls = ["oracle"] + (["ref"] if po.ref_available() else [])
    for impl in impls:
        f = po.Oracle(po.F32 if f32 else po.F64, po.LEAN, impl=impl)
        f.initialize(tr.cfg, tr.imu0)

        def check(f, stage, k):
            P = f.getCovariance()
            assert np.array_equal(P, P.T), (impl, stage, k)                       # exactly symmetric, as (P + P^T)/2 leaves it
            if stage in ("marginalize", "prune") and not f32:
                assert np.linalg.eigvalsh(P).min() > -1e-12 * np.abs(P).max(), (impl, stage, k)
        _frames(f, tr, N, N + 6, check)


@settings(**SET)
@given(traj=st.integers(0, 10_000), N=st.integers(6, 9), F=st.integers(6, 24))
def test_update_is_invariant_to_the_null_space_basis_under_isotropic_noise(oracle_lib, traj, N, F):
    """column-pivoted Householder Q (JacobiSVD's trailing U columns) vs unpivoted reflectors: different bases of the same
    left null space -> same gamma, same decisions, same state and covariance (isotropic noise only)"""
    po = oracle_lib
    tr = sc.Trajectory(2, traj, N, F, N + 5)
    a, b = po.Oracle(po.F64, po.LEAN), po.Oracle(po.F64, po.LEAN)
    b.setColPivNull(False)
    a.initialize(tr.cfg, tr.imu0); b.initialize(tr.cfg, tr.imu0)
    for k in range(N + 5):
        H.oracle_frame(a, tr, k, N); H.oracle_frame(b, tr, k, N)
        ta, tb = a.lastTracks(), b.lastTracks()
        assert np.array_equal(ta[:, :4], tb[:, :4]), k                          # motion / triangulation / gate decisions, rows
        assert np.allclose(ta[:, 4], tb[:, 4], rtol=1e-9, atol=1e-12), k        # gamma
        e = H.state_errors(a.getImuState(), b.getImuState(), a.getCamStates()[0], b.getCamStates()[0], a.getCovariance(), b.getCovariance())
        assert H.worst(e) < 1e-8, (k, e)


@settings(**SET)
@given(seed=st.integers(0, 2**31 - 1), M=st.integers(2, 29))
def test_left_null_space_of_the_reference_is_orthonormal_and_annihilates_h_f(oracle_lib, seed, M):
    """matrixU().rightCols(2M-3) of the shim's JacobiSVD (what msckf.h:954-955 reads)"""
    import ctypes as C
    po = oracle_lib
    if not po.ref_available():
        pytest.skip("no oracle/_ref/lib_ref.so")
    L = po.lib("ref")
    rng = np.random.default_rng(seed)
    Hf = np.asfortranarray(rng.standard_normal((2 * M, 3)) * rng.uniform(0.01, 10, 3))
    U = np.zeros((2 * M, 2 * M), order="F"); V = np.zeros((3, 3), order="F"); sv = np.zeros(3)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    L.shim_svd(1, 2 * M, 3, dp(Hf), dp(U), dp(V), dp(sv))
    A = U[:, 3:]
    assert np.abs(A.T @ Hf).max() < 1e-12 * np.abs(Hf).max() * 2 * M
    assert np.abs(A.T @ A - np.eye(2 * M - 3)).max() < 1e-12


@settings(**SET)
@given(seed=st.integers(0, 2**31 - 1), n=st.integers(2, 8), drop=st.integers(1, 3))
def test_pruning_keeps_the_covariance_symmetric_psd(oracle_lib, seed, n, drop):
    """square_slice / column_slice through dropOldest (oracle) == the same gather in numpy"""
    po = oracle_lib
    drop = min(drop, n - 1)
    rng = np.random.default_rng(seed)
    D = 15 + 6 * n
    B = rng.standard_normal((D, D)); P = B @ B.T / D
    tr = sc.Trajectory(2, 1, n, 0, 1)
    f = po.Oracle(po.F64, po.LEAN)
    f.initialize(tr.cfg, tr.imu0)
    for i in range(n):
        f.augmentState(i, 0.0)
    f.setCovariance(P)
    f.dropOldest(drop)
    keep = np.r_[0:15, 15 + 6 * drop:D]
    Q = f.getCovariance()
    assert np.array_equal(Q, P[np.ix_(keep, keep)])
    assert np.array_equal(Q, Q.T) and np.linalg.eigvalsh(Q).min() > 0


# ---------------------------------------------------------------------------------------------------- HIP path
@pytest.mark.gpu
@settings(max_examples=5, deadline=None, derandomize=True)
@given(traj=st.integers(0, 10_000), N=st.integers(5, 12), F=st.integers(4, 30), f32=st.booleans())
def test_hip_covariance_bit_symmetric_after_every_stage(traj, N, F, f32):
    from msckf_mono_amd import capi
    tr = sc.Trajectory(2, traj, N, F, N + 6)
    bt = capi.Batch(1, N, max(F, 1), max(N, 4), capi.F32 if f32 else capi.F64)
    bt.initialize(0, tr.cfg, tr.imu0)

    def check(stage, k):
        P = bt.covariance(0)
        assert np.array_equal(P, P.T), (stage, k)
        if stage in ("marginalize", "prune") and not f32:
            assert np.linalg.eigvalsh(P).min() > -1e-12 * np.abs(P).max(), (stage, k)
    for k in range(N + 6):
        bt.propagate_range(0, 1, tr.imu_for_frame(k)); check("propagate", k)
        bt.augment_range(0, 1); check("augment", k)
        fr = tr.frames[k]
        bt.set_tracks(0, fr["M"], fr["slots"], fr["obs"])
        if len(fr["M"]):
            bt.marginalize_range(0, 1); check("marginalize", k)
        if bt.num_cam_states(0) == N:
            bt.drop_oldest_range(0, 1, 1); check("prune", k)
    bt.close()
