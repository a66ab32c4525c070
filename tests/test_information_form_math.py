"""The algebra behind the information-form compression of kernels_gram.hip, checked on the CPU with the numpy twin
(no GPU): for the tracks of one marginalize call
    H_o^T H_o = blockdiag(sum h^T h) - sum_j B_j^T B_j ,  B_j = Q_f^T H_x_j ,
its Cholesky factor with semi-definite pivot skipping is R of the stacked QR up to row signs (T^T T = R^T R), the
skipped pivots are the unobservable directions of the window, and the Kalman update computed from [T | r_n] equals the
one computed from the QR's [R | Q^T r]."""
import numpy as np
import scipy.linalg as sla

from msckf_mono_amd import scenario as sc
import np_oracle


def _capture(N=8, F=24, nf=13, seed_traj=5):
    cfg = sc.filter_config(N)
    tr = sc.Trajectory(2, seed_traj, N, F, nf, cfg)
    flt = np_oracle.NpMSCKF(cfg, tr.imu0, nullspace="householder")
    cap = {}
    orig = flt.jac

    def jac(p_f, slots, obs):          # same maths as NpMSCKF.jac, keeping the unprojected blocks
        Ho, ro, A_j = orig(p_f, slots, obs)
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
        cap.setdefault("cur", []).append((Hx, Hf, r, Ho, ro))
        return Ho, ro, A_j
    flt.jac = jac
    orig_mu = flt.measurement_update
    out = {}

    def mu(H, r, R):
        out["H"], out["r"], out["P"], out["tracks"] = H.copy(), r.copy(), flt.P.copy(), cap.pop("cur")
        orig_mu(H, r, R)
    flt.measurement_update = mu
    for k in range(nf):
        for rd in tr.imu_for_frame(k):
            flt.propagate(rd)
        flt.augment(k)
        fr = tr.frames[k]
        cap.pop("cur", None)
        if len(fr["M"]):
            flt.set_tracks(fr["M"], fr["slots"], fr["obs"])
            flt.marginalize()
        if len(flt.cams) == N:
            flt.drop_oldest(1)
    return cfg, out


def _chol_skip(Lam_hat, n, tol=64 * np.finfo(float).eps):
    """lower Cholesky of the (n+1)x(n+1) matrix [H|r]^T[H|r] over the first n pivots, zero column for a skipped pivot"""
    A = Lam_hat.copy()
    L = np.zeros((n + 1, n))
    d0 = np.diag(A).copy()
    skipped = 0
    for k in range(n):
        if not A[k, k] > tol * d0[k]:
            skipped += 1
            continue
        L[k:, k] = A[k:, k] / np.sqrt(A[k, k])
        A[k:, k:] -= np.outer(L[k:, k], L[k:, k])
    return L, skipped


def test_gram_of_the_projected_stack_and_its_factor():
    cfg, c = _capture()
    H, r, P = c["H"], c["r"], c["P"]
    n = H.shape[1] - 15
    Hc = np.hstack([H[:, 15:], r[:, None]])
    assert sum(t[3].shape[0] for t in c["tracks"]) == H.shape[0]     # every track of this frame passed the gate
    # information form from the UNPROJECTED per-track blocks
    Lam = np.zeros((n + 1, n + 1))
    for Hx, Hf, rr, _, _ in c["tracks"]:
        Hh = np.hstack([Hx[:, 15:], rr[:, None]])
        Qf = np.linalg.qr(Hf, mode="reduced")[0]
        B = Qf.T @ Hh
        Lam += Hh.T @ Hh - B.T @ B
    G = Hc.T @ Hc
    assert np.linalg.norm(Lam - G) < 1e-12 * np.linalg.norm(G)
    # its factor against the QR of the stack
    L, skipped = _chol_skip(Lam, n)
    T, rn = L[:n].T, L[n]
    Rq = np.linalg.qr(Hc, mode="r")
    R, Qtr = np.triu(Rq[:n, :n]), Rq[:n, n]
    assert np.linalg.norm(T.T @ T - R.T @ R) < 1e-11 * np.linalg.norm(R.T @ R)
    assert np.linalg.norm(T.T @ rn - R.T @ Qtr) < 1e-11 * np.linalg.norm(R.T @ Qtr)
    # skipped pivots = rank deficiency of the stack (gauge freedoms + cameras no track of this frame sees)
    sv = np.linalg.svd(H[:, 15:], compute_uv=False)
    assert skipped == int(np.sum(sv < 1e-9 * sv[0])) > 0
    # and the update is the same
    sig2 = cfg["u_var_prime"]

    def update(Tm, rv):
        TH = np.zeros((Tm.shape[0], P.shape[0])); TH[:, 15:] = Tm
        S = TH @ P @ TH.T + sig2 * np.eye(Tm.shape[0])
        K = np.linalg.solve(S, TH @ P).T
        A = np.eye(P.shape[0]) - K @ TH
        return K @ rv, A @ P @ A.T + sig2 * K @ K.T
    dx1, P1 = update(T, rn)
    dx2, P2 = update(R, Qtr)
    assert np.linalg.norm(dx1 - dx2) < 1e-9 * np.linalg.norm(dx2)
    assert np.linalg.norm(P1 - P2) < 1e-10 * np.linalg.norm(P2)


def test_gate_pair_enumeration_covers_the_upper_triangle_once():
    """k_feature's G = H_x P_cc H_x^T loop (kernels_feature.hip) enumerates the camera pairs (a, b), a <= b, of a track as a
    rectangle of ceil(M / 2) rows of width M | 1 -- row k holds row k of the triangle followed by row M - 1 - k (M even) or
    M - k (M odd) -- with the row index taken from a float product: every pair exactly once, for every track length the
    kernel accepts, in the float arithmetic the kernel uses."""
    f32 = np.float32
    for M in range(2, 65):
        Wd = M | 1
        invW = f32(1.0) / f32(Wd)
        seen = set()
        for p in range(M * (M + 1) // 2):
            k = int((f32(p) + f32(0.5)) * invW)
            c = p - k * Wd
            first = c < M - k
            a = k if first else (M - k if (M & 1) else M - 1 - k)
            b = a + (c if first else c - (M - k))
            assert 0 <= a <= b < M, (M, p, a, b)
            assert (a, b) not in seen, (M, p, a, b)
            seen.add((a, b))
        assert len(seen) == M * (M + 1) // 2


def test_syrk_workgroup_enumeration_covers_the_upper_block_triangle_once():
    """k_gram's SYRK launch (kernels_gram.hip): workgroup index -> (block row ti, first tile tj0) with GT_MAX tiles per
    workgroup; with the shipped GT_MAX = 1 every 64 x 64 tile of the upper block triangle is one workgroup."""
    for GT_MAX in (1, 3):
        for np_cap in range(1, 7):
            npairs = sum((np_cap - ti + GT_MAX - 1) // GT_MAX for ti in range(np_cap))
            tiles = []
            for bx in range(npairs):
                rem, ti, tj0 = bx, 0, 0
                for ti in range(np_cap):
                    ng = (np_cap - ti + GT_MAX - 1) // GT_MAX
                    if rem < ng:
                        tj0 = ti + GT_MAX * rem
                        break
                    rem -= ng
                tiles += [(ti, tj) for tj in range(tj0, min(tj0 + GT_MAX, np_cap))]
            assert sorted(tiles) == [(i, j) for i in range(np_cap) for j in range(i, np_cap)], (GT_MAX, np_cap)


def test_xcd_item_mapping_keeps_a_trajectory_on_one_xcd_and_covers_every_item_once():
    """dev_common.h xcd_item / xcd_grid (k_feature, k_gram, the GEMM tiles, the gain Cholesky's parts, k_trsm_rows): MI355X
    places workgroup w on XCD w % 8; workgroup w = x + 8 j serves trajectory x + 8 (j // items), item j % items.  Every
    (trajectory, item) pair exactly once, all items of trajectory i on XCD i % 8, padding workgroups only where an XCD has
    fewer trajectories than the fullest one."""
    for nb in (1, 3, 8, 16, 21, 64, 65):
        for items in (1, 4, 10, 16, 200):
            grid = 8 * ((nb + 7) // 8) * items
            seen = {}
            for w in range(grid):
                x, j = w & 7, w >> 3
                q = j // items
                i, item = x + 8 * q, j - q * items
                if i >= nb:
                    continue                                  # padding workgroup: returns at once
                assert (i, item) not in seen
                seen[(i, item)] = w % 8
            assert len(seen) == nb * items
            assert all(xcd == i % 8 for (i, _), xcd in seen.items())


def test_shared_s_product_ownership_partitions_the_blocks():
    """kernels_chol.hip s_owner: which of a trajectory's NPART gain-solve workgroups forms the S blocks of register set
    (ii, jj) (block rows 4 ii .. 4 ii + 3, block columns 4 jj .. 4 jj + 3, jj <= ii).  Every set has exactly one owner below
    NPART, and with the cost of a block row (its number of k-blocks, NB - i: T_H is upper triangular) the most loaded part
    carries about a third of the product, not all of it."""
    def owner(ii, jj, NPART):
        return (0 if ii == 0 else (1 + jj if ii == 1 else (3 if jj == 2 else 1 + jj))) % NPART
    for NB, NPART in ((4, 3), (8, 3), (12, 4)):
        sets = [(ii, jj) for ii in range(NB // 4) for jj in range(ii + 1)]
        load = [0.0] * NPART
        for ii, jj in sets:
            o = owner(ii, jj, NPART)
            assert 0 <= o < NPART
            blocks = [(i, j) for i in range(4 * ii, 4 * ii + 4) for j in range(4 * jj, 4 * jj + 4) if j <= i]
            load[o] += sum(NB - i for i, j in blocks)
        total = sum(load)
        assert abs(total - sum(NB - i for i in range(NB) for j in range(i + 1))) < 1e-9
        if NB == 12:
            assert max(load) / total < 0.40, load       # 364 k-block products: 136 | 98 | 98 | 32


def test_counter_barrier_target_is_wrap_safe():
    """the gain solve's rendezvous (4 parts, or 3 with part 0 counting twice; the group of 16 is kept as a generic case): a
    workgroup draws `old` from a counter that only grows and waits until the counter has reached the next multiple of the
    group size GP above it, compared as (int)(counter - target) >= 0.  GP is a power of two, so launch after launch the
    workgroups of a trajectory release together -- also across the 2^32 wrap."""
    M32 = 1 << 32
    for adds in ([1] * 16, [1] * 4, [2, 1, 1]):           # what each arriving workgroup adds; GP = sum
        GP = sum(adds)
        assert GP & (GP - 1) == 0
        counter = (M32 - 3 * GP) % M32                    # wraps in the fourth launch
        for launch in range(8):
            targets = []
            for k, inc in enumerate(adds):
                old = counter
                counter = (counter + inc) % M32
                targets.append(((old // GP + 1) * GP) % M32)
                done = (counter - targets[-1]) % M32
                done = done - M32 if done >= (1 << 31) else done   # the (int) cast
                assert (done >= 0) == (k == len(adds) - 1), (adds, launch, k)
            assert len(set(targets)) == 1


def test_prune_on_the_downdate_index_map():
    """k_gemm_mfma<OP_DOWN> with Dev::Pout: dropping the nd oldest camera states sends row / column g of the covariance to
    g (IMU block), nowhere (15 <= g < 15 + 6 nd) or g - 6 nd -- the same as P[keep][:, keep] with keep = the IMU block and
    the camera states nd .. N - 1 (matrix_utils.h:58-87 square_slice)."""
    rng = np.random.default_rng(5)
    for N, nd in ((5, 0), (5, 1), (8, 3), (4, 4)):
        D = 15 + 6 * N
        P = rng.standard_normal((D, D)); P = P + P.T
        cut = 15 + 6 * nd
        out = np.full((D, D), np.nan)
        for gi in range(D):
            for gj in range(gi, D):                         # upper triangle, mirrored (as the tiles do)
                if (gi < 15 or gi >= cut) and (gj < 15 or gj >= cut):
                    di, dj = (gi if gi < 15 else gi - 6 * nd), (gj if gj < 15 else gj - 6 * nd)
                    out[dj, di] = out[di, dj] = P[gi, gj]
        keep = list(range(15)) + list(range(cut, D))
        ref = P[np.ix_(keep, keep)]
        Dn = len(keep)
        assert np.array_equal(out[:Dn, :Dn], ref)


def test_epilogues_of_the_update_kernels_hold_no_serialized_store_load_chains(tmp_path):
    """Stores count in vmcnt on gfx950, so `P[i] = P[i] - acc[i]` element by element compiles to load -> wait (also for the
    previous stores' acknowledgement) -> store, once per element: the covariance downdate and k_gram's epilogue were written
    that way (sixteen serial round trips each) until round 3.  hipcc cross-compiles here: the ISA of those kernels is
    checked for store -> load -> s_waitcnt vmcnt(0) sequences (scripts/experiments/store_load_chains.py)."""
    import os, shutil, subprocess, sys
    from concurrent.futures import ThreadPoolExecutor
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        import pytest
        pytest.skip("no hipcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "scripts", "experiments"))
    from store_load_chains import count_chains

    def isa(name):
        out = str(tmp_path / (name + ".s"))
        subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=fast", "-fno-slp-vectorize", "-S", "--cuda-device-only",
                        "-I" + os.path.join(root, "include"), "-o", out, os.path.join(root, "msckf_mono_amd", "csrc", name + ".hip")],
                       check=True, capture_output=True, timeout=600)
        return count_chains(out)
    with ThreadPoolExecutor(2) as ex:
        kal, gram = ex.map(isa, ["kernels_kalman", "kernels_gram"])
    checked = 0
    for name, n in list(kal.items()) + list(gram.items()):
        if ("k_gemm_mfma" in name and ("ILi8E" in name or "ILi9E" in name or "ILi0E" in name)) or name.startswith("_ZN5msckf6k_gramI"):
            assert n <= 1, (name, n)
            checked += 1
    assert checked >= 6, (checked, sorted(kal)[:5])
