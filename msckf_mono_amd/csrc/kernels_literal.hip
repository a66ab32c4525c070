// kernels_literal.hip -- the reference's compression under anisotropic pixel noise (u_var' != v_var'), built the reference's
// way on the device: R_o_j = A_j^T R_j A_j per track, HouseholderQR of the stack in column order with the zero-tail rule
// (rows 0..14 of H_o verbatim), R_n = Q_1^T R_o Q_1 (msckf.h:423-431, 1343-1366) -- see literal_core.h, which holds the
// algorithm (shared with the host build that tests/test_literal_core.py checks against the oracle on the CPU).
//
// What the update needs of (T_H, r_n, R_n) is Lam^ = [T_H | r_n]^T R_n^-1 [T_H | r_n], which depends on Q_1 only through its
// range: any basis of that range gives the same matrix (literal_core.h: literal_compact).  Three launches:
//   k_lit_pre    all trajectories, one wavefront per item: (a) per track the column-pivoted Householder QR of H_f_j
//                (lanes = observations, the reflectors in registers) -> V, T of A_j, and the six rows [Q_f^T H_x ; D] the track
//                contributes to Gam = sum_j (u-rows of (I - Q_f Q_f^T) H_x_j)^T (the same); (b) per camera slot the block-diagonal
//                part of Gam; (c) per trajectory the row / observation offsets of the stacked tracks in list order
//   k_lit_gamma  Gam on the f64 matrix cores: one wavefront per 32 x 32 tile of the lower triangle, twelve contraction rows (two
//                tracks) per three v_mfma_f64_16x16x4_f64 steps, operands straight from global memory (L2), tracks whose cameras
//                miss the tile skipped
//   k_literal    one workgroup of 1024 threads per trajectory: explicit rows, the sweep for its decisions (panels of 16 in
//                LDS), the basis, Z = [[Bs^T R_o Bs, .], [(Bs^T A)^T, 0]] and its blocked elimination
// The sweep over the dense stack (literal_general) stays selectable with MSCKF_HIP_LITERAL_ROUTE=1 for tests and A/B runs.  The
// output is the information matrix in the place where k_gram leaves H_o^T H_o, so that the blocked Cholesky and the Kalman
// stage run unchanged with sigma^2 = 1.
#include "dev_common.h"
#include "literal_core.h"

namespace msckf {

constexpr int LIT_LDS_DOUBLES = 12000;   // 94 KB: a panel of 16 steps of the sweep at a 30-camera window (one workgroup per compute unit)

long lit_ws_doubles(int n6, int m_cap, int r_cap) { return lit::compact_ws_doubles(n6, m_cap, r_cap, 0); }

template <class S>
__device__ __forceinline__ lit::Args<S> lit_args(const Dev<S>& d, int bi, int b) {
  const LitBufs& L = d.lit;
  const S* prm = d.prm + (long)b * PRM_STRIDE;
  const int n1 = d.n6cap + 1;
  lit::Args<S> a;
  a.F = d.trk_n[(long)bi * d.wl_stride_n];
  a.m_cap = d.m_cap;
  a.N = d.ncam[b];
  a.status = d.trk_status + (long)b * d.f_cap; a.inc_bit = ST_INCLUDED;
  a.M = d.trk_M + (long)bi * d.wl_stride_f;
  if (d.trk_off) { a.slots = d.trk_slots; a.off = d.trk_off + (long)bi * d.wl_stride_f; }
  else { a.slots = d.trk_slots + (long)bi * d.wl_stride_o; a.off = nullptr; }
  a.Hx = d.trk_Hx + (long)b * d.f_cap * d.m_cap * 12;
  a.rw = d.trk_rw + (long)b * d.f_cap * 2 * d.m_cap;
  a.u_var = (double)prm[PRM_UVAR]; a.v_var = (double)prm[PRM_VVAR]; a.tol = L.tol;
  a.ldx = L.ldx; a.X = L.X ? L.X + (long)b * L.ldx * n1 : nullptr;
  a.tau = L.tau + (long)b * (2 * n1 + 2);
  a.Vf = L.Vf + (long)b * d.f_cap * 2 * d.m_cap * 3; a.Tf = L.Tf + (long)b * d.f_cap * 9;
  a.row0 = L.row0 + (long)b * (d.f_cap + 1); a.obs0 = L.obs0 + (long)b * (d.f_cap + 1); a.otrk = L.otrk + (long)b * L.ldg;
  a.kept = L.kept + (long)b * L.kept_stride;
  a.r_cap = L.r_cap; a.TH = L.TH + (long)b * L.r_cap * n1;
  a.ldg = L.ldg; a.G = L.G ? L.G + (long)b * L.ldg * L.r_cap : nullptr;
  a.ldz = L.ldz; a.Z = L.Z + (long)b * L.ldz * L.ldz;
  a.Lam = d.Lam + (long)b * d.ldR * d.ldR; a.ldL = d.ldR;
  a.info = L.info + (long)b * 8;
  a.LamIn = a.Lam; a.lam_part = d.lam_part; a.gram_parts = (d.compress && d.ldR <= 192 && d.lam_part > 0) ? (d.gram_parts >= 3 ? d.gram_parts : 3) : 1;
  a.W2 = L.W2 + (long)b * L.w2_stride;
  a.Gam = L.Gam + (long)b * d.ldR * d.ldR; a.ldGam = d.ldR;
  return a;
}

// ---------------------------------------------------------------------------------------------------------------------
// k_lit_pre: one wavefront per item of a trajectory: items [0, f_cap) tracks, [f_cap, f_cap + n_cap) camera slots, the last
// one the offsets.  serial != 0 (MSCKF_HIP_LITERAL_SERIAL=1: A/B runs): the track items run literal_core.h's serial reference
// (track_null_space + gamma_rows) on one lane instead of the wavefront-parallel factorization.
template <class S>
__global__ __launch_bounds__(64) void k_lit_pre(Dev<S> d, int b0, int nb, int items, int serial) {
  int bi, item;
  if (!xcd_item(nb, items, bi, item)) return;
  const int b = b0 + bi, lane = threadIdx.x;
  const S* prm = d.prm + (long)b * PRM_STRIDE;
  if (prm[PRM_LIT] == S(0)) return;
  const int* st = d.stats + (long)b * STAT_STRIDE;
  if (st[STAT_MROWS] == 0) return;
  const LitBufs& L = d.lit;
  const int f_cap = d.f_cap, m_cap = d.m_cap, ldc = d.ldR;
  const int F = d.trk_n[(long)bi * d.wl_stride_n];
  const int* status = d.trk_status + (long)b * f_cap;
  const int* Mv = d.trk_M + (long)bi * d.wl_stride_f;
  if (item == items - 1) {
    // ---- (c) first stacked row / observation of every track in list order (msckf.h:404-441): a chunk of tracks per lane,
    // an exclusive scan over the lanes
    int* row0 = L.row0 + (long)b * (f_cap + 1); int* obs0 = L.obs0 + (long)b * (f_cap + 1);
    const int per = (F + 63) / 64, t_lo = lane * per, t_hi = min(F, t_lo + per);
    int rs = 0, os = 0;
    for (int t = t_lo; t < t_hi; ++t) if (status[t] & ST_INCLUDED) { rs += 2 * Mv[t] - 3; os += Mv[t]; }
    int rinc = rs, oinc = os;
    for (int off = 1; off < 64; off <<= 1) {
      const int r2 = __shfl_up(rinc, off), o2 = __shfl_up(oinc, off);
      if (lane >= off) { rinc += r2; oinc += o2; }
    }
    int r = rinc - rs, o = oinc - os;
    for (int t = t_lo; t < t_hi; ++t) {
      row0[t] = r; obs0[t] = o;
      if (status[t] & ST_INCLUDED) { r += 2 * Mv[t] - 3; o += Mv[t]; }
    }
    if (lane == 63) { row0[F] = rinc; obs0[F] = oinc; int* info = L.info + (long)b * 8; info[0] = rinc; for (int q = 1; q < 8; ++q) info[q] = 0; }
    return;
  }
  if (item >= f_cap) {
    // ---- (b) block-diagonal part of Gam: per camera slot the sum over the stacked tracks of h_u^T h_u (h_u: the u-row of the
    // observation's 2 x 6 Jacobian block), lanes over tracks
    const int s = item - f_cap;
    if (s >= d.ncam[b]) return;
    double acc[21];
#pragma unroll
    for (int q = 0; q < 21; ++q) acc[q] = 0.0;
    for (int t = lane; t < F; t += 64) {
      const long tb = (long)b * f_cap + t;
      const int oi = d.trk_inv[tb * d.n_cap + s];
      if (!(status[t] & ST_INCLUDED) || oi < 0) continue;
      const S* hx = d.trk_Hx + (tb * m_cap + oi) * 12;
      double h0[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) h0[k] = (double)hx[k];
      int q = 0;
#pragma unroll
      for (int x = 0; x < 6; ++x)
#pragma unroll
        for (int y = x; y < 6; ++y) acc[q++] += h0[x] * h0[y];
    }
    double* out = L.Du + ((long)b * d.n_cap + s) * 24;
#pragma unroll
    for (int q = 0; q < 21; ++q) { const double v = wave_sum(acc[q]); if (lane == 0) out[q] = v; }
    return;
  }
  // ---- (a) per track
  const int t = item;
  if (t >= F || !(status[t] & ST_INCLUDED)) return;
  const long tb = (long)b * f_cap + t;
  const int M = Mv[t];
  double* Vg = L.Vf + tb * 2 * m_cap * 3;
  double* Tg = L.Tf + tb * 9;
  double* rows6 = L.BD + tb * 6 * ldc;
  const int fl = d.trk_first[tb], s_lo = fl & 63, s_hi = (fl >> 8) & 63;
  if (serial) {
    lit::Args<S> a = lit_args(d, bi, b);
    for (int x = 6 * s_lo + lane; x < 6 * s_hi + 6; x += 64)
#pragma unroll
      for (int q = 0; q < 6; ++q) rows6[(long)q * ldc + x] = 0.0;
    __threadfence(); __builtin_amdgcn_wave_barrier();
    if (lane == 0) { lit::track_null_space(a, t); lit::gamma_rows(a, t, rows6, ldc); }
    return;
  }
  // Column-pivoted Householder QR of H_f = -H_x(:, 3:6) (2M x 3), lane o holds rows 2o, 2o + 1 (literal_core.h:
  // track_null_space is the serial statement of the same steps; the pivot rule is the oracle's: largest remaining squared
  // column norm over the rows from the pivot row down, first one wins)
  double x[2][3] = {{0, 0, 0}, {0, 0, 0}};
  double hh[12];
#pragma unroll
  for (int q = 0; q < 12; ++q) hh[q] = 0.0;
  for (int o0 = 0; o0 < M; o0 += 64) {     // (M <= 64: one pass)
    const int o = o0 + lane;
    if (o < M) {
      const S* hx = d.trk_Hx + (tb * m_cap + o) * 12;
#pragma unroll
      for (int q = 0; q < 12; ++q) hh[q] = (double)hx[q];
    }
    break;
  }
  const bool mine = lane < M;
  if (mine) {
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) { x[0][cc] = -hh[3 + cc]; x[1][cc] = -hh[9 + cc]; }
  }
  double tau[3] = {0, 0, 0};
#pragma unroll
  for (int k = 0; k < 3; ++k) {                     // (a stacked track has M >= 2: at least four rows, all three steps exist)
    // squared norms of the remaining columns over rows >= k
    const int rk0 = 2 * lane, rk1 = 2 * lane + 1;
    double best = -1.0; int big = k;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      if (j < k) continue;
      double sl = 0;
      if (mine) { if (rk0 >= k) sl += x[0][j] * x[0][j]; if (rk1 >= k) sl += x[1][j] * x[1][j]; }
      const double sj = wave_sum(sl);
      if (sj > best) { best = sj; big = j; }
    }
    if (big != k) {
#pragma unroll
      for (int j = 0; j < 3; ++j)
        if (j == big) {
#pragma unroll
          for (int i = 0; i < 2; ++i) { const double tmp = x[i][k]; x[i][k] = x[i][j]; x[i][j] = tmp; }
        }
    }
    double tl = 0;
    if (mine) { if (rk0 > k) tl += x[0][k] * x[0][k]; if (rk1 > k) tl += x[1][k] * x[1][k]; }
    const double tail2 = wave_sum(tl);
    const double c0 = wave_bcast(x[k & 1][k], k >> 1);
    if (tail2 <= 2.2250738585072014e-308) {
      tau[k] = 0;
      if (rk0 > k) x[0][k] = 0;
      if (rk1 > k) x[1][k] = 0;
      continue;
    }
    double beta = sqrt(c0 * c0 + tail2);
    if (c0 >= 0.0) beta = -beta;
    const double inv = 1.0 / (c0 - beta);
    if (rk0 > k) x[0][k] *= inv;
    if (rk1 > k) x[1][k] *= inv;
    tau[k] = (beta - c0) / beta;
    if (lane == (k >> 1)) x[k & 1][k] = beta;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      if (j <= k) continue;
      double sl = 0;
      if (mine) { if (rk0 > k) sl += x[0][k] * x[0][j]; if (rk1 > k) sl += x[1][k] * x[1][j]; }
      double s = wave_sum(sl) + wave_bcast(x[k & 1][j], k >> 1);
      s *= tau[k];
      if (lane == (k >> 1)) x[k & 1][j] -= s;
      if (rk0 > k) x[0][j] -= s * x[0][k];
      if (rk1 > k) x[1][j] -= s * x[1][k];
    }
  }
  // implicit structure of V (unit lower trapezoidal)
  auto vf = [&](int sub, int q) -> double { const int i = 2 * lane + sub; return i < q ? 0.0 : (i == q ? 1.0 : x[sub][q]); };
  double d01, d02, d12;
  {
    double a01 = 0, a02 = 0, a12 = 0;
    if (mine) {
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) { const double v0 = vf(sub, 0), v1 = vf(sub, 1), v2 = vf(sub, 2); a01 += v0 * v1; a02 += v0 * v2; a12 += v1 * v2; }
    }
    d01 = wave_sum(a01); d02 = wave_sum(a02); d12 = wave_sum(a12);
  }
  double T[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  T[0] = tau[0]; T[4] = tau[1]; T[8] = tau[2];
  T[1] = -tau[1] * T[0] * d01;
  T[2] = -tau[2] * (T[0] * d02 + T[1] * d12);
  T[5] = -tau[2] * T[4] * d12;
  if (mine) {
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int cc = 0; cc < 3; ++cc) Vg[(2 * lane + sub) * 3 + cc] = x[sub][cc];
  }
  if (lane == 0) {
#pragma unroll
    for (int q = 0; q < 9; ++q) Tg[q] = T[q];
  }
  // Z3 = T V(0:3, :)^T from the rows 0, 1 (lane 0) and 2 (lane 1)
  double v3[3][3];
#pragma unroll
  for (int cc = 0; cc < 3; ++cc)
#pragma unroll
    for (int q = 0; q < 3; ++q) { const double raw = wave_bcast(x[cc & 1][q], cc >> 1); v3[cc][q] = cc < q ? 0.0 : (cc == q ? 1.0 : raw); }
  double Z3[9];
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) { double acc = 0; for (int q = p; q < 3; ++q) acc += T[p * 3 + q] * v3[cc][q]; Z3[p * 3 + cc] = acc; }
  double q0[3], q1[3];
#pragma unroll
  for (int cc = 0; cc < 3; ++cc) {
    q0[cc] = (2 * lane == cc ? 1.0 : 0.0) - (vf(0, 0) * Z3[cc] + vf(0, 1) * Z3[3 + cc] + vf(0, 2) * Z3[6 + cc]);
    q1[cc] = (2 * lane + 1 == cc ? 1.0 : 0.0) - (vf(1, 0) * Z3[cc] + vf(1, 1) * Z3[3 + cc] + vf(1, 2) * Z3[6 + cc]);
  }
  if (!mine) { q0[0] = q0[1] = q0[2] = 0.0; q1[0] = q1[1] = q1[2] = 0.0; }
  double W[9];
#pragma unroll
  for (int a1 = 0; a1 < 3; ++a1)
#pragma unroll
    for (int a2 = a1; a2 < 3; ++a2) { const double v = wave_sum(q0[a1] * q0[a2]); W[a1 * 3 + a2] = v; W[a2 * 3 + a1] = v; }
  // the rows, lanes along the columns of the track's slot range (zeros where a slot inside the range is unobserved):
  // B = Q_f^T H_x,  D = Q_u^T H_xu - W B / 2; the observation's Q_f rows come from its lane through the crossbar
  // (every lane stays in the loop to its last round: a lane that left could not serve as the crossbar's source)
  for (int col0 = 6 * s_lo; col0 < 6 * s_hi + 6; col0 += 64) {
    const int col = col0 + lane;
    const bool live = col < 6 * s_hi + 6;
    const int s = live ? col / 6 : s_lo, kk = live ? col - 6 * s : 0;
    const int oi = d.trk_inv[tb * d.n_cap + s];
    const int src4 = 4 * (oi < 0 ? 0 : oi);
    double g0[3], g1[3];
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) { g0[cc] = lane_gather(q0[cc], src4); g1[cc] = lane_gather(q1[cc], src4); }
    double h0 = 0, h1 = 0;
    if (oi >= 0) { const S* hx = d.trk_Hx + (tb * m_cap + oi) * 12; h0 = (double)hx[kk]; h1 = (double)hx[6 + kk]; }
    double bb[3], cu[3];
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) { bb[cc] = oi >= 0 ? g0[cc] * h0 + g1[cc] * h1 : 0.0; cu[cc] = oi >= 0 ? g0[cc] * h0 : 0.0; }
    if (live) {
#pragma unroll
      for (int cc = 0; cc < 3; ++cc) {
        rows6[(long)cc * ldc + col] = bb[cc];
        rows6[(long)(3 + cc) * ldc + col] = cu[cc] - 0.5 * (W[cc * 3 + 0] * bb[0] + W[cc * 3 + 1] * bb[1] + W[cc * 3 + 2] * bb[2]);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// k_lit_gamma: Gam = Du - sum_j (B_j^T D_j + D_j^T B_j) (lower triangle), one wavefront per 32 x 32 tile.  With the rows of a
// track ordered [B ; D] on the left and [D ; B] on the right the sum is one product with six contraction rows per track;
// two tracks fill three k = 4 steps of v_mfma_f64_16x16x4_f64.
typedef double lg_v4d __attribute__((ext_vector_type(4)));
template <class S>
__global__ __launch_bounds__(64) void k_lit_gamma(Dev<S> d, int b0, int nb, int ntile) {
  int bi, tile;
  if (!xcd_item(nb, ntile, bi, tile)) return;
  const int b = b0 + bi, lane = threadIdx.x;
  const S* prm = d.prm + (long)b * PRM_STRIDE;
  if (prm[PRM_LIT] == S(0)) return;
  const int* st = d.stats + (long)b * STAT_STRIDE;
  if (st[STAT_MROWS] == 0) return;
  const int P = st[STAT_PASSED], n = 6 * d.ncam[b], ldc = d.ldR, f_cap = d.f_cap;
  int ti = (int)((sqrtf(8.0f * tile + 1.0f) - 1.0f) * 0.5f);
  while (ti * (ti + 1) / 2 > tile) --ti;
  while ((ti + 1) * (ti + 2) / 2 <= tile) ++ti;
  const int tj = tile - ti * (ti + 1) / 2;
  if (32 * tj >= n) return;                                  // (ti >= tj)
  const LitBufs& L = d.lit;
  const int* order = d.trk_order + (long)b * f_cap;
  const int lr = lane & 15, lk = lane >> 4;
  const int ci0 = 32 * ti + lr, ci1 = ci0 + 16, cj0 = 32 * tj + lr, cj1 = cj0 + 16;
  lg_v4d acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = lg_v4d{0.0, 0.0, 0.0, 0.0};
  const double* dummy = L.Du;                                // an always-valid word for the lanes whose column a track does not have
  // the sorted track list and the tracks' slot ranges, 64 per register (four registers: up to 256 stacked tracks are served by
  // v_readlane; a pair's operands then depend on nothing but registers -- two levels of dependent loads per pair otherwise).
  // (Requesting the next pair's operands before this pair's MFMAs did not pay: 0.215 -> 0.238 ms; the kernel is bound by the ~0.9 GB of
  // operands a launch of 128 trajectories moves out of L2, four loads per four MFMAs, not by one pair's latency.)
  int ordv[4], flv[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int e = lane + 64 * k;
    ordv[k] = e < P ? order[e] : 0;
    flv[k] = e < P ? d.trk_first[(long)b * f_cap + ordv[k]] : 0;
  }
  auto track_of = [&](int e, int& t, int& fl) {
    if (e < 256) {
      const int k = e >> 6, l = e & 63;
      const int ov = k == 0 ? ordv[0] : (k == 1 ? ordv[1] : (k == 2 ? ordv[2] : ordv[3]));
      const int fv = k == 0 ? flv[0] : (k == 1 ? flv[1] : (k == 2 ? flv[2] : flv[3]));
      t = wave_bcast(ov, l); fl = wave_bcast(fv, l);
    } else { t = order[e]; fl = d.trk_first[(long)b * f_cap + t]; }
  };
  for (int e0 = 0; e0 < P; e0 += 2) {
    // the pair of tracks (e0, e0 + 1): ranges, overlap with the tile's rows and columns
    int tt[2], lo[2], hi[2]; bool use[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = e0 + u;
      int fl = 0; tt[u] = 0;
      if (e < P) track_of(e, tt[u], fl);
      lo[u] = 6 * (fl & 63); hi[u] = 6 * ((fl >> 8) & 63) + 5;
      use[u] = e < P && lo[u] < 32 * ti + 32 && hi[u] >= 32 * ti && lo[u] < 32 * tj + 32 && hi[u] >= 32 * tj;
    }
    if (!use[0] && !use[1]) continue;
#pragma unroll
    for (int mstep = 0; mstep < 3; ++mstep) {
      const int kk = 4 * mstep + lk, u = kk >= 6 ? 1 : 0, xr = kk - 6 * u;       // contraction row: track u of the pair, row xr of [B ; D]
      const int tu = u ? tt[1] : tt[0], lou = u ? lo[1] : lo[0], hiu = u ? hi[1] : hi[0];
      const bool us = u ? use[1] : use[0];
      const double* Lrow = L.BD + (((long)b * f_cap + tu) * 6 + xr) * ldc;                      // left operand: [B ; D]
      const double* Rrow = L.BD + (((long)b * f_cap + tu) * 6 + (xr + 3) % 6) * ldc;            // right operand: [D ; B]
      const bool ai0 = us && ci0 >= lou && ci0 <= hiu, ai1 = us && ci1 >= lou && ci1 <= hiu;
      const bool bj0 = us && cj0 >= lou && cj0 <= hiu, bj1 = us && cj1 >= lou && cj1 <= hiu;
      const double a0r = *(ai0 ? Lrow + ci0 : dummy), a1r = *(ai1 ? Lrow + ci1 : dummy);
      const double b0r = *(bj0 ? Rrow + cj0 : dummy), b1r = *(bj1 ? Rrow + cj1 : dummy);
      const double a0 = ai0 ? a0r : 0.0, a1 = ai1 ? a1r : 0.0, bv0 = bj0 ? b0r : 0.0, bv1 = bj1 ? b1r : 0.0;
      acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, bv0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, bv1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, bv0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, bv1, acc[1][1], 0, 0, 0);
    }
  }
  // epilogue: C row = (lane >> 4) + 4 r (index of the LEFT operand's column), C column = lane & 15
  double* Gam = L.Gam + (long)b * ldc * ldc;
  const double* Du = L.Du + (long)b * d.n_cap * 24;
#pragma unroll
  for (int ib = 0; ib < 2; ++ib)
#pragma unroll
    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 32 * ti + 16 * ib + lk + 4 * r, j = 32 * tj + 16 * jb + lr;
        if (i >= n || j > i) continue;
        double term = 0.0;
        if (i / 6 == j / 6) { const int a6 = j % 6, c6 = i % 6; term = Du[(i / 6) * 24 + a6 * 6 - a6 * (a6 - 1) / 2 + (c6 - a6)]; }
        Gam[(long)i * ldc + j] = term - acc[ib][jb][r];
      }
}

// The trajectory's context of the four phase kernels / the dense-route kernel; false: nothing to do for this workgroup
template <class S>
__device__ __forceinline__ bool lit_ctx(const Dev<S>& d, int b0, int nb, double* red, double* lds, lit::Ctx& c, int& bi, int& b) {
  bi = blockIdx.x; b = b0 + bi;
  if (bi >= nb) return false;
  const S* prm = d.prm + (long)b * PRM_STRIDE;
  if (prm[PRM_LIT] == S(0)) return false;                    // isotropic (or pre-whitened) trajectory: k_gram's Lam^ stands
  const int* st = d.stats + (long)b * STAT_STRIDE;
  if (st[STAT_MROWS] == 0) return false;
  c.tid = threadIdx.x; c.nt = blockDim.x; c.lane = threadIdx.x & 63; c.wave = threadIdx.x >> 6; c.nw = blockDim.x >> 6; c.red = red;
  c.tim = d.lit.tim ? d.lit.tim + (long)b * LIT_TIM_SLOTS : nullptr; c.lds = lds; c.lds_doubles = LIT_LDS_DOUBLES;
  return true;
}

// The compact route as four launches, one workgroup of 1 024 threads per trajectory each (literal_core.h: compact_rows /
// _sweep / _basis / _eliminate).  As ONE kernel the compiler ran out of its 128 registers per thread and the panel loops
// reloaded spilled addresses from scratch memory inside their dependent chains; a launch boundary costs ~5 us.
template <class S, int PHASE>
__global__ __launch_bounds__(1024) void k_lit_phase(Dev<S> d, int b0, int nb) {
  __shared__ double red[40];                                 // two reductions at a time (literal_core.h: wg_sum2): 2 x 16 wavefronts
  extern __shared__ double lit_lds[];
  lit::Ctx c; int bi, b;
  if (!lit_ctx(d, b0, nb, red, lit_lds, c, bi, b)) return;
  const lit::Args<S> a = lit_args(d, bi, b);
  const int m = a.row0[a.F];
  if (m <= 0) return;
  if (PHASE == 0) {
    if (c.tim && c.tid == 0) { for (int q = 12; q < 16; ++q) c.tim[q] = 0; }     // accumulating phase timers of the sweep's panels
    lit::tick(c, 0);
    lit::compact_rows(c, a, m);
  } else if (PHASE == 1) lit::compact_sweep(c, a, m);
  else if (PHASE == 2) lit::compact_basis(c, a, m);
  else {
    lit::compact_eliminate(c, a);
    // the blocked Cholesky adds the split-K copies of Lam^ that k_gram leaves (Dev::lam_part apart): none here
    if (d.lam_part) {
      const int n1 = 6 * a.N + 1;
      for (int cpy = 1; cpy < 4; ++cpy) {
        double* Lc = d.Lam + cpy * d.lam_part + (long)b * d.ldR * d.ldR;
        for (int e = threadIdx.x; e < n1 * n1; e += blockDim.x) {
          const int hi = e / n1, lo = e - hi * n1;
          if (lo <= hi) Lc[(long)hi * d.ldR + lo] = 0.0;
        }
      }
    }
  }
}

// the sweep over the dense stack (MSCKF_HIP_LITERAL_ROUTE=1: the definition, tests and A/B runs)
template <class S>
__global__ __launch_bounds__(1024) void k_literal_dense(Dev<S> d, int b0, int nb) {
  __shared__ double red[40];
  extern __shared__ double lit_lds[];
  lit::Ctx c; int bi, b;
  if (!lit_ctx(d, b0, nb, red, lit_lds, c, bi, b)) return;
  const lit::Args<S> a = lit_args(d, bi, b);
  lit::tick(c, 0);
  const int m = lit::prepare(c, a);
  if (m <= 0) return;
  lit::literal_general(c, a, m, a.obs0[a.F]);
  if (d.lam_part) {
    const int n1 = 6 * a.N + 1;
    for (int cpy = 1; cpy < 4; ++cpy) {
      double* Lc = d.Lam + cpy * d.lam_part + (long)b * d.ldR * d.ldR;
      for (int e = threadIdx.x; e < n1 * n1; e += blockDim.x) {
        const int hi = e / n1, lo = e - hi * n1;
        if (lo <= hi) Lc[(long)hi * d.ldR + lo] = 0.0;
      }
    }
  }
}

// The phase kernels ask for LIT_LDS_DOUBLES * 8 = 94 KB of dynamic LDS per workgroup: granted on gfx950 (160 KB per compute unit), not on
// a part with 64 KB.  The result of every attribute call is kept: a device that refuses gets -ENOTSUP from the literal route's
// allocation (msckf_hip.hip: lit_alloc) instead of launches that fail one update later.
static bool g_lit_lds_ok = true;
template <class K> static void lit_lds_attr(K k) {
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LIT_LDS_DOUBLES * sizeof(double))) != hipSuccess) { (void)hipGetLastError(); g_lit_lds_ok = false; }
}
void literal_device_setup() {
  g_lit_lds_ok = true;
  lit_lds_attr(k_lit_phase<float, 0>); lit_lds_attr(k_lit_phase<float, 1>); lit_lds_attr(k_lit_phase<float, 2>); lit_lds_attr(k_lit_phase<float, 3>);
  lit_lds_attr(k_lit_phase<double, 0>); lit_lds_attr(k_lit_phase<double, 1>); lit_lds_attr(k_lit_phase<double, 2>); lit_lds_attr(k_lit_phase<double, 3>);
  lit_lds_attr(k_literal_dense<float>); lit_lds_attr(k_literal_dense<double>);
}
bool literal_lds_available() { return g_lit_lds_ok; }

// part: 0 all launches; 1 k_lit_pre, 2 k_lit_gamma, 3 the four phase kernels (k_lit_phase<., 0..3>) alone (the stage timers of a profiled run bracket each)
template <class S>
void launch_literal(const Dev<S>& d, int b0, int nb, hipStream_t st, int part) {
  if (nb <= 0 || !d.lit.W2) return;
  const bool dense = d.lit.route == 1;
  if (!dense && (part == 0 || part == 1)) {
    const int items = d.f_cap + d.n_cap + 1;
    hipLaunchKernelGGL(k_lit_pre<S>, dim3(xcd_grid(nb, items)), dim3(64), 0, st, d, b0, nb, items, d.lit.serial);
  }
  if (!dense && (part == 0 || part == 2)) {
    const int T = (d.n6cap + 31) / 32, ntile = T * (T + 1) / 2;
    hipLaunchKernelGGL(k_lit_gamma<S>, dim3(xcd_grid(nb, ntile)), dim3(64), 0, st, d, b0, nb, ntile);
  }
  if (part == 0 || part == 3) {
    const size_t lds = LIT_LDS_DOUBLES * sizeof(double);
    if (dense) hipLaunchKernelGGL(k_literal_dense<S>, dim3(nb), dim3(1024), lds, st, d, b0, nb);
    else {
      hipLaunchKernelGGL((k_lit_phase<S, 0>), dim3(nb), dim3(1024), lds, st, d, b0, nb);
      hipLaunchKernelGGL((k_lit_phase<S, 1>), dim3(nb), dim3(1024), lds, st, d, b0, nb);
      hipLaunchKernelGGL((k_lit_phase<S, 2>), dim3(nb), dim3(1024), lds, st, d, b0, nb);
      hipLaunchKernelGGL((k_lit_phase<S, 3>), dim3(nb), dim3(1024), lds, st, d, b0, nb);
    }
  }
}
template void launch_literal<float>(const Dev<float>&, int, int, hipStream_t, int);
template void launch_literal<double>(const Dev<double>&, int, int, hipStream_t, int);

}  // namespace msckf
