// literal_core.h -- the reference's measurement compression under ANISOTROPIC pixel noise (u_var' != v_var', the shipped
// EuRoC configuration asl_msckf.cpp:77-78), built the reference's way for one trajectory:
//
//   msckf.h:423        R_j     = diag(u', v', u', v', ...)
//   msckf.h:954-957    A_j     = trailing 2M-3 columns of JacobiSVD's full U of H_f_j   (= of the Q of a column-pivoted
//                                Householder QR: Eigen's QR preconditioner for a tall matrix), H_o_j = A_j^T H_x_j
//   msckf.h:430-431    r_o_j   = A_j^T r_j,   R_o_j = A_j^T R_j A_j
//   msckf.h:436-441    H_o, r_o, R_o stacked in the order of feature_tracks_to_residualize_
//   msckf.h:1343-1348  HouseholderQR(H_o) in column order; a step whose tail is zero is the identity (Eigen's
//                      makeHouseholder), so the 15 zero IMU columns hand rows 0..14 of H_o through VERBATIM and a
//                      dependent / zero camera column hands its row through (Q2); rows of R that are non-zero are kept
//   msckf.h:1365-1366  r_n = Q_1^T r_o,  R_n = Q_1^T R_o Q_1
//
// and then handed to the filter's update as the information matrix it stands for,
//       Lam^ = [T_H | r_n]^T R_n^-1 [T_H | r_n]            ((6N + 1) x (6N + 1), f64),
// whose Cholesky factor [T^ | r^] with unit noise gives the same K r_n and the same covariance as (T_H, r_n, R_n) in
// msckf.h:1368-1418 (posterior information P^-1 + T_H^T R_n^-1 T_H, information vector T_H^T R_n^-1 r_n): everything after
// the compression is the library's existing update with sigma^2 = 1.
//
// Zero tails: in floating point the tail of a column that depends on the previous ones (the window's gauge directions; a
// stack with fewer rows than columns) is rounding noise, not zero, and the reference then reflects along a direction that
// is rounding noise -- its own result moves by ~1e-4 (gyro bias) between two roundings (tests/test_ref_vs_oracle.py).
// `tol` > 0 treats a tail below tol * |column| as the zero it stands for and drops rows of R whose entries are all below
// tol * max|R| (exactly zero in exact arithmetic): the reference's algorithm in its exact-arithmetic limit, reproducible to
// rounding.  tol = 0 is the reference's rule to the letter (tail^2 <= numeric_limits::min).
//
// Written once for two compilers: hipcc (kernels_literal.hip: one workgroup per trajectory, phases separated by barriers)
// and g++ -DLIT_HOST (tests/cpp/literal_host.cpp: the same phases run serially, checked against the oracle on the CPU).
// All arithmetic in f64 whatever the filter's scalar type.
#ifndef MSCKF_LITERAL_CORE_H
#define MSCKF_LITERAL_CORE_H

#include <math.h>

namespace msckf {
namespace lit {

#ifdef LIT_HOST
#define LIT_FN inline
struct Ctx { int tid = 0, nt = 1, lane = 0, wave = 0, nw = 1; double* red = nullptr; };
LIT_FN void barrier(const Ctx&) {}
template <class F> LIT_FN void par_for(const Ctx&, long n, F f) { for (long i = 0; i < n; ++i) f(i); }
template <class F> LIT_FN double wg_sum(const Ctx&, long lo, long hi, F f) { double s = 0; for (long i = lo; i < hi; ++i) s += f(i); return s; }
template <class F> LIT_FN double wg_max(const Ctx&, long lo, long hi, F f) { double s = 0; for (long i = lo; i < hi; ++i) { const double v = f(i); s = v > s ? v : s; } return s; }
// wave_for: item j is handled by one whole wavefront; inside, lane_for / wave_sum spread a row range over its lanes
template <class F> LIT_FN void wave_for(const Ctx&, long lo, long hi, F f) { for (long j = lo; j < hi; ++j) f(j); }
template <class F> LIT_FN void lane_for(const Ctx&, long lo, long hi, F f) { for (long i = lo; i < hi; ++i) f(i); }
template <class F> LIT_FN double wave_sum_range(const Ctx&, long lo, long hi, F f) { double s = 0; for (long i = lo; i < hi; ++i) s += f(i); return s; }
// NV sums over a row range at once: f(i, v) adds row i's contribution to v[0..NV)
template <int NV, class F> LIT_FN void wave_sum_vec(const Ctx&, long lo, long hi, double (&out)[NV], F f) {
  for (int k = 0; k < NV; ++k) out[k] = 0;
  for (long i = lo; i < hi; ++i) f(i, out);
}
// lower triangle of G^T G (G: mobs x nr, column-major with leading dimension ldg): st(i, j, value) for every j <= i < nr
template <class ST> LIT_FN void syrk_lower(const Ctx&, const double* G, long ldg, int nr, int mobs, ST st) {
  for (int j = 0; j < nr; ++j)
    for (int i = j; i < nr; ++i) {
      double s = 0;
      for (int o = 0; o < mobs; ++o) s += G[o + ldg * i] * G[o + ldg * j];
      st(i, j, s);
    }
}
LIT_FN bool first_lane(const Ctx&) { return true; }
LIT_FN bool first_thread(const Ctx&) { return true; }
LIT_FN void tick(const Ctx&, int) {}
#else
#define LIT_FN __device__ __forceinline__
struct Ctx { int tid, nt, lane, wave, nw; double* red; long long* tim; double* lds; int lds_doubles; };   // red: LDS scratch, nw + 2 doubles; tim: phase stamps (100 MHz) or null; lds: staging area
LIT_FN void barrier(const Ctx&) { __syncthreads(); }
template <class F> LIT_FN void par_for(const Ctx& c, long n, F f) { for (long i = c.tid; i < n; i += c.nt) f(i); }
template <class F> LIT_FN double wg_sum(const Ctx& c, long lo, long hi, F f) {
  double s = 0;
  for (long i = lo + c.tid; i < hi; i += c.nt) s += f(i);
  s = wave_sum(s);
  __syncthreads();                       // red may still be read from the previous reduction
  if (c.lane == 0) c.red[c.wave] = s;
  __syncthreads();
  double t = 0;
  for (int w = 0; w < c.nw; ++w) t += c.red[w];   // same order in every thread: a uniform value
  return t;
}
template <class F> LIT_FN double wg_max(const Ctx& c, long lo, long hi, F f) {
  double s = 0;
  for (long i = lo + c.tid; i < hi; i += c.nt) { const double v = f(i); s = v > s ? v : s; }
  s = wave_max(s);
  __syncthreads();
  if (c.lane == 0) c.red[c.wave] = s;
  __syncthreads();
  double t = 0;
  for (int w = 0; w < c.nw; ++w) t = c.red[w] > t ? c.red[w] : t;
  return t;
}
template <class F> LIT_FN void wave_for(const Ctx& c, long lo, long hi, F f) { for (long j = lo + c.wave; j < hi; j += c.nw) f(j); }
template <class F> LIT_FN void lane_for(const Ctx& c, long lo, long hi, F f) { for (long i = lo + c.lane; i < hi; i += 64) f(i); }
template <class F> LIT_FN double wave_sum_range(const Ctx& c, long lo, long hi, F f) {
  double s = 0;
  for (long i = lo + c.lane; i < hi; i += 64) s += f(i);
  return wave_sum(s);
}
template <int NV, class F> LIT_FN void wave_sum_vec(const Ctx& c, long lo, long hi, double (&out)[NV], F f) {
  double v[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) v[k] = 0;
  for (long i = lo + c.lane; i < hi; i += 64) f(i, v);
#pragma unroll
  for (int k = 0; k < NV; ++k) out[k] = wave_sum(v[k]);
}
// G is read from global memory ONCE: chunks of rows are staged in LDS ([row][column], row stride nr | 1), every thread owns
// one 4 x 4 tile of the lower triangle (a second pass takes the tiles beyond the thread count) and keeps its sixteen sums in
// registers across the chunks.  (One wavefront per tile with lanes along the rows re-read eight columns per tile: 220 MB per
// trajectory at a 30-camera window, and at 128 trajectories per launch the step was bound by that traffic.)
template <class ST> LIT_FN void syrk_lower(const Ctx& c, const double* G, long ldg, int nr, int mobs, ST st) {
  const int ntile = (nr + 3) / 4, ntl = ntile * (ntile + 1) / 2, ldl = nr | 1;
  const int rows = c.lds_doubles / ldl;          // rows of G per chunk
  for (int e0 = 0; e0 < ntl; e0 += c.nt) {
    const int e = e0 + c.tid;
    int ti = 0, tj = 0;
    if (e < ntl) { ti = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5); while (ti * (ti + 1) / 2 > e) --ti; while ((ti + 1) * (ti + 2) / 2 <= e) ++ti; tj = e - ti * (ti + 1) / 2; }
    double acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0;
    for (int o0 = 0; o0 < mobs; o0 += rows) {
      const int nrow = mobs - o0 < rows ? mobs - o0 : rows;
      __syncthreads();
      for (long x = c.tid; x < (long)nrow * nr; x += c.nt) { const int k = (int)(x / nrow), o = (int)(x - (long)k * nrow); c.lds[o * ldl + k] = G[o0 + o + ldg * k]; }
      __syncthreads();
      if (e < ntl) {
        const double* li = c.lds + 4 * ti; const double* lj = c.lds + 4 * tj;
        const int i3 = 4 * ti + 3 < nr ? 3 : nr - 1 - 4 * ti, j3 = 4 * tj + 3 < nr ? 3 : nr - 1 - 4 * tj;   // clamp inside the matrix (edge tiles)
        for (int o = 0; o < nrow; ++o) {
          const double* ri = li + o * ldl; const double* rj = lj + o * ldl;
          const double x0 = ri[0], x1 = ri[i3 < 1 ? i3 : 1], x2 = ri[i3 < 2 ? i3 : 2], x3 = ri[i3];
          const double y0 = rj[0], y1 = rj[j3 < 1 ? j3 : 1], y2 = rj[j3 < 2 ? j3 : 2], y3 = rj[j3];
          acc[0] += x0 * y0; acc[1] += x0 * y1; acc[2] += x0 * y2; acc[3] += x0 * y3;
          acc[4] += x1 * y0; acc[5] += x1 * y1; acc[6] += x1 * y2; acc[7] += x1 * y3;
          acc[8] += x2 * y0; acc[9] += x2 * y1; acc[10] += x2 * y2; acc[11] += x2 * y3;
          acc[12] += x3 * y0; acc[13] += x3 * y1; acc[14] += x3 * y2; acc[15] += x3 * y3;
        }
      }
    }
    if (e < ntl)
      for (int qi = 0; qi < 4; ++qi)
        for (int qj = 0; qj < 4; ++qj) { const int i = 4 * ti + qi, j = 4 * tj + qj; if (i < nr && j <= i) st(i, j, acc[qi * 4 + qj]); }
  }
  __syncthreads();
}
LIT_FN bool first_lane(const Ctx& c) { return c.lane == 0; }
LIT_FN bool first_thread(const Ctx& c) { return c.tid == 0; }
LIT_FN void tick(const Ctx& c, int slot) { if (c.tim && c.tid == 0) c.tim[slot] = (long long)wall_clock64(); }
#endif

// One trajectory's inputs (what k_feature / k_select left behind) and work space.  HT: scalar type of the Jacobian blocks.
template <class HT>
struct Args {
  // ---- inputs
  int F;                    // tracks in the work-list
  int m_cap;                // observations per track the per-track arrays are laid out for
  int N;                    // camera states in the window; n = 6 N state columns
  const int* status;        // [F] bit `inc_bit` set: the track's rows enter the stack (msckf.h:352-441)
  int inc_bit;
  const int* M;             // [F]
  const int* slots;         // slot of observation o of track t at slots[first(t) + o]
  const int* off;           // first(t) = off ? off[t] : t * m_cap
  const HT* Hx;             // [F][m_cap][12]: rows 2o, 2o+1 of H_x_j as 2 x 6 (camera columns of slot o)   msckf.h:915-950
  const HT* rw;             // [F][2 m_cap]: r_j                                                             msckf.h:960-978
  double u_var, v_var, tol;
  // ---- work space (f64), all per trajectory
  int ldx;                  // row capacity of X (>= stacked rows m)
  double* X;                // [ldx x (n + 1)] column-major: [H_o(:, 15:) | r_o], then R / reflectors, then Q'
  double* tau;              // [n]
  double* Vf;               // [F][2 m_cap][3] reflectors of H_f_j (unit lower trapezoidal, implicit ones) -> A_j
  double* Tf;               // [F][9] compact-WY T of those
  int* row0;                // [F + 1] first stacked row of track t (list order), row0[F] = m
  int* obs0;                // [F + 1] first observation index of track t among the stacked tracks
  int* otrk;                // [ldg] track of stacked observation g
  int* kept;                // [6 (n + 16) + 64] kept rows of R (msckf.h:1347) + flag / index scratch behind them
  int r_cap;                // >= n + 15 (row capacity of TH / G / Z)
  double* TH;               // [r_cap x (n + 1)] column-major: kept rows of [R | Q^T r_o]
  int ldg;                  // row capacity of G (>= stacked observations)
  double* G;                // [ldg x r_cap] column-major: u-rows of A Q_1   (R_n = v' I + (u' - v') G^T G)
  int ldz;                  // r_cap + n + 1
  double* Z;                // [ldz x ldz] column-major lower triangle: [[R_n, .], [TH^T, 0]] -> Schur complement -Lam^
  // ---- fast path only (literal_compress_fast): H_o^T H_o as k_gram left it, the slot -> observation map, scratch
  const double* LamIn;      // [H_o | r_o]^T [H_o | r_o], element (hi, lo), lo <= hi <= n, at LamIn[hi * ldL + lo] (+ split-K copies)
  long lam_part; int gram_parts;   // copies of LamIn lam_part doubles apart: block column lo / 64 came in min(lo / 64 + parts - 2, parts) partial sums
  const signed char* inv; int inv_stride;   // observation index of camera slot s in track t at inv[t * inv_stride + s], -1 = not observed
  double* W;                // scratch: (n + 1)^2 + ZCAP (n + 1) + ZCAP 2 m_cap + F 18 m_cap + 3 ldg doubles
  // ---- outputs
  double* Lam; int ldL;     // Lam^(hi, lo), lo <= hi <= n, at Lam[hi * ldL + lo]  (what k_chol_mfma / lam_hat read)
  int* info;                // [8]: stacked rows m, kept rows r, reflected steps, steps skipped by the tolerance, route (1 fast, 2 general), rows handed
                            // through verbatim, fast route's shape check: -100 log10 of the smallest independent / the largest dependent pivot ratio
};

template <class HT> LIT_FN int first_obs(const Args<HT>& a, int t) { return a.off ? a.off[t] : t * a.m_cap; }
// V of a track's H_f factorization with its implicit structure
LIT_FN double vf_at(const double* V, int i, int q) { return i < q ? 0.0 : (i == q ? 1.0 : V[i * 3 + q]); }

// ---------------------------------------------------------------------------------------------------------------------
// per track: column-pivoted Householder QR of H_f_j = -H_x_j(:, 3:6) (2M x 3), in place in V (essential parts below the
// diagonal), compact-WY T with Q = H_0 H_1 H_2 = I - V T V^T.  The pivot rule is the oracle's (and Eigen's, away from
// ties): the remaining column of largest squared norm over rows k.., first one wins.  Serial: one thread per track.
template <class HT>
LIT_FN void track_null_space(const Args<HT>& a, int t) {
  const int M = a.M[t], R2 = 2 * M;
  double* V = a.Vf + (long)t * 2 * a.m_cap * 3;
  double* T = a.Tf + (long)t * 9;
  const HT* hx = a.Hx + (long)t * a.m_cap * 12;
  for (int i = 0; i < R2; ++i)
    for (int c = 0; c < 3; ++c) V[i * 3 + c] = -(double)hx[(i >> 1) * 12 + (i & 1) * 6 + 3 + c];
  double tau[3] = {0, 0, 0};
  const int steps = R2 < 3 ? R2 : 3;
  for (int k = 0; k < steps; ++k) {
    int big = k; double best = -1.0;
    for (int j = k; j < 3; ++j) {
      double s = 0;
      for (int i = k; i < R2; ++i) s += V[i * 3 + j] * V[i * 3 + j];
      if (s > best) { best = s; big = j; }
    }
    if (big != k) for (int i = 0; i < R2; ++i) { const double x = V[i * 3 + k]; V[i * 3 + k] = V[i * 3 + big]; V[i * 3 + big] = x; }
    double tail2 = 0;
    for (int i = k + 1; i < R2; ++i) tail2 += V[i * 3 + k] * V[i * 3 + k];
    const double c0 = V[k * 3 + k];
    if (tail2 <= 2.2250738585072014e-308) { tau[k] = 0; for (int i = k + 1; i < R2; ++i) V[i * 3 + k] = 0; continue; }
    double beta = sqrt(c0 * c0 + tail2);
    if (c0 >= 0.0) beta = -beta;
    const double inv = 1.0 / (c0 - beta);
    for (int i = k + 1; i < R2; ++i) V[i * 3 + k] *= inv;
    tau[k] = (beta - c0) / beta;
    V[k * 3 + k] = beta;
    for (int j = k + 1; j < 3; ++j) {
      double s = V[k * 3 + j];
      for (int i = k + 1; i < R2; ++i) s += V[i * 3 + k] * V[i * 3 + j];
      s *= tau[k];
      V[k * 3 + j] -= s;
      for (int i = k + 1; i < R2; ++i) V[i * 3 + j] -= s * V[i * 3 + k];
    }
  }
  double d01 = 0, d02 = 0, d12 = 0;
  for (int i = 0; i < R2; ++i) {
    const double v0 = vf_at(V, i, 0), v1 = vf_at(V, i, 1), v2 = vf_at(V, i, 2);
    d01 += v0 * v1; d02 += v0 * v2; d12 += v1 * v2;
  }
  for (int i = 0; i < 9; ++i) T[i] = 0;
  T[0] = tau[0]; T[4] = tau[1]; T[8] = tau[2];
  T[1] = -tau[1] * T[0] * d01;                       // T(0,1)
  T[2] = -tau[2] * (T[0] * d02 + T[1] * d12);        // T(0,2)
  T[5] = -tau[2] * T[4] * d12;                       // T(1,2)
}

// Tail of both routes: R_n = v' I + (u' - v') G^T G (msckf.h:1366), then Z = [[R_n, .], [[T_H | r_n]^T, 0]] (lower triangle);
// eliminating the nr pivots of R_n leaves -[T_H | r_n]^T R_n^-1 [T_H | r_n] in the trailing block = -Lam^.
template <class HT>
LIT_FN void information_from_compressed(const Ctx& c, const Args<HT>& a, int n, int nr, int mobs) {
  const int rc = a.r_cap;
  const int nz = nr + n + 1;
  const long ldz = a.ldz;
  double* Z = a.Z;
  const double dlt = a.u_var - a.v_var;
  syrk_lower(c, a.G, a.ldg, nr, mobs, [&](int i, int j, double sgg) { Z[i + ldz * j] = dlt * sgg + (i == j ? a.v_var : 0.0); });
  par_for(c, (long)(n + 1) * nz, [&](long e) {
    const int j = (int)(e / (n + 1)), cc = (int)(e - (long)j * (n + 1));
    Z[(nr + cc) + ldz * j] = j < nr ? a.TH[j + (long)rc * cc] : 0.0;
  });
  barrier(c);
  tick(c, 7);
  for (int k = 0; k < nr; ++k) {
    const double dk = Z[k + ldz * k];
    const double dinv = 1.0 / dk;
    // one wavefront per trailing column j, lanes along its rows i >= j: Z(i, j) -= Z(i, k) Z(j, k) / d
    const double* zk = Z + ldz * k;
    wave_for(c, k + 1, nz, [&](long j) {
      const double ljk = zk[j] * dinv;
      if (ljk == 0.0) return;
      double* zj = Z + ldz * j;
      lane_for(c, j, nz, [&](long i) { zj[i] -= zk[i] * ljk; });
    });
    barrier(c);
  }
  tick(c, 8);
  // ---- Lam^ (lower triangle incl. row n) where the blocked Cholesky reads it
  par_for(c, (long)(n + 1) * (n + 1), [&](long e) {
    const int hi = (int)(e / (n + 1)), lo = (int)(e - (long)hi * (n + 1));
    if (lo > hi) return;
    a.Lam[(long)hi * a.ldL + lo] = -Z[(nr + hi) + ldz * (nr + lo)];
  });
  barrier(c);
}

// Stacked row / observation offsets in list order (msckf.h:404-441) and A_j per track.  Returns the stacked rows m.
template <class HT>
LIT_FN int prepare(const Ctx& c, const Args<HT>& a) {
  const int F = a.F;
  if (first_thread(c)) {
    int r = 0, o = 0;
    for (int t = 0; t < F; ++t) {
      a.row0[t] = r; a.obs0[t] = o;
      if (a.status[t] & a.inc_bit) { r += 2 * a.M[t] - 3; o += a.M[t]; }
    }
    a.row0[F] = r; a.obs0[F] = o;
    a.info[0] = r; a.info[1] = 0; a.info[2] = 0; a.info[3] = 0; a.info[4] = 0; a.info[5] = 0; a.info[6] = 0; a.info[7] = 0;
  }
  barrier(c);
  if (a.row0[F] <= 0) return 0;
  // A_j: null space of H_f_j^T per track (msckf.h:954-955); observation -> track map
  par_for(c, F, [&](long t) {
    if (!(a.status[t] & a.inc_bit)) return;
    track_null_space(a, (int)t);
    for (int o = 0; o < a.M[t]; ++o) a.otrk[a.obs0[t] + o] = (int)t;
  });
  barrier(c);
  return a.row0[F];
}

// The general route: the reference's sequence to the letter on the dense stack (any shape of stack).
template <class HT>
LIT_FN void literal_general(const Ctx& c, const Args<HT>& a, const int m, const int mobs) {
  const int n = 6 * a.N, D = 15 + n, F = a.F;
  const long ldx = a.ldx;
  double* X = a.X;
  par_for(c, (long)m * (n + 1), [&](long e) { const long j = e / m, i = e - j * m; X[i + ldx * j] = 0.0; });
  barrier(c);

  // ---- H_o_j = A_j^T H_x_j and r_o_j = A_j^T r_j written to their place in the stack (msckf.h:957, :430, :436-437):
  // (Q_f^T h)_i = h_i - V(i, :) T^T V^T h, rows 3.. ; a column of H_x_j has two non-zero entries (rows 2o, 2o+1)
  {
    const int cper = 6 * a.m_cap + 1;    // columns of [H_x_j | r_j] (padded)
    par_for(c, (long)F * cper, [&](long e) {
      const int t = (int)(e / cper), cc = (int)(e - (long)t * cper);
      if (!(a.status[t] & a.inc_bit)) return;
      const int M = a.M[t], R2 = 2 * M;
      if (cc > 6 * M) return;
      const double* V = a.Vf + (long)t * 2 * a.m_cap * 3;
      const double* T = a.Tf + (long)t * 9;
      const HT* hx = a.Hx + (long)t * a.m_cap * 12;
      double s[3] = {0, 0, 0};
      int o = -1, col; double h0 = 0, h1 = 0;
      if (cc < 6 * M) {
        o = cc / 6; const int kk = cc - 6 * o;
        h0 = (double)hx[o * 12 + kk]; h1 = (double)hx[o * 12 + 6 + kk];
        for (int q = 0; q < 3; ++q) s[q] = vf_at(V, 2 * o, q) * h0 + vf_at(V, 2 * o + 1, q) * h1;
        col = 6 * a.slots[first_obs(a, t) + o] + kk;
      } else {
        const HT* r = a.rw + (long)t * 2 * a.m_cap;
        for (int i = 0; i < R2; ++i) for (int q = 0; q < 3; ++q) s[q] += vf_at(V, i, q) * (double)r[i];
        col = n;
      }
      double w[3];
      for (int q = 0; q < 3; ++q) { double x = 0; for (int p = 0; p <= q; ++p) x += T[p * 3 + q] * s[p]; w[q] = x; }   // T^T s
      double* xc = X + ldx * col + a.row0[t];
      for (int i = 3; i < R2; ++i) {
        double h;
        if (cc < 6 * M) h = (i == 2 * o) ? h0 : ((i == 2 * o + 1) ? h1 : 0.0);
        else h = (double)(a.rw + (long)t * 2 * a.m_cap)[i];
        xc[i - 3] = h - (vf_at(V, i, 0) * w[0] + vf_at(V, i, 1) * w[1] + vf_at(V, i, 2) * w[2]);
      }
    });
  }
  barrier(c);

  // ---- HouseholderQR(H_o) in column order (msckf.h:1343).  Steps 0..14 meet the zero IMU columns: identity.  Step 15 + k
  // works on camera column k, rows 15 + k.. ; r_o (column n) rides along, so that column n ends as Q^T r_o.
  const int steps_total = m < D ? m : D;
  const int msteps = steps_total - 15 > 0 ? steps_total - 15 : 0;
  const double tol2 = a.tol * a.tol;
  int n_reflect = 0, n_skip_tol = 0;
  for (int k = 0; k < msteps; ++k) {
    const int p = 15 + k;
    double* xk = X + ldx * k;
    const double tail2 = wg_sum(c, p + 1, m, [&](long i) { return xk[i] * xk[i]; });
    double zero2 = 2.2250738585072014e-308;
    if (a.tol > 0) {
      const double head2 = wg_sum(c, 0, p + 1, [&](long i) { return xk[i] * xk[i]; });
      const double z = tol2 * (head2 + tail2);
      zero2 = z > zero2 ? z : zero2;
    }
    const double c0 = xk[p];
    barrier(c);
    if (tail2 <= zero2) {
      if (tail2 > 2.2250738585072014e-308) ++n_skip_tol;
      if (first_thread(c)) a.tau[k] = 0.0;
      par_for(c, m - (p + 1), [&](long i) { xk[p + 1 + i] = 0.0; });
      barrier(c);
      continue;
    }
    ++n_reflect;
    double beta = sqrt(c0 * c0 + tail2);
    if (c0 >= 0.0) beta = -beta;
    const double inv = 1.0 / (c0 - beta), tk = (beta - c0) / beta;
    par_for(c, m - (p + 1), [&](long i) { xk[p + 1 + i] *= inv; });
    if (first_thread(c)) { xk[p] = beta; a.tau[k] = tk; }
    barrier(c);
    wave_for(c, k + 1, n + 1, [&](long j) {
      double* xj = X + ldx * j;
      double s = wave_sum_range(c, p + 1, m, [&](long i) { return xk[i] * xj[i]; });
      s = (s + xj[p]) * tk;
      lane_for(c, p + 1, m, [&](long i) { xj[i] -= s * xk[i]; });
      if (first_lane(c)) xj[p] -= s;
    });
    barrier(c);
  }

  // ---- rows of R that are kept (msckf.h:1345-1348: the upper-triangular view, a row with any non-zero entry)
  double rmax = 0;
  if (a.tol > 0) {
    rmax = wg_max(c, 0, (long)steps_total * n, [&](long e) {
      const long j = e / steps_total, i = e - j * steps_total;
      return (j + 15 >= i) ? fabs(X[i + ldx * j]) : 0.0;
    });
  }
  barrier(c);
  int* flag = a.kept + (n + 16);   // scratch behind the kept list: [steps_total] flags
  par_for(c, steps_total, [&](long i) {
    int any = 0;
    const int c_lo = i >= 15 ? (int)i - 15 : 0;
    for (int j = c_lo; j < n && !any; ++j) { const double v = fabs(X[i + ldx * j]); any = a.tol > 0 ? (v > a.tol * rmax) : (v != 0.0); }
    flag[i] = any;
  });
  barrier(c);
  if (first_thread(c)) {
    int nr = 0;
    for (int i = 0; i < steps_total; ++i) if (flag[i]) a.kept[nr++] = i;
    a.info[1] = nr; a.info[2] = n_reflect; a.info[3] = n_skip_tol; a.info[4] = 2; a.info[5] = steps_total - msteps;
  }
  barrier(c);
  const int nr = a.info[1];
  const int rc = a.r_cap;
  // [T_H | r_n]: kept rows of the upper-triangular view and of Q^T r_o (msckf.h:1351-1365)
  par_for(c, (long)nr * (n + 1), [&](long e) {
    const int j = (int)(e / nr), k = (int)(e - (long)j * nr), row = a.kept[k];
    double v = X[row + ldx * j];
    if (j < n && j + 15 < row) v = 0.0;
    a.TH[k + (long)rc * j] = v;
  });
  barrier(c);

  // ---- Q' = H_15 H_16 ... (first msteps columns), generated in place of the reflectors (backward accumulation); its
  // columns are zero in rows 0..14, and column k is zero above row 15 + k before H_k .. H_15 reach it.
  par_for(c, (long)msteps * 15, [&](long e) { const long j = e / 15, i = e - j * 15; if (i < m) X[i + ldx * j] = 0.0; });
  barrier(c);
  for (int k = msteps - 1; k >= 0; --k) {
    const int p = 15 + k;
    double* xk = X + ldx * k;
    const double tk = a.tau[k];
    if (tk != 0.0) {
      wave_for(c, k + 1, msteps, [&](long j) {
        double* xj = X + ldx * j;
        double s = wave_sum_range(c, p + 1, m, [&](long i) { return xk[i] * xj[i]; });
        s = (s + xj[p]) * tk;
        lane_for(c, p + 1, m, [&](long i) { xj[i] -= s * xk[i]; });
        if (first_lane(c)) xj[p] -= s;
      });
    }
    barrier(c);
    par_for(c, m - 15, [&](long ii) {
      const long i = 15 + ii;
      if (i < p) xk[i] = 0.0; else if (i == p) xk[i] = 1.0 - tk; else xk[i] = -tk * xk[i];
    });
    barrier(c);
  }

  // ---- G = u-rows of A Q_1 (stacked observations x kept rows): R_n = Q_1^T R_o Q_1 = v' I + (u' - v') G^T G
  // (msckf.h:423, 431, 1366).  Column k of Q_1 is e_row for a kept row < 15, else column row - 15 of Q'.
  // A_j q = Q_f [0; q] = q~ - V T (V^T q~), q~ = [0, 0, 0, q]
  par_for(c, (long)F * nr, [&](long e) {
    const int t = (int)(e / nr), k = (int)(e - (long)t * nr);
    if (!(a.status[t] & a.inc_bit)) return;
    const int M = a.M[t], rho = 2 * M - 3, r0 = a.row0[t], row = a.kept[k];
    const double* V = a.Vf + (long)t * 2 * a.m_cap * 3;
    const double* T = a.Tf + (long)t * 9;
    const double* q = row >= 15 ? X + ldx * (row - 15) + r0 : nullptr;
    auto qv = [&](int i) -> double { return q ? q[i] : ((r0 + i == row) ? 1.0 : 0.0); };   // i in [0, rho)
    double s[3] = {0, 0, 0};
    for (int i = 0; i < rho; ++i) { const double x = qv(i); if (x != 0.0) for (int qq = 0; qq < 3; ++qq) s[qq] += vf_at(V, i + 3, qq) * x; }
    double w[3];
    for (int pp = 0; pp < 3; ++pp) { double x = 0; for (int qq = pp; qq < 3; ++qq) x += T[pp * 3 + qq] * s[qq]; w[pp] = x; }   // T s
    double* g = a.G + (long)a.ldg * k + a.obs0[t];
    for (int o = 0; o < M; ++o) {
      const int i = 2 * o;
      const double qt = i >= 3 ? qv(i - 3) : 0.0;
      g[o] = qt - (vf_at(V, i, 0) * w[0] + vf_at(V, i, 1) * w[1] + vf_at(V, i, 2) * w[2]);
    }
  });
  barrier(c);
  information_from_compressed(c, a, n, nr, mobs);
}

// ---------------------------------------------------------------------------------------------------------------------
// The fast route, for the usual shape of a stack: many more rows than columns, every camera of the window (but for leading
// ones nobody saw) observed, rank deficiency only in the window's gauge directions.  Then HouseholderQR(H_o) does nothing
// special after handing through the first z = 15 + 6 c0 rows (c0 = leading cameras without an observation): every later
// step reflects until the columns that depend on the previous ones are reached, and those are the last ones, whose steps
// find nothing but zeros below (rows dropped).  So Q_1 = [e_0 .. e_(z-1) | Q'] with Q' ANY orthonormal basis of the column
// space of H' = H_o(z:, :), and the update depends on Q_1 only through its span (T_H, r_n, R_n transform together under a
// rotation of the kept rows).  With H' = Q' R':
//     R'      = chol(H'^T H'),  H'^T H' = H_o^T H_o - H_o(:z)^T H_o(:z);  H_o^T H_o is what k_gram accumulates in f64
//               (sum_j [H_x | r]^T (I - Q_f Q_f^T) [H_x | r]: independent of the null-space basis)
//     Q'^T r' = R'^-T H'^T r'                         (the augmented column of the same factorization)
//     u-rows of A Q_1 = [ u-rows of A(:, :z) | (u-rows of A_b A_b^T H_x) R'^-1 ],  A_b = the columns of A below row z:
//               A_b A_b^T = I - Q_f(:, :d) Q_f(:, :d)^T per track, d = 3 + (rows of the track among the first z)
// -- no m x n stack, no reflector sweep: O(n^3 + (sum M) n^2) instead of O(m n^2) passes over 8 MB per trajectory.
// Whether the stack has that shape is checked on the factorization itself (a column found dependent -- pivot below
// tol^2 |column|^2, the Householder tail rule in Gram form -- followed by an independent one, or a pivot too close to the
// threshold to call, or m <= z, or z > LIT_ZCAP): if not, the caller runs literal_general.  Returns whether it applied.
constexpr int LIT_ZCAP = 63;

template <class HT>
LIT_FN double lam_in(const Args<HT>& a, int hi, int lo) {   // hi >= lo
  const double* p = a.LamIn + (long)hi * a.ldL + lo;
  double v = p[0];
  if (a.lam_part && a.gram_parts >= 3) {
    const int nc0 = lo / 64 + a.gram_parts - 2, nc = nc0 < a.gram_parts ? nc0 : a.gram_parts;
    for (int cpy = 1; cpy < nc; ++cpy) v += p[cpy * a.lam_part];
  }
  return v;
}

template <class HT>
LIT_FN bool literal_fast(const Ctx& c, const Args<HT>& a, const int m, const int mobs) {
  const int n = 6 * a.N, F = a.F, n1 = n + 1;
  double* C = a.W;                              // (n + 1)^2 column-major, lower triangle: Lam' -> L = R'^T (row n: Q'^T r')
  double* Xt = C + (long)n1 * n1;               // [LIT_ZCAP][n + 1] row-major: rows 0..z-1 of [H_o | r_o]
  double* At = Xt + (long)LIT_ZCAP * n1;        // [LIT_ZCAP][2 m_cap]: column i - row0 of A_j, for the z top rows
  double* Bt = At + (long)LIT_ZCAP * 2 * a.m_cap;   // [F][3][6 m_cap]: rows 0..2 of Q_f^T H_x_j (compact: column 6 o + kk)
  double* Qfu = Bt + (long)F * 18 * a.m_cap;    // [ldg][3]: Q_f(2o, 0..2) per stacked observation
  double* dcol = a.tau;                         // [n] |column|^2 of H_o (incl. the top rows)
  const int ks = n + 16;
  int* flag = a.kept + ks;                      // [ks] kept flags of top rows
  int* skip = a.kept + 2 * ks;                  // [ks] column found dependent
  int* kidx = a.kept + 3 * ks;                  // [ks] ordinal of a column among the independent ones, -1
  int* topt = a.kept + 4 * ks;                  // [LIT_ZCAP + 1] track of top row i
  int* shared = a.kept + 5 * ks;                // z, ok, r', skipped-active count
  tick(c, 1);
  // ---- z and the tracks of the top rows
  if (first_thread(c)) {
    unsigned long long seen = 0;
    for (int t = 0; t < F; ++t)
      if (a.status[t] & a.inc_bit) for (int o = 0; o < a.M[t]; ++o) seen |= 1ull << (a.slots[first_obs(a, t) + o] & 63);
    int c0 = 0;
    while (c0 < a.N && !((seen >> c0) & 1ull)) ++c0;
    const int z = 15 + 6 * c0;
    int ok = (z <= LIT_ZCAP && m > z && a.LamIn != nullptr) ? 1 : 0;
    if (ok) {
      int t = 0;
      for (int i = 0; i < z; ++i) {
        while (!(a.status[t] & a.inc_bit) || a.row0[t] + 2 * a.M[t] - 3 <= i) ++t;
        topt[i] = t;
      }
    }
    shared[0] = z; shared[1] = ok;
  }
  barrier(c);
  const int z = shared[0];
  if (!shared[1]) return false;
  // ---- a_i = A_j e_(i - row0) = Q_f e_(3 + i - row0), the top rows of [H_o | r_o] (msckf.h:957, :430)
  par_for(c, (long)z * n1, [&](long e) { Xt[e] = 0.0; });
  barrier(c);
  par_for(c, z, [&](long i) {
    const int t = topt[i], M = a.M[t], R2 = 2 * M, q = 3 + (int)i - a.row0[t];
    const double* V = a.Vf + (long)t * 2 * a.m_cap * 3;
    const double* T = a.Tf + (long)t * 9;
    double* ai = At + i * 2 * a.m_cap;
    double sv[3], w[3];
    for (int p = 0; p < 3; ++p) sv[p] = vf_at(V, q, p);
    for (int p = 0; p < 3; ++p) { double x = 0; for (int qq = p; qq < 3; ++qq) x += T[p * 3 + qq] * sv[qq]; w[p] = x; }   // T V(q, :)^T
    for (int r = 0; r < R2; ++r) ai[r] = (r == q ? 1.0 : 0.0) - (vf_at(V, r, 0) * w[0] + vf_at(V, r, 1) * w[1] + vf_at(V, r, 2) * w[2]);
    const HT* hx = a.Hx + (long)t * a.m_cap * 12;
    const HT* rr = a.rw + (long)t * 2 * a.m_cap;
    double* xr = Xt + i * n1;
    double sr = 0;
    for (int o = 0; o < M; ++o) {
      const int col = 6 * a.slots[first_obs(a, t) + o];
      for (int kk = 0; kk < 6; ++kk) xr[col + kk] = ai[2 * o] * (double)hx[o * 12 + kk] + ai[2 * o + 1] * (double)hx[o * 12 + 6 + kk];
      sr += ai[2 * o] * (double)rr[2 * o] + ai[2 * o + 1] * (double)rr[2 * o + 1];
    }
    xr[n] = sr;
  });
  barrier(c);
  tick(c, 2);
  // ---- Lam' = H_o^T H_o - (top rows)^T (top rows), lower triangle incl. row n
  par_for(c, (long)n1 * n1, [&](long e) {
    const int lo = (int)(e / n1), hi = (int)(e - (long)lo * n1);
    if (hi < lo) return;
    const double full = (hi == n && lo == n) ? 0.0 : lam_in(a, hi, lo);
    double s = 0;
    for (int i = 0; i < z; ++i) s += Xt[i * n1 + hi] * Xt[i * n1 + lo];
    C[hi + (long)n1 * lo] = full - s;
    if (hi == lo && hi < n) dcol[hi] = full;
  });
  barrier(c);
  tick(c, 3);
  // ---- Cholesky with the zero-tail rule in Gram form: the pivot of column k IS |tail|^2 of Householder step 15 + k
  // The pivot of column k is |tail|^2 of Householder step 15 + k, so the Householder rule itself decides: dependent iff
  // pivot <= tol^2 |column|^2.  What the f64 Gram matrix resolves (measured over the benchmark's sequences at a 30-camera
  // window): exactly dependent columns come out at 3e-11 |column|^2 in good geometry and up to 3e-8 in the weakest (small
  // earlier pivots amplify the rounding), columns that depend on the others only up to the float rounding of H_x at up to
  // 2e-7, independent ones from 2e-6.  Hence tol^2 no finer than 1e-7 here, and a pivot within 10 % of the threshold is left
  // to the general route's tail test.
  // (Float Jacobians: the two populations overlap in weak geometry -- whichever way such a column is called is inside the
  // float filter's own rounding, so the threshold decides and nothing is handed to the general route.)
  const double t2a = a.tol * a.tol, lo2 = t2a > 1e-7 ? t2a : 1e-7, band = sizeof(HT) == 4 ? 0.0 : 0.1 * lo2;
  int ok = 1, seen_dep = 0;
  double min_ind = 1.0, max_dep = 1e-300;
  for (int k = 0; k < n; ++k) {
    double* ck = C + (long)n1 * k;
    const double piv = ck[k], dk = dcol[k];
    if (dk > 0.0) { const double ratio = piv / dk; if (ratio > lo2) min_ind = ratio < min_ind ? ratio : min_ind; else if (ratio > max_dep) max_dep = ratio; }
    bool indep = dk > 0.0 && piv > lo2 * dk;
    if (dk > 0.0 && fabs(piv / dk - lo2) < band) ok = 0;                         // too close to call
    // float Jacobians: once the dependent columns have begun, a pivot that clears the threshold by less than a factor 30 is
    // the same rounding of H_x that the threshold is there to catch (a column that is really independent again -- a stack
    // for the general route -- clears it by orders of magnitude)
    if (sizeof(HT) == 4 && indep && seen_dep && piv < 30 * lo2 * dk) indep = false;
    if (dk > 0.0 && !indep) seen_dep = 1;
    barrier(c);
    if (!indep) {
      if (first_thread(c)) skip[k] = 1;
      par_for(c, n1 - k, [&](long i) { ck[k + i] = 0.0; });
      barrier(c);
      continue;
    }
    const double dinv = 1.0 / sqrt(piv);
    if (first_thread(c)) skip[k] = 0;
    par_for(c, n1 - k, [&](long i) { ck[k + i] *= dinv; });       // column k of L (the diagonal becomes sqrt(piv))
    barrier(c);
    wave_for(c, k + 1, n, [&](long j) {
      const double ljk = ck[j];
      if (ljk == 0.0) return;
      double* cj = C + (long)n1 * j;
      lane_for(c, j, n1, [&](long i) { cj[i] -= ck[i] * ljk; });
    });
    barrier(c);
  }
  tick(c, 4);
  // ---- shape of the stack: the dependent columns must be the last of the observed ones
  if (first_thread(c)) {
    int rp = 0, nsk = 0, seen_skip = 0;
    for (int k = 0; k < n; ++k) {
      if (dcol[k] > 0.0) {
        if (skip[k]) { seen_skip = 1; ++nsk; }
        else if (seen_skip) ok = 0;
      }
      kidx[k] = skip[k] ? -1 : rp;
      if (!skip[k]) ++rp;
    }
    shared[1] = ok; shared[2] = rp; shared[3] = nsk;
    a.info[6] = (int)(-100.0 * log10(min_ind)); a.info[7] = (int)(-100.0 * log10(max_dep));
  }
  barrier(c);
  if (!shared[1]) return false;
  const int rp = shared[2];
  // ---- rows that are kept (msckf.h:1345-1348) and [T_H | r_n]
  double rmax = 0;
  if (a.tol > 0) {
    const double r1 = wg_max(c, 0, (long)z * n, [&](long e) { const long i = e / n, j = e - i * n; return (j + 15 >= i) ? fabs(Xt[i * n1 + j]) : 0.0; });
    const double r2 = wg_max(c, 0, (long)n * n, [&](long e) { const long k = e / n, j = e - k * n; return (j >= k && !skip[k]) ? fabs(C[j + (long)n1 * k]) : 0.0; });
    rmax = r1 > r2 ? r1 : r2;
  }
  barrier(c);
  par_for(c, z, [&](long i) {
    int any = 0;
    const int c_lo = i >= 15 ? (int)i - 15 : 0;
    for (int j = c_lo; j < n && !any; ++j) { const double v = fabs(Xt[i * n1 + j]); any = a.tol > 0 ? (v > a.tol * rmax) : (v != 0.0); }
    flag[i] = any;
  });
  barrier(c);
  if (first_thread(c)) {
    int nr = 0;
    for (int i = 0; i < z; ++i) if (flag[i]) a.kept[nr++] = i;
    shared[4] = nr;                                   // kept top rows
    for (int k = 0; k < n; ++k) if (!skip[k]) a.kept[nr++] = z + k;   // (row labels only: z + column)
    a.info[1] = nr; a.info[2] = rp; a.info[3] = shared[3]; a.info[4] = 1; a.info[5] = z;
  }
  barrier(c);
  const int nr = a.info[1], ztk = shared[4];
  const int rc = a.r_cap;
  par_for(c, (long)nr * n1, [&](long e) {
    const int j = (int)(e / nr), k = (int)(e - (long)j * nr);
    double v;
    if (k < ztk) { const int row = a.kept[k]; v = (j < n && j + 15 < row) ? 0.0 : Xt[row * n1 + j]; }
    else { const int col = a.kept[k] - z; v = j >= col ? C[j + (long)n1 * col] : 0.0; }
    a.TH[k + (long)rc * j] = v;
  });
  // ---- G, top columns: u-rows of A e_i (non-zero inside the row's own track)
  par_for(c, (long)ztk * mobs, [&](long e) {
    const int k = (int)(e / mobs), g = (int)(e - (long)k * mobs), i = a.kept[k], t = topt[i];
    const int o = g - a.obs0[t];
    a.G[(long)a.ldg * k + g] = (a.otrk[g] == t) ? At[i * 2 * a.m_cap + 2 * o] : 0.0;
  });
  tick(c, 5);
  // ---- rows 0..2 of Q_f^T H_x_j per track and Q_f(2o, 0..2) per observation: with them a u-row of the projected Jacobian
  // is three products per entry for every track that has no row among the first z (d = 3)
  par_for(c, (long)F * 6 * a.m_cap, [&](long e) {
    const int t = (int)(e / (6 * a.m_cap)), cc = (int)(e - (long)t * 6 * a.m_cap);
    if (!(a.status[t] & a.inc_bit) || cc >= 6 * a.M[t]) return;
    const double* V = a.Vf + (long)t * 2 * a.m_cap * 3;
    const double* T = a.Tf + (long)t * 9;
    const HT* hx = a.Hx + (long)t * a.m_cap * 12;
    const int op = cc / 6, kk = cc - 6 * op;
    const double h0 = (double)hx[op * 12 + kk], h1 = (double)hx[op * 12 + 6 + kk];
    double sv[3], w[3];
    for (int p2 = 0; p2 < 3; ++p2) sv[p2] = vf_at(V, 2 * op, p2) * h0 + vf_at(V, 2 * op + 1, p2) * h1;
    for (int q = 0; q < 3; ++q) { double x = 0; for (int p2 = 0; p2 <= q; ++p2) x += T[p2 * 3 + q] * sv[p2]; w[q] = x; }
    for (int q = 0; q < 3; ++q) {
      const double hq = q == 2 * op ? h0 : (q == 2 * op + 1 ? h1 : 0.0);
      Bt[((long)t * 3 + q) * 6 * a.m_cap + cc] = hq - (vf_at(V, q, 0) * w[0] + vf_at(V, q, 1) * w[1] + vf_at(V, q, 2) * w[2]);
    }
  });
  par_for(c, mobs, [&](long g) {
    const int t = a.otrk[g], o = (int)g - a.obs0[t];
    const double* V = a.Vf + (long)t * 2 * a.m_cap * 3;
    const double* T = a.Tf + (long)t * 9;
    double tv[3];
    for (int p2 = 0; p2 < 3; ++p2) tv[p2] = vf_at(V, 2 * o, 0) * T[0 * 3 + p2] + vf_at(V, 2 * o, 1) * T[1 * 3 + p2] + vf_at(V, 2 * o, 2) * T[2 * 3 + p2];
    for (int q = 0; q < 3; ++q) Qfu[g * 3 + q] = (q == 2 * o ? 1.0 : 0.0) - (tv[0] * vf_at(V, q, 0) + tv[1] * vf_at(V, q, 1) + tv[2] * vf_at(V, q, 2));
  });
  barrier(c);
  // ---- G, the other columns: x = (u-row of A_b A_b^T H_x) R'^-1 per stacked observation, 16 columns at a time
  par_for(c, mobs, [&](long g) {
    const int t = a.otrk[g], o = (int)g - a.obs0[t], M = a.M[t], R2 = 2 * M, rho = R2 - 3;
    int kt = z - a.row0[t]; kt = kt < 0 ? 0 : (kt > rho ? rho : kt);
    const int d = 3 + kt;
    const double* V = a.Vf + (long)t * 2 * a.m_cap * 3;
    const double* T = a.Tf + (long)t * 9;
    const HT* hx = a.Hx + (long)t * a.m_cap * 12;
    const signed char* inv = a.inv + (long)t * a.inv_stride;
    double* gout = a.G + (long)a.ldg * ztk + g;
    if (d >= R2) { for (int k = 0; k < rp; ++k) gout[(long)a.ldg * k] = 0.0; return; }
    double tv[3], Sd[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, ev[3];
    for (int p = 0; p < 3; ++p) tv[p] = vf_at(V, 2 * o, 0) * T[0 * 3 + p] + vf_at(V, 2 * o, 1) * T[1 * 3 + p] + vf_at(V, 2 * o, 2) * T[2 * 3 + p];
    for (int q = 0; q < d; ++q) for (int p1 = 0; p1 < 3; ++p1) for (int p = 0; p < 3; ++p) Sd[p1][p] += vf_at(V, q, p1) * vf_at(V, q, p);
    for (int p = 0; p < 3; ++p) ev[p] = (2 * o < d ? vf_at(V, 2 * o, p) : 0.0) - (tv[0] * Sd[0][p] + tv[1] * Sd[1][p] + tv[2] * Sd[2][p]);
    auto qf = [&](int q) -> double { return (q == 2 * o ? 1.0 : 0.0) - (tv[0] * vf_at(V, q, 0) + tv[1] * vf_at(V, q, 1) + tv[2] * vf_at(V, q, 2)); };   // Q_f(2o, q)
    auto hhat = [&](int col) -> double {
      const int slot = col / 6, kk = col - 6 * slot, op = inv[slot];
      if (op < 0) return 0.0;
      const double h0 = (double)hx[op * 12 + kk], h1 = (double)hx[op * 12 + 6 + kk];
      double sv[3], val = op == o ? h0 : 0.0;
      for (int p = 0; p < 3; ++p) sv[p] = vf_at(V, 2 * op, p) * h0 + vf_at(V, 2 * op + 1, p) * h1;
      for (int q = 0; q < 3; ++q) { double w = 0; for (int p = 0; p <= q; ++p) w += T[p * 3 + q] * sv[p]; val += ev[q] * w; }
      if (2 * op < d) val -= qf(2 * op) * h0;
      if (2 * op + 1 < d) val -= qf(2 * op + 1) * h1;
      return val;
    };
    int smin = 1 << 30;
    for (int o2 = 0; o2 < M; ++o2) { const int sl = a.slots[first_obs(a, t) + o2]; smin = sl < smin ? sl : smin; }
    const int cb0 = (6 * smin / 16) * 16;
    for (int k = 0; k < n; ++k) if (k < cb0 && kidx[k] >= 0) gout[(long)a.ldg * kidx[k]] = 0.0;
    for (int cb = cb0; cb < n; cb += 16) {
      double acc[16];
      if (d == 3) {
        const double q0 = Qfu[g * 3], q1 = Qfu[g * 3 + 1], q2 = Qfu[g * 3 + 2];
        const double* b0 = Bt + (long)t * 3 * 6 * a.m_cap; const double* b1 = b0 + 6 * a.m_cap; const double* b2 = b1 + 6 * a.m_cap;
        for (int j = 0; j < 16; ++j) {
          const int col = cb + j;
          double val = 0.0;
          if (col < n) {
            const int slot = col / 6, kk = col - 6 * slot, op = inv[slot];
            if (op >= 0) { const int cc = 6 * op + kk; val = (op == o ? (double)hx[op * 12 + kk] : 0.0) - (q0 * b0[cc] + q1 * b1[cc] + q2 * b2[cc]); }
          }
          acc[j] = val;
        }
      } else {
        for (int j = 0; j < 16; ++j) acc[j] = cb + j < n ? hhat(cb + j) : 0.0;
      }
      // four earlier columns per pass: their x and their rows of R' are requested together (one column per pass left every
      // load waiting for the previous one's use: 8 of the fast route's 15 ms); a dependent column contributes x = 0
      const int nj = n - cb < 16 ? n - cb : 16;
      int cp = cb0;
      for (; cp + 4 <= cb; cp += 4) {
        const int k0 = kidx[cp], k1 = kidx[cp + 1], k2 = kidx[cp + 2], k3 = kidx[cp + 3];
        const double x0 = k0 >= 0 ? gout[(long)a.ldg * k0] : 0.0, x1 = k1 >= 0 ? gout[(long)a.ldg * k1] : 0.0;
        const double x2 = k2 >= 0 ? gout[(long)a.ldg * k2] : 0.0, x3 = k3 >= 0 ? gout[(long)a.ldg * k3] : 0.0;
        const double* r0 = C + (long)n1 * cp + cb; const double* r1 = r0 + n1; const double* r2 = r1 + n1; const double* r3 = r2 + n1;   // R'(cp + i, cb + j)
        if (nj == 16) {
#pragma unroll
          for (int j = 0; j < 16; ++j) acc[j] -= x0 * r0[j] + x1 * r1[j] + x2 * r2[j] + x3 * r3[j];
        } else {
          for (int j = 0; j < nj; ++j) acc[j] -= x0 * r0[j] + x1 * r1[j] + x2 * r2[j] + x3 * r3[j];
        }
      }
      for (; cp < cb; ++cp) {
        if (kidx[cp] < 0) continue;
        const double xc = gout[(long)a.ldg * kidx[cp]];
        const double* rrow = C + (long)n1 * cp + cb;          // R'(cp, cb + j)
        for (int j = 0; j < nj; ++j) acc[j] -= xc * rrow[j];
      }
      for (int j = 0; j < 16; ++j) {
        const int col = cb + j;
        if (col >= n || kidx[col] < 0) continue;
        const double* rrow = C + (long)n1 * col + cb;
        const double x = acc[j] / rrow[j];
        gout[(long)a.ldg * kidx[col]] = x;
        for (int j2 = j + 1; j2 < 16; ++j2) if (cb + j2 < n) acc[j2] -= x * rrow[j2];
      }
    }
  });
  barrier(c);
  tick(c, 6);
  information_from_compressed(c, a, n, nr, mobs);
  tick(c, 9);
  return true;
}

// route: 0 = fast when the stack has the shape for it, else general; 1 = general; 2 = fast only (tests: Lam^ is left
// untouched and info[4] = 0 when the shape check fails)
template <class HT>
LIT_FN void literal_compress(const Ctx& c, const Args<HT>& a, const int route = 0) {
  tick(c, 0);
  const int m = prepare(c, a);
  if (m <= 0) return;
  const int mobs = a.obs0[a.F];
  if (route != 1 && literal_fast(c, a, m, mobs)) return;
  if (route == 2) return;
  literal_general(c, a, m, mobs);
}

}  // namespace lit
}  // namespace msckf
#endif
