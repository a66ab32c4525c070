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
// Two routes compute the same thing.  literal_general builds the dense stack and sweeps the reflectors over it, exactly as
// written above: O(m n) memory passes per step -- the definition, kept for tests and A/B runs.  literal_compact (default)
// runs the SAME sequence of Householder steps on a compressed representation (the rows that can become pivot rows
// explicitly, the rest through their Gram matrix) in O(n^2) per step; see there.
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
struct Ctx { int tid = 0, nt = 1, lane = 0, wave = 0, nw = 1; double* red = nullptr; double* lds = nullptr; int lds_doubles = 0; };   // lds: the device's staging area, a heap block here
LIT_FN void barrier(const Ctx&) {}
template <class F> LIT_FN void par_for(const Ctx&, long n, F f) { for (long i = 0; i < n; ++i) f(i); }
template <class F> LIT_FN double wg_sum(const Ctx&, long lo, long hi, F f) { double s = 0; for (long i = lo; i < hi; ++i) s += f(i); return s; }
template <class F> LIT_FN double wg_max(const Ctx&, long lo, long hi, F f) { double s = 0; for (long i = lo; i < hi; ++i) { const double v = f(i); s = v > s ? v : s; } return s; }
// two sums over one range in one reduction: f(i, a, b) adds item i's contributions to a and b
template <class F> LIT_FN void wg_sum2(const Ctx&, long lo, long hi, double& sa, double& sb, F f) { sa = 0; sb = 0; for (long i = lo; i < hi; ++i) f(i, sa, sb); }
// wave_for: item j is handled by one whole wavefront; inside, lane_for / wave_sum spread a row range over its lanes
template <class F> LIT_FN void wave_for(const Ctx&, long lo, long hi, F f) { for (long j = lo; j < hi; ++j) f(j); }
template <class F> LIT_FN void lane_for(const Ctx&, long lo, long hi, F f) { for (long i = lo; i < hi; ++i) f(i); }
template <class F> LIT_FN double wave_sum_range(const Ctx&, long lo, long hi, F f) { double s = 0; for (long i = lo; i < hi; ++i) s += f(i); return s; }
// row_for: item j is handled by one ROW of a wavefront (16 lanes; four items per wavefront at a time); inside, rowlane_for /
// row_sum_range spread a range over the row's lanes
template <class F> LIT_FN void row_for(const Ctx&, long lo, long hi, F f) { for (long j = lo; j < hi; ++j) f(j); }
template <class F> LIT_FN void rowlane_for(const Ctx&, long lo, long hi, F f) { for (long i = lo; i < hi; ++i) f(i); }
template <class F> LIT_FN double row_sum_range(const Ctx&, long lo, long hi, F f) { double s = 0; for (long i = lo; i < hi; ++i) s += f(i); return s; }
LIT_FN bool first_rowlane(const Ctx&) { return true; }
// rowlane_update: x(i) <- upd(i) over a range, where upd reads what it needs of item i (the device version reads four items
// before it writes the first: a store followed by the next item's loads is a full memory round trip when the compiler cannot
// rule out that they alias)
template <class FU, class FS> LIT_FN void rowlane_update(const Ctx&, long lo, long hi, FU upd, FS st) { for (long i = lo; i < hi; ++i) st(i, upd(i)); }
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
// A^T B for A (kd x ma) and B (kd x nb), both column-major with the contraction index along the columns: st(i, j, value)
template <class ST> LIT_FN void atb(const Ctx&, const double* A, long lda, int ma, const double* B, long ldb, int nb, int kd, ST st) {
  for (int j = 0; j < nb; ++j)
    for (int i = 0; i < ma; ++i) {
      double s = 0;
      for (int l = 0; l < kd; ++l) s += A[l + lda * i] * B[l + ldb * j];
      st(i, j, s);
    }
}
LIT_FN bool first_lane(const Ctx&) { return true; }
LIT_FN bool first_thread(const Ctx&) { return true; }
template <class ST> LIT_FN void atb_lower(const Ctx& c, const double* A, long lda, int ma, const double* B, long ldb, int kd, ST st) {
  atb(c, A, lda, ma, B, ldb, ma, kd, [&](int i, int j, double v) { if (i >= j) st(i, j, v); });
}
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
template <class F> LIT_FN void wg_sum2(const Ctx& c, long lo, long hi, double& sa, double& sb, F f) {
  double a = 0, b = 0;
  for (long i = lo + c.tid; i < hi; i += c.nt) f(i, a, b);
  a = wave_sum(a); b = wave_sum(b);
  __syncthreads();                       // red may still be read from the previous reduction
  if (c.lane == 0) { c.red[c.wave] = a; c.red[c.nw + c.wave] = b; }
  __syncthreads();
  sa = 0; sb = 0;
  for (int w = 0; w < c.nw; ++w) { sa += c.red[w]; sb += c.red[c.nw + w]; }
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
// A column per ROW of a wavefront (16 lanes) instead of per wavefront: four columns of a wavefront have their loads in flight
// together, and a workgroup walks 64 columns at a time -- the per-column steps (dot product -> update) are chains of dependent
// global round trips, and a wavefront that takes its columns one after the other pays every one of them (sweep: 5.1 -> ms,
// columns of Q: 3.3 -> ms at a 30-camera window)
template <class F> LIT_FN void row_for(const Ctx& c, long lo, long hi, F f) { for (long j = lo + 4 * c.wave + (c.lane >> 4); j < hi; j += 4 * c.nw) f(j); }
template <class F> LIT_FN void rowlane_for(const Ctx& c, long lo, long hi, F f) { for (long i = lo + (c.lane & 15); i < hi; i += 16) f(i); }
template <class F> LIT_FN double row_sum_range(const Ctx& c, long lo, long hi, F f) {
  double s = 0;
  for (long i = lo + (c.lane & 15); i < hi; i += 16) s += f(i);
  // the four DPP steps inside a row of 16 lanes: every lane of the row ends up with the row's sum (dev_common.h: wave_sum's first half)
  s += dpp_x<DPP_QUAD_X1>(s); s += dpp_x<DPP_QUAD_X2>(s); s += dpp_x<DPP_HALF_MIRROR>(s); s += dpp_x<DPP_ROW_MIRROR>(s);
  return s;
}
LIT_FN bool first_rowlane(const Ctx& c) { return (c.lane & 15) == 0; }
template <class FU, class FS> LIT_FN void rowlane_update(const Ctx& c, long lo, long hi, FU upd, FS st) {
  for (long i0 = lo + (c.lane & 15); i0 < hi; i0 += 64) {
    double v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const long i = i0 + 16 * u; v[u] = i < hi ? upd(i) : 0.0; }
#pragma unroll
    for (int u = 0; u < 4; ++u) { const long i = i0 + 16 * u; if (i < hi) st(i, v[u]); }
  }
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
template <int TS, class ST> LIT_FN void syrk_lower_ts(const Ctx& c, const double* G, long ldg, int nr, int mobs, ST st) {
  const int ntile = (nr + TS - 1) / TS, ntl = ntile * (ntile + 1) / 2, ldl = nr | 1;
  const int rows = c.lds_doubles / ldl;          // rows of G per chunk
  for (int e0 = 0; e0 < ntl; e0 += c.nt) {
    const int e = e0 + c.tid;
    int ti = 0, tj = 0;
    if (e < ntl) { ti = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5); while (ti * (ti + 1) / 2 > e) --ti; while ((ti + 1) * (ti + 2) / 2 <= e) ++ti; tj = e - ti * (ti + 1) / 2; }
    double acc[TS * TS];
#pragma unroll
    for (int k = 0; k < TS * TS; ++k) acc[k] = 0;
    for (int o0 = 0; o0 < mobs; o0 += rows) {
      const int nrow = mobs - o0 < rows ? mobs - o0 : rows;
      __syncthreads();
      for (long x = c.tid; x < (long)nrow * nr; x += c.nt) { const int k = (int)(x / nrow), o = (int)(x - (long)k * nrow); c.lds[o * ldl + k] = G[o0 + o + ldg * k]; }
      __syncthreads();
      if (e < ntl) {
        const double* li = c.lds + TS * ti; const double* lj = c.lds + TS * tj;
        int io[TS], jo[TS];                       // offsets clamped inside the matrix (edge tiles)
#pragma unroll
        for (int q = 0; q < TS; ++q) { io[q] = TS * ti + q < nr ? q : nr - 1 - TS * ti; jo[q] = TS * tj + q < nr ? q : nr - 1 - TS * tj; }
        for (int o = 0; o < nrow; ++o) {
          const double* ri = li + o * ldl; const double* rj = lj + o * ldl;
          double xv[TS], yv[TS];
#pragma unroll
          for (int q = 0; q < TS; ++q) { xv[q] = ri[io[q]]; yv[q] = rj[jo[q]]; }
#pragma unroll
          for (int qi = 0; qi < TS; ++qi)
#pragma unroll
            for (int qj = 0; qj < TS; ++qj) acc[qi * TS + qj] += xv[qi] * yv[qj];
        }
      }
    }
    if (e < ntl)
#pragma unroll
      for (int qi = 0; qi < TS; ++qi)
#pragma unroll
        for (int qj = 0; qj < TS; ++qj) { const int i = TS * ti + qi, j = TS * tj + qj; if (i < nr && j <= i) st(i, j, acc[qi * TS + qj]); }
  }
  __syncthreads();
}
// The same product on the f64 matrix cores (v_mfma_f64_16x16x4_f64) for up to 192 columns and sixteen wavefronts: the 16 x 16
// blocks of the lower triangle (78 at 180 columns) are dealt round-robin to the wavefronts, five accumulators each; a chunk of
// rows is staged [row][column] (zero beyond the matrix, row stride 208: rows k and k + 1 half the banks apart) and every MFMA
// takes its two operands straight from there.  The tile version is bound by its LDS reads (ten 8-byte reads per 25 FMAs and
// thread: 2.3 ms for the 3 200 x 180 H_u of a 30-camera window); an MFMA reads two operands per 1 024 FMAs.
typedef double lit_v4d __attribute__((ext_vector_type(4)));
template <class ST> LIT_FN void syrk_lower_mfma(const Ctx& c, const double* G, long ldg, int nr, int mobs, ST st) {
  constexpr int LDL = 208, MAXB = 5;
  const int nbk = (nr + 15) / 16, nblk = nbk * (nbk + 1) / 2;
  const int rows = (c.lds_doubles / LDL) & ~3;                 // rows of G per chunk (a multiple of the MFMA's k = 4)
  int bi[MAXB], bj[MAXB];
  lit_v4d acc[MAXB];
#pragma unroll
  for (int q = 0; q < MAXB; ++q) {
    const int e = c.wave + c.nw * q;
    int ti = 0, tj = 0;
    if (e < nblk) { ti = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5); while (ti * (ti + 1) / 2 > e) --ti; while ((ti + 1) * (ti + 2) / 2 <= e) ++ti; tj = e - ti * (ti + 1) / 2; }
    bi[q] = ti; bj[q] = tj;
    acc[q] = lit_v4d{0.0, 0.0, 0.0, 0.0};
  }
  const int lr = c.lane & 15, lk = c.lane >> 4;
  for (int o0 = 0; o0 < mobs; o0 += rows) {
    const int nrow = mobs - o0 < rows ? mobs - o0 : rows, nrow4 = (nrow + 3) & ~3;
    __syncthreads();
    for (long x = c.tid; x < (long)nrow4 * (16 * nbk); x += c.nt) {
      const int k = (int)(x / nrow4), o = (int)(x - (long)k * nrow4);
      c.lds[o * LDL + k] = (o < nrow && k < nr) ? G[o0 + o + ldg * k] : 0.0;
    }
    __syncthreads();
    for (int k0 = 0; k0 < nrow4; k0 += 4) {
      const double* row = c.lds + (k0 + lk) * LDL + lr;
#pragma unroll
      for (int q = 0; q < MAXB; ++q) {
        if (c.wave + c.nw * q >= nblk) break;
        // C(i, j) += sum_k G(k, 16 bi + i) G(k, 16 bj + j): the A operand's lane index runs along i, the B operand's along j
        acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(row[16 * bi[q]], row[16 * bj[q]], acc[q], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int q = 0; q < MAXB; ++q) {
    if (c.wave + c.nw * q >= nblk) break;
#pragma unroll
    for (int r = 0; r < 4; ++r) {                              // C/D layout: row = (lane >> 4) + 4 r, column = lane & 15
      const int i = 16 * bi[q] + lk + 4 * r, j = 16 * bj[q] + lr;
      if (i < nr && j <= i) st(i, j, acc[q][r]);
    }
  }
  __syncthreads();
}
// 4 x 4 tiles per thread; 5 x 5 when the 4 x 4 tiles outnumber the threads (a second pass re-stages the whole of G: the Gram
// matrix of 180 columns is 1 035 tiles for 1 024 threads); the matrix cores for tall operands
template <class ST> LIT_FN void syrk_lower(const Ctx& c, const double* G, long ldg, int nr, int mobs, ST st) {
  const int nbk = (nr + 15) / 16;
  if (mobs >= 256 && c.nw == 16 && nbk * (nbk + 1) / 2 <= 5 * 16 && 8 * 208 <= c.lds_doubles) { syrk_lower_mfma(c, G, ldg, nr, mobs, st); return; }
  const int nt4 = (nr + 3) / 4;
  if (nt4 * (nt4 + 1) / 2 <= c.nt) syrk_lower_ts<4>(c, G, ldg, nr, mobs, st);
  else syrk_lower_ts<5>(c, G, ldg, nr, mobs, st);
}
// A^T B with chunks of the contraction index staged in LDS ([row][column of A | column of B]), one 4 x 4 tile of the result
// per thread and pass (the operands are read from global memory once per pass)
template <int TS, bool LOWER, class ST> LIT_FN void atb_ts(const Ctx& c, const double* A, long lda, int ma, const double* B, long ldb, int nb, int kd, ST st) {
  const int ta = (ma + TS - 1) / TS, tb = (nb + TS - 1) / TS, ntl = LOWER ? ta * (ta + 1) / 2 : ta * tb, ldl = (ma + nb) | 1;
  const int rows = c.lds_doubles / ldl;
  for (int e0 = 0; e0 < ntl; e0 += c.nt) {
    const int e = e0 + c.tid;
    int ti = 0, tj = 0;
    if (e < ntl) {
      if (LOWER) { ti = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5); while (ti * (ti + 1) / 2 > e) --ti; while ((ti + 1) * (ti + 2) / 2 <= e) ++ti; tj = e - ti * (ti + 1) / 2; }
      else { ti = e % ta; tj = e / ta; }
    }
    double acc[TS * TS];
#pragma unroll
    for (int k = 0; k < TS * TS; ++k) acc[k] = 0;
    for (int l0 = 0; l0 < kd; l0 += rows) {
      const int nrow = kd - l0 < rows ? kd - l0 : rows;
      __syncthreads();
      for (long x = c.tid; x < (long)nrow * ma; x += c.nt) { const int k = (int)(x / nrow), l = (int)(x - (long)k * nrow); c.lds[l * ldl + k] = A[l0 + l + lda * k]; }
      for (long x = c.tid; x < (long)nrow * nb; x += c.nt) { const int k = (int)(x / nrow), l = (int)(x - (long)k * nrow); c.lds[l * ldl + ma + k] = B[l0 + l + ldb * k]; }
      __syncthreads();
      if (e < ntl) {
        const double* li = c.lds + TS * ti; const double* lj = c.lds + ma + TS * tj;
        int io[TS], jo[TS];                       // offsets clamped inside the matrices (edge tiles)
#pragma unroll
        for (int q = 0; q < TS; ++q) { io[q] = TS * ti + q < ma ? q : ma - 1 - TS * ti; jo[q] = TS * tj + q < nb ? q : nb - 1 - TS * tj; }
        for (int l = 0; l < nrow; ++l) {
          const double* ri = li + l * ldl; const double* rj = lj + l * ldl;
          double xv[TS], yv[TS];
#pragma unroll
          for (int q = 0; q < TS; ++q) { xv[q] = ri[io[q]]; yv[q] = rj[jo[q]]; }
#pragma unroll
          for (int qi = 0; qi < TS; ++qi)
#pragma unroll
            for (int qj = 0; qj < TS; ++qj) acc[qi * TS + qj] += xv[qi] * yv[qj];
        }
      }
    }
    if (e < ntl)
#pragma unroll
      for (int qi = 0; qi < TS; ++qi)
#pragma unroll
        for (int qj = 0; qj < TS; ++qj) { const int i = TS * ti + qi, j = TS * tj + qj; if (i < ma && j < nb && (!LOWER || i >= j)) st(i, j, acc[qi * TS + qj]); }
  }
  __syncthreads();
}
// 4 x 4 tiles per thread, 5 x 5 when that saves a pass over the operands (every pass re-stages A and B); atb_lower: only the
// entries (i, j), i >= j, of a square result (half the tiles)
template <class ST> LIT_FN void atb(const Ctx& c, const double* A, long lda, int ma, const double* B, long ldb, int nb, int kd, ST st) {
  const int t4 = ((ma + 3) / 4) * ((nb + 3) / 4), t5 = ((ma + 4) / 5) * ((nb + 4) / 5);
  if ((t4 + c.nt - 1) / c.nt <= (t5 + c.nt - 1) / c.nt) atb_ts<4, false>(c, A, lda, ma, B, ldb, nb, kd, st);
  else atb_ts<5, false>(c, A, lda, ma, B, ldb, nb, kd, st);
}
template <class ST> LIT_FN void atb_lower(const Ctx& c, const double* A, long lda, int ma, const double* B, long ldb, int kd, ST st) {
  const int a4 = (ma + 3) / 4, a5 = (ma + 4) / 5, t4 = a4 * (a4 + 1) / 2, t5 = a5 * (a5 + 1) / 2;
  if ((t4 + c.nt - 1) / c.nt <= (t5 + c.nt - 1) / c.nt) atb_ts<4, true>(c, A, lda, ma, B, ldb, ma, kd, st);
  else atb_ts<5, true>(c, A, lda, ma, B, ldb, ma, kd, st);
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
  double* X;                // dense route only: [ldx x (n + 1)] column-major: [H_o(:, 15:) | r_o], then R / reflectors, then Q'
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
  double* G;                // dense route only: [ldg x r_cap] column-major: u-rows of A Q_1   (R_n = v' I + (u' - v') G^T G)
  int ldz;                  // r_cap + n + 1
  double* Z;                // [ldz x ldz] column-major lower triangle: [[R_n, .], [TH^T, 0]] -> Schur complement -Lam^
  // ---- compact route: H_o^T H_o as k_gram left it, scratch
  const double* LamIn;      // [H_o | r_o]^T [H_o | r_o], element (hi, lo), lo <= hi <= n, at LamIn[hi * ldL + lo] (+ split-K copies)
  long lam_part; int gram_parts;   // copies of LamIn lam_part doubles apart: block column lo / 64 came in min(lo / 64 + parts - 2, parts) partial sums
  double* W2;               // scratch of the compact route: compact_ws_doubles(6 n_cap, m_cap, r_cap)
  // ---- outputs
  double* Lam; int ldL;     // Lam^(hi, lo), lo <= hi <= n, at Lam[hi * ldL + lo]  (what k_chol_mfma / lam_hat read)
  int* info;                // [8]: stacked rows m, kept rows r, reflected steps, steps skipped by the tolerance, route (3 compact, 2 dense sweep),
                            // leading steps that reflect nothing (15), 0, 0
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

template <class HT> LIT_FN void information_from_rn(const Ctx& c, const Args<HT>& a, int n, int nr);

// Tail of both routes: R_n = v' I + (u' - v') G^T G (msckf.h:1366), then Z = [[R_n, .], [[T_H | r_n]^T, 0]] (lower triangle);
// eliminating the nr pivots of R_n leaves -[T_H | r_n]^T R_n^-1 [T_H | r_n] in the trailing block = -Lam^.
template <class HT>
LIT_FN void information_from_compressed(const Ctx& c, const Args<HT>& a, int n, int nr, int mobs) {
  const long ldz = a.ldz;
  double* Z = a.Z;
  const double dlt = a.u_var - a.v_var;
  syrk_lower(c, a.G, a.ldg, nr, mobs, [&](int i, int j, double sgg) { Z[i + ldz * j] = dlt * sgg + (i == j ? a.v_var : 0.0); });
  barrier(c);
  information_from_rn(c, a, n, nr);
}

// Z(0:nr, 0:nr) holds the lower triangle of R_n: append [T_H | r_n]^T, eliminate, store Lam^
template <class HT>
LIT_FN void information_from_rn(const Ctx& c, const Args<HT>& a, int n, int nr) {
  const int rc = a.r_cap;
  const int nz = nr + n + 1;
  const long ldz = a.ldz;
  double* Z = a.Z;
  par_for(c, (long)(n + 1) * nz, [&](long e) {
    const int j = (int)(e / (n + 1)), cc = (int)(e - (long)j * (n + 1));
    Z[(nr + cc) + ldz * j] = j < nr ? a.TH[j + (long)rc * cc] : 0.0;
  });
  barrier(c);
  tick(c, 10);
  // Elimination of the nr pivots, Z(i, j) -= Z(i, k) (Z(j, k) / d_k) for i >= j > k, in panels of PB pivots: the panel's columns
  // (rows k0 .., all that is left of them) are staged in LDS and eliminated against each other there, then ONE pass over the
  // trailing triangle in global memory applies the panel's PB updates to every element, in pivot order -- the same operands,
  // the same operations in the same order as pivot by pivot (which took a barrier and ~20 dependent global round trips per pivot:
  // 3.4 ms of 182 pivots on a 363-square matrix; this form: 12 passes).  The eliminated columns are not written back (only the
  // trailing block is read afterwards).
  {
    int PB = 16;
    while (PB > 1 && (long)(nz + 1) * PB + PB > c.lds_doubles) PB >>= 1;
    double* sD = c.lds;                                   // [PB] 1 / d_k of the panel's pivots
    double* sP = c.lds + PB;                              // [rows k0 .. nz)[PB], row-major with PB + (PB < 16 ? 0 : 1) padding
    const int ldp = PB + (PB >= 16 ? 1 : 0);
    if ((long)(nz + 1) * ldp + PB > c.lds_doubles) PB = 0;   // (cannot happen with the sizes the callers allocate: fall back below)
    for (int k0 = 0; PB > 0 && k0 < nr; k0 += PB) {
      const int pb = nr - k0 < PB ? nr - k0 : PB, mrow = nz - k0;
      // stage: element (k0 + i, k0 + cc); inside the panel's diagonal block the upper half is filled from the mirror image
      par_for(c, (long)mrow * pb, [&](long e) {
        const int cc = (int)(e / mrow), i = (int)(e - (long)cc * mrow);
        sP[(long)i * ldp + cc] = i >= cc ? Z[(k0 + i) + ldz * (k0 + cc)] : Z[(k0 + cc) + ldz * (k0 + i)];
      });
      barrier(c);
      for (int cc = 0; cc < pb; ++cc) {
        // column cc against the panel's later columns: rows i >= c2 of column c2 > cc
        const double dinv = 1.0 / sP[(long)cc * ldp + cc];
        if (first_thread(c)) sD[cc] = dinv;
        const int nc = pb - 1 - cc;
        par_for(c, (long)mrow * nc, [&](long e) {
          const int q = (int)(e / mrow), i = (int)(e - (long)q * mrow), c2 = cc + 1 + q;
          if (i < c2) return;
          const double ljk = sP[(long)c2 * ldp + cc] * dinv;
          sP[(long)i * ldp + c2] -= sP[(long)i * ldp + cc] * ljk;
        });
        barrier(c);
      }
      // trailing triangle: columns j >= k0 + pb, rows i >= j (a rectangle of indices, the upper half skipped)
      const int j0 = k0 + pb, mt = nz - j0;
      par_for(c, (long)mt * mt, [&](long e) {
        const int jj = (int)(e / mt), ii = (int)(e - (long)jj * mt);
        if (ii < jj) return;
        const double* pi = sP + (long)(pb + ii) * ldp; const double* pj = sP + (long)(pb + jj) * ldp;
        double z = Z[(j0 + ii) + ldz * (j0 + jj)];
        for (int cc = 0; cc < pb; ++cc) z -= pi[cc] * (pj[cc] * sD[cc]);
        Z[(j0 + ii) + ldz * (j0 + jj)] = z;
      });
      barrier(c);
    }
    if (PB == 0)
      for (int k = 0; k < nr; ++k) {
        const double dinv = 1.0 / Z[k + ldz * k];
        const double* zk = Z + ldz * k;
        wave_for(c, k + 1, nz, [&](long j) {
          const double ljk = zk[j] * dinv;
          if (ljk == 0.0) return;
          double* zj = Z + ldz * j;
          lane_for(c, j, nz, [&](long i) { zj[i] -= zk[i] * ljk; });
        });
        barrier(c);
      }
  }
  tick(c, 11);
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

// H_o^T H_o as k_gram left it: the lower triangle in up to `gram_parts` partial sums
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

// ---------------------------------------------------------------------------------------------------------------------
// The compact route: HouseholderQR(H_o) in the reference's order, STEP BY STEP, for any shape of stack, without the stack.
// Only the rows 0 .. e-1, e = min(m, 15 + n), can ever be pivot rows (step 15 + k pivots on row 15 + k): they are kept
// explicitly (E, e x (n + 1)).  Of the rows below (B) a reflector needs only inner products of columns -- the Gram matrix
// Gb = B^T B = [H_o | r_o]^T [H_o | r_o] - E^T E, which k_gram already accumulates in f64 -- and leaves B as B0 Y for a
// coefficient matrix Y it updates by column operations:
//     |tail|^2 = sum_(i>p) E(i,k)^2 + Gb(k,k)                        v = [.. 1 | E(i>p,k) | B(:,k)] / (c0 - beta)
//     v^T x_j  = E(p,j) + sum_(i>p) v_i E(i,j) + Gb(k,j) / (c0-beta)  s_j = tau v^T x_j,  a_j = s_j / (c0 - beta)
//     E(p,j) -= s_j,  E(i>p,j) -= s_j v_i,  B(:,j) -= a_j B(:,k):   Y(:,j) -= a_j Y(:,k),
//     Gb(j,l) -= a_l Gb(j,k) + a_j Gb(k,l) - a_j a_l Gb(k,k)
// -- the same numbers as the sweep over the dense stack (literal_general), O(e n + n^2) per step instead of O(m n).  A
// vector of the stack's row space is then [t ; B0 y]: the columns of Q_1 = H_0 H_1 .. e_row are built in that form from the
// stored reflectors (v^T [t ; B0 y] = v_E^T t + (Gb0 y_v)^T y), and the u-rows of A Q_1 that R_n = Q_1^T R_o Q_1 needs are
//     G = G_E Tq + Hu Yq,   G_E(:, i) = u-rows of A e_i (explicit rows),  Hu = u-rows of A_B A_B^T H_x (rows in B),
// so that G^T G = Tq^T (G_E^T G_E) Tq + Tq^T (G_E^T Hu) Yq + (.)^T + Yq^T (Hu^T Hu) Yq with three small matrices accumulated
// per track (Hu^T Hu is block-local: a track touches its own cameras' columns).  Nothing of size m x n exists.
LIT_FN long compact_ws_doubles(int n, int m_cap, int r_cap, int ldg) {
  const long n1 = n + 1, ec = 15 + n;
  return ec * n1 + ec * 2L * m_cap + n1 * n1 + 4L * n * n + 2 * (ec * (long)r_cap + (long)n * r_cap) + ec * ec + 4 * ec * (long)n + ec * (long)r_cap + (long)n * r_cap + (long)ldg * n + 64;
}

template <class HT>
LIT_FN void literal_compact(const Ctx& c, const Args<HT>& a, const int m, const int mobs) {
  const int n = 6 * a.N, F = a.F, n1 = n + 1, D = 15 + n;
  const int e = m < D ? m : D;                  // explicit rows
  const int steps_total = e, msteps = steps_total - 15 > 0 ? steps_total - 15 : 0;
  const long ec = 15 + n, rc = a.r_cap;
  double* E = a.W2;                             // [ec x n1] column-major (leading dimension ec)
  double* At = E + ec * n1;                     // [ec][2 m_cap]: a_i = A_j e_(i - row0) for the explicit rows
  double* Gb = At + ec * 2L * a.m_cap;          // [n1 x n1] symmetric, both triangles
  double* Gb0 = Gb + (long)n1 * n1;             // [n x n] Gb before the sweep (Jacobian columns)
  double* Y = Gb0 + (long)n * n;                // [n x n] B = B0 Y
  double* Yv = Y + (long)n * n;                 // [n x n] column k: B-part of reflector k in coordinates of B0's columns
  double* Gv = Yv + (long)n * n;                // [n x n] column k: Gb0 Yv(:, k)
  double* Tq = Gv + (long)n * n;                // [ec x rc] explicit-row part of the kept columns of Q
  double* Yq = Tq + ec * rc;                    // [n x rc] B-part of the kept columns of Q (coordinates of B0's columns)
  double* See = Yq + (long)n * rc;              // [ec x ec] G_E^T G_E
  double* Seb = See + ec * ec;                  // [ec x n] G_E^T Hu
  double* SebT = Seb + ec * n;                  // [n x ec] the same transposed (each of the two products that use it reads it along its contraction index)
  double* P1 = SebT + ec * n;                   // [ec x rc] See Tq + Seb Yq
  double* P3 = P1 + ec * rc;                    // [n x rc] Gam Yq + Seb^T Tq
  double* TqT = P3 + (long)n * rc;              // [rc x ec] Tq transposed (the last product reads it along the kept columns)
  double* YqT = TqT + rc * ec;                  // [rc x n]
  double* VE = YqT + rc * (long)n;              // [ec x n] explicit parts of the reflectors as a matrix (zero above the pivot row, 1 in it)
  double* VET = VE + ec * (long)n;              // [n x ec] the same transposed
  double* Hh = VET + ec * (long)n;              // [ldg x n] u-rows of every track's projected Jacobian (stacked observations x state columns)
  double* Gam = Gv;                             // Hu^T Hu reuses Gv's space once the columns of Q are built
  const int ks = n + 16;
  int* flag = a.kept + ks;
  int* topt = a.kept + 2 * ks;                  // [ec] track of explicit row i
  tick(c, 1);
  // ---- explicit rows: a_i = Q_f e_(3 + i - row0) and row i of [H_o | r_o] (msckf.h:957, :430)
  if (first_thread(c)) {
    int t = 0;
    for (int i = 0; i < e; ++i) {
      while (!(a.status[t] & a.inc_bit) || a.row0[t] + 2 * a.M[t] - 3 <= i) ++t;
      topt[i] = t;
    }
  }
  par_for(c, (long)e * n1, [&](long x) { const long j = x / e, i = x - j * e; E[i + ec * j] = 0.0; });
  barrier(c);
  par_for(c, e, [&](long i) {
    const int t = topt[i], M = a.M[t], R2 = 2 * M, q = 3 + (int)i - a.row0[t];
    const double* V = a.Vf + (long)t * 2 * a.m_cap * 3;
    const double* T = a.Tf + (long)t * 9;
    double* ai = At + i * 2 * a.m_cap;
    double sv[3], w[3];
    for (int p = 0; p < 3; ++p) sv[p] = vf_at(V, q, p);
    for (int p = 0; p < 3; ++p) { double x = 0; for (int qq = p; qq < 3; ++qq) x += T[p * 3 + qq] * sv[qq]; w[p] = x; }
    for (int r = 0; r < R2; ++r) ai[r] = (r == q ? 1.0 : 0.0) - (vf_at(V, r, 0) * w[0] + vf_at(V, r, 1) * w[1] + vf_at(V, r, 2) * w[2]);
    const HT* hx = a.Hx + (long)t * a.m_cap * 12;
    const HT* rr = a.rw + (long)t * 2 * a.m_cap;
    double sr = 0;
    for (int o = 0; o < M; ++o) {
      const int col = 6 * a.slots[first_obs(a, t) + o];
      for (int kk = 0; kk < 6; ++kk) E[i + ec * (col + kk)] = ai[2 * o] * (double)hx[o * 12 + kk] + ai[2 * o + 1] * (double)hx[o * 12 + 6 + kk];
      sr += ai[2 * o] * (double)rr[2 * o] + ai[2 * o + 1] * (double)rr[2 * o + 1];
    }
    E[i + ec * n] = sr;
  });
  barrier(c);
  tick(c, 2);
  // ---- Gb = [H_o | r_o]^T [H_o | r_o] - E^T E (zero when every row is explicit), Y = I
  // (E^T E through the LDS-staged product: one thread per entry walking two columns of E in global memory took 1.9 ms)
  auto gb_store = [&](int hi, int lo, double ete) {
    const double full = (hi == n && lo == n) ? 0.0 : lam_in(a, hi, lo);
    const double v = m > e ? full - ete : 0.0;
    Gb[hi + (long)n1 * lo] = v; Gb[lo + (long)n1 * hi] = v;
    if (hi < n) { Gb0[hi + (long)n * lo] = v; Gb0[lo + (long)n * hi] = v; }
  };
  if (m > e) syrk_lower(c, E, ec, n1, e, gb_store);
  else par_for(c, (long)n1 * n1, [&](long x) { const int lo = (int)(x / n1), hi = (int)(x - (long)lo * n1); if (hi >= lo) gb_store(hi, lo, 0.0); });
  par_for(c, (long)n * n, [&](long x) { const long j = x / n, i = x - j * n; Y[x] = i == j ? 1.0 : 0.0; Yv[x] = 0.0; });
  barrier(c);
  tick(c, 3);
  // ---- the sweep (msckf.h:1343): steps 0..14 meet the zero IMU columns; step 15 + k works on camera column k, pivot row 15 + k
  const double tol2 = a.tol * a.tol;
  int n_reflect = 0, n_skip_tol = 0;
  for (int k = 0; k < msteps; ++k) {
    const int p = 15 + k;
    double* ek = E + ec * k;
    const double gkk = Gb[k + (long)n1 * k] > 0.0 ? Gb[k + (long)n1 * k] : 0.0;
    // |head|^2 (rows 0 .. p) and |tail|^2 (rows below p, + the Gram entry for the rows in B) of column k in one reduction
    double head2, tail2;
    wg_sum2(c, 0, e, head2, tail2, [&](long i, double& hs, double& ts) { const double v = ek[i] * ek[i]; if (i <= p) hs += v; else ts += v; });
    tail2 += gkk;
    double zero2 = 2.2250738585072014e-308;
    // The part of the tail that lives in B comes out of the Gram matrix, which resolves |tail|^2 to ~1e-8 |column|^2 at best
    // (measured over the benchmark's sequences: exactly dependent columns leave 3e-11 .. 3e-8, independent ones 2e-6 and
    // more): no finer a threshold than 1e-7 while rows below the explicit ones exist -- also when tol = 0 asks for the
    // reference's rule to the letter, which only the sweep over the dense stack can honour
    const double t2 = (m > e && tol2 < 1e-7) ? 1e-7 : tol2;
    if (t2 > 0) {
      const double z = t2 * (head2 + tail2);
      zero2 = z > zero2 ? z : zero2;
    }
    const double c0 = ek[p];
    if (tail2 <= zero2) {
      if (tail2 > 2.2250738585072014e-308) ++n_skip_tol;
      if (first_thread(c)) a.tau[k] = 0.0;
      par_for(c, e - (p + 1), [&](long i) { ek[p + 1 + i] = 0.0; });
      barrier(c);
      continue;
    }
    ++n_reflect;
    double beta = sqrt(c0 * c0 + tail2);
    if (c0 >= 0.0) beta = -beta;
    const double dn = 1.0 / (c0 - beta), tk = (beta - c0) / beta;
    // a_j for the columns to the right (r_o rides along as column n); kept in tau's tail [n .. 2n].  The reflector's entries
    // are ek[i] dn, formed on the fly from the unscaled column (the same product the store below rounds): column k itself is
    // scaled, and Yv, beta, tau are written, in the NEXT phase, which does not read E -- one phase and one barrier less per step
    double* aj = a.tau + n1;
    row_for(c, k + 1, n1, [&](long j) {
      double* ej = E + ec * j;
      double sdot = row_sum_range(c, p + 1, e, [&](long i) { return (ek[i] * dn) * ej[i]; });
      sdot = (sdot + ej[p] + Gb[j + (long)n1 * k] * dn) * tk;
      rowlane_update(c, p + 1, e, [&](long i) { return ej[i] - sdot * (ek[i] * dn); }, [&](long i, double v) { ej[i] = v; });
      if (first_rowlane(c)) { ej[p] -= sdot; aj[j] = sdot * dn; }
    });
    barrier(c);
    par_for(c, e - (p + 1), [&](long i) { ek[p + 1 + i] *= dn; });
    par_for(c, n, [&](long i) { Yv[i + (long)n * k] = Y[i + (long)n * k] * dn; });
    if (first_thread(c)) { ek[p] = beta; a.tau[k] = tk; }
    // B(:, j) -= a_j B(:, k): Gram and coefficients (columns k+1 .. n; Y only over the Jacobian columns).  One row of a
    // wavefront per column j, its lanes along l >= j / along the rows of Y: contiguous loads and stores, no index divisions,
    // and only the lower triangle of Gb is kept up to date (every reader asks for (hi, lo)) -- one thread per entry with the
    // mirror image stored too took most of the sweep's 27 us per step (a memory round trip per entry: a store, then the next
    // entry's loads)
    row_for(c, k + 1, n1, [&](long j) {
      const double cj = Gb[j + (long)n1 * k], ajj = aj[j];
      double* gj = Gb + (long)n1 * j; const double* gk = Gb + (long)n1 * k;
      rowlane_update(c, j, n1, [&](long l) { return gj[l] - aj[l] * cj - ajj * gk[l] + ajj * aj[l] * gkk; }, [&](long l, double v) { gj[l] = v; });
      if (j < n) {
        double* yj = Y + (long)n * j; const double* yk = Y + (long)n * k;
        rowlane_update(c, 0, k + 1, [&](long i) { return yj[i] - ajj * yk[i]; }, [&](long i, double v) { yj[i] = v; });   // column k of Y has its support in rows 0..k
      }
    });
    barrier(c);
  }
  tick(c, 4);
  // ---- rows of R that are kept (msckf.h:1345-1348) and [T_H | r_n]
  double rmax = 0;
  if (a.tol > 0) rmax = wg_max(c, 0, (long)steps_total * n, [&](long x) { const long j = x / steps_total, i = x - j * steps_total; return (j + 15 >= i) ? fabs(E[i + ec * j]) : 0.0; });
  barrier(c);
  par_for(c, steps_total, [&](long i) {
    int any = 0;
    const int c_lo = i >= 15 ? (int)i - 15 : 0;
    for (int j = c_lo; j < n && !any; ++j) { const double v = fabs(E[i + ec * j]); any = a.tol > 0 ? (v > a.tol * rmax) : (v != 0.0); }
    flag[i] = any;
  });
  barrier(c);
  if (first_thread(c)) {
    int nr = 0;
    for (int i = 0; i < steps_total; ++i) if (flag[i]) a.kept[nr++] = i;
    a.info[1] = nr; a.info[2] = n_reflect; a.info[3] = n_skip_tol; a.info[4] = 3; a.info[5] = steps_total - msteps;
  }
  barrier(c);
  const int nr = a.info[1];
  par_for(c, (long)nr * n1, [&](long x) {
    const int j = (int)(x / nr), k = (int)(x - (long)j * nr), row = a.kept[k];
    double v = E[row + ec * j];
    if (j < n && j + 15 < row) v = 0.0;
    a.TH[k + rc * j] = v;
  });
  // ---- Gv(:, k) = Gb0 Yv(:, k) for the reflectors that exist (Yv(:, k) has its support in rows 0..k)
  // (as a product: Yv is zero below its diagonal and in the columns of the steps that did not reflect, Gb0 is symmetric; one
  // thread per entry walking a column of Yv paid a memory round trip per term: 0.7 ms)
  if (m > e && msteps > 0) atb(c, Gb0, n, n, Yv, n, msteps, n, [&](int i, int k, double v) { Gv[i + (long)n * k] = v; });
  else par_for(c, (long)n * msteps, [&](long x) { Gv[x] = 0.0; });
  barrier(c);
  tick(c, 5);
  // ---- kept columns of Q = H_0 H_1 ..: q = H_0 .. H_k e_(15 + k) as [t ; B0 y]; a column of a row < 15 is e_row itself.
  // Compact WY form: Q = I - V T V^T with T^-1 = striu(V^T V) + diag(1 / tau) (V: the reflectors that exist, columns
  // [explicit part ; B0 yv]; v_i^T v_j = VE_i^T VE_j + yv_i^T Gb0 yv_j), so that Q e_c = e_c - V X with T^-1 X = V^T e_c: two
  // products for V^T V, one back substitution over the steps, two products for [Tq ; Yq].  (Reflector by reflector over every
  // column that needs it -- a dot product and an update per column and step, each a chain of dependent global round trips --
  // took 3.1 ms of 12 at a 30-camera window; one thread per column walking all its reflectors 21 ms.)
  {
    const int ms = msteps;
    double* YvT = Y;                    // [ms x n] (Y is free after the sweep)
    double* Uc = Gb;                    // [ms x ms] column-major, strictly upper part: Uc(r, j) = v_r^T v_j, r < j (Gb is free after the sweep)
    double* Wx = P1;                    // [ms x nr] V^T e_c, then X (P1 is formed later)
    if (ms > 0) {
      par_for(c, (long)e * ms, [&](long x) {
        const int j = (int)(x / e), i = (int)(x - (long)j * e), pj = 15 + j;
        const double v = a.tau[j] == 0.0 ? 0.0 : (i < pj ? 0.0 : (i == pj ? 1.0 : E[i + ec * j]));
        VE[i + ec * j] = v; VET[j + (long)ms * i] = v;
      });
      par_for(c, (long)n * ms, [&](long x) { const int j = (int)(x / n), l = (int)(x - (long)j * n); YvT[j + (long)ms * l] = a.tau[j] == 0.0 ? 0.0 : Yv[l + (long)n * j]; });
      par_for(c, (long)ms * nr, [&](long x) {
        const int ka = (int)(x / ms), j = (int)(x - (long)ka * ms), pj = 15 + j, row = a.kept[ka];
        Wx[x] = a.tau[j] == 0.0 ? 0.0 : (row < pj ? 0.0 : (row == pj ? 1.0 : E[row + ec * j]));
      });
      barrier(c);
      syrk_lower(c, VE, ec, ms, e, [&](int i, int j, double v) { if (i > j) Uc[j + (long)ms * i] = v; });
      if (m > e) atb_lower(c, Yv, n, ms, Gv, n, n, [&](int i, int j, double v) { if (i > j) Uc[j + (long)ms * i] += v; });
      barrier(c);
      // back substitution, last step first: X(j, :) = tau_j W(j, :), W(r, :) -= U(r, j) X(j, :) for r < j
      for (int j = ms - 1; j >= 1; --j) {
        const double tj = a.tau[j];
        if (tj == 0.0) continue;
        const double* uj = Uc + (long)ms * j;
        row_for(c, 0, nr, [&](long ka) {
          double* w = Wx + (long)ms * ka;
          const double xj = tj * w[j];
          if (xj == 0.0) return;
          rowlane_update(c, 0, j, [&](long r) { return w[r] - uj[r] * xj; }, [&](long r, double v) { w[r] = v; });
        });
        barrier(c);
      }
      par_for(c, (long)ms * nr, [&](long x) { const int j = (int)(x % ms); Wx[x] *= a.tau[j]; });
      barrier(c);
      atb(c, VET, ms, e, Wx, ms, nr, ms, [&](int i, int ka, double v) { Tq[i + ec * ka] = (i == a.kept[ka] ? 1.0 : 0.0) - v; });
      if (m > e) atb(c, YvT, ms, n, Wx, ms, nr, ms, [&](int l, int ka, double v) { Yq[l + (long)n * ka] = -v; });
      else par_for(c, (long)n * nr, [&](long x) { Yq[x] = 0.0; });
    } else {
      par_for(c, (long)e * nr, [&](long x) { const int ka = (int)(x / e), i = (int)(x - (long)ka * e); Tq[i + ec * ka] = i == a.kept[ka] ? 1.0 : 0.0; });
      par_for(c, (long)n * nr, [&](long x) { Yq[x] = 0.0; });
    }
    barrier(c);
  }
  tick(c, 6);
  // ---- G_E^T G_E, G_E^T Hu, Hu^T Hu (Gam takes Gv's place).  Hu -- the u-rows of every track's projected Jacobian -- is
  // laid out once as a dense (stacked observations) x n matrix (one wavefront per track), and Hu^T Hu is then the same
  // LDS-staged product as G^T G of the dense route: Hu is read once.  (Track by track with two barriers each: 13 ms; one
  // thread per entry gathering over the tracks: 26 ms; this: ~3 ms.)
  double* Hu = Hh;                                      // [ldg x n] column-major
  const long ldh = a.ldg;
  par_for(c, ec * ec, [&](long x) { See[x] = 0.0; });
  par_for(c, ec * (long)n, [&](long x) { Seb[x] = 0.0; SebT[x] = 0.0; });
  par_for(c, (long)mobs * n, [&](long x) { const long j = x / mobs, g = x - j * mobs; Hu[g + ldh * j] = 0.0; });
  barrier(c);
  tick(c, 12);
  if (m > e)
    row_for(c, 0, F, [&](long t) {                         // a row of 16 lanes per track: 64 tracks of the workgroup at a time
      if (!(a.status[t] & a.inc_bit)) return;
      const int M = a.M[t], rho = 2 * M - 3, r0 = a.row0[t];
      int kE = e - r0; kE = kE < 0 ? 0 : (kE > rho ? rho : kE);      // rows of the track that are explicit
      const int d = 3 + kE;
      if (d >= 2 * M) return;                                          // every row of the track is explicit: nothing of it in B
      const double* V = a.Vf + (long)t * 2 * a.m_cap * 3;
      const double* T = a.Tf + (long)t * 9;
      const HT* hx = a.Hx + (long)t * a.m_cap * 12;
      const int so = first_obs(a, t), g0 = a.obs0[t];
      // u-rows of (I - Q_f(:, :d) Q_f(:, :d)^T) H_x_j, scattered to the state columns of the track's cameras: lane = observation o
      // (what depends on o alone -- e_o^T Q_f(:, :d) Q_f(:, :d)^T in the reflectors' coordinates -- is formed once per lane, not
      // once per entry: one lane per entry redid it 6 M times, 1.3 ms of this phase), the 6 M columns in a loop; consecutive
      // lanes write consecutive rows of H_u
      rowlane_for(c, 0, M, [&](long ol) {
        const int o = (int)ol;
        double tv[3], ev[3] = {0, 0, 0};
        for (int p2 = 0; p2 < 3; ++p2) tv[p2] = vf_at(V, 2 * o, 0) * T[0 * 3 + p2] + vf_at(V, 2 * o, 1) * T[1 * 3 + p2] + vf_at(V, 2 * o, 2) * T[2 * 3 + p2];
        auto qf = [&](int q) -> double { return (q == 2 * o ? 1.0 : 0.0) - (tv[0] * vf_at(V, q, 0) + tv[1] * vf_at(V, q, 1) + tv[2] * vf_at(V, q, 2)); };
        for (int q = 0; q < d; ++q) { const double f = qf(q); for (int p2 = 0; p2 < 3; ++p2) ev[p2] += f * vf_at(V, q, p2); }
        for (int op = 0; op < M; ++op) {
          const double q0 = 2 * op < d ? qf(2 * op) : 0.0, q1 = 2 * op + 1 < d ? qf(2 * op + 1) : 0.0;
          double* hrow = Hu + (g0 + o) + ldh * (6 * a.slots[so + op]);
          // everything of observation op is fetched before the first store (a store, then the next column's loads, is a memory
          // round trip per column)
          double hh[12], vv[6], val[6];
          for (int x = 0; x < 12; ++x) hh[x] = (double)hx[op * 12 + x];
          for (int p2 = 0; p2 < 3; ++p2) { vv[p2] = vf_at(V, 2 * op, p2); vv[3 + p2] = vf_at(V, 2 * op + 1, p2); }
          for (int kk = 0; kk < 6; ++kk) {
            const double h0 = hh[kk], h1 = hh[6 + kk];
            double sv[3], vl = op == o ? h0 : 0.0;
            for (int p2 = 0; p2 < 3; ++p2) sv[p2] = vv[p2] * h0 + vv[3 + p2] * h1;
            for (int q = 0; q < 3; ++q) { double w = 0; for (int p2 = 0; p2 <= q; ++p2) w += T[p2 * 3 + q] * sv[p2]; vl += ev[q] * w; }
            if (2 * op < d) vl -= q0 * h0;
            if (2 * op + 1 < d) vl -= q1 * h1;
            val[kk] = vl;
          }
          for (int kk = 0; kk < 6; ++kk) hrow[ldh * kk] = val[kk];
        }
      });
    });
  barrier(c);
  tick(c, 13);
  if (m > e) syrk_lower(c, Hu, ldh, n, mobs, [&](int i, int j, double sv2) { Gam[i + (long)n * j] = sv2; Gam[j + (long)n * i] = sv2; });
  else par_for(c, (long)n * n, [&](long x) { Gam[x] = 0.0; });
  tick(c, 14);
  // the tracks that own explicit rows: G_E^T G_E blocks, and G_E^T Hu for the one with rows on both sides of e
  for (int t = topt[0]; t <= topt[e - 1]; ++t) {
    if (!(a.status[t] & a.inc_bit)) continue;
    const int M = a.M[t], rho = 2 * M - 3, r0 = a.row0[t], g0 = a.obs0[t];
    int kE = e - r0; kE = kE < 0 ? 0 : (kE > rho ? rho : kE);
    par_for(c, (long)kE * kE, [&](long x) {
      const int i2 = (int)(x / kE), i1 = (int)(x - (long)i2 * kE);
      double sacc = 0;
      for (int o = 0; o < M; ++o) sacc += At[(long)(r0 + i1) * 2 * a.m_cap + 2 * o] * At[(long)(r0 + i2) * 2 * a.m_cap + 2 * o];
      See[(r0 + i1) + ec * (r0 + i2)] = sacc;
    });
    if (m > e && kE < rho)
      par_for(c, (long)kE * n, [&](long x) {
        const int col = (int)(x / kE), i1 = (int)(x - (long)col * kE);
        const double* hh = Hu + g0 + ldh * col;
        double sacc = 0;
        for (int o = 0; o < M; ++o) sacc += At[(long)(r0 + i1) * 2 * a.m_cap + 2 * o] * hh[o];
        Seb[(r0 + i1) + ec * col] = sacc; SebT[col + (long)n * (r0 + i1)] = sacc;
      });
  }
  barrier(c);
  tick(c, 7);
  // ---- G^T G = Tq^T (See Tq + Seb Yq) + Yq^T (Seb^T Tq + Gam Yq)
  // See is block diagonal (an explicit row meets only the rows of its own track) and Seb has rows only for the one track
  // that has rows on both sides of e
  // All as LDS-staged products over the dense matrices (the entries outside See's blocks and Seb's rows are exact zeros: the
  // sums are the same): one thread per entry with a loop over the rows of the entry's track paid a memory round trip per term
  // (1.4 ms for P1 alone)
  const int ts = topt[e - 1], rs = a.row0[ts];           // the track of the last explicit row: the only one that can straddle
  const bool straddle = m > e && rs + 2 * a.M[ts] - 3 > e;
  atb(c, See, ec, e, Tq, ec, nr, e, [&](int i, int ka, double v) { P1[i + ec * ka] = v; });
  barrier(c);
  tick(c, 15);
  if (straddle) atb(c, SebT, n, e, Yq, n, nr, n, [&](int i, int ka, double v) { P1[i + ec * ka] += v; });
  // P3 = Gam Yq (+ Seb^T Tq from the rows of the one track that has rows on both sides of e); G^T G = Tq^T P1 + Yq^T P3
  if (m > e) atb(c, Gam, n, n, Yq, n, nr, n, [&](int i, int ka, double v) { P3[i + (long)n * ka] = v; });
  else par_for(c, (long)n * nr, [&](long x) { P3[x] = 0.0; });
  barrier(c);
  if (straddle) atb(c, Seb, ec, n, Tq, ec, nr, e, [&](int i, int ka, double v) { P3[i + (long)n * ka] += v; });
  barrier(c);
  const long ldz = a.ldz;
  const double dlt = a.u_var - a.v_var;
  atb_lower(c, Tq, ec, nr, P1, ec, e, [&](int ka, int kb, double v) { a.Z[ka + ldz * kb] = v; });
  barrier(c);
  if (m > e) atb_lower(c, Yq, n, nr, P3, n, n, [&](int ka, int kb, double v) { a.Z[ka + ldz * kb] += v; });
  barrier(c);
  par_for(c, (long)nr * nr, [&](long x) {
    const int kb = (int)(x / nr), ka = (int)(x - (long)kb * nr);
    if (ka >= kb) a.Z[ka + ldz * kb] = dlt * a.Z[ka + ldz * kb] + (ka == kb ? a.v_var : 0.0);
  });
  barrier(c);
  tick(c, 8);
  information_from_rn(c, a, n, nr);
  tick(c, 9);
}

// route: 0 (default) = the compact route; 1 = the sweep over the dense stack (needs its work space X, G: tests and A/B runs)
template <class HT>
LIT_FN void literal_compress(const Ctx& c, const Args<HT>& a, const int route = 0) {
  tick(c, 0);
  const int m = prepare(c, a);
  if (m <= 0) return;
  const int mobs = a.obs0[a.F];
  if (route != 1 || !a.X || !a.G) { literal_compact(c, a, m, mobs); return; }
  literal_general(c, a, m, mobs);
}

}  // namespace lit
}  // namespace msckf
#endif
