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
// uses that Lam^ depends on Q_1 only through range(Q_1): the SAME sequence of Householder steps runs, for its decisions, on a
// compressed representation (the rows that can become pivot rows explicitly, all rows from the pivot row down through their
// Gram matrix), and Lam^ follows from a basis of that range that needs no Q; see there.
//
// Written once for two compilers: hipcc (kernels_literal.hip: one workgroup per trajectory and phase kernel, phases separated
// by barriers) and g++ -DLIT_HOST (tests/cpp/literal_host.cpp: the same phases run serially, checked against the oracle on
// the CPU).  Device-only pieces (DPP recurrences, ballot compaction, the chip-wide kernels of kernels_literal.hip) have a
// serial statement beside them here.
// All arithmetic in f64 whatever the filter's scalar type.
#ifndef MSCKF_LITERAL_CORE_H
#define MSCKF_LITERAL_CORE_H

#include <math.h>
#include <type_traits>

namespace msckf {
namespace lit {

#ifdef LIT_HOST
#define LIT_FN inline
#define LIT_HD inline
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
LIT_FN void tick_acc(const Ctx&, int, long long&) {}
LIT_FN long long tick_now(const Ctx&) { return 0; }
LIT_FN double lit_rsqrt(double x) { return 1.0 / sqrt(x); }
LIT_FN double lit_rcp(double x) { return 1.0 / x; }
// par_for with a 32-bit index (the device divides 32-bit indices ~10x faster than 64-bit ones)
template <class F> LIT_FN void par_for32(const Ctx&, int n, F f) { for (int i = 0; i < n; ++i) f(i); }
// st(i, val(i)) for i in [0, n): the device evaluates four items before it stores the first (a store followed by the next
// item's dependent loads is a memory round trip when the compiler cannot rule out that they alias)
template <class FV, class FS> LIT_FN void par_map4(const Ctx&, int n, FV val, FS st) { for (int i = 0; i < n; ++i) st(i, val(i)); }
// a section run by ONE wavefront with wave_sync between its dependent steps (no workgroup barrier inside)
LIT_FN bool first_wave(const Ctx&) { return true; }
LIT_FN void wave_sync(const Ctx&) {}
LIT_FN void wave_mem_sync(const Ctx&) {}
// out[0 .. count) = the i in [0, n) with pred(i), ascending; returns count (uniform)
template <class P> LIT_FN int compact_list(const Ctx&, int n, int* out, P pred) { int cnt = 0; for (int i = 0; i < n; ++i) if (pred(i)) out[cnt++] = i; return cnt; }
#else
#define LIT_FN __device__ __forceinline__
#define LIT_HD __host__ __device__ inline
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
LIT_FN bool first_lane(const Ctx& c) { return c.lane == 0; }
LIT_FN bool first_thread(const Ctx& c) { return c.tid == 0; }
LIT_FN void tick(const Ctx& c, int slot) { if (c.tim && c.tid == 0) c.tim[slot] = (long long)wall_clock64(); }
// accumulating phase timer (phases that repeat per panel): slot += now - prev, prev = now
LIT_FN void tick_acc(const Ctx& c, int slot, long long& prev) { if (c.tim && c.tid == 0) { const long long now = (long long)wall_clock64(); c.tim[slot] += now - prev; prev = now; } }
LIT_FN long long tick_now(const Ctx& c) { return (c.tim && c.tid == 0) ? (long long)wall_clock64() : 0; }
// hardware seed + Newton steps (dev_common.h): the IEEE sqrt / division sequences are ~30 dependent instructions each, and the
// panel cores are chains of them
LIT_FN double lit_rsqrt(double x) { return fast_rsqrt(x); }
LIT_FN double lit_rcp(double x) { return fast_rcp(x); }
template <class F> LIT_FN void par_for32(const Ctx& c, int n, F f) { for (int i = c.tid; i < n; i += c.nt) f(i); }
template <class FV, class FS> LIT_FN void par_map4(const Ctx& c, int n, FV val, FS st) {
  // (measured: four items per thread in flight -- four inlined copies of the value function -- gained nothing on the Gram
  // start and the basis products and cost registers: one item at a time)
  for (int i = c.tid; i < n; i += c.nt) st(i, val(i));
}
LIT_FN bool first_wave(const Ctx& c) { return c.wave == 0; }
// LDS hand-over between the lanes of one wavefront: the fences keep the compiler from moving reads above writes
LIT_FN void wave_sync(const Ctx&) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
// GLOBAL-memory hand-over between the lanes of one wavefront (a lane reads what another lane of its wavefront stored): the
// stores are released to the device's coherence point and the loads that follow do not come out of a stale first-level line
LIT_FN void wave_mem_sync(const Ctx&) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
// stream compaction by the first wavefront (ballot + prefix popcount, 64 candidates per round); the count reaches the other
// wavefronts through c.red.  A thread-0 loop (load flag, store index, next load) paid a memory round trip per candidate.
template <class P> LIT_FN int compact_list(const Ctx& c, int n, int* out, P pred) {
  __syncthreads();                       // red may still be read from the previous reduction
  if (c.wave == 0) {
    int cnt = 0;
    for (int base = 0; base < n; base += 64) {
      const int i = base + c.lane;
      const bool v = i < n && pred(i);
      const unsigned long long mask = __ballot(v);
      if (v) out[cnt + __popcll(mask & ((1ull << c.lane) - 1ull))] = i;
      cnt += __popcll(mask);
    }
    if (c.lane == 0) c.red[0] = (double)cnt;
  }
  __syncthreads();
  const int r = (int)c.red[0];
  __syncthreads();
  return r;
}
#endif

// ---- the panels' per-row / per-column recurrences.  A row below a panel (a column right of it) takes its PB entries through
// the panel's pivots one after the other: entry q2 changes at every pivot q < q2, so one thread per row walks a chain of
// PB x (LDS read, FMA, LDS write) round trips (16 us per panel at PB = 16).  On the device a row of 16 lanes takes an item
// instead, lane = entry: the pivot's value reaches the other lanes by a DPP row_share (no memory, a few cycles), what the
// step multiplies it with sits in registers -- sixteen dependent FMAs per item.  (PB <= 16; the host build keeps the plain loops.)
#ifdef LIT_HOST
// sweep, rows below the panel: E(p0 + r, k0 + q) at sEC[q * lde + r] -> reflector entries v_q(r)
LIT_FN void panel_rows_sweep(const Ctx&, double* sEC, long lde, const double* sS, int PB, const double* sDn, int pb, int r_lo, int r_hi) {
  for (int r = r_lo; r < r_hi; ++r) {
    double* row = sEC + r;
    for (int q = 0; q < pb; ++q) {
      const double vq = row[q * lde] * sDn[q];          // 0 for a step that did not reflect
      for (int q2 = q + 1; q2 < pb; ++q2) row[q2 * lde] -= vq * sS[q * PB + q2];
      row[q * lde] = vq;
    }
  }
}
// sweep, columns right of the panel: (E(p0 + q, j), Gh(k0 + q, j)) at (sEP, sGP)[q * n1 + jj] -> (s_q(j), R(p0 + q, j))
LIT_FN void panel_cols_sweep(const Ctx&, double* sEP, double* sGP, long n1, const double* sEC, long lde, const double* sRf, const double* sBi, int pb, int ntr) {
  for (int jj = 0; jj < ntr; ++jj) {
    double* yc = sEP + jj; double* gc = sGP + jj;
    for (int q = 0; q < pb; ++q) {
      const bool rf = sRf[q] != 0.0;
      const double yq = yc[q * n1];
      const double r = rf ? gc[q * n1] * sBi[q] : yq;
      const double sj = rf ? yq - r : 0.0;
      for (int q2 = q + 1; q2 < pb; ++q2) { gc[q2 * n1] -= sEC[q2 * lde + q] * r; yc[q2 * n1] -= sEC[q * lde + q2] * sj; }
      gc[q * n1] = r; yc[q * n1] = sj;
    }
  }
}
// elimination, rows below the panel's diagonal block: Z(k0 + r, k0 + q) at sP[q * ldr + r], left unscaled
LIT_FN void panel_rows_elim(const Ctx&, double* sP, long ldr, const double* sD, int pb, int r_lo, int r_hi) {
  for (int r = r_lo; r < r_hi; ++r) {
    double* row = sP + r;
    for (int q = 0; q + 1 < pb; ++q) {
      const double yq = row[q * ldr] * sD[q];
      for (int q2 = q + 1; q2 < pb; ++q2) row[q2 * ldr] -= yq * sP[q * ldr + q2];
    }
  }
}
#else
template <int Q> LIT_FN double row_share(double v) { return dpp_x<0x150 + Q>(v); }     // the value of lane Q of this row of 16 lanes
LIT_FN void panel_rows_sweep(const Ctx& c, double* sEC, long lde, const double* sS, int PB, const double* sDn, int pb, int r_lo, int r_hi) {
  const int g = c.lane & 15;
  const bool on = g < pb;
  double sg[16];                                         // column g of S: s_q(k0 + g), q < g
#pragma unroll
  for (int q = 0; q < 16; ++q) sg[q] = (on && q < g) ? sS[q * PB + g] : 0.0;
  const double dng = on ? sDn[g] : 0.0;
  for (int r = r_lo + 4 * c.wave + (c.lane >> 4); r < r_hi; r += 4 * c.nw) {
    double y = on ? sEC[g * lde + r] : 0.0;
    auto step = [&](auto qc) {
      constexpr int q = decltype(qc)::value;
      const double vq = row_share<q>(y * dng);
      y = g == q ? vq : y - vq * sg[q];                  // (sg[q] = 0 for the lanes at or left of the pivot)
    };
    step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{}); step(std::integral_constant<int, 3>{});
    step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{}); step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{});
    step(std::integral_constant<int, 8>{}); step(std::integral_constant<int, 9>{}); step(std::integral_constant<int, 10>{}); step(std::integral_constant<int, 11>{});
    step(std::integral_constant<int, 12>{}); step(std::integral_constant<int, 13>{}); step(std::integral_constant<int, 14>{}); step(std::integral_constant<int, 15>{});
    if (on) sEC[g * lde + r] = y;
  }
}
LIT_FN void panel_cols_sweep(const Ctx& c, double* sEP, double* sGP, long n1, const double* sEC, long lde, const double* sRf, const double* sBi, int pb, int ntr) {
  const int g = c.lane & 15;
  const bool on = g < pb;
  double rg[16], vg[16];                                 // R(p0 + q, k0 + g) and v_q(p0 + g), q < g
#pragma unroll
  for (int q = 0; q < 16; ++q) { const bool use = on && q < g; rg[q] = use ? sEC[g * lde + q] : 0.0; vg[q] = use ? sEC[q * lde + g] : 0.0; }
  const bool rfg = on && sRf[on ? g : 0] != 0.0;
  const double big = rfg ? sBi[g] : 0.0;
  for (int jj = 4 * c.wave + (c.lane >> 4); jj < ntr; jj += 4 * c.nw) {
    double gq = on ? sGP[g * n1 + jj] : 0.0, yq = on ? sEP[g * n1 + jj] : 0.0;
    auto step = [&](auto qc) {
      constexpr int q = decltype(qc)::value;
      const double rl = rfg ? gq * big : yq, sl = rfg ? yq - rl : 0.0;     // what this lane would publish as the pivot
      const double r = row_share<q>(rl), sj = row_share<q>(sl);
      gq = g == q ? rl : gq - rg[q] * r;
      yq = g == q ? sl : yq - vg[q] * sj;
    };
    step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{}); step(std::integral_constant<int, 3>{});
    step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{}); step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{});
    step(std::integral_constant<int, 8>{}); step(std::integral_constant<int, 9>{}); step(std::integral_constant<int, 10>{}); step(std::integral_constant<int, 11>{});
    step(std::integral_constant<int, 12>{}); step(std::integral_constant<int, 13>{}); step(std::integral_constant<int, 14>{}); step(std::integral_constant<int, 15>{});
    if (on) { sGP[g * n1 + jj] = gq; sEP[g * n1 + jj] = yq; }
  }
}
LIT_FN void panel_rows_elim(const Ctx& c, double* sP, long ldr, const double* sD, int pb, int r_lo, int r_hi) {
  const int g = c.lane & 15;
  const bool on = g < pb;
  double lg[16];                                         // L(k0 + g, k0 + q), q < g (unscaled)
#pragma unroll
  for (int q = 0; q < 16; ++q) lg[q] = (on && q < g) ? sP[q * ldr + g] : 0.0;
  const double dg = on ? sD[g] : 0.0;
  for (int r = r_lo + 4 * c.wave + (c.lane >> 4); r < r_hi; r += 4 * c.nw) {
    double y = on ? sP[g * ldr + r] : 0.0;
    auto step = [&](auto qc) {
      constexpr int q = decltype(qc)::value;
      const double yq = row_share<q>(y * dg);
      y -= yq * lg[q];                                   // (lg[q] = 0 for the lanes at or left of the pivot: they keep their value)
    };
    step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{}); step(std::integral_constant<int, 3>{});
    step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{}); step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{});
    step(std::integral_constant<int, 8>{}); step(std::integral_constant<int, 9>{}); step(std::integral_constant<int, 10>{}); step(std::integral_constant<int, 11>{});
    step(std::integral_constant<int, 12>{}); step(std::integral_constant<int, 13>{}); step(std::integral_constant<int, 14>{});
    if (on) sP[g * ldr + r] = y;
  }
}
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
  const double* Gam; int ldGam;    // Gam(hi, lo), lo <= hi < n, at Gam[hi * ldGam + lo]: sum over the stacked tracks of H_x^T P D_u P H_x, P = I - Q_f Q_f^T
                            // (the u-rows of every track's projected Jacobian, squared: independent of the null-space basis); gamma_rows / the caller's product
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

template <class HT> LIT_FN void information_from_rn(const Ctx& c, const Args<HT>& a, int n, int nr, bool appended = false);

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
LIT_FN void information_from_rn(const Ctx& c, const Args<HT>& a, int n, int nr, bool appended) {
  const int rc = a.r_cap, n1 = n + 1;
  const int nz = nr + n1;
  const long ldz = a.ldz;
  double* Z = a.Z;
  // appended: the caller wrote [T_H | r_n]^T below R_n itself (rows nr .., columns < nr); the block right of it starts as zero
  par_for32(c, n1 * (appended ? n1 : nz), [&](int x) {
    const int jq = x / n1, cc = x - jq * n1, j = appended ? nr + jq : jq;
    Z[(nr + cc) + ldz * j] = j < nr ? a.TH[j + (long)rc * cc] : 0.0;
  });
  barrier(c);
  tick(c, 10);
  // Elimination of the nr pivots, Z(i, j) -= Z(i, k) Z(j, k) / d_k for i >= j > k, in panels of PB pivots staged in LDS (all
  // rows that are left of the panel's columns): the PB x PB diagonal block by ONE wavefront (no workgroup barrier between
  // its pivots), then one thread per row below it takes the row through the panel's pivots (a triangular recurrence over its
  // PB entries), then ONE pass over the trailing triangle in global memory, a 4 x 4 tile per thread, applies the panel's PB
  // updates.  (Pivot by pivot: a barrier and ~20 dependent global round trips per pivot, 3.4 ms of 182 pivots on a 363-square
  // matrix; panels with a barrier per pivot and one thread per trailing entry: 0.95 ms.)  The eliminated columns are not written
  // back (only the trailing block is read afterwards).
  {
    int PB = 16;
    const long ldr = (nz + 4) | 1;                        // panel storage [q][row]: rows contiguous (a thread's four rows are one 32-byte read, no bank conflicts)
    while (PB > 1 && ldr * PB + 2 * PB > c.lds_doubles) PB >>= 1;
    double* sD = c.lds;                                   // [PB] 1 / d_k of the panel's pivots
    double* sP = c.lds + 2 * PB;                          // [PB][ldr]: Z(k0 + r, k0 + q) at sP[q * ldr + r], unscaled
    if (ldr * PB + 2 * PB > c.lds_doubles) PB = 0;        // (cannot happen with the sizes the callers allocate: fall back below)
    for (int k0 = 0; PB > 0 && k0 < nr; k0 += PB) {
      const int pb = nr - k0 < PB ? nr - k0 : PB, mrow = nz - k0;
      // stage: element (k0 + i, k0 + q); inside the panel's diagonal block the upper half is filled from the mirror image
      par_for32(c, mrow * pb, [&](int x) {
        const int q = x / mrow, i = x - q * mrow;
        sP[q * ldr + i] = i >= q ? Z[(k0 + i) + ldz * (k0 + q)] : Z[(k0 + q) + ldz * (k0 + i)];
      });
      barrier(c);
      if (first_wave(c)) {
        for (int q = 0; q < pb; ++q) {
          const double dinv = lit_rcp(sP[q * ldr + q]);
          if (first_lane(c)) sD[q] = dinv;
          // rows r > q of the block, columns q2 in (q, r]: lane = (row offset mod 4, column offset)
          lane_for(c, 0, 64, [&](long ln) {
            const int q2 = q + 1 + ((int)ln & 15);
            for (int r = q + 1 + ((int)ln >> 4); r < pb; r += 4)
              if (q2 <= r) sP[q2 * ldr + r] -= sP[q * ldr + r] * (sP[q * ldr + q2] * dinv);
          });
          wave_sync(c);
        }
      }
      barrier(c);
      // rows below the block: entry q2 of the row after the pivots q < q2 of the panel
      panel_rows_elim(c, sP, ldr, sD, pb, pb, mrow);
      barrier(c);
      // trailing triangle: columns j >= k0 + pb, rows i >= j: Z(i, j) -= sum_q L(i, q) L(j, q) / d_q; tiles numbered down the
      // tile columns (consecutive threads: consecutive rows of the column-major Z)
      const int j0 = k0 + pb, mt = nz - j0, tt = (mt + 3) / 4, ntile = tt * (tt + 1) / 2;
#ifndef LIT_HOST
      // ... on the f64 matrix cores where the panel is full: a 16 x 16 block of the triangle per wavefront and step, its two
      // operands straight out of the staged panel ([pivot][row]: the MFMA's k along the pivots) -- two LDS reads per lane and
      // k-step instead of the tiles' eight per entry and panel (the tile pass was bound by them and by its global round trip)
      if (pb == 16) {
        const int tb = (mt + 15) / 16, nblk = tb * (tb + 1) / 2;
        const int lr = c.lane & 15, lk = c.lane >> 4;
        const double* below = sP + pb;                  // below[q * ldr + i]: Z(j0 + i, k0 + q)
        for (int t = c.wave; t < nblk; t += c.nw) {
          const double w2 = 2.0 * tb + 1.0;
          int bj = (int)((w2 - sqrt(w2 * w2 - 8.0 * t)) * 0.5);
          while (bj > 0 && bj * tb - bj * (bj - 1) / 2 > t) --bj;
          while ((bj + 1) * tb - (bj + 1) * bj / 2 <= t) ++bj;
          const int bi = bj + (t - (bj * tb - bj * (bj - 1) / 2));
          const int ii = 16 * bi + lr;                  // row of Z (the B operand's index, the result's column index)
          double old[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) { const int jj = 16 * bj + lk + 4 * r; old[r] = (ii < mt && jj <= ii) ? Z[(j0 + ii) + ldz * (j0 + jj)] : 0.0; }
          lit_v4d acc = lit_v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            const int q = 4 * kk + lk;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(below[q * ldr + 16 * bj + lr] * sD[q], below[q * ldr + ii], acc, 0, 0, 0);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) { const int jj = 16 * bj + lk + 4 * r; if (ii < mt && jj <= ii) Z[(j0 + ii) + ldz * (j0 + jj)] = old[r] - acc[r]; }
        }
      } else
#endif
      par_for32(c, ntile, [&](int t) {
        const double w2 = 2.0 * tt + 1.0;
        int bj = (int)((w2 - sqrt(w2 * w2 - 8.0 * t)) * 0.5);
        while (bj > 0 && bj * tt - bj * (bj - 1) / 2 > t) --bj;
        while ((bj + 1) * tt - (bj + 1) * bj / 2 <= t) ++bj;
        const int bi = bj + (t - (bj * tt - bj * (bj - 1) / 2)), i0 = 4 * bi, c0 = 4 * bj;      // rows i0 .. (>= columns c0 ..)
        int io[4], jo[4];
#pragma unroll
        for (int z = 0; z < 4; ++z) { io[z] = i0 + z < mt ? i0 + z : mt - 1; jo[z] = c0 + z < mt ? c0 + z : mt - 1; }
        double acc[16];
#pragma unroll
        for (int z = 0; z < 16; ++z) acc[z] = 0.0;
        for (int q = 0; q < pb; ++q) {
          double av[4], bv[4];
          const double dq = sD[q];
          const double* col = sP + q * ldr + pb;
#pragma unroll
          for (int z = 0; z < 4; ++z) { av[z] = col[io[z]]; bv[z] = col[jo[z]] * dq; }
#pragma unroll
          for (int zi = 0; zi < 4; ++zi)
#pragma unroll
            for (int zj = 0; zj < 4; ++zj) acc[zj * 4 + zi] += av[zi] * bv[zj];
        }
        double old[16];
#pragma unroll
        for (int zj = 0; zj < 4; ++zj)
#pragma unroll
          for (int zi = 0; zi < 4; ++zi) { const int hi = io[zi] > jo[zj] ? io[zi] : jo[zj], lo = io[zi] > jo[zj] ? jo[zj] : io[zi]; old[zj * 4 + zi] = Z[(j0 + hi) + ldz * (j0 + lo)]; }
#pragma unroll
        for (int zj = 0; zj < 4; ++zj)
#pragma unroll
          for (int zi = 0; zi < 4; ++zi) if (i0 + zi < mt && c0 + zj <= i0 + zi) Z[(j0 + i0 + zi) + ldz * (j0 + c0 + zj)] = old[zj * 4 + zi] - acc[zj * 4 + zi];
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
  par_map4(c, n1 * n1, [&](int x) -> double {
    const int lo = x / n1, hi = x - lo * n1;
    return lo > hi ? 0.0 : -Z[(nr + hi) + ldz * (nr + lo)];
  }, [&](int x, double v) {
    const int lo = x / n1, hi = x - lo * n1;
    if (lo <= hi) a.Lam[(long)hi * a.ldL + lo] = v;
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
// The compact route (default): the reference's compression as a PROJECTION, without the stack and without Q.
//
// What the filter sees of (T_H, r_n, R_n) = (Q_1^T H_o, Q_1^T r_o, Q_1^T R_o Q_1) (msckf.h:1351-1366) is
//     Lam^ = [T_H | r_n]^T R_n^-1 [T_H | r_n] = A^T Q_1 (Q_1^T R_o Q_1)^-1 Q_1^T A ,    A = [H_o | r_o],
// which depends on Q_1 only through its range S:  Lam^ = A^T Bs (Bs^T R_o Bs)^-1 Bs^T A  for ANY basis Bs of S.  And S is
//     span{ e_i : kept rows i < 15 }                  (the 15 zero IMU columns hand rows 0..14 through verbatim)
//   + span{ x'_c : camera columns c whose step REFLECTED }     (x'_c = column c of H_o with its first 15 rows zeroed:
//                                                               H_o = Q R, and R has no entry in a dropped row)
//   + span{ q_h = H_15 H_16 .. e_h : handed-through rows h that are kept }   (a dependent column in the middle of the sweep
//                                                               whose row still has entries to its right; rare -- typically the
//                                                               oldest camera seen by one or two tracks -- but any number of them)
// so the Householder sweep is needed only for its DECISIONS (which steps reflect, which rows are kept) and for the few q_h.
// It runs on a compressed representation: the rows that can become pivot rows (the first e = min(m, 15 + n)) explicitly
// (E), all rows from the pivot row down through their Gram matrix Gh, which a step downdates by the finished row of R:
//     c0 = E(p, k),  |tail|^2 = Gh(k, k) - c0^2,   R(p, j) = Gh(k, j) / beta          (reflecting step; beta^2 = Gh(k, k))
//     s_j = E(p, j) - R(p, j),  E(i > p, j) -= s_j v_i  (v_i = E(i, k) / (c0 - beta)),  Gh(j, l) -= R(p, j) R(p, l)
//     a skipped step (zero tail) hands row p through: R(p, :) = E(p, :), the same downdate of Gh
// -- no inner product over rows, no B = B0 Y coefficient matrix; Gh starts as k_gram's f64 H_o^T H_o minus the first 15 rows.
// The steps run in panels of 16 staged in LDS: the 16 x 16 core by one wavefront, then one thread per row below the panel
// (its 16 reflector entries) and per column to the right (its 16 entries of R and of s) -- each a short triangular recurrence
// -- then ONE rank-16 pass over the trailing parts of E and Gh in global memory.  When every row is explicit (m <= 15 + n:
// few tracks) the tails are summed exactly instead (the Gram matrix resolves a tail to ~1e-8 |column|^2 at best) by the plain
// step-by-step sweep over E.
//
// With the basis in hand:  Bs^T A has rows E(i, :), Gh0(c, :) = (A^T A)(c, :) - first 15 rows, R(h, :);
// Bs^T R_o Bs = v' Bs^T Bs + (u' - v') (G Bs)^T (G Bs), G = u-rows of blockdiag(A_j), and for a basis vector [t ; B0 y]
// (B0: the rows below the explicit ones)  G b = G_E t~ + H_u y,  t~ = t - E0 y,  H_u = the u-rows of every track's projected
// Jacobian (I - Q_f Q_f^T) H_x -- so that everything is made of three small matrices:  See = G_E^T G_E (block diagonal by
// track), Xe = G_E^T H_u, Gam = H_u^T H_u (n x n, independent of the basis A_j: computed for all trajectories of a launch
// on the matrix cores, kernels_literal.hip: k_lit_gamma).  For e_i and x'_c only the first 15 rows of t~ are non-zero and
// y is a unit vector: Gam enters through its (C, C) submatrix, no product.  Lam^ then comes out of the blocked elimination
// of Z = [[Bs^T R_o Bs, .], [(Bs^T A)^T, 0]] (information_from_rn).
LIT_HD long compact_ws_doubles(int n, int m_cap, int r_cap, int /*ldg*/) {
  const long n1 = n + 1, ec = 15 + n;
  return 2 * ec * n1 + ec * 2L * m_cap + 3 * n1 * n1 + 35 * n1 + ec * ec + ec * (long)n + 15L * n + 4 * ec
       + 3L * n * n + 2 * ec * (long)n + 2L * n * n + 55L * (2 * ec + m_cap) + 128;
}

// row r of Q_f = (I - V T V^T)(:, 0:3) of a track, Z3 = T V(0:3, :)^T
LIT_FN void qf_z3(const double* V, const double* T, double (&Z3)[9]) {
  for (int p = 0; p < 3; ++p)
    for (int cc = 0; cc < 3; ++cc) { double x = 0; for (int q = p; q < 3; ++q) x += T[p * 3 + q] * vf_at(V, cc, q); Z3[p * 3 + cc] = x; }
}
LIT_FN void qf_row(const double* V, const double (&Z3)[9], int r, double (&out)[3]) {
  const double v0 = vf_at(V, r, 0), v1 = vf_at(V, r, 1), v2 = vf_at(V, r, 2);
  for (int cc = 0; cc < 3; ++cc) out[cc] = (r == cc ? 1.0 : 0.0) - (v0 * Z3[cc] + v1 * Z3[3 + cc] + v2 * Z3[6 + cc]);
}

// The six rows a track contributes to Gam = Du - sum_j (B_j^T D_j + D_j^T B_j):  B = Q_f^T H_x (3 x n),
// D = Q_u^T H_xu - W B / 2  (Q_u, H_xu: the u-rows of Q_f and H_x; W = Q_u^T Q_u), from the track's V, T; rows6 is
// [6][ldc], only the columns of the track's cameras are written (the caller zeroed the rest).  Serial: the reference
// implementation of what k_lit_pre does with a wavefront per track.
template <class HT>
LIT_FN void gamma_rows(const Args<HT>& a, int t, double* rows6, long ldc) {
  const int M = a.M[t];
  const double* V = a.Vf + (long)t * 2 * a.m_cap * 3;
  const double* T = a.Tf + (long)t * 9;
  const HT* hx = a.Hx + (long)t * a.m_cap * 12;
  double Z3[9]; qf_z3(V, T, Z3);
  double W[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int o = 0; o < M; ++o) { double q0[3]; qf_row(V, Z3, 2 * o, q0); for (int x = 0; x < 3; ++x) for (int y = 0; y < 3; ++y) W[x * 3 + y] += q0[x] * q0[y]; }
  for (int o = 0; o < M; ++o) {
    double q0[3], q1[3]; qf_row(V, Z3, 2 * o, q0); qf_row(V, Z3, 2 * o + 1, q1);
    const int col = 6 * a.slots[first_obs(a, t) + o];
    for (int kk = 0; kk < 6; ++kk) {
      const double h0 = (double)hx[o * 12 + kk], h1 = (double)hx[o * 12 + 6 + kk];
      double b[3], cu[3];
      for (int x = 0; x < 3; ++x) { b[x] = q0[x] * h0 + q1[x] * h1; cu[x] = q0[x] * h0; }
      for (int x = 0; x < 3; ++x) {
        rows6[x * ldc + col + kk] = b[x];
        rows6[(3 + x) * ldc + col + kk] = cu[x] - 0.5 * (W[x * 3 + 0] * b[0] + W[x * 3 + 1] * b[1] + W[x * 3 + 2] * b[2]);
      }
    }
  }
}

// See / Xe for the explicit rows [i_lo, i_hi): See(i, i') = sum_o a_i[2o] a_i'[2o] inside a track, Xe(i, :) = (P D_u a_i)^T H_x
template <class HT>
LIT_FN void explicit_row_products(const Ctx& c, const Args<HT>& a, int e, int n, long ec, int i_lo, int i_hi, const int* topt, const double* At, double* See, double* Xe, double* G3) {
  const int nrow = i_hi - i_lo;
  if (nrow <= 0) return;
  const long mc2 = 2L * a.m_cap;
  par_for32(c, nrow * n, [&](int x) { const int j = x / nrow, i = i_lo + (x - j * nrow); Xe[i + ec * j] = 0.0; });
  // g3_i = Q_f^T (D_u a_i)
  par_for(c, nrow, [&](long ii) {
    const int i = i_lo + (int)ii, t = topt[i], M = a.M[t];
    const double* V = a.Vf + (long)t * 2 * a.m_cap * 3;
    double Z3[9]; qf_z3(V, a.Tf + (long)t * 9, Z3);
    double g[3] = {0, 0, 0};
    for (int o = 0; o < M; ++o) { double q0[3]; qf_row(V, Z3, 2 * o, q0); const double au = At[i * mc2 + 2 * o]; for (int x = 0; x < 3; ++x) g[x] += q0[x] * au; }
    for (int x = 0; x < 3; ++x) G3[i * 3 + x] = g[x];
  });
  barrier(c);
  par_for32(c, nrow * a.m_cap, [&](int x) {
    const int i = i_lo + x / a.m_cap, o = x % a.m_cap, t = topt[i];
    if (o >= a.M[t]) return;
    const double* V = a.Vf + (long)t * 2 * a.m_cap * 3;
    double Z3[9]; qf_z3(V, a.Tf + (long)t * 9, Z3);
    double q0[3], q1[3]; qf_row(V, Z3, 2 * o, q0); qf_row(V, Z3, 2 * o + 1, q1);
    const double* g = G3 + i * 3;
    const double w0 = At[i * mc2 + 2 * o] - (q0[0] * g[0] + q0[1] * g[1] + q0[2] * g[2]);
    const double w1 = -(q1[0] * g[0] + q1[1] * g[1] + q1[2] * g[2]);
    const HT* hx = a.Hx + (long)t * a.m_cap * 12 + o * 12;
    double hh[12];
    for (int kk = 0; kk < 12; ++kk) hh[kk] = (double)hx[kk];
    const int col = 6 * a.slots[first_obs(a, t) + o];
    for (int kk = 0; kk < 6; ++kk) Xe[i + ec * (col + kk)] = w0 * hh[kk] + w1 * hh[6 + kk];
  });
  // See: pairs of explicit rows of one track (rows of [i_lo, i_hi) against every explicit row of their track)
  par_for32(c, nrow * e, [&](int x) {
    const int i = i_lo + x / e, i2 = x % e;
    const int t = topt[i];
    double sacc = 0;
    if (topt[i2] == t) { const int M = a.M[t]; for (int o = 0; o < M; ++o) sacc += At[i * mc2 + 2 * o] * At[i2 * mc2 + 2 * o]; }
    See[i + ec * i2] = sacc; See[i2 + ec * i] = sacc;
  });
  barrier(c);
}

// One step of the sweep's per-column bookkeeping, shared by the three forms of the sweep


// 4 x 4 tiles of the lower triangle of an N x N index space, a thread per tile: f(i0, j0), i0 >= j0 multiples of four.  Where an
// entry is a short inner product of two staged vectors, a tile reads 8 vectors for 16 entries instead of 32 -- the entry-wise
// passes over the basis products and the Gram start were bound by their LDS reads (60 / 30 doubles per entry)
template <class F> LIT_FN void par_tiles_lower4(const Ctx& c, int N, F f) {
  const int nt = (N + 3) / 4;
  par_for(c, (long)nt * (nt + 1) / 2, [&](long el) {
    const int e = (int)el;
    int ti = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
    while (ti * (ti + 1) / 2 > e) --ti;
    while ((ti + 1) * (ti + 2) / 2 <= e) ++ti;
    f(4 * ti, 4 * (e - ti * (ti + 1) / 2));
  });
}
// C(i, j) = sum_(klo(i) <= k < khi(j0)) fa(i, k) fb(k, j), i < M, j < N: a thread per (i, four consecutive j = j0 .. j0 + 3), i fastest
// (fa coalesced over i, fb the same address in every lane of an i-run), four k in flight -- the plain loop per element waits
// out a memory round trip per term (Gram of the start x column operations of a 160-step prefix: 0.6 ms)
template <class FL, class FH, class FA, class FB, class ST>
LIT_FN void par_gemm4(const Ctx& c, int M, int N, FL klo, FH khi, FA fa, FB fb, ST st) {
  const int nq = (N + 3) / 4;
  par_for(c, (long)M * nq, [&](long x) {
    const int q = (int)(x / M), i = (int)(x - (long)q * M), j0 = 4 * q;
    const int k1 = khi(j0);
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    int k = klo(i);
    for (; k + 4 <= k1; k += 4) {
      double av[4], bv[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        av[u] = fa(i, k + u);
#pragma unroll
        for (int v = 0; v < 4; ++v) bv[u][v] = fb(k + u, j0 + v);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[v] += av[u] * bv[u][v];
    }
    for (; k < k1; ++k) {
      const double av = fa(i, k);
#pragma unroll
      for (int v = 0; v < 4; ++v) acc[v] += av * fb(k, j0 + v);
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) if (j0 + v < N) st(i, j0 + v, acc[v]);
  });
}

// The reflectors of the steps before k_h, last one first, on [t ; y] (q_h = [t ; B0 y]); a wavefront per handed-through row.
// t and y live in memory: every step is load -> reduce -> store -> next load (3.4 us a step: 0.55 ms for a row 160 steps in)
template <class HT>
LIT_FN void reflector_chain_mem(const Ctx& c, const Args<HT>& a, int ah, int kh, int kmax, int e, int n, long ec, const double* E, const double* Gv, const double* Yk, const int* refl, double* t, double* y) {
  (void)ah;
  lane_for(c, 0, e, [&](long i) { t[i] = i == 15 + kh ? 1.0 : 0.0; });
  lane_for(c, 0, n, [&](long l) { y[l] = 0.0; });
  for (int j = kh - 1; j >= 0; --j) {
    if (!refl[j]) continue;
    const int pj = 15 + j;
    const double* ej = E + ec * j;
    const double d1 = wave_sum_range(c, 0, e, [&](long i) { return i < pj ? 0.0 : (i == pj ? t[i] : ej[i] * t[i]); });
    const double d2 = wave_sum_range(c, 0, kmax, [&](long l) { return Gv[l + (long)n * j] * y[l]; });
    const double al = a.tau[j] * (d1 + d2);
    lane_for(c, 0, e, [&](long i) { if (i >= pj) t[i] -= al * (i == pj ? 1.0 : ej[i]); });
    lane_for(c, 0, kmax, [&](long l) { if (l <= j) y[l] -= al * Yk[l + (long)n * j]; });
  }
}
#ifndef LIT_HOST
// ... in registers (lane l holds entries l, l + 64, l + 128, l + 192), the next reflector's columns and tau loaded while this
// one's dot product reduces, the step after that looked up in the list of reflected steps (rl, ascending, nrl entries) one
// iteration ahead: one reduction per step is what is left of the chain.  (Walking refl[] for the next reflected step put a
// dependent load in front of every step's loads: 1 us a step.)
template <class HT>
LIT_FN void reflector_chain_regs(const Ctx& c, const Args<HT>& a, int kh, int kmax, int e, int n, long ec, const double* E, const double* Gv, const double* Yk, const int* rl, int nrl, double* t, double* y) {
  double tr[4], yr[4], ev[4], gv[4], yk[4], evn[4], gvn[4], ykn[4], tau = 0, taun = 0;
#pragma unroll
  for (int u = 0; u < 4; ++u) { tr[u] = (c.lane + 64 * u) == 15 + kh ? 1.0 : 0.0; yr[u] = 0.0; ev[u] = gv[u] = yk[u] = evn[u] = gvn[u] = ykn[u] = 0.0; }
  auto load = [&](int j, double (&ev_)[4], double (&gv_)[4], double (&yk_)[4], double& tau_) {
    const int pj = 15 + j;
    tau_ = a.tau[j];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = c.lane + 64 * u;
      ev_[u] = (i < e && i > pj) ? E[i + ec * j] : (i == pj ? 1.0 : 0.0);
      gv_[u] = i < kmax ? Gv[i + (long)n * j] : 0.0;
      yk_[u] = i <= j ? Yk[i + (long)n * j] : 0.0;
    }
  };
  int idx = (int)wave_sum_range(c, 0, nrl, [&](long q) { return rl[q] < kh ? 1.0 : 0.0; }) - 1;    // reflected steps before k_h: rl[0 .. idx]
  int j = idx >= 0 ? rl[idx] : -1, jn = idx >= 1 ? rl[idx - 1] : -1;
  if (j >= 0) load(j, ev, gv, yk, tau);
  while (j >= 0) {
    const int jnn = idx >= 2 ? rl[idx - 2] : -1;
    if (jn >= 0) load(jn, evn, gvn, ykn, taun);
    double d = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) d += ev[u] * tr[u] + gv[u] * yr[u];
    const double al = tau * wave_sum(d);
#pragma unroll
    for (int u = 0; u < 4; ++u) { tr[u] -= al * ev[u]; yr[u] -= al * yk[u]; ev[u] = evn[u]; gv[u] = gvn[u]; yk[u] = ykn[u]; }
    tau = taun; j = jn; jn = jnn; --idx;
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) { const int i = c.lane + 64 * u; if (i < e) t[i] = tr[u]; if (i < n) y[i] = yr[u]; }
}
#endif
template <class HT>
LIT_FN void reflector_chain(const Ctx& c, const Args<HT>& a, int ah, int kh, int kmax, int e, int n, long ec, const double* E, const double* Gv, const double* Yk, const int* refl, const int* rl, int nrl, double* t, double* y) {
#ifndef LIT_HOST
  if (e <= 256 && n <= 256) { reflector_chain_regs(c, a, kh, kmax, e, n, ec, E, Gv, Yk, rl, nrl, t, y); return; }
#endif
  (void)rl; (void)nrl;
  reflector_chain_mem(c, a, ah, kh, kmax, e, n, ec, E, Gv, Yk, refl, t, y);
}

// sum_(k < len) pa[k sa] pb[k sb], four terms in flight (the plain loop waits out a memory round trip per term)
LIT_FN double dot_strided(const double* pa, long sa, const double* pb, long sb, int len) {
  double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  int k = 0;
  for (; k + 4 <= len; k += 4) {
    const double a0 = pa[k * sa], a1 = pa[(k + 1) * sa], a2 = pa[(k + 2) * sa], a3 = pa[(k + 3) * sa];
    const double b0 = pb[k * sb], b1 = pb[(k + 1) * sb], b2 = pb[(k + 2) * sb], b3 = pb[(k + 3) * sb];
    s0 += a0 * b0; s1 += a1 * b1; s2 += a2 * b2; s3 += a3 * b3;
  }
  for (; k < len; ++k) s0 += pa[k * sa] * pb[k * sb];
  return (s0 + s1) + (s2 + s3);
}
// Ph = See t~ + Xe y and Qh = Xe^T t~ + Gam y for the nh handed-through rows (t~ = Th, y = Yh) WITHOUT the full See / Xe (a pass
// over every explicit row: 0.35 ms on every launch that had one such row among its trajectories).  Per track t that owns
// explicit rows, with g = sum_i a_i[u-rows] t~(i) (a_i = row i of A_t^T), z = H_x y, P = I - Q_f Q_f^T:
//     (See t~)(i) = a_i[u]^T g,   (Xe y)(i) = a_i[u]^T (P z)[u],   Xe^T t~ = H_x^T P [g on the u-rows ; 0 on the v-rows]
// in passes over (row, observation of those tracks) pairs -- a workgroup's worth of threads each -- eight rows at a time.
// GS: [6 nEc] rows 2o, 2o + 1 of Q_f | eight x [6 nEc] g, z0 -> (P u)_2o, z1 -> (P u)_2o+1, Q_f^T u and Q_f^T z at the
// track's first two observations | [nEc] ints: track, slot (when the staging area is too small for them); nEc = 2 ec + m_cap observations at most (a track of M has 2 M - 3 >= M / 2 rows)
template <class HT>
LIT_FN void extras_products(const Ctx& c, const Args<HT>& a, int e, int n, long ec, int nh, int kmax, const int* topt, const double* At, const double* Th, const double* Yh, double* Ph, double* Qh, double* GS) {
  const long mc2 = 2L * a.m_cap, nEc = 2 * ec + a.m_cap;
  const int t0 = topt[0], t1 = topt[e - 1], ob = a.obs0[t0], nE = a.obs0[t1] + a.M[t1] - ob;
  double* QF = GS;
  double* CH = GS + 6 * nEc;
  // observation -> track, slot, first observation of the track: in the staging area when it fits (read in every inner loop below)
  int* obs_t = 3 * nEc <= 2L * c.lds_doubles ? reinterpret_cast<int*>(c.lds) : reinterpret_cast<int*>(GS + 54 * nEc);
  int* obs_s = obs_t + nEc;
  par_for(c, t1 - t0 + 1, [&](long tl) {
    const int t = t0 + (int)tl;
    if (!(a.status[t] & a.inc_bit)) return;
    const int og0 = a.obs0[t] - ob, so = first_obs(a, t);
    for (int o = 0; o < a.M[t]; ++o) { obs_t[og0 + o] = t; obs_s[og0 + o] = a.slots[so + o]; }
  });
  barrier(c);
  par_for(c, nE, [&](long og) {
    const int t = obs_t[og], o = (int)og - (a.obs0[t] - ob);
    const double* V = a.Vf + (long)t * 2 * a.m_cap * 3;
    double Z3[9], q0[3], q1[3]; qf_z3(V, a.Tf + (long)t * 9, Z3);
    qf_row(V, Z3, 2 * o, q0); qf_row(V, Z3, 2 * o + 1, q1);
    for (int k = 0; k < 3; ++k) { QF[6 * og + k] = q0[k]; QF[6 * og + 3 + k] = q1[k]; }
  });
  barrier(c);
  for (int h0 = 0; h0 < nh; h0 += 8) {
    const int nc = nh - h0 < 8 ? nh - h0 : 8;
    // g, z per (row, observation)
    par_for(c, (long)nc * nE, [&](long x) {
      const int hc = (int)(x / nE), og = (int)(x - (long)hc * nE), ah = h0 + hc;
      const int t = obs_t[og], o = og - (a.obs0[t] - ob), r0 = a.row0[t];
      int r1 = r0 + 2 * a.M[t] - 3; r1 = r1 < e ? r1 : e;
      const double* tt = Th + ec * ah; const double* yy = Yh + (long)n * ah;
      const double g = dot_strided(At + r0 * mc2 + 2 * o, mc2, tt + r0, 1, r1 - r0);
      const HT* hx = a.Hx + ((long)t * a.m_cap + o) * 12;
      const int col = 6 * obs_s[og];
      double z0 = 0, z1 = 0;
      for (int kk = 0; kk < 6; ++kk) { const double yv = yy[col + kk]; z0 += (double)hx[kk] * yv; z1 += (double)hx[6 + kk] * yv; }
      double* ch = CH + 6 * nEc * hc;
      ch[og] = g; ch[nEc + og] = z0; ch[2 * nEc + og] = z1;
    });
    barrier(c);
    // Q_f^T u, Q_f^T z per (row, track): a wavefront each
    wave_for(c, 0, (long)nc * (t1 - t0 + 1), [&](long x) {
      const int hc = (int)(x / (t1 - t0 + 1)), t = t0 + (int)(x - (long)hc * (t1 - t0 + 1));
      if (!(a.status[t] & a.inc_bit)) return;
      const int og0 = a.obs0[t] - ob, M = a.M[t];
      double* ch = CH + 6 * nEc * hc;
      double cu[3], cz[3];                         // (the stores after all six sums: the loads of the six are then issued together)
      for (int k = 0; k < 3; ++k) {
        cu[k] = wave_sum_range(c, 0, M, [&](long o) { return QF[6 * (og0 + o) + k] * ch[og0 + o]; });
        cz[k] = wave_sum_range(c, 0, M, [&](long o) { return QF[6 * (og0 + o) + k] * ch[nEc + og0 + o] + QF[6 * (og0 + o) + 3 + k] * ch[2 * nEc + og0 + o]; });
      }
      if (first_lane(c)) for (int k = 0; k < 3; ++k) { ch[(3 + k) * nEc + og0] = cu[k]; ch[(3 + k) * nEc + og0 + 1] = cz[k]; }
    });
    barrier(c);
    // (P u) rows 2o, 2o + 1 and g + (P z)_2o
    par_for(c, (long)nc * nE, [&](long x) {
      const int hc = (int)(x / nE), og = (int)(x - (long)hc * nE);
      const int t = obs_t[og], og0 = a.obs0[t] - ob;
      double* ch = CH + 6 * nEc * hc;
      double su = 0, sv = 0, sz = 0;
      for (int k = 0; k < 3; ++k) {
        const double cu = ch[(3 + k) * nEc + og0], cz = ch[(3 + k) * nEc + og0 + 1];
        su += QF[6 * og + k] * cu; sv += QF[6 * og + 3 + k] * cu; sz += QF[6 * og + k] * cz;
      }
      const double g = ch[og], z0 = ch[nEc + og];
      ch[nEc + og] = g - su; ch[2 * nEc + og] = -sv; ch[og] = g + z0 - sz;
    });
    barrier(c);
    par_for(c, (long)nc * e, [&](long x) {
      const int hc = (int)(x / e), i = (int)(x - (long)hc * e), ah = h0 + hc;
      const int t = topt[i], og0 = a.obs0[t] - ob, M = a.M[t];
      const double* ch = CH + 6 * nEc * hc;
      Ph[i + ec * ah] = dot_strided(At + i * mc2, 2, ch + og0, 1, M);
    });
    par_for(c, (long)nc * n, [&](long x) {
      const int hc = (int)(x / n), cc = (int)(x - (long)hc * n), ah = h0 + hc, sl = cc / 6, kk = cc - 6 * sl;
      const double* ch = CH + 6 * nEc * hc;
      double sacc = 0;
      for (int og = 0; og < nE; ++og) {
        if (obs_s[og] != sl) continue;
        const int t = obs_t[og], o = og - (a.obs0[t] - ob);
        const HT* hx = a.Hx + ((long)t * a.m_cap + o) * 12;
        sacc += (double)hx[kk] * ch[nEc + og] + (double)hx[6 + kk] * ch[2 * nEc + og];
      }
      const double* yy = Yh + (long)n * ah;
      const int lsplit = cc < kmax ? cc : kmax;     // Gam(cc, l): row cc up to the diagonal, column cc below it
      sacc += dot_strided(a.Gam + (long)cc * a.ldGam, 1, yy, 1, lsplit);
      if (kmax > lsplit) sacc += dot_strided(a.Gam + (long)lsplit * a.ldGam + cc, a.ldGam, yy + lsplit, 1, kmax - lsplit);
      Qh[cc + (long)n * ah] = sacc;
    });
    barrier(c);
  }
}


#ifndef LIT_HOST
// The 16 x 16 core of the sweep's panels on ONE wavefront with the block in REGISTERS: lane j < 16 holds column j of the panel's
// rows and of its Gram block (sixteen entries each), a pivot's values reach the other lanes by v_readlane -- no LDS round trip
// and no wave_sync between the pivots (the LDS form: ~1 800 cycles a pivot, three hand-overs through LDS each; this one ~1 000,
// the rsqrt / rcp chain and thirty broadcasts).  Same arithmetic, entry by entry, as the loop it replaces (sweep_gram_blocked
// keeps that one for the host build and for panels narrower than sixteen).  Not inlined: inside the phase kernel its 64
// registers of block pushed the panel loops' addresses into scratch memory.  (The elimination's diagonal block the same
// way: no gain -- that routine's panels are bound by their staging and trailing passes -- and left as it was.)
// C(j, r) = sEC[j * lde + r], G(r, j) = sG[r * 16 + j] (both triangles); results back where the loop leaves them
__device__ __attribute__((noinline)) void sweep_core_regs(const Ctx& c, double* sEC, long lde, double* sG, double* sS, double* sDn, double* sBi, double* sTau, double* sRf, double* sCnt, const double* sLd, int pb, double t2) {
  const int j = c.lane & 15;
  const bool mine = c.lane < 16 && j < pb;
  double Cc[16], Gc[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) { Cc[r] = (mine && r < pb) ? sEC[j * lde + r] : 0.0; Gc[r] = (mine && r < pb) ? sG[r * 16 + j] : 0.0; }
  const double ldj = mine ? sLd[j] : 0.0;
  int nref = 0, nskt = 0;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    if (q < pb) {
      const double c0 = wave_bcast(Cc[q], q), gq0 = wave_bcast(Gc[q], q), ldq = wave_bcast(ldj, q);
      const double gq = gq0 > 0.0 ? gq0 : 0.0;
      double tail2 = gq - c0 * c0; tail2 = tail2 > 0.0 ? tail2 : 0.0;
      double zero2 = t2 * ldq; zero2 = zero2 > 2.2250738585072014e-308 ? zero2 : 2.2250738585072014e-308;
      const bool reflect = tail2 > zero2;
      double sq = 0.0;
      if (reflect) {
        ++nref;
        const double g2 = c0 * c0 + tail2, rs = lit_rsqrt(g2);
        const double binv = c0 >= 0.0 ? -rs : rs, beta = g2 * binv;
        const double dn = lit_rcp(c0 - beta);
        if (c.lane == 0) { sDn[q] = dn; sBi[q] = binv; sTau[q] = (beta - c0) * binv; sRf[q] = 1.0; }
        if (mine && j >= q) { const double rr = j == q ? beta : Gc[q] * binv; sq = Cc[q] - rr; Cc[q] = rr; }
        if (c.lane == q) {
#pragma unroll
          for (int r = q + 1; r < 16; ++r) Cc[r] *= dn;
        }
      } else {
        if (tail2 > 2.2250738585072014e-308) ++nskt;
        if (c.lane == 0) { sDn[q] = 0.0; sBi[q] = 0.0; sTau[q] = 0.0; sRf[q] = 0.0; }
        if (c.lane == q) {
#pragma unroll
          for (int r = q + 1; r < 16; ++r) Cc[r] = 0.0;
        }
      }
      if (mine && j >= q) sS[q * 16 + j] = sq;
      const double rown = Cc[q];                     // R(p0 + q, k0 + j): row q of this lane's column, final from here on
#pragma unroll
      for (int r = q + 1; r < 16; ++r) {
        const double vr = wave_bcast(Cc[r], q);      // column q, row r (the reflector's entry, 0 when the step was skipped)
        const double rr = wave_bcast(rown, r);       // row q of column r
        if (j > q) { Cc[r] -= vr * sq; Gc[r] -= rr * rown; }
      }
    }
  }
  if (mine) {
#pragma unroll
    for (int r = 0; r < 16; ++r) if (r < pb) { sEC[j * lde + r] = Cc[r]; sG[r * 16 + j] = Gc[r]; }
  }
  if (c.lane == 0) { sCnt[0] = (double)nref; sCnt[1] = (double)nskt; }
}
#endif

struct SweepOut { int n_reflect, n_skip_tol; };

// ---- the sweep when rows exist below the explicit ones, step by step (the definition of the blocked form; runs when the
// staging area is too small for a panel)
template <class HT>
LIT_FN SweepOut sweep_gram_steps(const Ctx& c, const Args<HT>& a, int e, int n, int msteps, long ec, double* E, double* Gh, const double* Ld, double* Ac, double* dnv, int* refl, double* sv) {
  const int n1 = n + 1;
  const double tol2 = a.tol * a.tol, t2 = tol2 < 1e-7 ? 1e-7 : tol2;
  SweepOut so = {0, 0};
  double* rr = a.tau + n1;                         // [n1] the finished row of R
  for (int k = 0; k < msteps; ++k) {
    const int p = 15 + k;
    double* ek = E + ec * k;
    const double c0 = ek[p], gkk = Gh[k + (long)n1 * k] > 0.0 ? Gh[k + (long)n1 * k] : 0.0;
    double tail2 = gkk - c0 * c0; tail2 = tail2 > 0.0 ? tail2 : 0.0;
    double zero2 = t2 * Ld[k]; zero2 = zero2 > 2.2250738585072014e-308 ? zero2 : 2.2250738585072014e-308;
    const bool reflect = tail2 > zero2;
    barrier(c);                                    // everybody has read column k's state
    double beta = 0, dn = 0, binv = 0;
    if (reflect) {
      ++so.n_reflect;
      beta = sqrt(c0 * c0 + tail2); if (c0 >= 0.0) beta = -beta;
      dn = 1.0 / (c0 - beta); binv = 1.0 / beta;
      if (first_thread(c)) { a.tau[k] = (beta - c0) / beta; dnv[k] = dn; refl[k] = 1; }
    } else {
      if (tail2 > 2.2250738585072014e-308) ++so.n_skip_tol;
      if (first_thread(c)) { a.tau[k] = 0.0; dnv[k] = 0.0; refl[k] = 0; }
    }
    // finished row p of R (columns k..n), s_j, the reflector's entries in place of column k's tail
    par_for(c, n1 - k, [&](long jj) {
      const int j = k + (int)jj;
      const double ep = E[p + ec * j];
      const double r = reflect ? (j == k ? beta : Gh[j + (long)n1 * k] * binv) : ep;
      rr[j] = r;
      sv[j] = reflect ? ep - r : 0.0;
      if (j > k) Ac[j + (long)n1 * k] = reflect ? (ep - r) * dn : 0.0;
    });
    par_for(c, e - (p + 1), [&](long i) { ek[p + 1 + i] = reflect ? ek[p + 1 + i] * dn : 0.0; });
    barrier(c);
    // E(i > p, j > k) -= s_j v_i ;  E(p, j) = R(p, j) ;  Gh(j, l) -= R(p, j) R(p, l) for l >= j > k (lower triangle, column-major)
    if (reflect) {
      const int ni = e - (p + 1), nj = n1 - (k + 1);
      par_for(c, (long)ni * nj, [&](long x) {
        const int jj = (int)(x / ni), i = p + 1 + (int)(x - (long)jj * ni), j = k + 1 + jj;
        E[i + ec * j] -= sv[j] * ek[i];
      });
    }
    {
      const int nj = n1 - (k + 1);
      par_for(c, (long)nj * nj, [&](long x) {
        const int lj = (int)(x / nj), li = (int)(x - (long)lj * nj);
        if (li < lj) return;
        const int j = k + 1 + li, l = k + 1 + lj;
        Gh[j + (long)n1 * l] -= rr[j] * rr[l];
      });
    }
    barrier(c);
    par_for(c, n1 - k, [&](long jj) { const int j = k + (int)jj; E[p + ec * j] = rr[j]; });
  }
  barrier(c);
  return so;
}

// ---- the same steps in panels of PB <= 16 staged in LDS
template <class HT>
LIT_FN SweepOut sweep_gram_blocked(const Ctx& c, const Args<HT>& a, int e, int n, int msteps, long ec, double* E, double* Gh, const double* Ld, double* Ac, double* dnv, int* refl, double* sv) {
  const int n1 = n + 1;
  int PB = 16;
  const long lde = (e - 15 + 4) | 1;                 // the panel's columns [q][row]: rows contiguous (a thread's four rows are one 32-byte read, no bank conflicts)
  auto need = [&](int pb) -> long { return lde * pb + 2L * pb * n1 + 2L * pb * pb + 9L * pb + 16; };
  while (PB > 2 && need(PB) > c.lds_doubles) PB >>= 1;
  if (need(PB) > c.lds_doubles) return sweep_gram_steps(c, a, e, n, msteps, ec, E, Gh, Ld, Ac, dnv, refl, sv);
  double* sEC = c.lds;                               // [PB][lde]: E(p0 + r, k0 + q) at sEC[q * lde + r]; rows r < PB x the PB columns are the panel's core
  double* sEP = sEC + lde * PB;                      // [PB][n1]: E(p0 + q, k0 + PB + jj) -> s_q(j)
  double* sGP = sEP + (long)PB * n1;                 // [PB][n1]: Gh(k0 + q, k0 + PB + jj) -> R(p0 + q, j)
  double* sG = sGP + (long)PB * n1;                  // [PB][PB] Gram block of the panel's columns (both triangles)
  double* sS = sG + PB * PB;                         // [PB][PB] s_q(k0 + q'), q' > q
  double* sDn = sS + PB * PB;                        // [PB] 1 / (c0 - beta)   (0: the step did not reflect)
  double* sBi = sDn + PB;                            // [PB] 1 / beta
  double* sTau = sBi + PB;                           // [PB]
  double* sRf = sTau + PB;                           // [PB] 1.0 reflected / 0.0 skipped
  double* sCnt = sRf + PB;                           // [2] counts of the panel
  double* sLd = sCnt + 2;                            // [PB] squared norms of the panel's columns (a global load per step sat on the core's chain)
  const double tol2 = a.tol * a.tol, t2 = tol2 < 1e-7 ? 1e-7 : tol2;
  SweepOut so = {0, 0};
  long long tprev = tick_now(c);       // phase timers over the panels: slots 12 stage, 13 core, 14 rows / columns, 15 results + trailing pass
  for (int k0 = 0; k0 < msteps; k0 += PB) {
    const int pb = msteps - k0 < PB ? msteps - k0 : PB, p0 = 15 + k0;
    const int nrow = e - p0;                         // rows p0 .. e-1
    const int ntr = n1 - (k0 + pb);                  // trailing columns k0 + pb .. n
    // ---- stage
    par_for32(c, nrow * pb, [&](int x) { const int q = x / nrow, r = x - q * nrow; sEC[(q) * lde + (r)] = E[(p0 + r) + ec * (k0 + q)]; });
    par_for32(c, ntr * pb, [&](int x) {
      const int q = x / ntr, jj = x - q * ntr, j = k0 + pb + jj;
      sEP[(long)q * n1 + jj] = E[(p0 + q) + ec * j];
      sGP[(long)q * n1 + jj] = Gh[j + (long)n1 * (k0 + q)];
    });
    par_for32(c, pb * pb, [&](int x) {
      const int q = x / pb, q2 = x - q * pb, hi = q > q2 ? q : q2, lo = q > q2 ? q2 : q;
      sG[q * PB + q2] = Gh[(k0 + hi) + (long)n1 * (k0 + lo)];
      if (q2 == 0) sLd[q] = Ld[k0 + q];
    });
    barrier(c);
    tick_acc(c, 12, tprev);
    // ---- core: the panel's columns against each other, one wavefront
#ifndef LIT_HOST
    if (PB == 16) { if (first_wave(c)) sweep_core_regs(c, sEC, lde, sG, sS, sDn, sBi, sTau, sRf, sCnt, sLd, pb, t2); }
    else
#endif
    if (first_wave(c)) {
      int nref = 0, nskt = 0;
      for (int q = 0; q < pb; ++q) {
        const double c0 = sEC[(q) * lde + (q)], gq = sG[q * PB + q] > 0.0 ? sG[q * PB + q] : 0.0;
        double tail2 = gq - c0 * c0; tail2 = tail2 > 0.0 ? tail2 : 0.0;
        double zero2 = t2 * sLd[q]; zero2 = zero2 > 2.2250738585072014e-308 ? zero2 : 2.2250738585072014e-308;
        const bool reflect = tail2 > zero2;
        wave_sync(c);                                // every lane has read the step's inputs
        if (reflect) {
          ++nref;
          const double g2 = c0 * c0 + tail2, rs = lit_rsqrt(g2);
          const double binv = c0 >= 0.0 ? -rs : rs, beta = g2 * binv;      // beta = -sign(c0) sqrt(g2)
          const double dn = lit_rcp(c0 - beta);
          if (first_lane(c)) { sDn[q] = dn; sBi[q] = binv; sTau[q] = (beta - c0) * binv; sRf[q] = 1.0; }
          lane_for(c, q, pb, [&](long q2) {
            const double r = q2 == q ? beta : sG[q * PB + q2] * binv;
            sS[q * PB + q2] = sEC[(q2) * lde + (q)] - r;
            sEC[(q2) * lde + (q)] = r;
          });
          lane_for(c, q + 1, pb, [&](long r) { sEC[q * lde + r] *= dn; });
        } else {
          if (tail2 > 2.2250738585072014e-308) ++nskt;
          if (first_lane(c)) { sDn[q] = 0.0; sBi[q] = 0.0; sTau[q] = 0.0; sRf[q] = 0.0; }
          lane_for(c, q, pb, [&](long q2) { sS[q * PB + q2] = 0.0; });
          lane_for(c, q + 1, pb, [&](long r) { sEC[q * lde + r] = 0.0; });
        }
        wave_sync(c);
        lane_for(c, 0, 64, [&](long ln) {              // lane = (row offset mod 4, column offset)
          const int q2 = q + 1 + ((int)ln & 15);
          if (q2 >= pb) return;
          for (int r = q + 1 + ((int)ln >> 4); r < pb; r += 4) {
            sEC[q2 * lde + r] -= sEC[q * lde + r] * sS[q * PB + q2];
            sG[r * PB + q2] -= sEC[r * lde + q] * sEC[q2 * lde + q];
          }
        });
        wave_sync(c);
      }
      if (first_lane(c)) { sCnt[0] = (double)nref; sCnt[1] = (double)nskt; }
    }
    barrier(c);
    tick_acc(c, 13, tprev);
    so.n_reflect += (int)sCnt[0]; so.n_skip_tol += (int)sCnt[1];
    // ---- one thread per row below the panel (its reflector entries) and per column to the right (its entries of R and s)
    panel_rows_sweep(c, sEC, lde, sS, PB, sDn, pb, pb, nrow);
    panel_cols_sweep(c, sEP, sGP, n1, sEC, lde, sRf, sBi, pb, ntr);
    barrier(c);
    tick_acc(c, 14, tprev);
    // ---- results of the panel, and ONE pass over the trailing parts: E(i, j) -= sum_q v_q(i) s_q(j),  Gh(j, l) -= sum_q R_q(j) R_q(l)
    par_for32(c, pb, [&](int q) { a.tau[k0 + q] = sTau[q]; dnv[k0 + q] = sDn[q]; refl[k0 + q] = sRf[q] != 0.0 ? 1 : 0; });
    par_for32(c, nrow * pb, [&](int x) { const int q = x / nrow, r = x - q * nrow; E[(p0 + r) + ec * (k0 + q)] = sEC[(q) * lde + (r)]; });
    par_for32(c, ntr * pb, [&](int x) {
      const int q = x / ntr, jj = x - q * ntr, j = k0 + pb + jj;
      E[(p0 + q) + ec * j] = sGP[(long)q * n1 + jj];
      Ac[j + (long)n1 * (k0 + q)] = sEP[(long)q * n1 + jj] * sDn[q];
    });
    par_for32(c, pb * pb, [&](int x) { const int q = x / pb, q2 = x - q * pb; if (q2 > q) Ac[(k0 + q2) + (long)n1 * (k0 + q)] = sS[q * PB + q2] * sDn[q]; });
    {
      const int ni = nrow - pb, ti = (ni + 3) / 4, tj = (ntr + 3) / 4, ntE = ti * tj, ntG = tj * (tj + 1) / 2;
#ifndef LIT_HOST
      // ... on the f64 matrix cores where the panel is full (as the elimination's trailing pass, information_from_rn): 16 x 16
      // blocks per wavefront, operands straight out of the staged panel -- [step][row] and [step][column] are the MFMA's k-major
      // layouts as they stand; the result's contiguous index (the row of E, the row of Gh) rides on the B operand
      if (pb == 16) {
        const int bE = (ni + 15) / 16, bT = (ntr + 15) / 16, nbE = bE * bT, nbG = bT * (bT + 1) / 2;
        const int lr = c.lane & 15, lk = c.lane >> 4;
        for (int t = c.wave; t < nbE + nbG; t += c.nw) {
          const bool isE = t < nbE;
          int bi, bj;
          if (isE) { bj = t / bE; bi = t - bj * bE; }
          else {
            const int tg = t - nbE;
            const double w2 = 2.0 * bT + 1.0;
            bj = (int)((w2 - sqrt(w2 * w2 - 8.0 * tg)) * 0.5);
            while (bj > 0 && bj * bT - bj * (bj - 1) / 2 > tg) --bj;
            while ((bj + 1) * bT - (bj + 1) * bj / 2 <= tg) ++bj;
            bi = bj + (tg - (bj * bT - bj * (bj - 1) / 2));
          }
          const int ii = 16 * bi + lr, ilim = isE ? ni : ntr;
          double* dst = isE ? E + (p0 + pb + ii) + ec * (k0 + pb) : Gh + (k0 + pb + ii) + (long)n1 * (k0 + pb);
          const long ldd = isE ? ec : (long)n1;
          double old[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) { const int jj = 16 * bj + lk + 4 * r; old[r] = (ii < ilim && jj < ntr && (isE || jj <= ii)) ? dst[ldd * jj] : 0.0; }
          const double* pa = isE ? sEP + 16 * bj + lr : sGP + 16 * bj + lr;           // A: the result's column index (the step's s / R entries)
          const double* pb_ = isE ? sEC + pb + ii : sGP + ii;                       // B: the result's row index
          const long sa = n1, sb = isE ? lde : (long)n1;
          lit_v4d acc = lit_v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) { const int q = 4 * kk + lk; acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[sa * q], pb_[sb * q], acc, 0, 0, 0); }
#pragma unroll
          for (int r = 0; r < 4; ++r) { const int jj = 16 * bj + lk + 4 * r; if (ii < ilim && jj < ntr && (isE || jj <= ii)) dst[ldd * jj] = old[r] - acc[r]; }
        }
      } else
#endif
      par_for32(c, ntE + ntG, [&](int x) {
        double acc[16];
#pragma unroll
        for (int z = 0; z < 16; ++z) acc[z] = 0.0;
        if (x < ntE) {
          const int bj = x / ti, bi = x - bj * ti, i0 = 4 * bi, j0 = 4 * bj;
          int io[4], jo[4];
#pragma unroll
          for (int z = 0; z < 4; ++z) { io[z] = i0 + z < ni ? i0 + z : ni - 1; jo[z] = j0 + z < ntr ? j0 + z : ntr - 1; }
          for (int q = 0; q < pb; ++q) {
            double av[4], bv[4];
#pragma unroll
            for (int z = 0; z < 4; ++z) { av[z] = sEC[(q) * lde + ((pb + io[z]))]; bv[z] = sEP[(long)q * n1 + jo[z]]; }
#pragma unroll
            for (int zi = 0; zi < 4; ++zi)
#pragma unroll
              for (int zj = 0; zj < 4; ++zj) acc[zj * 4 + zi] += av[zi] * bv[zj];
          }
          double old[16];
#pragma unroll
          for (int zj = 0; zj < 4; ++zj)
#pragma unroll
            for (int zi = 0; zi < 4; ++zi) old[zj * 4 + zi] = E[(p0 + pb + io[zi]) + ec * (k0 + pb + jo[zj])];
#pragma unroll
          for (int zj = 0; zj < 4; ++zj)
#pragma unroll
            for (int zi = 0; zi < 4; ++zi) if (i0 + zi < ni && j0 + zj < ntr) E[(p0 + pb + i0 + zi) + ec * (k0 + pb + j0 + zj)] = old[zj * 4 + zi] - acc[zj * 4 + zi];
        } else {
          // tiles of the lower triangle numbered down the tile columns (consecutive threads: consecutive rows of the column-major Gh)
          const int t = x - ntE;
          const double w2 = 2.0 * tj + 1.0;
          int bj = (int)((w2 - sqrt(w2 * w2 - 8.0 * t)) * 0.5);
          while (bj > 0 && bj * tj - bj * (bj - 1) / 2 > t) --bj;
          while ((bj + 1) * tj - (bj + 1) * bj / 2 <= t) ++bj;
          const int bi = bj + (t - (bj * tj - bj * (bj - 1) / 2)), i0 = 4 * bi, j0 = 4 * bj;      // rows i0.. (>= columns j0..)
          int io[4], jo[4];
#pragma unroll
          for (int z = 0; z < 4; ++z) { io[z] = i0 + z < ntr ? i0 + z : ntr - 1; jo[z] = j0 + z < ntr ? j0 + z : ntr - 1; }
          for (int q = 0; q < pb; ++q) {
            double av[4], bv[4];
#pragma unroll
            for (int z = 0; z < 4; ++z) { av[z] = sGP[(long)q * n1 + io[z]]; bv[z] = sGP[(long)q * n1 + jo[z]]; }
#pragma unroll
            for (int zi = 0; zi < 4; ++zi)
#pragma unroll
              for (int zj = 0; zj < 4; ++zj) acc[zj * 4 + zi] += av[zi] * bv[zj];
          }
          double old[16];
#pragma unroll
          for (int zj = 0; zj < 4; ++zj)
#pragma unroll
            for (int zi = 0; zi < 4; ++zi) { const int hi = io[zi] > jo[zj] ? io[zi] : jo[zj], lo = io[zi] > jo[zj] ? jo[zj] : io[zi]; old[zj * 4 + zi] = Gh[(k0 + pb + hi) + (long)n1 * (k0 + pb + lo)]; }
#pragma unroll
          for (int zj = 0; zj < 4; ++zj)
#pragma unroll
            for (int zi = 0; zi < 4; ++zi) if (i0 + zi < ntr && j0 + zj <= i0 + zi) Gh[(k0 + pb + i0 + zi) + (long)n1 * (k0 + pb + j0 + zj)] = old[zj * 4 + zi] - acc[zj * 4 + zi];
        }
      });
    }
    barrier(c);
    tick_acc(c, 15, tprev);
  }
  return so;
}

// ---- the sweep when every row is explicit (m <= 15 + n): tails and inner products summed over the rows themselves
template <class HT>
LIT_FN SweepOut sweep_explicit(const Ctx& c, const Args<HT>& a, int e, int n, int msteps, long ec, double* E, double* Ac, double* dnv, int* refl) {
  const int n1 = n + 1;
  const double tol2 = a.tol * a.tol;
  SweepOut so = {0, 0};
  for (int k = 0; k < msteps; ++k) {
    const int p = 15 + k;
    double* ek = E + ec * k;
    double head2, tail2;
    wg_sum2(c, 0, e, head2, tail2, [&](long i, double& hs, double& ts) { const double v = ek[i] * ek[i]; if (i <= p) hs += v; else ts += v; });
    double zero2 = tol2 * (head2 + tail2); zero2 = zero2 > 2.2250738585072014e-308 ? zero2 : 2.2250738585072014e-308;
    const double c0 = ek[p];
    if (tail2 <= zero2) {
      if (tail2 > 2.2250738585072014e-308) ++so.n_skip_tol;
      if (first_thread(c)) { a.tau[k] = 0.0; dnv[k] = 0.0; refl[k] = 0; }
      par_for(c, e - (p + 1), [&](long i) { ek[p + 1 + i] = 0.0; });
      par_for(c, n1 - (k + 1), [&](long jj) { Ac[(k + 1 + jj) + (long)n1 * k] = 0.0; });
      barrier(c);
      continue;
    }
    ++so.n_reflect;
    double beta = sqrt(c0 * c0 + tail2); if (c0 >= 0.0) beta = -beta;
    const double dn = 1.0 / (c0 - beta), tk = (beta - c0) / beta;
    row_for(c, k + 1, n1, [&](long j) {
      double* ej = E + ec * j;
      double sdot = row_sum_range(c, p + 1, e, [&](long i) { return (ek[i] * dn) * ej[i]; });
      sdot = (sdot + ej[p]) * tk;
      rowlane_update(c, p + 1, e, [&](long i) { return ej[i] - sdot * (ek[i] * dn); }, [&](long i, double v) { ej[i] = v; });
      if (first_rowlane(c)) { ej[p] -= sdot; Ac[j + (long)n1 * k] = sdot * dn; }
    });
    barrier(c);
    par_for(c, e - (p + 1), [&](long i) { ek[p + 1 + i] *= dn; });
    if (first_thread(c)) { ek[p] = beta; a.tau[k] = tk; dnv[k] = dn; refl[k] = 1; }
    barrier(c);
  }
  return so;
}

// Work space of the compact route, carved out of a.W2 (and the int scratch behind a.kept); every phase below starts with it:
//   const int e = m < D ? m : D;                  // explicit rows
//   double* E = a.W2;                             // [ec x n1] column-major: the explicit rows; ends as R (rows) and the reflectors' explicit parts (below the pivots)
//   double* E0 = E + ec * n1;                     // [ec x n1] the explicit rows as they were
//   double* At = E0 + ec * n1;                    // [ec][2 m_cap]: a_i = A_j e_(i - row0) for the explicit rows
//   double* Gh = At + ec * mc2;                   // [n1 x n1] lower triangle, column-major: Gram matrix of the rows from the pivot row down
//   double* Ac = Gh + (long)n1 * n1;              // [n1 x n1] Ac[j + n1 k] = s_j / (c0 - beta) of step k (column operations of the sweep), j > k
//   double* G0s = Ac + (long)n1 * n1;             // [n1 x n1] lower triangle, column-major: Gh as it starts (the Gram matrix of the rows from row 15 down)
//   double* Stg = G0s + (long)n1 * n1;            // [2][n1][16] stand-in for the LDS staging of the first 15 rows when the staging area is too small
//   double* Ld = Stg + 33L * n1;                  // [n1] squared column norms
//   double* See = Ld + n1;                        // [ec x ec] G_E^T G_E (blocks of the tracks that own explicit rows)
//   double* Xe = See + ec * ec;                   // [ec x n] G_E^T H_u
//   double* Ut = Xe + ec * (long)n;               // [15 x n] See15 E15 / 2 - Xe15
//   double* G3 = Ut + 15L * n;                    // [ec][3]
//   double* W3 = G3 + 3 * ec;                     // [ec] s_j of the step at hand (step-by-step form of the sweep)
//   double* dnv = W3 + ec;                        // [n1]
//   double* Yk = dnv + n1 + 63;                   // [n x n] extras: Y(:, j), then Yv = Y dn
//   double* Gb0 = Yk + (long)n * n;               // [n x n] extras: Gram matrix of the rows below the explicit ones (leading block)
//   double* Gv = Gb0 + (long)n * n;               // [n x n] extras: Gb0 Yv
//   double* Th = Gv + (long)n * n;                // [ec x n] extras: t_h, then t~_h
//   double* Ph = Th + ec * (long)n;               // [ec x n] extras: See t~_h + Xe y_h
//   double* Yh = Ph + ec * (long)n;               // [n x n] extras: y_h
//   double* Qh = Yh + (long)n * n;                // [n x n] extras: Xe^T t~_h + Gam y_h
//   double* GS = Qh + (long)n * n;                // [55 (2 ec + m_cap)] extras: per observation of the tracks that own explicit rows (extras_products)
//   int* flag = a.kept + ks;                      // [e]
//   int* topt = a.kept + 2 * ks;                  // [e] track of explicit row i
//   int* refl = a.kept + 3 * ks;                  // [msteps]
//   int* bidx = a.kept + 4 * ks;                  // [<= e] the basis: kept rows < 15, reflected steps, kept handed-through rows
//   (a.kept + 5 * ks: the reflected steps before the last kept handed-through row, for the reflector chains of compact_basis)
//   double* sE = 33L * n1 <= c.lds_doubles ? c.lds : Stg;   // (+ n1 / 2 doubles: the basis list of the last phase)
#define LIT_COMPACT_LAYOUT \
  const int n = 6 * a.N, F = a.F, n1 = n + 1, D = 15 + n; \
  const int e = m < D ? m : D; \
  const int msteps = e - 15 > 0 ? e - 15 : 0; \
  const bool gram = m > e; \
  const long ec = 15 + n, mc2 = 2L * a.m_cap; \
  double* E = a.W2; \
  double* E0 = E + ec * n1; \
  double* At = E0 + ec * n1; \
  double* Gh = At + ec * mc2; \
  double* Ac = Gh + (long)n1 * n1; \
  double* G0s = Ac + (long)n1 * n1; \
  double* Stg = G0s + (long)n1 * n1; \
  double* Ld = Stg + 33L * n1; \
  double* See = Ld + n1; \
  double* Xe = See + ec * ec; \
  double* Ut = Xe + ec * (long)n; \
  double* G3 = Ut + 15L * n; \
  double* W3 = G3 + 3 * ec; \
  double* dnv = W3 + ec; \
  double* Yk = dnv + n1 + 63; \
  double* Gb0 = Yk + (long)n * n; \
  double* Gv = Gb0 + (long)n * n; \
  double* Th = Gv + (long)n * n; \
  double* Ph = Th + ec * (long)n; \
  double* Yh = Ph + ec * (long)n; \
  double* Qh = Yh + (long)n * n; \
  double* GS = Qh + (long)n * n; \
  const int ks = n + 16; \
  int* flag = a.kept + ks; \
  int* topt = a.kept + 2 * ks; \
  int* refl = a.kept + 3 * ks; \
  int* bidx = a.kept + 4 * ks; \
  const double dlt = a.u_var - a.v_var; \
  const int e15 = e < 15 ? e : 15; \
  double* sE = 33L * n1 <= c.lds_doubles ? c.lds : Stg; \
  double* sU = sE + 16L * n1; \
  auto stage15 = [&]() { \
    par_for32(c, 16 * n1, [&](int x) { \
      const int j = x >> 4, l = x & 15; \
      sE[x] = l < e15 ? E0[l + ec * j] : 0.0; \
      sU[x] = (l < 15 && j < n) ? Ut[l + 15L * j] : 0.0; \
    }); \
    barrier(c); \
  };

// ---- phase A: explicit rows, their products, the Gram matrix the sweep starts from
template <class HT>
LIT_FN void compact_rows(const Ctx& c, const Args<HT>& a, const int m) {
  LIT_COMPACT_LAYOUT
  tick(c, 1);
  // ---- explicit rows: a_i = Q_f e_(3 + i - row0) and row i of [H_o | r_o] (msckf.h:957, :430)
  par_for(c, F, [&](long t) {
    if (!(a.status[t] & a.inc_bit)) return;
    const int r0 = a.row0[t], r1 = r0 + 2 * a.M[t] - 3;
    for (int i = r0; i < r1 && i < e; ++i) topt[i] = (int)t;
  });
  par_for32(c, e * n1, [&](int x) { const int j = x / e, i = x - j * e; E[i + ec * j] = 0.0; E0[i + ec * j] = 0.0; });
  barrier(c);
  par_for(c, e, [&](long i) {
    const int t = topt[i], q = 3 + (int)i - a.row0[t];
    const double* V = a.Vf + (long)t * 2 * a.m_cap * 3;
    const double* T = a.Tf + (long)t * 9;
    double sv[3];
    for (int p = 0; p < 3; ++p) sv[p] = vf_at(V, q, p);
    for (int p = 0; p < 3; ++p) { double x = 0; for (int qq = p; qq < 3; ++qq) x += T[p * 3 + qq] * sv[qq]; G3[i * 3 + p] = x; }
  });
  barrier(c);
  par_for32(c, e * a.m_cap, [&](int x) {
    const int i = x / a.m_cap, o = x % a.m_cap, t = topt[i];
    if (o >= a.M[t]) return;
    const int q = 3 + i - a.row0[t];
    const double* V = a.Vf + (long)t * 2 * a.m_cap * 3;
    const double* w = G3 + i * 3;
    const double a0 = (2 * o == q ? 1.0 : 0.0) - (vf_at(V, 2 * o, 0) * w[0] + vf_at(V, 2 * o, 1) * w[1] + vf_at(V, 2 * o, 2) * w[2]);
    const double a1 = (2 * o + 1 == q ? 1.0 : 0.0) - (vf_at(V, 2 * o + 1, 0) * w[0] + vf_at(V, 2 * o + 1, 1) * w[1] + vf_at(V, 2 * o + 1, 2) * w[2]);
    const HT* hx = a.Hx + (long)t * a.m_cap * 12 + o * 12;
    double hh[12];
    for (int kk = 0; kk < 12; ++kk) hh[kk] = (double)hx[kk];
    const int col = 6 * a.slots[first_obs(a, t) + o];
    At[i * mc2 + 2 * o] = a0; At[i * mc2 + 2 * o + 1] = a1;
    for (int kk = 0; kk < 6; ++kk) { const double v = a0 * hh[kk] + a1 * hh[6 + kk]; E[i + ec * (col + kk)] = v; E0[i + ec * (col + kk)] = v; }
  });
  barrier(c);
  par_for(c, e, [&](long i) {
    const int t = topt[i], M = a.M[t];
    const HT* rr = a.rw + (long)t * mc2;
    double sr = 0;
    for (int o = 0; o < M; ++o) sr += At[i * mc2 + 2 * o] * (double)rr[2 * o] + At[i * mc2 + 2 * o + 1] * (double)rr[2 * o + 1];
    E[i + ec * n] = sr; E0[i + ec * n] = sr;
  });
  barrier(c);
  explicit_row_products(c, a, e, n, ec, 0, e15, topt, At, See, Xe, G3);
  // Ut = See15 E15 / 2 - Xe15  (zero rows beyond e15)
  par_for32(c, 15 * n, [&](int x) {
    const int cc = x / 15, l = x - 15 * cc;
    double s = 0;
    if (l < e15) { for (int l2 = 0; l2 < e15; ++l2) s += See[l + ec * l2] * E[l2 + ec * cc]; s = 0.5 * s - Xe[l + ec * cc]; }
    Ut[l + 15L * cc] = s;
  });
  barrier(c);
  tick(c, 2);
  // ---- Gh = [H_o | r_o]^T [H_o | r_o] minus the first 15 rows (lower triangle; the corner (n, n) is never a pivot), kept a
  // second time (G0s) for the basis products; the first 15 rows staged [column][16]
  stage15();
  const int ldt = (n1 + 7) & ~3;                   // the same rows [l][column] for the tiles (four columns: one 32-byte read; zero beyond n1)
  if (33L * n1 + 15L * ldt <= c.lds_doubles) {
    double* sEt = sU + 16L * n1;
    par_for32(c, 15 * ldt, [&](int x) { const int l = x / ldt, j = x - l * ldt; sEt[x] = (l < e15 && j < n1) ? E0[l + ec * j] : 0.0; });
    barrier(c);
    par_tiles_lower4(c, n1, [&](int i0, int j0) {
      double acc[4][4];
#pragma unroll
      for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[p][q] = 0.0;
#pragma unroll 1
      for (int l = 0; l < 15; ++l) {               // (unrolled, the compiler requests all 15 rows' vectors at once: 580 B of scratch per lane)
        const double* r = sEt + l * ldt;
        double ei[4], ej[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) { ei[p] = r[i0 + p]; ej[p] = r[j0 + p]; }
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[p][q] += ei[p] * ej[q];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int hi = i0 + p, lo = j0 + q;
          if (hi >= n1 || lo > hi) continue;
          const double v = (hi == n && lo == n) ? 0.0 : lam_in(a, hi, lo) - acc[p][q];
          G0s[hi + (long)n1 * lo] = v;
          if (gram) Gh[hi + (long)n1 * lo] = v;
        }
    });
  } else
  par_map4(c, n1 * n1, [&](int x) -> double {
    const int lo = x / n1, hi = x - lo * n1;
    if (hi < lo || (hi == n && lo == n)) return 0.0;
    double v = lam_in(a, hi, lo);
    const double* eh = sE + 16 * hi; const double* el = sE + 16 * lo;
#pragma unroll
    for (int l = 0; l < 15; ++l) v -= eh[l] * el[l];
    return v;
  }, [&](int x, double v) {
    const int lo = x / n1, hi = x - lo * n1;
    if (hi < lo) return;
    G0s[hi + (long)n1 * lo] = v;
    if (gram) Gh[hi + (long)n1 * lo] = v;
  });
  if (gram) par_for32(c, n, [&](int k) { Ld[k] = lam_in(a, k, k); });
  barrier(c);
}

// ---- phase B: the sweep (msckf.h:1343) for its decisions
template <class HT>
LIT_FN void compact_sweep(const Ctx& c, const Args<HT>& a, const int m) {
  LIT_COMPACT_LAYOUT
  tick(c, 3);
  // ---- the sweep (msckf.h:1343): steps 0..14 meet the zero IMU columns; step 15 + k works on camera column k, pivot row 15 + k
  SweepOut so = {0, 0};
  if (msteps > 0) so = gram ? sweep_gram_blocked(c, a, e, n, msteps, ec, E, Gh, Ld, Ac, dnv, refl, W3)
                            : sweep_explicit(c, a, e, n, msteps, ec, E, Ac, dnv, refl);
  if (first_thread(c)) { a.info[2] = so.n_reflect; a.info[3] = so.n_skip_tol; }
  barrier(c);
  tick(c, 4);
}

// ---- phase C: kept rows, the basis of range(Q_1), Z = [[Bs^T R_o Bs, .], [(Bs^T A)^T, .]]
template <class HT>
LIT_FN void compact_basis(const Ctx& c, const Args<HT>& a, const int m) {
  LIT_COMPACT_LAYOUT
  // ---- rows of R that are kept (msckf.h:1345-1348): a row with an entry above tol * max|R| in its upper-triangular part
  double rmax = 0;
  if (a.tol > 0) rmax = wg_max(c, 0, (long)e * n, [&](long xl) { const int x = (int)xl, j = x / e, i = x - j * e; return (j + 15 >= i) ? fabs(E[i + ec * j]) : 0.0; });
  par_for(c, e, [&](long i) { flag[i] = 0; });
  barrier(c);
  par_for(c, (long)e * n, [&](long xl) {         // an entry per thread (a thread per column walked 195 rows one load at a time); every writer of a flag writes 1
    const int x = (int)xl, j = x / e, i = x - j * e;
    if (j + 15 < i) return;
    const double v = fabs(E[i + ec * j]);
    if (a.tol > 0 ? (v > a.tol * rmax) : (v != 0.0)) flag[i] = 1;
  });
  barrier(c);
  const int nr_kept = compact_list(c, e, a.kept, [&](int i) { return flag[i] != 0; });
  const int na = compact_list(c, e15, bidx, [&](int i) { return flag[i] != 0; });
  // (a step that reflected with a pivot below tol * max|R| loses its row like any other: its column then lies in the span of
  // the basis to within the tolerance, and is left out of it)
  const int nb = compact_list(c, msteps, bidx + na, [&](int k) { return refl[k] != 0 && flag[15 + k] != 0; });
  const int nh = compact_list(c, msteps, bidx + na + nb, [&](int k) { return refl[k] == 0 && flag[15 + k] != 0; });   // steps whose row was handed through and kept
  const int nr = na + nb + nh;
  if (first_thread(c)) { a.info[1] = nr_kept; a.info[4] = 3; a.info[5] = e - msteps; a.info[6] = nh; a.info[7] = nr; }
  barrier(c);
  tick(c, 5);
  // ---- the kept handed-through rows: q_h = H_first .. H_(k_h - 1) e_h as [t ; B0 y] (the reflectors of the steps before k_h,
  // last one first); reflector j is [0 .. 1 (row 15 + j) | E(i, j) below | B0 Y(:, j) dn_j] with Y = (I + A)^-1 the column
  // operations of the sweep (A(i, j) = Ac[j + n1 i])
  if (nh > 0) {
    const int* hk = bidx + na + nb;               // steps k_h, ascending
    const int kmax = hk[nh - 1];                  // reflectors 0 .. kmax - 1 can act
    // Y(r, j) = [r == j] - sum_(i < j) Y(r, i) A(i, j), sixteen columns at a time: the part of the sum over the columns before
    // the block as a product for all rows at once, the part inside the block per row in registers.  (A thread per row walking
    // all its columns: k^2 / 2 dependent round trips, 3.9 ms for a handed-through row 160 steps in.)
    // (rows of A of the steps that did not reflect were never written: zero, so that the products below need no test)
    par_for(c, (long)kmax * n1, [&](long x) { const int i = (int)(x / n1); if (!refl[i]) Ac[x] = 0.0; });
    barrier(c);
    for (int J0 = 0; J0 < kmax; J0 += 16) {
      const int jn = kmax - J0 < 16 ? kmax - J0 : 16;
      if (J0 > 0)
        par_gemm4(c, J0, jn, [&](int r) { return r; }, [&](int) { return J0; },
                  [&](int r, int i) { return Yk[r + (long)n * i]; },
                  [&](int i, int jj) { return Ac[(J0 + jj) + (long)n1 * i]; },
                  [&](int r, int jj, double v) { Yk[r + (long)n * (J0 + jj)] = -v; });
      double* Ab = c.lds_doubles >= 256 ? c.lds : Stg;      // the block's own A, zero where a step did not reflect
      par_for(c, 256, [&](long x) { const int ii = (int)(x >> 4), jj = (int)(x & 15); Ab[x] = (ii < jj && jj < jn && refl[J0 + ii]) ? Ac[(J0 + jj) + (long)n1 * (J0 + ii)] : 0.0; });
      barrier(c);
      par_for(c, J0 + jn, [&](long r) {
        double yv[16];
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) {
          const int j = J0 + jj;
          double sv = 0.0;
          if (jj < jn && j >= r) {
            sv = r < J0 ? Yk[r + (long)n * j] : (j == r ? 1.0 : 0.0);
#pragma unroll
            for (int ii = 0; ii < jj; ++ii) sv -= yv[ii] * Ab[ii * 16 + jj];
            if (!refl[j]) sv = 0.0;
            Yk[r + (long)n * j] = sv;
          }
          yv[jj] = sv;
        }
      });
      barrier(c);
    }
    tick(c, 16);
    // Gb0 = B0^T B0 (the rows that were never explicit) = Lam_in - E0^T E0 over the first kmax columns
    if (gram && kmax <= 8)        // (a short prefix: a wavefront per entry beats staging the rows for three threads' worth of tiles)
      wave_for(c, 0, (long)kmax * kmax, [&](long x) {
        const int lo = (int)(x / kmax), hi = (int)(x - (long)lo * kmax);
        if (hi < lo) return;
        const double v = wave_sum_range(c, 0, e, [&](long i) { return E0[i + ec * hi] * E0[i + ec * lo]; });
        if (first_lane(c)) { const double g = lam_in(a, hi, lo) - v; Gb0[hi + (long)n * lo] = g; Gb0[lo + (long)n * hi] = g; }
      });
    else if (gram) syrk_lower(c, E0, ec, kmax, e, [&](int hi, int lo, double v) { const double g = lam_in(a, hi, lo) - v; Gb0[hi + (long)n * lo] = g; Gb0[lo + (long)n * hi] = g; });
    else par_for(c, (long)kmax * kmax, [&](long x) { const int lo = (int)(x / kmax), hi = (int)(x - (long)lo * kmax); Gb0[hi + (long)n * lo] = 0.0; });
    barrier(c);
    tick(c, 17);
    {                             // Yv = Y diag(dn), zero below the diagonal and in the columns up to the next multiple of four
      const int k4 = (kmax + 3) & ~3, kc = k4 < n ? k4 : n;
      par_for(c, (long)kc * kc, [&](long x) { const int j = (int)(x / kc), r = (int)(x - (long)j * kc); Yk[r + (long)n * j] = (r <= j && j < kmax) ? Yk[r + (long)n * j] * dnv[j] : 0.0; });
    }
    barrier(c);
    par_gemm4(c, kmax, kmax, [&](int) { return 0; }, [&](int j0) { return j0 + 4 < kmax ? j0 + 4 : kmax; },
              [&](int r, int l) { return Gb0[r + (long)n * l]; },
              [&](int l, int j) { return Yk[l + (long)n * (j < n ? j : n - 1)]; },
              [&](int r, int j, double v) { Gv[r + (long)n * j] = v; });
    barrier(c);
    tick(c, 18);
    int* rlist = a.kept + 5 * ks;                 // the reflected steps below kmax, ascending
    const int nrl = compact_list(c, kmax, rlist, [&](int k) { return refl[k] != 0; });
    barrier(c);
    wave_for(c, 0, nh, [&](long ah) { reflector_chain(c, a, (int)ah, hk[ah], kmax, e, n, ec, E, Gv, Yk, refl, rlist, nrl, Th + ec * ah, Yh + (long)n * ah); });
    barrier(c);
    tick(c, 19);
    // t~ = t - E0 y
    par_for(c, (long)e * nh, [&](long x) {
      const int ah = (int)(x / e), i = (int)(x - (long)ah * e);
      double sacc = Th[i + ec * ah];
      for (int l = 0; l < kmax; ++l) sacc -= E0[i + ec * l] * Yh[l + (long)n * ah];
      Th[i + ec * ah] = sacc;
    });
    barrier(c);
    tick(c, 20);
    extras_products(c, a, e, n, ec, nh, kmax, topt, At, Th, Yh, Ph, Qh, GS);
    tick(c, 21);
    // (q_h, q_h') products of the Z fill below: a wavefront per pair
    wave_for(c, 0, (long)nh * nh, [&](long x) {
      const int ah = (int)(x / nh), a2 = (int)(x - (long)ah * nh);
      if (a2 > ah) return;
      const double s1 = wave_sum_range(c, 0, e, [&](long i) { return Th[i + ec * a2] * Ph[i + ec * ah]; });
      const double s2 = wave_sum_range(c, 0, n, [&](long l) { return Yh[l + (long)n * a2] * Qh[l + (long)n * ah]; });
      if (first_lane(c)) Gb0[a2 + (long)n * ah] = s1 + s2;
    });
    barrier(c);
    tick(c, 22);
  }
  tick(c, 6);
  // ---- Z(0:nr, 0:nr) = Bs^T R_o Bs (lower triangle) and TH = Bs^T A for the basis [e_i | x'_c | q_h]
  {
    const long ldz = a.ldz;
    stage15();
    int* sB = reinterpret_cast<int*>(sU + 16L * n1);      // the basis list beside the staged rows
    par_for32(c, nr, [&](int k) { sB[k] = bidx[k]; });
    barrier(c);
    tick(c, 7);
    const int nab = na + nb;
    const int ldb = (nb + 7) & ~3;                  // the staged rows of the basis' columns, gathered, [l][k] (zero beyond nb)
    const bool tiles = 33L * n1 + 104 + 30L * ldb <= c.lds_doubles;
    double* sEg = sU + 16L * n1 + 104; double* sUg = sEg + 15L * ldb;
    if (tiles) {
      par_for32(c, 15 * ldb, [&](int x) {
        const int l = x / ldb, k = x - l * ldb;
        const int cc = k < nb ? sB[na + k] : 0;
        sEg[x] = k < nb ? sE[16 * cc + l] : 0.0; sUg[x] = k < nb ? sU[16 * cc + l] : 0.0;
      });
      barrier(c);
    }
    // entries with an e_i in them -- and all of them when the staging area has no room for the tiles' copy
    par_map4(c, tiles ? nab * na : nab * nab, [&](int x) -> double {
      const int kb = x / nab, ka = x - kb * nab;
      if (ka < kb) return 0.0;
      double val;
      if (ka < na) {                               // (e_i, e_i')
        const int i = sB[ka], i2 = sB[kb];
        val = (i == i2 ? a.v_var : 0.0) + dlt * See[i + ec * i2];
      } else {
        const int cc = sB[ka];
        const double* ec_ = sE + 16 * cc; const double* uc_ = sU + 16 * cc;
        if (kb < na) {                             // (x'_c, e_i)
          const int i = sB[kb];
          double sacc = Xe[i + ec * cc];
          for (int l = 0; l < e15; ++l) sacc -= See[i + ec * l] * ec_[l];
          val = dlt * sacc;
        } else {                                   // (x'_c, x'_c'), c >= c'
          const int c2 = sB[kb];
          const double* e2 = sE + 16 * c2; const double* u2 = sU + 16 * c2;
          double sacc = a.Gam[(long)cc * a.ldGam + c2];
#pragma unroll
          for (int l = 0; l < 15; ++l) sacc += ec_[l] * u2[l] + uc_[l] * e2[l];
          val = a.v_var * G0s[cc + (long)n1 * c2] + dlt * sacc;
        }
      }
      return val;
    }, [&](int x, double val) {
      const int kb = x / nab, ka = x - kb * nab;
      if (ka >= kb) a.Z[ka + ldz * kb] = val;
    });
    // (x'_c, x'_c'), c >= c': v' G^0(c, c') + (u' - v') (Gam(c, c') + sum_l E(l, c) U~(l, c') + U~(l, c) E(l, c'))
    if (tiles) par_tiles_lower4(c, nb, [&](int i0, int j0) {
      double acc[4][4];
#pragma unroll
      for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[p][q] = 0.0;
#pragma unroll 1
      for (int l = 0; l < 15; ++l) {
        const double* er = sEg + l * ldb; const double* ur = sUg + l * ldb;
        double ei[4], ui[4], ej[4], uj[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) { ei[p] = er[i0 + p]; ui[p] = ur[i0 + p]; ej[p] = er[j0 + p]; uj[p] = ur[j0 + p]; }
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[p][q] += ei[p] * uj[q] + ui[p] * ej[q];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int ka = i0 + p, kb = j0 + q;
          if (ka >= nb || kb > ka) continue;
          const int cc = sB[na + ka], c2 = sB[na + kb];
          a.Z[(na + ka) + ldz * (na + kb)] = a.v_var * G0s[cc + (long)n1 * c2] + dlt * (a.Gam[(long)cc * a.ldGam + c2] + acc[p][q]);
        }
    });
    barrier(c);
    tick(c, 23);
    // the handed-through rows' part in a pass of its own (inside the pass above, every wavefront walked their branch -- loops over
    // global memory -- for the one or two lanes that had such an entry: +45 us on a launch with one such trajectory)
    par_for(c, (long)nh * nr, [&](long x) {
      const int ah = (int)(x / nr), kb = (int)(x - (long)ah * nr), ka = nab + ah, h = 15 + sB[ka];
      if (kb > ka) return;
      double val;
      if (kb < na) val = dlt * Ph[sB[kb] + ec * ah];                   // (q_h, e_i)
      else if (kb < nab) {                         // (q_h, x'_c): q_h^T x'_c = R(h, c)
        const int cc = sB[kb];
        double sacc = Qh[cc + (long)n * ah];
        for (int l = 0; l < e15; ++l) sacc -= sE[16 * cc + l] * Ph[l + ec * ah];
        val = a.v_var * (cc + 15 >= h ? E[h + ec * cc] : 0.0) + dlt * sacc;
      } else {                                     // (q_h, q_h'): Th^T Ph + Yh^T Qh, left in Gb0 by the block above
        const int a2 = kb - nab;
        val = (ah == a2 ? a.v_var : 0.0) + dlt * Gb0[a2 + (long)n * ah];
      }
      a.Z[ka + ldz * kb] = val;
    });
    par_map4(c, nr * n1, [&](int x) -> double {
      const int k = x / n1, j = x - k * n1;
      double val;
      if (k < na) val = E0[sB[k] + ec * j];
      else if (k < na + nb) { const int cc = sB[k]; val = cc >= j ? G0s[cc + (long)n1 * j] : G0s[j + (long)n1 * cc]; }
      else { const int h = 15 + sB[k]; val = (j == n || j + 15 >= h) ? E[h + ec * j] : 0.0; }
      return val;
    }, [&](int x, double val) {
      const int k = x / n1, j = x - k * n1;
      a.Z[(nr + j) + ldz * k] = val;               // [T_H | r_n]^T below R_n, where the elimination wants it
    });
  }
  barrier(c);
  tick(c, 8);
}

// ---- phase D: elimination of the basis' pivots leaves -Lam^
template <class HT>
LIT_FN void compact_eliminate(const Ctx& c, const Args<HT>& a) {
  information_from_rn(c, a, 6 * a.N, a.info[7], true);
  tick(c, 9);
}

// The four phases in one call (the host build; the device runs them as four launches -- kernels_literal.hip -- so that each
// gets registers of its own: as one kernel at 1 024 threads the panel loops reloaded spilled addresses from scratch memory)
template <class HT>
LIT_FN void literal_compact(const Ctx& c, const Args<HT>& a, const int m, const int mobs) {
  (void)mobs;
  compact_rows(c, a, m);
  compact_sweep(c, a, m);
  compact_basis(c, a, m);
  barrier(c);
  compact_eliminate(c, a);
}

// route: 0 (default) = the compact route; 1 = the sweep over the dense stack (needs its work space X, G: tests and A/B runs)
template <class HT>
LIT_FN void literal_compress(const Ctx& c, const Args<HT>& a, const int route = 0, const bool prepared = false) {
  tick(c, 0);
  // prepared: the offsets, V / T of every stacked track and the info block are there already (kernels_literal.hip: k_lit_pre)
  const int m = prepared ? a.row0[a.F] : prepare(c, a);
  if (m <= 0) return;
  const int mobs = a.obs0[a.F];
  if (route != 1 || !a.X || !a.G) { literal_compact(c, a, m, mobs); return; }
  literal_general(c, a, m, mobs);
}

}  // namespace lit
}  // namespace msckf
#endif
