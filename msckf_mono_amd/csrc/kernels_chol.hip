// kernels_chol.hip -- blocked right-looking Cholesky on the matrix cores, one workgroup (4 wavefronts) per matrix.
//
// Two uses, one kernel template:
//   GRAM (T = double): [T_H | r_n] = chol(Lam^), Lam^ = [H_o | r_o]^T [H_o | r_o] accumulated by k_gram -- the
//        compression of the stacked Jacobian (HouseholderQR + Q_1^T r_o of measurementUpdate, msckf.h:1338-1366) in
//        information form; semi-definite pivot skipping as in k_chol_T (a direction the stack says nothing about gives
//        a zero row of T_H).
//   GAIN (T = float):  S = L L^T for S = T_H P T_H^T + R_n (msckf.h:1369) with the rows [P T_H^T ; r_n^T] appended, which
//        the factorization turns into W = P T_H^T L^-T and z^T = (L^-1 r_n)^T; dx = K r_n = W z (msckf.h:1370-1373 without
//        the explicit inverse).  NPART workgroups per trajectory share the appended rows (each redoes the factorization).
//
// The trailing matrix lives in MFMA accumulators: 16 x 16 blocks, 2 x 2 block-cyclic over the four wavefronts.  Per panel
// of 16 columns:
//   (1) the owners drop the panel's blocks into LDS;
//   (2) ONE wavefront factors the 16 x 16 diagonal block with lane = row, pivots and multipliers broadcast by v_readlane,
//       and runs the SAME eliminations on the rows of the identity: that yields M with  y = x M  for the forward
//       substitution of any row x against the block (zero columns for skipped pivots included);
//   (3) all wavefronts form the panel below the diagonal as L21 = A21 M on the matrix cores (the per-row substitution of
//       the older k_chol_blk -- 136 dependent FMAs fed by LDS reads per thread -- is gone);
//   (4) finished rows of T_H / columns of W go to global memory, dx accumulates;
//   (5) rank-16 update of the trailing blocks, operands straight from the LDS panel.
// 12 panels x 4 barriers instead of 180 steps x (LDS exchange + barrier); the O(n^3) part runs at MFMA rate.
#include <utility>

#include "dev_common.h"

namespace msckf {

typedef double cd4 __attribute__((ext_vector_type(4)));
typedef float cf4 __attribute__((ext_vector_type(4)));

template <class T> struct Mf;
template <> struct Mf<double> {   // v_mfma_f64_16x16x4_f64: C/D row = (lane >> 4) + 4 r, col = lane & 15
  typedef cd4 V;
  static __device__ __forceinline__ V mma(double a, double b, V c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) + 4 * r; }
};
template <> struct Mf<float> {    // v_mfma_f32_16x16x4_f32: C/D row = 4 (lane >> 4) + r, col = lane & 15
  typedef cf4 V;
  static __device__ __forceinline__ V mma(float a, float b, V c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ int row(int lane, int r) { return 4 * (lane >> 4) + r; }
};

template <class F, int... Ps>
__device__ __forceinline__ void cstatic_for_impl(F&& f, std::integer_sequence<int, Ps...>) { (f(std::integral_constant<int, Ps>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void cstatic_for(F&& f) { cstatic_for_impl(f, std::make_integer_sequence<int, N>{}); }

enum { CH_GRAM = 0, CH_GAIN = 1 };

#ifdef MSCKF_ABLATE
// phase timers of the -DMSCKF_ABLATE build: shader-clock cycles of workgroup 0's thread 0 per phase, summed over launches
// [mode][0 load, 1 panel->LDS, 2 diagonal block, 3 L21, 4 outputs, 5 trailing update, 6 launches]
__device__ unsigned long long g_chol_cycles[2][8];
#define CH_TICK(slot) do { if (tid == 0) { const long long t_ = clock64(); cyc[slot] += t_ - tlast; tlast = t_; } } while (0)
#else
#define CH_TICK(slot) do {} while (0)
#endif

// NB: 16-column blocks of the factored matrix; NA: appended 16-row blocks held by ONE workgroup (GAIN), 0 for GRAM
template <class T, class SO, int NB, int NA, int MODE, int NPART>
__global__ __launch_bounds__(256) void k_chol_mfma(Dev<SO> d, int b0) {
  typedef typename Mf<T>::V V;
  constexpr int NR = NB + NA, HR = (NR + 1) / 2, HC = (NB + 1) / 2, LP = 17;
  const int b = b0 + (MODE == CH_GAIN ? blockIdx.y : blockIdx.x), part = MODE == CH_GAIN ? (int)blockIdx.x : 0;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int pi = w >> 1, pj = w & 1;
  __builtin_amdgcn_s_setprio(3);   // latency-bound chain: win instruction arbitration against co-resident throughput waves of the other slice
  int* st = d.stats + (long)b * STAT_STRIDE;
  if (st[STAT_MROWS] == 0) return;
  const int N = d.ncam[b], n = 6 * N, D = 15 + n;
  // rows of the main block that exist: GRAM n + 1 (row n = H_o^T r_o), GAIN n; appended rows of this part (GAIN): a slice
  // of the D rows of P T_H^T plus, as the LAST row of the slice's blocks, r_n^T
  const int main_rows = MODE == CH_GRAM ? n + 1 : n;
  const int app_per = MODE == CH_GAIN ? 16 * NA - 1 : 0;              // rows of PHt per part (the last appended row is r_n^T)
  const int app_lo = part * app_per, app_hi = min(D, app_lo + app_per);   // [app_lo, app_hi) rows of PHt
  const int zrow = 16 * NR - 1;                                          // panel row of r_n^T
  __shared__ T sP[16 * NR][LP];   // current panel: every row, 16 columns
  __shared__ T sM[16][LP];        // M: y = x M solves y L11^T = x (zero columns for skipped pivots)
  __shared__ T sD0[16 * NB];      // GRAM: original diagonal (pivot tolerance)

  // ---- sources
  const double* Lam = nullptr; const double* Dg = nullptr;
  const SO* Sm = nullptr; const SO* PHtT = nullptr; const SO* R0 = d.Rbuf + ((long)b * d.nchunk) * (long)d.n6cap * d.ldR;
  if (MODE == CH_GRAM) { Lam = d.Lam + (long)b * d.ldR * d.ldR; Dg = d.Dg + (long)b * d.n_cap * DG_STRIDE; }
  else { Sm = d.Smat + (long)b * d.n6cap * d.n6cap; PHtT = d.K + (long)b * d.ld * d.n6cap; }
  // Initial accumulators.  (Measured: hoisting the loads of all blocks into one basic block -- clamped addresses, masks --
  // runs the kernel out of registers next to 60 live accumulator blocks and is slower than one round trip per block.)
  auto element = [&](int row, int col) -> T {
    if (MODE == CH_GRAM) return (T)lam_hat(Lam, Dg, d.ldR, n, d.n_cap, row, col);
    if (col >= n) return T(0);
    if (row < 16 * NB) return row < n ? (T)Sm[(long)row * d.n6cap + col] : T(0);   // S (OP_S wrote both triangles): lanes run along col
    if (row == zrow) return (T)R0[(long)col * d.ldR + n];   // r_n[col]
    const int ar = app_lo + row - 16 * NB;
    return ar < app_hi ? (T)PHtT[(long)ar * d.n6cap + col] : T(0);   // row-major copy of P T_H^T left by OP_PHT
  };
#ifdef MSCKF_ABLATE
  long long cyc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = clock64();
#endif
  V acc[HR][HC];
#pragma unroll
  for (int ii = 0; ii < HR; ++ii)
#pragma unroll
    for (int jj = 0; jj < HC; ++jj) {
      acc[ii][jj] = V{0, 0, 0, 0};
      if (2 * ii + 1 < 2 * jj && 2 * ii + 1 < NB) continue;   // compile-time: main block row above the column block
      const int i = 2 * ii + pi, j = 2 * jj + pj;
      if (i >= NR || j >= NB || (i < NB && j > i) || 16 * j >= n) continue;
      if (i < NB && 16 * i >= main_rows) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[ii][jj][r] = element(16 * i + Mf<T>::row(lane, r), 16 * j + (lane & 15));
    }
  if (MODE == CH_GRAM)
    for (int t = tid; t < 16 * NB; t += 256) { const double dv = lam_hat(Lam, Dg, d.ldR, n, d.n_cap, t, t); sD0[t] = t < n ? (T)dv : T(0); }
  const T tol = T(64.0 * 2.220446049250313e-16);
  int nskip = 0;
  T dxacc = 0;   // GAIN: thread t < app rows accumulates dx[app_lo + t] = sum_k W(., k) z_k
  SO* Rt = d.Rbuf + ((long)b * d.nchunk) * (long)d.n6cap * d.ldR;
  SO* Wg = d.W + (long)b * d.ld * d.n6cap;

  auto panel = [&](auto pc) __attribute__((always_inline)) {
    constexpr int p = decltype(pc)::value;
    if (16 * p < n) {
      __syncthreads();                                   // the previous panel's operands are no longer read
      CH_TICK(p == 0 ? 0 : 5);
      // ---- (1) blocks (i, p), i >= p, from the accumulators to the LDS panel
      if (pj == (p & 1)) {
#pragma unroll
        for (int ii = 0; ii < HR; ++ii) {
          if (2 * ii + 1 < p && 2 * ii + 1 < NB) continue;   // compile-time: block row above the panel
          const int i = 2 * ii + pi;
          if (i < p || i >= NR) continue;
#pragma unroll
          for (int r = 0; r < 4; ++r) sP[16 * i + Mf<T>::row(lane, r)][lane & 15] = acc[ii][p >> 1][r];
        }
      }
      __syncthreads();
      CH_TICK(1);
      // ---- (2) diagonal block on ONE wavefront, all 64 lanes busy: lane (r = lane & 15, g = lane >> 4) holds row r,
      // columns 4q + g of the block and of the identity image v.  Per pivot: rsqrt on the owner, dinv by v_readlane, the scaled
      // pivot column and the multipliers L(j, k) reach the other lanes through the LDS crossbar (ds_bpermute), 4 + 4 FMAs/lane.
      if (w == 0) {
        const int kcount = min(16, n - 16 * p);
        const int r = lane & 15, g = lane >> 4;
        T x[4], v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { x[q] = sP[16 * p + r][4 * q + g]; v[q] = (4 * q + g == r) ? T(1) : T(0); }
        const T d0 = MODE == CH_GRAM ? sD0[16 * p + r] : T(0);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          constexpr int dummy = 0; (void)dummy;
          const int gk = k & 3, qk = k >> 2, src = 16 * gk + k;     // pivot (k, k) lives in lane src, register x[qk]
          bool skip_l;
          T piv = x[qk];
          if (MODE == CH_GRAM) skip_l = (k >= kcount) || !(piv > tol * d0);
          else { skip_l = k >= kcount; piv = piv > T(0) ? piv : Lim<T>::tiny(); }
          const T dinv_l = skip_l ? T(0) : fast_rsqrt(piv);
          const T dinv = wave_bcast(dinv_l, src);
          const T pv = wave_bcast(piv, src);
          if (MODE == CH_GRAM && k < kcount && dinv == T(0)) ++nskip;
          // column k of L (valid in the lanes of group gk): L(r, k)
          const T c_own = (r == k) ? pv * dinv : (r > k ? x[qk] * dinv : T(0));
          const T vk_own = v[qk] * dinv;
          if (g == gk) { x[qk] = c_own; v[qk] = vk_own; }
          const T c = __shfl(c_own, 16 * gk + r, 64);     // L(r, k) for every group
          const T vk = __shfl(vk_own, 16 * gk + r, 64);   // v(r, k)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            if (4 * q + 3 <= k) continue;                 // compile-time: all of this register's columns are <= k
            const int j = 4 * q + g;
            const T ljk = __shfl(c_own, 16 * gk + (j & 15), 64);   // L(j, k)
            if (j > k) {
              if (r >= j) x[q] -= c * ljk;
              v[q] -= vk * ljk;
            }
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int j = 4 * q + g; sP[16 * p + r][j] = j <= r ? x[q] : T(0); sM[r][j] = v[q]; }
      }
      __syncthreads();
      CH_TICK(2);
      // ---- (3) panel below the diagonal block: L21 = A21 M on the matrix cores, one 16-row block per wavefront at a time
      {
        T bq[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) bq[s4] = sM[(lane >> 4) + 4 * s4][lane & 15];
        for (int i = p + 1 + w; i < NR; i += 4) {
          if (i < NB && 16 * i >= main_rows) continue;
          T a[4];
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) a[s4] = sP[16 * i + (lane & 15)][(lane >> 4) + 4 * s4];
          V y = V{0, 0, 0, 0};
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) y = Mf<T>::mma(a[s4], bq[s4], y);
#pragma unroll
          for (int r = 0; r < 4; ++r) sP[16 * i + Mf<T>::row(lane, r)][lane & 15] = y[r];
        }
      }
      __syncthreads();
      CH_TICK(3);
      // ---- (4) results of this panel
      if (MODE == CH_GRAM) {
        // rows 16p .. 16p+15 of T = L^T: T[k][c] = L(c, k), zero left of the diagonal and beyond column n
        if (tid < 16 * NB) {
          const int c = tid;
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            const int k = 16 * p + j;
            const T val = (c >= k && c <= n) ? sP[c][j] : T(0);
            if (k < n && c < d.ldR) Rt[(long)k * d.ldR + c] = (SO)val;
          }
        }
      } else {
        // columns 16p .. 16p+15 of W for this part's rows; dx += W(:, k) z_k
        if (tid < app_per) {
          const int ar = app_lo + tid;
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            const int k = 16 * p + j;
            const T wv = sP[16 * NB + tid][j];
            if (k < n && ar < app_hi) { Wg[(long)k * d.ld + ar] = (SO)wv; dxacc += wv * sP[zrow][j]; }
          }
        }
      }
      CH_TICK(4);
      // ---- (5) rank-16 update of the trailing blocks: acc(i, j) -= L(i, p) L(j, p)^T
#pragma unroll
      for (int jj = 0; jj < HC; ++jj) {
        if (2 * jj + 1 <= p) continue;                    // compile-time: at or left of the panel for either parity
        const int j = 2 * jj + pj;
        if (j <= p || j >= NB || 16 * j >= n) continue;
        T bq[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) bq[s4] = sP[16 * j + (lane & 15)][4 * s4 + (lane >> 4)];
#pragma unroll
        for (int ii = 0; ii < HR; ++ii) {
          if (2 * ii + 1 < 2 * jj && 2 * ii + 1 < NB) continue;   // compile-time: main block row above the column block
          const int i = 2 * ii + pi;
          if (i >= NR || (i < NB && (i < j || 16 * i >= main_rows))) continue;
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) {
            const T a = -sP[16 * i + (lane & 15)][4 * s4 + (lane >> 4)];
            acc[ii][jj] = Mf<T>::mma(a, bq[s4], acc[ii][jj]);
          }
        }
      }
    }
  };
  cstatic_for<NB>(panel);
#ifdef MSCKF_ABLATE
  CH_TICK(5);
  if (tid == 0 && blockIdx.x == 0 && blockIdx.y == 0) {
    for (int q = 0; q < 6; ++q) atomicAdd(&g_chol_cycles[MODE][q], (unsigned long long)cyc[q]);
    atomicAdd(&g_chol_cycles[MODE][6], 1ull);
  }
#endif
  if (MODE == CH_GRAM) { if (tid == 0) st[STAT_RROWS] = n - nskip; }
  else if (tid < app_per && app_lo + tid < app_hi) d.dx[(long)b * d.ld + app_lo + tid] = (SO)dxacc;
}

template <class S>
bool launch_chol_gram(const Dev<S>& d, int b0, int nb, hipStream_t st) {
  if (nb <= 0) return true;
  switch (d.ldR / 16) {
    case 4: hipLaunchKernelGGL((k_chol_mfma<double, S, 4, 0, CH_GRAM, 1>), dim3(nb), dim3(256), 0, st, d, b0); return true;
    case 8: hipLaunchKernelGGL((k_chol_mfma<double, S, 8, 0, CH_GRAM, 1>), dim3(nb), dim3(256), 0, st, d, b0); return true;
    case 12: hipLaunchKernelGGL((k_chol_mfma<double, S, 12, 0, CH_GRAM, 1>), dim3(nb), dim3(256), 0, st, d, b0); return true;
    default: return false;
  }
}

// GAIN: parts x NA blocks of 16 rows must cover the D + 1 appended rows (each part also carries r_n^T)
template <int NB, int NA, int NPART>
static void gain_launch(const Dev<float>& d, int b0, int nb, hipStream_t st) {
  hipLaunchKernelGGL((k_chol_mfma<float, float, NB, NA, CH_GAIN, NPART>), dim3(NPART, nb), dim3(256), 0, st, d, b0);
}
bool launch_chol_gain(const Dev<float>& d, int b0, int nb, hipStream_t st) {
  if (nb <= 0) return true;
  const int nbn = (d.n6cap + 15) / 16;
  // rows per part = 16 NA - 1; D = 15 + 6 n_cap <= NPART (16 NA - 1)
  if (nbn <= 4) { gain_launch<4, 2, 3>(d, b0, nb, st); return true; }          // D <= 79 <= 3 * 31
  if (nbn <= 8) { gain_launch<8, 3, 3>(d, b0, nb, st); return true; }          // D <= 143 <= 3 * 47
  if (nbn <= 12) { gain_launch<12, 4, 4>(d, b0, nb, st); return true; }        // D <= 207 <= 4 * 63
  return false;
}

#ifdef MSCKF_ABLATE
void chol_cycles_read(unsigned long long* out16, int reset) {
  (void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_chol_cycles), sizeof(unsigned long long) * 16);
  if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_chol_cycles), z, sizeof(z)); }
}
#endif

template bool launch_chol_gram<float>(const Dev<float>&, int, int, hipStream_t);
template bool launch_chol_gram<double>(const Dev<double>&, int, int, hipStream_t);

}  // namespace msckf
