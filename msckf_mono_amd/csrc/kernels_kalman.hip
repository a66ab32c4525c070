// kernels_kalman.hip -- Kalman gain, state correction and Joseph-form covariance update (gfx950).
//
// Reference: MSCKF::measurementUpdate, msckf.h:1368-1418 (+ buildUpdateQuat :851-872).  Inputs are the
// compressed measurement [T | r_n] left by the compression stage (kernels_gram.hip or kernels_qr.hip) in Rbuf[b][0]
// (T upper triangular n x n over the camera columns, n = 6N, possibly with zero rows; the IMU columns of T_H are
// zero, msckf.h:949) and R_n = sigma^2 I.
//   PHt = P[:,15:] T^T                      (D x n)
//   S   = T PHt[15:,:] + sigma^2 I          (n x n)     :1369
// Default ("square-root gain") form -- the gain K = P T_H^T S^-1 (:1370, explicit .inverse()) is never formed:
//   S = L L^T by a register-resident Cholesky that carries [PHt ; r_n^T] along as appended rows, which the same
//   rank-1 eliminations turn into [W ; z^T] = [PHt L^-T ; (L^-1 r_n)^T]
//   dx = K r_n = W z, injected into the IMU and every camera state          :1373-1391
//   P <- P - W W^T  ( = (I - K T_H) P, BASELINE.json north_star's form; with K from this factorization it equals
//        the Joseph form (I - K T_H) P (I - K T_H)^T + K R_n K^T of :1394-1403 identically in exact arithmetic:
//        K PHt^T = K S K^T = W W^T), one symmetric rank-n downdate written to both triangles from the same values,
//        so the symmetrisation of :1401-1403 is implicit
// Joseph form (msckf_hip_set_covariance_update(h, 1); the reference's literal sequence, kept for A/B):
//   [W ; E] = [PHt L^-T ; L^-T], K = W E^T, A = I - K T_H, P <- sym(A P A^T + sigma^2 K K^T)
// Every product is a batched 64x64-tile GEMM, LDS-staged, on the matrix cores: v_mfma_f32_32x32x2_f32 in float
// (zero k-tiles of the triangular operands skipped, symmetric results computed once), v_mfma_f64_16x16x4_f64 in
// double; trajectories with no gated-in rows are skipped.
#include "dev_common.h"

namespace msckf {

enum { OP_PHT = 0, OP_S, OP_W, OP_K, OP_A, OP_AP, OP_X, OP_RETIRED7 /* (K = W E^T of the split Joseph solve; the number stays: kernel names in profiles) */, OP_DOWN,
       OP_PHTT };   // PHt = P[:,15:] T^T written ONLY as its row-major copy (float MFMA kernel): all the blocked gain solve reads

template <class S>
struct KView {
  int D, n, ld, ldn, ldR;
  S sig2;
  const S* P; const S* R0;
  S* PHt; S* Sm; S* Linv; S* W; S* K; S* A; S* AP; S* X;
  S* PHtT;   // square-root gain form: row-major copy of PHt (D x n, row stride ldn) in the K buffer, for the blocked solve
};
template <class S>
__device__ __forceinline__ KView<S> make_view(const Dev<S>& d, int b) {
  KView<S> v;
  v.n = 6 * d.ncam[b]; v.D = 15 + v.n; v.ld = d.ld; v.ldn = d.n6cap; v.ldR = d.ldR;
  v.sig2 = d.prm[(long)b * PRM_STRIDE + PRM_SIG2];
  const long pl = (long)d.ld * d.ld, nl = (long)d.n6cap * d.n6cap, dn = (long)d.ld * d.n6cap;
  v.P = d.P + b * pl;
  v.R0 = d.Rbuf + ((long)b * d.nchunk) * (long)d.n6cap * d.ldR;
  v.PHt = d.PHt + b * dn; v.Sm = d.Smat + b * nl; v.Linv = d.Linv + b * nl;
  v.W = d.W + b * dn; v.K = d.K + b * dn; v.A = d.A + b * pl; v.AP = d.AP + b * pl; v.X = d.X + b * pl;
  v.PHtT = d.joseph == 0 ? v.K : nullptr;
  return v;
}

template <class S, int OP> __device__ __forceinline__ void op_dims(const KView<S>& v, int& M, int& N, int& K) {
  if (OP == OP_PHT) { M = v.D; N = v.n; K = v.n; }
  else if (OP == OP_S) { M = v.n; N = v.n; K = v.n; }
  else if (OP == OP_W || OP == OP_K) { M = v.D; N = v.n; K = v.n; }
  else if (OP == OP_A) { M = v.D; N = v.D; K = v.n; }
  else if (OP == OP_AP) { M = v.D; N = v.D; K = v.D; }
  else if (OP == OP_DOWN) { M = v.D; N = v.D; K = v.n; }
  else { M = v.D; N = v.D; K = v.D + v.n; }
}
template <class S, int OP> __device__ __forceinline__ S op_a(const KView<S>& v, int i, int k) {
  if (OP == OP_PHT) return v.P[(long)(15 + k) * v.ld + i];
  if (OP == OP_S) return k >= i ? v.R0[(long)i * v.ldR + k] : S(0);
  if (OP == OP_W) return v.PHt[(long)k * v.ld + i];
  if (OP == OP_K) return v.W[(long)k * v.ld + i];
  if (OP == OP_A) return v.K[(long)k * v.ld + i];
  if (OP == OP_AP) return v.A[(long)k * v.ld + i];
  if (OP == OP_DOWN) return v.W[(long)k * v.ld + i];
  return k < v.D ? v.AP[(long)k * v.ld + i] : v.sig2 * v.K[(long)(k - v.D) * v.ld + i];
}
template <class S, int OP> __device__ __forceinline__ S op_b(const KView<S>& v, int k, int j) {
  if (OP == OP_PHT) return k >= j ? v.R0[(long)j * v.ldR + k] : S(0);            // T[j][k]
  if (OP == OP_S) return v.PHt[(long)j * v.ld + 15 + k];
  if (OP == OP_W) return k <= j ? v.Linv[(long)k * v.ldn + j] : S(0);             // Linv(j,k)
  if (OP == OP_K) return j <= k ? v.Linv[(long)j * v.ldn + k] : S(0);             // Linv(k,j)
  if (OP == OP_A) return (j >= 15 && j - 15 >= k) ? v.R0[(long)k * v.ldR + (j - 15)] : S(0);   // T_H[k][j]
  if (OP == OP_AP) return v.P[(long)j * v.ld + k];
  if (OP == OP_DOWN) return v.W[(long)k * v.ld + j];
  return k < v.D ? v.A[(long)k * v.ld + j] : v.K[(long)(k - v.D) * v.ld + j];
}
template <class S, int OP> __device__ __forceinline__ void op_store(const KView<S>& v, int i, int j, S acc) {
  if (OP == OP_PHT) { v.PHt[(long)j * v.ld + i] = acc; if (v.PHtT) v.PHtT[(long)i * v.ldn + j] = acc; }
  else if (OP == OP_S) {   // lower tiles are computed; the mirror image makes S readable row-wise
    if (i >= j) {          // one writer per location: T (P T^T) is not symmetric to the last bit
      const S val = acc + (i == j ? v.sig2 : S(0));
      v.Sm[(long)j * v.ldn + i] = val; v.Sm[(long)i * v.ldn + j] = val;
    }
  }
  else if (OP == OP_W) v.W[(long)j * v.ld + i] = acc;
  else if (OP == OP_K) v.K[(long)j * v.ld + i] = acc;
  else if (OP == OP_A) v.A[(long)j * v.ld + i] = (i == j ? S(1) : S(0)) - acc;
  else if (OP == OP_AP) v.AP[(long)j * v.ld + i] = acc;
  else if (OP == OP_DOWN) {   // P <- P - W W^T, called for i <= j only: both triangles get the same value
    S* Pw = const_cast<S*>(v.P);
    const S val = Pw[(long)j * v.ld + i] - acc;
    Pw[(long)j * v.ld + i] = val; Pw[(long)i * v.ld + j] = val;
  }
  else v.X[(long)j * v.ld + i] = acc;
}

// f32 tile GEMM on the matrix cores: v_mfma_f32_32x32x2_f32 (f32 in / f32 accumulate -- bit-identical to an fmaf
// chain, MI355X_MICROARCH.md).  64 x 64 output tile per workgroup, one 32 x 32 accumulator per wavefront.  The
// product is formed transposed (MFMA "A" operand = B-tile, "B" operand = A-tile) so that the accumulator's
// lane index runs along the output ROW index i: stores to the column-major work matrices are coalesced.
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Operand-B staging order per product: true when the B element (k, j) is contiguous in j (then lanes run along j),
// false when it is contiguous in k (lanes run along k) -- keeps the global loads of the tile coalesced.
template <int OP> struct BContigJ { static constexpr bool value = (OP == OP_A || OP == OP_X || OP == OP_W || OP == OP_DOWN); };

// Operand access split into {address, validity, scale} so that the prefetch is branch-free: every load of a
// k-tile is issued unconditionally from a clamped address (all in flight together -- a conditional per element makes
// the compiler drain vmcnt between them) and masked / scaled when it is staged to LDS.
template <int OP> __device__ __forceinline__ const float* opa_ptr(const KView<float>& v, int i, int k) {
  if (OP == OP_PHT) return v.P + (long)(15 + k) * v.ld + i;
  if (OP == OP_S) return v.R0 + (long)i * v.ldR + k;
  if (OP == OP_W) return v.PHt + (long)k * v.ld + i;
  if (OP == OP_K) return v.W + (long)k * v.ld + i;
  if (OP == OP_A) return v.K + (long)k * v.ld + i;
  if (OP == OP_AP) return v.A + (long)k * v.ld + i;
  if (OP == OP_DOWN) return v.W + (long)k * v.ld + i;
  return k < v.D ? v.AP + (long)k * v.ld + i : v.K + (long)(k - v.D) * v.ld + i;
}
template <int OP> __device__ __forceinline__ float opa_fix(const KView<float>& v, int i, int k, float x) {
  if (OP == OP_S) return k >= i ? x : 0.f;
  if (OP == OP_X) return k < v.D ? x : v.sig2 * x;
  return x;
}
template <int OP> __device__ __forceinline__ const float* opb_ptr(const KView<float>& v, int k, int j) {
  if (OP == OP_PHT) return v.R0 + (long)j * v.ldR + k;
  if (OP == OP_S) return v.PHt + (long)j * v.ld + 15 + k;
  if (OP == OP_W) return v.Linv + (long)k * v.ldn + j;
  if (OP == OP_K) return v.Linv + (long)j * v.ldn + k;
  if (OP == OP_A) return v.R0 + (long)k * v.ldR + (j >= 15 ? j - 15 : 0);
  if (OP == OP_AP) return v.P + (long)j * v.ld + k;
  if (OP == OP_DOWN) return v.W + (long)k * v.ld + j;
  return k < v.D ? v.A + (long)k * v.ld + j : v.K + (long)(k - v.D) * v.ld + j;
}
template <int OP> __device__ __forceinline__ float opb_fix(const KView<float>& v, int k, int j, float x) {
  if (OP == OP_PHT) return k >= j ? x : 0.f;
  if (OP == OP_W) return k <= j ? x : 0.f;
  if (OP == OP_K) return j <= k ? x : 0.f;
  if (OP == OP_A) return (j >= 15 && j - 15 >= k) ? x : 0.f;
  return x;
}

// State injection from a finished dx (msckf.h:1373-1391, buildUpdateQuat :851-872), for one trajectory, by the threads of one
// workgroup: the square-root gain form runs it inside the first tile's workgroup of the covariance downdate (one launch less).
template <class S>
__device__ __forceinline__ void inject_from_dx(const Dev<S>& d, int b, int tid, int nthreads, const S* dx_in = nullptr) {
  const S* dx = dx_in ? dx_in : d.dx + (long)b * d.ld;
  S* imu = d.imu + (long)b * IMU_STRIDE;
  // every load of a thread is issued before its first store (a store to imu / cam may alias dx as far as the compiler knows:
  // interleaved, each += became its own load -> wait -> store round trip, ~12 in a row on the thread that also owns a GEMM tile)
  if (tid == 0) {
    const V3<S> th = mk3(dx[0], dx[1], dx[2]);
    const Q4<S> q0 = ldq(imu + IQ);
    stq(imu + IQ, qmul(update_quat(th), q0));   // not re-normalised (:1376-1378)
  }
  if (tid >= 3 && tid < 15) imu[tid + 1] += dx[tid];   // b_g v b_a p follow q in the state block in dx's order
  static_assert(IBG == 4 && IV == 7 && IBA == 10 && IP == 13, "IMU state block order");
  const int N = d.ncam[b];
  for (int c = tid; c < N; c += nthreads) {
    S* cs = d.cam + ((long)b * d.n_cap + c) * CAM_STRIDE;
    const V3<S> th = mk3(dx[15 + 6 * c], dx[16 + 6 * c], dx[17 + 6 * c]);
    const V3<S> dp = mk3(dx[18 + 6 * c], dx[19 + 6 * c], dx[20 + 6 * c]);
    const Q4<S> q0 = ldq(cs);
    const V3<S> p0 = ld3(cs + 4);
    stq(cs, qnormalized(qmul(update_quat(th), q0)));
    st3(cs + 4, p0 + dp);
  }
}

#ifdef MSCKF_ABLATE
// phase timers of the -DMSCKF_ABLATE build (scripts/chol_phases.py): shader-clock cycles of thread 0 of tile (1, 1) of the PHt
// (row 0) and downdate (row 1) products: set-up, first k-tile landed, k loop, epilogue (stores acknowledged), launches
__device__ unsigned long long g_gemm_cycles[2][8];
void gemm_cycles_read(unsigned long long* out16, int reset) {
  (void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_gemm_cycles), sizeof(unsigned long long) * 16);
  if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_cycles), z, sizeof(z)); }
}
// wall-clock (100 MHz) start / end of every workgroup of the last PHt (row 0) / downdate (row 1) launch
__device__ unsigned long long g_gemm_trace[2][4096][2];
void gemm_trace_read(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gemm_trace), sizeof(unsigned long long) * 2 * 4096 * 2); }
#define GM_TRACE(which) do { if ((OP == OP_PHT || OP == OP_DOWN) && threadIdx.x == 0) { const int wg_ = (int)blockIdx.x; if (wg_ < 4096) g_gemm_trace[OP == OP_DOWN][wg_][which] = wall_clock64(); } } while (0)
#define GM_TICK(slot) do { if ((OP == OP_PHT || OP == OP_DOWN) && threadIdx.x == 0 && bx == 0 && by == 1) { const long long t_ = clock64(); atomicAdd(&g_gemm_cycles[OP == OP_DOWN][slot], (unsigned long long)(t_ - gm_t)); gm_t = t_; } } while (0)
#else
#define GM_TICK(slot) do { } while (0)
#define GM_TRACE(which) do { } while (0)
#endif
constexpr int GT = 64;   // k-tile of the MFMA GEMM (the staging maps below assume 64)

template <int OPX>
__global__ __launch_bounds__(256) void k_gemm_mfma(Dev<float> d, int b0, int nb, int tiles_x, int tiles_y) {
  using S = float;
  // OP_PHTT: the PHt product with the accumulator holding the block itself instead of its transpose, so that the lanes run along
  // the row-major copy's contiguous index j and that copy is the only output (the column-major PHt has no reader when S is
  // formed inside the blocked gain solve): half the bytes, and the one store that was strided is gone
  constexpr bool ROWMAJ = OPX == OP_PHTT;
  constexpr int OP = ROWMAJ ? (int)OP_PHT : OPX;
  // workgroup -> (trajectory, tile) with all tiles of a trajectory on one XCD (xcd_item): the tiles of a product share their
  // operand panels, and an XCD's L2 is private -- with the tiles of a trajectory dealt round-robin over the eight XCDs every
  // XCD pulled every trajectory's operands over the fabric
  int bi_, tile_;
  if (!xcd_item(nb, tiles_x * tiles_y, bi_, tile_)) return;
  const int b = b0 + bi_, bx = tile_ % tiles_x, by = tile_ / tiles_x;
#ifdef MSCKF_ABLATE
  long long gm_t = clock64();
  GM_TRACE(0); GM_TRACE(1);
#endif
  if ((OP == OP_X || OP == OP_DOWN) && bx > by) return;   // symmetric: the epilogue mirrors the upper tiles into P
  if (OP == OP_S && bx < by) return;   // S is symmetric: the gain solve reads its lower triangle
  const int mrows_ = d.stats[(long)b * STAT_STRIDE + STAT_MROWS];   // loaded together with the window size (independent scalar loads, one wait)
  const KView<S> v = make_view(d, b);
  // downdate with the frame's prune riding on it (Dev::Pout): a trajectory without an update still has its covariance moved
  const bool fused_prune = OP == OP_DOWN && d.Pout != nullptr;
  if (mrows_ == 0 && !fused_prune) return;
  int nd_ = 0;
  if (fused_prune) { nd_ = d.fuse_drop[bi_]; nd_ = nd_ < 0 ? 0 : (nd_ > v.n / 6 ? v.n / 6 : nd_); }
  if (OP == OP_DOWN && bx == 0 && by == 0) {
    if (mrows_ != 0) inject_from_dx<S>(d, b, threadIdx.x, 256);
    if (fused_prune) {
      __syncthreads();   // the camera states just corrected are compacted by other threads
      prune_bookkeeping<S>(d, b, threadIdx.x, v.n / 6, v.n / 6 - nd_, nd_, 0);
    }
  }
  int M, N, K;
  op_dims<S, OP>(v, M, N, K);
  if (mrows_ == 0) K = 0;   // (fused prune only) nothing to subtract
  const int i0 = bx * 64, j0 = by * 64;
  if (i0 >= M || j0 >= N) return;
  __shared__ S sbuf[2 * GT * 65];
  S (*sA)[65] = reinterpret_cast<S (*)[65]>(sbuf);
  S (*sB)[65] = reinterpret_cast<S (*)[65]>(sbuf + GT * 65);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = w & 1, wn = w >> 1;
  const bool wave_live = (i0 + 32 * wm < M) && (j0 + 32 * wn < N);   // edge tiles: whole wave sub-tile out of range
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // register-staged software pipeline: the global loads of tile t+1 are in flight while tile t runs on the MFMAs
  // (measured and rejected, round 3: a second register stage -- tile t+2 in flight behind two tiles of MFMAs: 80 registers,
  // PHt / S / downdate 21.8 / 23.6 / 31.7 -> 23.2 / 24.9 / 32.4 us; with ~0.5 us of MFMA per tile and six tiles per workgroup
  // these launches are bound by their fixed start-up and drain, not by exposed load latency)
  // (also rejected: all six k-tiles of a short product requested at once, 96 staging registers: downdate 31.7 -> 39.9 us, PHt
  // 22.0 -> 24.7 us -- what bound the downdate was its epilogue, see below)
  constexpr int NQ = GT / 4;
  S ra[NQ], rb[NQ];
  const int a_i = i0 + (tid & 63), a_ic = min(a_i, M - 1);
  auto fetch = [&](int k0) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int gk = min(k0 + (tid >> 6) + 4 * q, K - 1);
      ra[q] = *opa_ptr<OP>(v, a_ic, gk);
      int kb, jj;
      if (BContigJ<OP>::value) { jj = tid & 63; kb = (tid >> 6) + 4 * q; }
      else { kb = tid & 63; jj = (tid >> 6) + 4 * q; }   // GT = 64: a wavefront reads 64 consecutive k of one column
      rb[q] = *opb_ptr<OP>(v, min(k0 + kb, K - 1), min(j0 + jj, N - 1));
    }
  };
  auto stage = [&](int k0) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int kk = (tid >> 6) + 4 * q, gk = k0 + kk;
      sA[kk][tid & 63] = (a_i < M && gk < K) ? opa_fix<OP>(v, a_i, gk, ra[q]) : 0.f;
      int kb, jj;
      if (BContigJ<OP>::value) { jj = tid & 63; kb = kk; }
      else { kb = tid & 63; jj = (tid >> 6) + 4 * q; }   // GT = 64: a wavefront reads 64 consecutive k of one column
      const int gj = j0 + jj, gk2 = k0 + kb;
      sB[kb][jj] = (gj < N && gk2 < K) ? opb_fix<OP>(v, gk2, gj, rb[q]) : 0.f;
    }
  };
  // triangular operands: T, E = L^-T are upper triangular, so whole k-tiles of the product are zero
  int kbeg = 0;
  if (OP == OP_PHT) kbeg = (j0 / GT) * GT;                           // B(k, j) = T[j][k]: zero for k < j
  if (OP == OP_S) kbeg = (i0 / GT) * GT;                             // A(i, k) = T[i][k]: zero for k < i
  if (OP == OP_A) K = min(K, max(j0 + 64 - 15, 0));                  // B(k, j) = T_H[k][j]: zero for k > j - 15
  GM_TICK(0);
  if (kbeg < K) fetch(kbeg);
  for (int k0 = kbeg; k0 < K; k0 += GT) {
    stage(k0);
    __syncthreads();
    if (k0 == kbeg) GM_TICK(1);
    if (k0 + GT < K) fetch(k0 + GT);
    if (wave_live) {
      // rows of the tile beyond K were staged as zeros: no bound inside the tile, so that the operand reads of all its MFMAs
      // can be issued ahead of the chain (a branch per MFMA kept each read -> wait -> MFMA step apart)
#pragma unroll
      for (int kk = 0; kk < GT; kk += 2) {
        const S bj = sB[kk + (lane >> 5)][32 * wn + (lane & 31)];   // MFMA A operand: rows of the result = j
        const S ai = sA[kk + (lane >> 5)][32 * wm + (lane & 31)];   // MFMA B operand: cols of the result = i
        acc = ROWMAJ ? __builtin_amdgcn_mfma_f32_32x32x2f32(ai, bj, acc, 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x2f32(bj, ai, acc, 0, 0, 0);
      }
    }
    __syncthreads();
  }
  GM_TICK(2);
  if (ROWMAJ) {
    const int gj = j0 + 32 * wn + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int gi2 = i0 + 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (gi2 < M && gj < N) v.PHtT[(long)gi2 * v.ldn + gj] = acc[r];
    }
#ifdef MSCKF_ABLATE
    __builtin_amdgcn_s_waitcnt(0);
    GM_TICK(3);
    GM_TRACE(1);
    if (threadIdx.x == 0 && bx == 0 && by == 1) atomicAdd(&g_gemm_cycles[0][4], 1ull);
#endif
    return;
  }
  const int gi = i0 + 32 * wm + (lane & 31);
  if (OP == OP_X) {
    // P <- (X + X^T)/2 (msckf.h:1418) fused into the product: X is not materialised.  Tiles above the block diagonal
    // write their values to both halves of P; a diagonal tile averages with its own transpose through LDS.
    S* Pw = const_cast<S*>(v.P);
    if (bx != by) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int gj = j0 + 32 * wn + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (gi < M && gj < N) { Pw[(long)gj * v.ld + gi] = acc[r]; Pw[(long)gi * v.ld + gj] = acc[r]; }
      }
    } else {
      S (*sT)[65] = reinterpret_cast<S (*)[65]>(sbuf);   // [64][65]: the k-loop ended with a barrier
#pragma unroll
      for (int r = 0; r < 16; ++r) sT[32 * wm + (lane & 31)][32 * wn + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)] = acc[r];
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int lj = 32 * wn + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), li = 32 * wm + (lane & 31);
        const int gj = j0 + lj;
        if (gi < M && gj < N) Pw[(long)gj * v.ld + gi] = (sT[li][lj] + sT[lj][li]) * 0.5f;
      }
    }
    return;
  }
  if (OP == OP_DOWN) {
    // P <- P - W W^T on the upper triangle (diagonal tiles), both halves written from the same value.  All sixteen old values
    // are requested before the first store: interleaved (load, subtract, two stores per element) the compiler must assume
    // that a store changes the next element's load and waits for the stores' acknowledgement every time -- sixteen serial
    // round trips (31.7 -> 29.4 us).  (Requesting them in front of the k loop instead, so that they arrive under it: no change,
    // 26.4 vs 26.5 us.)
    S* Pw = const_cast<S*>(v.P);
    S old[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int gj = j0 + 32 * wn + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      old[r] = (gi <= gj && gj < N) ? Pw[(long)gj * v.ld + gi] : 0.f;
    }
    // Rows / columns of the nd_ oldest camera states vanish, later ones move up by 6 nd_ (fused prune: into the other buffer;
    // nd_ = 0 otherwise).  The value goes to (row, column) = (i, j) with the lanes along i, and to its mirror image (j, i)
    // through an LDS transpose so that the lanes run along j there too: the mirror store straight from the accumulator
    // layout was a 4-byte store per lane with a stride of a whole column (PHt's row-major copy had the same: 20.7 -> 15.1 us)
    S* Po = fused_prune ? d.Pout + (long)b * v.ld * v.ld : Pw;
    const int cut = 15 + 6 * nd_;
    const int di = gi < 15 ? gi : gi - 6 * nd_;
    const bool keep_i = gi < 15 || gi >= cut;
    S (*sT)[65] = reinterpret_cast<S (*)[65]>(sbuf);   // [64][65]: the k loop ended with a barrier
    const int li = 32 * wm + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int lj = 32 * wn + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), gj = j0 + lj;
      const int dj = gj < 15 ? gj : gj - 6 * nd_;
      const S val = old[r] - acc[r];
      sT[li][lj] = val;
      if (gi <= gj && gj < N && keep_i && (gj < 15 || gj >= cut)) Po[(long)dj * v.ld + di] = val;
    }
    __syncthreads();
    {
      const int lj = tid & 63, gj = j0 + lj;
      const int dj = gj < 15 ? gj : gj - 6 * nd_;
      const bool keep_j = gj < N && (gj < 15 || gj >= cut);
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int li2 = (tid >> 6) + 4 * q, gi2 = i0 + li2;
        const int di2 = gi2 < 15 ? gi2 : gi2 - 6 * nd_;
        if (keep_j && gi2 <= gj && (gi2 < 15 || gi2 >= cut)) Po[(long)di2 * v.ld + dj] = sT[li2][lj];
      }
    }
#ifdef MSCKF_ABLATE
    __builtin_amdgcn_s_waitcnt(0);
    GM_TICK(3);
    GM_TRACE(1);
    if (threadIdx.x == 0 && bx == 0 && by == 1) atomicAdd(&g_gemm_cycles[1][4], 1ull);
#endif
    return;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int gj = j0 + 32 * wn + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (gi < M && gj < N) op_store<S, OP>(v, gi, gj, acc[r]);
  }
#ifdef MSCKF_ABLATE
  __builtin_amdgcn_s_waitcnt(0);
  GM_TICK(3);
  GM_TRACE(1);
  if (OP == OP_PHT && threadIdx.x == 0 && bx == 0 && by == 1) atomicAdd(&g_gemm_cycles[0][4], 1ull);
#endif
}

// f64 tile GEMM on the matrix cores: v_mfma_f64_16x16x4_f64, 64 x 64 output tile per workgroup, each wavefront a
// 32 x 32 sub-tile as 2 x 2 accumulators.  As in the f32 kernel the product is formed transposed (MFMA "A" operand =
// B-tile) so that the accumulator's lane index runs along the output row index i and the stores are coalesced.
typedef double f64x4 __attribute__((ext_vector_type(4)));
template <int OP>
__global__ __launch_bounds__(256) void k_gemm_mfma64(Dev<double> d, int b0) {
  using S = double;
  const int b = b0 + blockIdx.z;
  if (OP == OP_DOWN && blockIdx.x > blockIdx.y) return;   // symmetric downdate: upper tiles, mirrored by the store
  if (OP == OP_S && blockIdx.x < blockIdx.y) return;       // (X is needed in full: k_symmetrize averages X and X^T)
  const int mrows_ = d.stats[(long)b * STAT_STRIDE + STAT_MROWS];   // loaded together with the window size (independent scalar loads, one wait)
  const KView<S> v = make_view(d, b);
  const bool fused_prune = OP == OP_DOWN && d.Pout != nullptr;   // see the float kernel
  if (mrows_ == 0 && !fused_prune) return;
  int nd_ = 0;
  if (fused_prune) { nd_ = d.fuse_drop[blockIdx.z]; nd_ = nd_ < 0 ? 0 : (nd_ > v.n / 6 ? v.n / 6 : nd_); }
  if (OP == OP_DOWN && blockIdx.x == 0 && blockIdx.y == 0) {
    if (mrows_ != 0) inject_from_dx<S>(d, b, threadIdx.x, 256);
    if (fused_prune) {
      __syncthreads();
      prune_bookkeeping<S>(d, b, threadIdx.x, v.n / 6, v.n / 6 - nd_, nd_, 0);
    }
  }
  int M, N, K;
  op_dims<S, OP>(v, M, N, K);
  if (mrows_ == 0) K = 0;
  const int i0 = blockIdx.x * 64, j0 = blockIdx.y * 64;
  if (i0 >= M || j0 >= N) return;
  constexpr int KT = 16;
  __shared__ S sA[KT][65];
  __shared__ S sB[KT][65];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wm = w & 1, wn = w >> 1;
  f64x4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c) acc[a][c] = f64x4{0.0, 0.0, 0.0, 0.0};
  int kbeg = 0;
  if (OP == OP_PHT) kbeg = (j0 / KT) * KT;
  if (OP == OP_S) kbeg = (i0 / KT) * KT;
  if (OP == OP_A) K = min(K, max(j0 + 64 - 15, 0));
  for (int k0 = kbeg; k0 < K; k0 += KT) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ii = tid & 63, kk = (tid >> 6) + 4 * q;
      const int gi = i0 + ii, gk = k0 + kk;
      sA[kk][ii] = (gi < M && gk < K) ? op_a<S, OP>(v, gi, gk) : S(0);
      int kb, jj;
      if (BContigJ<OP>::value) { jj = tid & 63; kb = (tid >> 6) + 4 * q; }
      else { kb = tid & 15; jj = (tid >> 4) + 16 * q; }
      const int gj = j0 + jj, gk2 = k0 + kb;
      sB[kb][jj] = (gj < N && gk2 < K) ? op_b<S, OP>(v, gk2, gj) : S(0);
    }
    __syncthreads();
#pragma unroll
    for (int k4 = 0; k4 < KT; k4 += 4) {
      const int kr = k4 + (lane >> 4), cc = lane & 15;
      const S bj0 = sB[kr][32 * wn + cc], bj1 = sB[kr][32 * wn + 16 + cc];   // MFMA A operand: rows of the result = j
      const S ai0 = sA[kr][32 * wm + cc], ai1 = sA[kr][32 * wm + 16 + cc];   // MFMA B operand: cols of the result = i
      acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(bj0, ai0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(bj0, ai1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(bj1, ai0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(bj1, ai1, acc[1][1], 0, 0, 0);
    }
    __syncthreads();
  }
  // acc[jb][ib][r] = C(i = i0 + 32 wm + 16 ib + (lane & 15), j = j0 + 32 wn + 16 jb + (lane >> 4) + 4 r)
  if (OP == OP_DOWN) {   // old values of P first, all loads in flight together, then the stores (see the float kernel)
    S* Pw = const_cast<S*>(v.P);
#pragma unroll
    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
      for (int ib = 0; ib < 2; ++ib)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int gi = i0 + 32 * wm + 16 * ib + (lane & 15), gj = j0 + 32 * wn + 16 * jb + (lane >> 4) + 4 * r;
          const S old = (gi <= gj && gj < N) ? Pw[(long)gj * v.ld + gi] : S(0);
          acc[jb][ib][r] = old - acc[jb][ib][r];
        }
    S* Po = fused_prune ? d.Pout + (long)b * v.ld * v.ld : Pw;
    const int cut = 15 + 6 * nd_;
#pragma unroll
    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
      for (int ib = 0; ib < 2; ++ib)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int gi = i0 + 32 * wm + 16 * ib + (lane & 15), gj = j0 + 32 * wn + 16 * jb + (lane >> 4) + 4 * r;
          const int di = gi < 15 ? gi : gi - 6 * nd_, dj = gj < 15 ? gj : gj - 6 * nd_;
          if (gi <= gj && gj < N && (gi < 15 || gi >= cut) && (gj < 15 || gj >= cut)) { Po[(long)dj * v.ld + di] = acc[jb][ib][r]; Po[(long)di * v.ld + dj] = acc[jb][ib][r]; }
        }
    return;
  }
#pragma unroll
  for (int jb = 0; jb < 2; ++jb)
#pragma unroll
    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gi = i0 + 32 * wm + 16 * ib + (lane & 15), gj = j0 + 32 * wn + 16 * jb + (lane >> 4) + 4 * r;
        if (gi < M && gj < N) op_store<S, OP>(v, gi, gj, acc[jb][ib][r]);
      }
}

// Cholesky S = L L^T and in-place triangular inverse, one workgroup per trajectory.  The matrix is staged
// in LDS when it fits (n*(n+1) scalars), otherwise it is factored in place in global memory.
template <class S, bool LDS>
__global__ __launch_bounds__(256) void k_chol_inv(Dev<S> d, int b0) {
  const int b = b0 + blockIdx.x, tid = threadIdx.x;
  const int mrows_ = d.stats[(long)b * STAT_STRIDE + STAT_MROWS];   // loaded together with the window size (independent scalar loads, one wait)
  const KView<S> v = make_view(d, b);
  if (mrows_ == 0) return;
  const int n = v.n;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  S* L; int ldl;
  if (LDS) {
    L = reinterpret_cast<S*>(smem_raw); ldl = n + 1;
    for (int e = tid; e < n * n; e += 256) { const int i = e % n, j = e / n; L[(long)j * ldl + i] = v.Sm[(long)j * v.ldn + i]; }
  } else { L = v.Sm; ldl = v.ldn; }
  __syncthreads();
  // right-looking Cholesky on the lower triangle (column-major: element (i,j) at L[j*ldl + i])
  for (int k = 0; k < n; ++k) {
    const S dkk = L[(long)k * ldl + k];
    const S dd = dsqrt(dkk > S(0) ? dkk : Lim<S>::tiny());
    if (!(dkk > S(0)) && tid == 0) atomicOr(&d.stats[(long)b * STAT_STRIDE + STAT_ERR], STAT_ERR_PIVOT);
    __syncthreads();
    const S dinv = S(1) / dd;
    for (int i = k + tid; i < n; i += 256) L[(long)k * ldl + i] = (i == k) ? dd : L[(long)k * ldl + i] * dinv;
    __syncthreads();
    // trailing update: columns j = k+1.., rows i >= j
    const int rem = n - k - 1;
    for (int e = tid; e < rem * rem; e += 256) {
      const int jj = e / rem, ii = e % rem;
      if (ii >= jj) {
        const int i = k + 1 + ii, j = k + 1 + jj;
        L[(long)j * ldl + i] -= L[(long)k * ldl + i] * L[(long)k * ldl + j];
      }
    }
    __syncthreads();
  }
  // in-place inverse of the lower-triangular L (unblocked trtri, columns right to left)
  for (int j = n - 1; j >= 0; --j) {
    const S ljj = S(1) / L[(long)j * ldl + j];
    __syncthreads();
    // x = -Linv[j+1:, j+1:] * L[j+1:, j] * ljj ; rows processed in parallel, reading the original column j
    S xs[4]; int cnt = 0;
    for (int i = j + 1 + tid; i < n; i += 256) {   // n <= 1024: at most four rows per thread
      S s = 0;
      for (int k = j + 1; k <= i; ++k) s += L[(long)k * ldl + i] * L[(long)j * ldl + k];
      xs[cnt++] = -s * ljj;
    }
    __syncthreads();
    cnt = 0;
    for (int i = j + 1 + tid; i < n; i += 256) L[(long)j * ldl + i] = xs[cnt++];
    if (tid == 0) L[(long)j * ldl + j] = ljj;
    __syncthreads();
  }
  // publish Linv (lower triangle, zeros above)
  for (int e = tid; e < n * n; e += 256) {
    const int i = e % n, j = e / n;
    v.Linv[(long)j * v.ldn + i] = (i >= j) ? L[(long)j * ldl + i] : S(0);
  }
}

// K = PHt S^-1 with S and PHt resident in REGISTERS of one workgroup (256 threads as a 16 x 16 grid, 2-D
// block-cyclic: thread (tx,ty) owns S(16a+tx, 16b+ty) and PHt(16a+tx, 16b+ty)).
//   phase 1: right-looking Cholesky S = L L^T; PHt rides along as appended rows, so the same rank-1
//            eliminations turn it into W = PHt L^-T (the explicit S^-1 of msckf.h:1370 is never formed);
//   phase 2: K = W L^-1 by a backward column sweep (K[:,c] = W[:,c]/L[c][c]; W[:,j<c] -= K[:,c] L[c][j]).
// Per step one LDS exchange (pivot column / pivot row) + one barrier; all FMAs run on registers.
template <class S, int NBN>
__global__ __launch_bounds__(256) void k_gain(Dev<S> d, int b0) {
  constexpr int G = 16, NBD = NBN + 1;
  const int b = b0 + blockIdx.x, tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int mrows_ = d.stats[(long)b * STAT_STRIDE + STAT_MROWS];   // loaded together with the window size (independent scalar loads, one wait)
  const KView<S> v = make_view(d, b);
  if (mrows_ == 0) return;
  const int n = v.n, D = v.D;
  __shared__ S sCol[2][G * NBN];     // pivot column of S (phase 1) / pivot row of L (phase 2)
  __shared__ S sW[2][G * NBD];       // pivot column of the appended block
  S A[NBN][NBN], Wm[NBD][NBN];
#pragma unroll
  for (int a = 0; a < NBN; ++a)
#pragma unroll
    for (int bb = 0; bb < NBN; ++bb) {
      const int i = G * a + tx, j = G * bb + ty;
      A[a][bb] = (a >= bb && i < n && j < n) ? v.Sm[(long)j * v.ldn + i] : S(0);
    }
#pragma unroll
  for (int a = 0; a < NBD; ++a)
#pragma unroll
    for (int bb = 0; bb < NBN; ++bb) {
      const int i = G * a + tx, j = G * bb + ty;
      Wm[a][bb] = (i < D && j < n) ? v.PHt[(long)j * v.ld + i] : S(0);
    }
  int buf = 0;
  // ---------------- phase 1
#pragma unroll
  for (int kb = 0; kb < NBN; ++kb) {
    const int kk_hi = min(G, n - G * kb);
    for (int kk = 0; kk < kk_hi; ++kk) {
      const int k = G * kb + kk;
      if (ty == kk) {
#pragma unroll
        for (int a = kb; a < NBN; ++a) sCol[buf][G * a + tx] = A[a][kb];
#pragma unroll
        for (int a = 0; a < NBD; ++a) sW[buf][G * a + tx] = Wm[a][kb];
      }
      __syncthreads();
      const S dkk = sCol[buf][k];
      const S dpos = dkk > S(0) ? dkk : Lim<S>::tiny();
      if (!(dkk > S(0)) && tid == 0) atomicOr(&d.stats[(long)b * STAT_STRIDE + STAT_ERR], STAT_ERR_PIVOT);   // S not positive definite: reported, run continues
      const S dinv = fast_rsqrt(dpos);
      const S dd = dpos * dinv;
      S li[NBN], lj[NBN], wi[NBD];
#pragma unroll
      for (int a = kb; a < NBN; ++a) li[a] = (a > kb || tx > kk) ? sCol[buf][G * a + tx] * dinv : S(0);
#pragma unroll
      for (int bb = kb; bb < NBN; ++bb) lj[bb] = (bb > kb || ty > kk) ? sCol[buf][G * bb + ty] * dinv : S(0);
#pragma unroll
      for (int a = 0; a < NBD; ++a) wi[a] = sW[buf][G * a + tx] * dinv;
#pragma unroll
      for (int a = kb; a < NBN; ++a)
#pragma unroll
        for (int bb = kb; bb <= a; ++bb) A[a][bb] -= li[a] * lj[bb];
#pragma unroll
      for (int a = 0; a < NBD; ++a)
#pragma unroll
        for (int bb = kb; bb < NBN; ++bb) Wm[a][bb] -= wi[a] * lj[bb];
      if (ty == kk) {   // finalise column k of L and of W
#pragma unroll
        for (int a = kb; a < NBN; ++a) {
          if (a > kb || tx > kk) A[a][kb] = li[a];
          else if (a == kb && tx == kk) A[a][kb] = dd;
        }
#pragma unroll
        for (int a = 0; a < NBD; ++a) Wm[a][kb] = wi[a];
      }
      buf ^= 1;
    }
  }
  // ---------------- phase 2
#pragma unroll
  for (int cb = NBN - 1; cb >= 0; --cb) {
    const int cc_hi = min(G, n - G * cb);
    for (int cc = cc_hi - 1; cc >= 0; --cc) {
      const int c = G * cb + cc;
      if (tx == cc) {   // row c of L: L(c, j) lives in A[cb][b] of the threads with tx == cc
#pragma unroll
        for (int bb = 0; bb <= cb; ++bb) sCol[buf][G * bb + ty] = A[cb][bb];
      }
      if (ty == cc) {
#pragma unroll
        for (int a = 0; a < NBD; ++a) sW[buf][G * a + tx] = Wm[a][cb];
      }
      __syncthreads();
      const S dinv = S(1) / sCol[buf][c];
      S kc[NBD], lr[NBN];
#pragma unroll
      for (int a = 0; a < NBD; ++a) kc[a] = sW[buf][G * a + tx] * dinv;
#pragma unroll
      for (int bb = 0; bb <= cb; ++bb) lr[bb] = (bb < cb || ty < cc) ? sCol[buf][G * bb + ty] : S(0);
#pragma unroll
      for (int a = 0; a < NBD; ++a)
#pragma unroll
        for (int bb = 0; bb <= cb; ++bb) Wm[a][bb] -= kc[a] * lr[bb];
      if (ty == cc) {
#pragma unroll
        for (int a = 0; a < NBD; ++a) Wm[a][cb] = kc[a];
      }
      buf ^= 1;
    }
  }
#pragma unroll
  for (int a = 0; a < NBD; ++a)
#pragma unroll
    for (int bb = 0; bb < NBN; ++bb) {
      const int i = G * a + tx, j = G * bb + ty;
      if (i < D && j < n) v.K[(long)j * v.ld + i] = Wm[a][bb];
    }
  // dx = K r_n (msckf.h:1373) while K is still in registers: per-thread partial sums over this thread's columns,
  // combined over the 16 column residues in a fixed order (deterministic)
  __shared__ S sDx[G][G * NBD + 1];
  {
    S rn[NBN];
#pragma unroll
    for (int bb = 0; bb < NBN; ++bb) { const int j = G * bb + ty; rn[bb] = j < n ? v.R0[(long)j * v.ldR + n] : S(0); }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < NBD; ++a) {
      S p = 0;
#pragma unroll
      for (int bb = 0; bb < NBN; ++bb) p += Wm[a][bb] * rn[bb];
      sDx[ty][G * a + tx] = p;
    }
    __syncthreads();
    for (int i = tid; i < D; i += 256) {
      S sm = 0;
#pragma unroll
      for (int y = 0; y < G; ++y) sm += sDx[y][i];
      d.dx[(long)b * d.ld + i] = sm;
    }
  }
}

// Square-root gain: NPART workgroups per trajectory.  Every workgroup factors S = L L^T in registers (redundantly -- the
// chip has idle CUs at these batch sizes) and carries 1/NPART of the rows of PHt plus the row r_n^T through the same
// rank-1 eliminations, which turns them into W = PHt L^-T and z^T = (L^-1 r_n)^T.  Then dx = K r_n = W z for the
// workgroup's rows, in a fixed summation order.  Neither S^-1 (msckf.h:1370) nor K is formed.
template <class S, int NBN, int NPART>
__global__ __launch_bounds__(256) void k_gain_w(Dev<S> d, int b0) {
  constexpr int G = 16, NBD = NBN + 1, NBQ = (NBD + NPART - 1) / NPART, NBW = NBQ + 1;
  const int b = b0 + blockIdx.y, part = blockIdx.x, tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  __builtin_amdgcn_s_setprio(3);   // latency-bound chain: win instruction arbitration against co-resident throughput waves of the other slice
  const int mrows_ = d.stats[(long)b * STAT_STRIDE + STAT_MROWS];   // loaded together with the window size (independent scalar loads, one wait)
  const KView<S> v = make_view(d, b);
  if (mrows_ == 0) return;
  const int n = v.n, D = v.D;
  __shared__ S sCol[2][G * NBN];
  __shared__ S sW[2][G * NBW];
  S A[NBN][NBN], Wm[NBW][NBN];
#pragma unroll
  for (int a = 0; a < NBN; ++a)
#pragma unroll
    for (int bb = 0; bb < NBN; ++bb) {
      const int i = G * a + tx, j = G * bb + ty;
      A[a][bb] = (a >= bb && i < n && j < n) ? v.Sm[(long)j * v.ldn + i] : S(0);
    }
#pragma unroll
  for (int a = 0; a < NBW; ++a) {
    const int ab = part * NBQ + a;            // row block of PHt; a == NBQ: the block whose row 0 is r_n^T
#pragma unroll
    for (int bb = 0; bb < NBN; ++bb) {
      const int j = G * bb + ty;
      S val = 0;
      if (a < NBQ) { const int i = G * ab + tx; if (ab < NBD && i < D && j < n) val = v.PHt[(long)j * v.ld + i]; }
      else if (tx == 0 && j < n) val = v.R0[(long)j * v.ldR + n];
      Wm[a][bb] = val;
    }
  }
  int buf = 0;
#pragma unroll
  for (int kb = 0; kb < NBN; ++kb) {
    const int kk_hi = min(G, n - G * kb);
    for (int kk = 0; kk < kk_hi; ++kk) {
      const int k = G * kb + kk;
      if (ty == kk) {
#pragma unroll
        for (int a = kb; a < NBN; ++a) sCol[buf][G * a + tx] = A[a][kb];
#pragma unroll
        for (int a = 0; a < NBW; ++a) sW[buf][G * a + tx] = Wm[a][kb];
      }
      __syncthreads();
      const S dkk = sCol[buf][k];
      const S dpos = dkk > S(0) ? dkk : Lim<S>::tiny();
      if (!(dkk > S(0)) && tid == 0) atomicOr(&d.stats[(long)b * STAT_STRIDE + STAT_ERR], STAT_ERR_PIVOT);   // S not positive definite: reported, run continues
      const S dinv = fast_rsqrt(dpos);
      const S dd = dpos * dinv;
      S li[NBN], lj[NBN], wi[NBW];
#pragma unroll
      for (int a = kb; a < NBN; ++a) li[a] = (a > kb || tx > kk) ? sCol[buf][G * a + tx] * dinv : S(0);
#pragma unroll
      for (int bb = kb; bb < NBN; ++bb) lj[bb] = (bb > kb || ty > kk) ? sCol[buf][G * bb + ty] * dinv : S(0);
#pragma unroll
      for (int a = 0; a < NBW; ++a) wi[a] = sW[buf][G * a + tx] * dinv;
#pragma unroll
      for (int a = kb; a < NBN; ++a)
#pragma unroll
        for (int bb = kb; bb <= a; ++bb) A[a][bb] -= li[a] * lj[bb];
#pragma unroll
      for (int a = 0; a < NBW; ++a)
#pragma unroll
        for (int bb = kb; bb < NBN; ++bb) Wm[a][bb] -= wi[a] * lj[bb];
      if (ty == kk) {
#pragma unroll
        for (int a = kb; a < NBN; ++a) {
          if (a > kb || tx > kk) A[a][kb] = li[a];
          else if (a == kb && tx == kk) A[a][kb] = dd;
        }
#pragma unroll
        for (int a = 0; a < NBW; ++a) Wm[a][kb] = wi[a];
      }
      buf ^= 1;
    }
  }
  // W rows of this part
#pragma unroll
  for (int a = 0; a < NBQ; ++a) {
    const int ab = part * NBQ + a, i = G * ab + tx;
#pragma unroll
    for (int bb = 0; bb < NBN; ++bb) {
      const int j = G * bb + ty;
      if (ab < NBD && i < D && j < n) v.W[(long)j * v.ld + i] = Wm[a][bb];
    }
  }
  // z lives in row 0 of the last block: thread (0, ty) holds z[G bb + ty]
  __shared__ S sZ[G * NBN];
  __shared__ S sDx[G][G * NBQ + 1];
  __syncthreads();
  if (tx == 0) {
#pragma unroll
    for (int bb = 0; bb < NBN; ++bb) sZ[G * bb + ty] = Wm[NBQ][bb];
  }
  __syncthreads();
  {
    S zz[NBN];
#pragma unroll
    for (int bb = 0; bb < NBN; ++bb) zz[bb] = sZ[G * bb + ty];
#pragma unroll
    for (int a = 0; a < NBQ; ++a) {
      S p = 0;
#pragma unroll
      for (int bb = 0; bb < NBN; ++bb) p += Wm[a][bb] * zz[bb];
      sDx[ty][G * a + tx] = p;
    }
    __syncthreads();
    for (int e = tid; e < G * NBQ; e += 256) {
      const int i = G * NBQ * part + e;
      if (i >= D) continue;
      S sm = 0;
#pragma unroll
      for (int y = 0; y < G; ++y) sm += sDx[y][e];
      d.dx[(long)b * d.ld + i] = sm;
    }
  }
}

// dx = W (L^-1 r_n) for the windows whose S does not fit the register grid (k_chol_inv left L^-1 in Linv, OP_W left
// W = PHt L^-T); one workgroup per trajectory.
template <class S>
__global__ __launch_bounds__(256) void k_dx_w(Dev<S> d, int b0) {
  const int b = b0 + blockIdx.x, tid = threadIdx.x;
  const int mrows_ = d.stats[(long)b * STAT_STRIDE + STAT_MROWS];   // loaded together with the window size (independent scalar loads, one wait)
  const KView<S> v = make_view(d, b);
  if (mrows_ == 0) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  S* srn = reinterpret_cast<S*>(smem_raw);
  S* sz = srn + v.ldn;
  for (int a = tid; a < v.n; a += 256) srn[a] = v.R0[(long)a * v.ldR + v.n];
  __syncthreads();
  for (int j = tid; j < v.n; j += 256) {
    S s = 0;
    for (int k = 0; k <= j; ++k) s += v.Linv[(long)k * v.ldn + j] * srn[k];     // Linv(j, k), lower triangular
    sz[j] = s;
  }
  __syncthreads();
  for (int i = tid; i < v.D; i += 256) {
    S s = 0;
    for (int j = 0; j < v.n; ++j) s += v.W[(long)j * v.ld + i] * sz[j];
    d.dx[(long)b * d.ld + i] = s;
  }
}

// dx = K r_n and state injection (msckf.h:1373-1391); one workgroup per trajectory.
template <class S, bool HAVE_DX>
__global__ __launch_bounds__(256) void k_inject(Dev<S> d, int b0) {
  const int b = b0 + blockIdx.x, tid = threadIdx.x;
  const int mrows_ = d.stats[(long)b * STAT_STRIDE + STAT_MROWS];   // loaded together with the window size (independent scalar loads, one wait)
  const KView<S> v = make_view(d, b);
  if (mrows_ == 0) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  S* sdx = reinterpret_cast<S*>(smem_raw);
  // r_n (column n of [T | r_n]) once into LDS; dx = K r_n with 12 independent accumulators per row so that a dozen
  // loads are in flight per thread (the product is pure latency: 140 KB per trajectory)
  S* srn = sdx + d.ld;
  if (!HAVE_DX) {
    for (int a = tid; a < v.n; a += 256) srn[a] = v.R0[(long)a * v.ldR + v.n];
    __syncthreads();
  }
  for (int i = tid; i < v.D; i += 256) {
    S s = 0;
    if (HAVE_DX) s = d.dx[(long)b * d.ld + i];
    else {
      S acc[12];
#pragma unroll
      for (int u = 0; u < 12; ++u) acc[u] = 0;
      int a = 0;
      for (; a + 11 < v.n; a += 12) {
#pragma unroll
        for (int u = 0; u < 12; ++u) acc[u] += v.K[(long)(a + u) * v.ld + i] * srn[a + u];
      }
      for (; a < v.n; ++a) acc[0] += v.K[(long)a * v.ld + i] * srn[a];
      s = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7])) + ((acc[8] + acc[9]) + (acc[10] + acc[11]));
      d.dx[(long)b * d.ld + i] = s;
    }
    sdx[i] = s;
  }
  __syncthreads();
  S* imu = d.imu + (long)b * IMU_STRIDE;
  if (tid == 0) {
    const Q4<S> q = qmul(update_quat(mk3(sdx[0], sdx[1], sdx[2])), ldq(imu + IQ));   // not re-normalised (:1376-1378)
    stq(imu + IQ, q);
    for (int k = 0; k < 3; ++k) { imu[IBG + k] += sdx[3 + k]; imu[IV + k] += sdx[6 + k]; imu[IBA + k] += sdx[9 + k]; imu[IP + k] += sdx[12 + k]; }
  }
  const int N = d.ncam[b];
  for (int c = tid; c < N; c += 256) {
    S* cs = d.cam + ((long)b * d.n_cap + c) * CAM_STRIDE;
    const Q4<S> q = qnormalized(qmul(update_quat(mk3(sdx[15 + 6 * c], sdx[16 + 6 * c], sdx[17 + 6 * c])), ldq(cs)));
    stq(cs, q);
    for (int k = 0; k < 3; ++k) cs[4 + k] += sdx[18 + 6 * c + k];
  }
}

// P <- (X + X^T)/2 (msckf.h:1418).  The MFMA path computes only the 64x64 tiles of X on or above the block diagonal
// (TRI): below it the mirrored element is taken, inside a diagonal tile the two halves are averaged as the reference does.
template <class S, bool TRI>
__global__ __launch_bounds__(256) void k_symmetrize(Dev<S> d, int b0) {
  const int b = b0 + blockIdx.y;
  if (d.stats[(long)b * STAT_STRIDE + STAT_MROWS] == 0) return;
  const int D = 15 + 6 * d.ncam[b], ld = d.ld;
  const S* X = d.X + (long)b * ld * ld;
  S* P = d.P + (long)b * ld * ld;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < (long)D * D; e += (long)gridDim.x * 256) {
    const int i = (int)(e % D), j = (int)(e / D);
    S val;
    if (TRI && (i >> 6) != (j >> 6)) val = (i >> 6) < (j >> 6) ? X[(long)j * ld + i] : X[(long)i * ld + j];
    else val = (X[(long)j * ld + i] + X[(long)i * ld + j]) / S(2);
    P[(long)j * ld + i] = val;
  }
}

template <int OP>
static void gemm_launch(const Dev<float>& d, int b0, int nb, int Mmax, int Nmax, hipStream_t st) {
  const int tx = (Mmax + 63) / 64, ty = (Nmax + 63) / 64;
  hipLaunchKernelGGL((k_gemm_mfma<OP>), dim3(xcd_grid(nb, tx * ty)), dim3(256), 0, st, d, b0, nb, tx, ty);
}
template <int OP>
static void gemm_launch(const Dev<double>& d, int b0, int nb, int Mmax, int Nmax, hipStream_t st) {
  hipLaunchKernelGGL((k_gemm_mfma64<OP>), dim3((Mmax + 63) / 64, (Nmax + 63) / 64, nb), dim3(256), 0, st, d, b0);
}
template <class S, int OP>
static void gemm(const Dev<S>& d, int b0, int nb, int Mmax, int Nmax, hipStream_t st) {
  gemm_launch<OP>(d, b0, nb, Mmax, Nmax, st);
}

static void gemm_pht_rowmajor(const Dev<float>& d, int b0, int nb, int Mmax, int Nmax, hipStream_t st) { gemm_launch<OP_PHTT>(d, b0, nb, Mmax, Nmax, st); }
static void gemm_pht_rowmajor(const Dev<double>& d, int b0, int nb, int Mmax, int Nmax, hipStream_t st) { gemm_launch<OP_PHT>(d, b0, nb, Mmax, Nmax, st); }   // (never taken: s_fused is float only)

template <class S, int NBN>
static void launch_gain_w(const Dev<S>& d, int b0, int nb, hipStream_t st) {
  // more workgroups per trajectory while the batch leaves CUs idle (the Cholesky of S is redone by each of them)
  if (nb <= 64) hipLaunchKernelGGL((k_gain_w<S, NBN, 8>), dim3(8, nb), dim3(256), 0, st, d, b0);
  else if (nb <= 128) hipLaunchKernelGGL((k_gain_w<S, NBN, 4>), dim3(4, nb), dim3(256), 0, st, d, b0);
  else hipLaunchKernelGGL((k_gain_w<S, NBN, 2>), dim3(2, nb), dim3(256), 0, st, d, b0);
}

static bool gain_blocked(const Dev<float>& d, int b0, int nb, hipStream_t st) { return launch_chol_gain(d, b0, nb, st); }
static bool gain_blocked(const Dev<double>&, int, int, hipStream_t) { return false; }

// ---------------------------------------------------------------- k_update_small: the whole update of a SMALL window in one launch
// Single filters with short windows (BASELINE.json configs[1]: 10 cameras, D = 75 .. 81) ran their update as a chain of six
// launches after k_select_diag -- k_gram, k_chol_mfma, two GEMMs, k_gain_w, the downdate GEMM, each a latency chain of its own
// plus a launch boundary: 104 us of kernels for 1.5 MFLOP.  Everything fits ONE workgroup's LDS at these sizes:
//   A  Lam^ = blockdiag(sum h^T h | sum h^T r) - sum_j [B_j | c_j]^T [B_j | c_j]      (msckf.h:404-431 in information form, f64)
//   B  [T | r_n] = chol(Lam^)     a pivot below 64 eps x its original diagonal is a gauge direction: zero row (kernels_chol.hip's rule)
//   C  PHt = P[:, 15:] T^T        D  S = T PHt[15:, :] + sigma^2 I                                                      (:1369)
//   E  S = L L^T with the rows [PHt ; r_n^T] riding along: they come out as W = PHt L^-T and z^T = (L^-1 r_n)^T           (:1370)
//   G  dx = W z, state injection (:1373-1391)     H  P <- P - W W^T
// The products (A, C, D, H) run on the matrix cores in 16 x 16 tiles (v_mfma_*_16x16x4, operands straight from LDS); the two
// factorizations keep the matrix in REGISTERS -- a wavefront task is 64 rows x 8 columns, lane = row -- and per pivot only the
// pivot's column crosses LDS: the owners publish it unscaled together with 1 / sqrt(pivot), one barrier, every task updates
// its columns right of the pivot (x -= (c_i d^2) c_j: one per-lane read, eight wave-uniform reads, eight FMAs).  P[:, 15:] is
// fetched into LDS once, while A runs.  First form of the kernel (scalar loops over LDS, 512 threads): ~200 us.
#ifdef MSCKF_ABLATE
// phase timers of the -DMSCKF_ABLATE build (scripts/us_phases.py): shader-clock cycles of thread 0 per phase of k_update_small, summed over launches
// [0 A Gram, 1 P to LDS + loads of B, 2 B chol(Lam^), 3 C PHt, 4 D S, 5 E loads, 6 E chol(S) + W, 7 G dx, 8 H downdate, 9 injection, 15 launches]
__device__ unsigned long long g_us_cycles[16];
void us_cycles_read(unsigned long long* out16, int reset) {
  (void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_us_cycles), sizeof(unsigned long long) * 16);
  if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_us_cycles), z, sizeof(z)); }
}
#define US_TICK(slot) do { if (tid == 0 && blockIdx.x == 0) { const long long t_ = clock64(); atomicAdd(&g_us_cycles[slot], (unsigned long long)(t_ - us_t)); us_t = t_; } } while (0)
#else
#define US_TICK(slot) do {} while (0)
#endif
typedef double us_d4 __attribute__((ext_vector_type(4)));
typedef float us_f4 __attribute__((ext_vector_type(4)));
template <class T> struct UsM;
template <> struct UsM<double> {
  typedef us_d4 V;
  static __device__ __forceinline__ V mma(double a, double b, V c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) + 4 * r; }
  static __device__ __forceinline__ double clamp() { return 1e-300; }    // a non-positive pivot (reported): its rsqrt squared stays finite
};
template <> struct UsM<float> {
  typedef us_f4 V;
  static __device__ __forceinline__ V mma(float a, float b, V c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ int row(int lane, int r) { return 4 * (lane >> 4) + r; }
  static __device__ __forceinline__ float clamp() { return 1e-30f; }
};
constexpr int US_UMAX = 2;     // tasks per wavefront of the factorizations
constexpr int US_SEG_E = 12;   // columns per task of chol(S) with its D + 1 riding rows (10 cameras: 3 row chunks x 5 segments = 15 tasks); chol(Lam^): SEGB = 4 (61 rows: one chunk, 15 tasks) or 8
// scalars of the PHt / W block [15 + n_max][n_max | 1]: it also stages 24 rows of B^ (doubles, 16 ceil((n_max + 1) / 16) + 1 columns) in
// phase A, the larger need for the first few frames of a window; and of the P staging block [n_max][(15 + n_max) | 1], later S [n_max][n_max | 1]
__host__ __device__ inline size_t update_small_ph_elems(int n_max, size_t scalar) {
  const size_t w = (size_t)(15 + n_max) * (n_max | 1), nb1 = 16 * (((size_t)n_max + 1 + 15) / 16) + 1, stg = (24 * nb1 * sizeof(double) + scalar - 1) / scalar;
  return ((w > stg ? w : stg) + 3) & ~(size_t)3;
}
// (phase A keeps two ints per included track there: f_cap of them at most)
__host__ __device__ inline size_t update_small_pt_elems(int n_max, int f_cap, size_t scalar) {
  const size_t a = (size_t)n_max * ((15 + n_max) | 1), b = (size_t)n_max * (n_max | 1), c = ((size_t)f_cap * 2 * sizeof(int) + scalar - 1) / scalar;
  const size_t m = a > b ? (a > c ? a : c) : (b > c ? b : c);
  return (m + 3) & ~(size_t)3;
}
// one 16 x 16 tile of sum_k a(16 ti + m, k) b(k, 16 tj + c), k in [k0, k1) in steps of 4 (the functors return zero out of range)
template <class T, class FA, class FB>
__device__ __forceinline__ typename UsM<T>::V us_tile(int ti, int tj, int k0, int k1, int lane, FA&& fa, FB&& fb) {
  typename UsM<T>::V acc = {0, 0, 0, 0};
  const int m = lane & 15, g = lane >> 4;
  for (int k = k0; k < k1; k += 4) acc = UsM<T>::mma(fa(16 * ti + m, k + g), fb(k + g, 16 * tj + m), acc);
  return acc;
}
// Right-looking Cholesky of the leading n x n block of a tall matrix (R rows) held in registers, rows n .. R-1 riding along.
// Task t = w + 16 u of wavefront w: column segment cs = t % ncs (columns SEG cs .. SEG cs + SEG - 1), row chunk t / ncs (row 64 rc +
// lane).  sCol[(k & 1) * cstride + i]: column k, unscaled, as its owners left it; slot R of it: 1 / sqrt(pivot k) (0: skipped).
// store(i, k, L(i, k)) is called for the entries i >= k of the finished column k by the task that owns it.
// What a step costs is the number of instructions its slowest wavefront issues between two barriers (one wavefront issues an
// instruction every ~6 cycles here), so: every LDS read of the step in one round trip; no masks -- registers of finished
// columns (j <= k), of columns beyond n and of rows beyond R are dead and may take garbage; the pivot's tolerance waits in a
// register of its row's lane; the column to publish is picked by a wave-uniform register index.
template <class T, int SEG, bool SEMIDEF, class Store>
__device__ __forceinline__ void us_chol_tall(T (&x)[US_UMAX][SEG], const int n, const int R, const int ncs, const int ntask, const int w, const int lane,
                                             T* sCol, const int cstride, const double* sD0, int* sFlag, Store&& store) {
  int cs_u[US_UMAX], i_u[US_UMAX]; bool on[US_UMAX]; T told[US_UMAX];
#pragma unroll
  for (int u = 0; u < US_UMAX; ++u) {
    const int t = w + 16 * u; on[u] = t < ntask; cs_u[u] = t % ncs; i_u[u] = 64 * (t / ncs) + lane;
    told[u] = SEMIDEF ? T(64.0 * 2.220446049250313e-16) * (T)sD0[min(i_u[u], n)] : T(0);
  }
  // column j1 (register jj of task u) as it stands, with its pivot's reciprocal square root
  auto publish = [&](int u, const T v, int j1) __attribute__((always_inline)) {
    T* col = sCol + (j1 & 1) * cstride;
    if (i_u[u] < R) col[i_u[u]] = v;
    if (i_u[u] == j1) {
      T piv = v; bool skip = false;
      if (SEMIDEF) { skip = !(piv > told[u]); if (skip) atomicAdd(&sFlag[0], 1); }
      else { if (!(piv > T(0))) sFlag[1] = 1; piv = piv > UsM<T>::clamp() ? piv : UsM<T>::clamp(); }
      col[R] = skip ? T(0) : fast_rsqrt(piv);
    }
  };
#pragma unroll
  for (int u = 0; u < US_UMAX; ++u) if (on[u] && cs_u[u] == 0) publish(u, x[u][0], 0);
  __syncthreads();
  for (int k = 0; k < n; ++k) {
    const T* col = sCol + (k & 1) * cstride;
    const int kc = k / SEG;
#pragma unroll
    for (int u = 0; u < US_UMAX; ++u) {
      if (!on[u] || cs_u[u] < kc) continue;                  // wave-uniform: every column of the task is finished
      const int i = i_u[u];
      const int j0 = SEG * cs_u[u];
      const T dinv = col[R];
      const T ci = col[min(i, R - 1)];
      T cj[SEG];
#pragma unroll
      for (int jj = 0; jj < SEG; ++jj) cj[jj] = col[j0 + jj];        // wave-uniform addresses (SEG ncs <= cstride)
      const T li = ci * dinv;                                // L(i, k)
      if (cs_u[u] == kc && (unsigned)(i - k) < (unsigned)(R - k)) store(i, k, li);
      const T li2 = li * dinv;
#pragma unroll
      for (int jj = 0; jj < SEG; ++jj) x[u][jj] -= li2 * cj[jj];
      const int jj1 = __builtin_amdgcn_readfirstlane(k + 1 - j0);
      if (k + 1 < n && jj1 >= 0 && jj1 < SEG) {
        T v = x[u][0];
#pragma unroll
        for (int jj = 1; jj < SEG; ++jj) v = jj == jj1 ? x[u][jj] : v;
        publish(u, v, k + 1);
      }
    }
    __syncthreads();
  }
}

template <class S, int US_SEG_B>
__global__ __launch_bounds__(1024) void k_update_small(Dev<S> d, int b0, int n_max) {
  typedef typename UsM<S>::V VS;
  typedef typename UsM<double>::V VD;
  const int b = b0 + blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  int* st = d.stats + (long)b * STAT_STRIDE;
  const int mrows_ = st[STAT_MROWS], npass = st[STAT_PASSED], N = d.ncam[b];
  // run_frames: the frame's prune rides on the downdate (Dev::Pout, as in the tile GEMM's OP_DOWN): P - W W^T goes straight to its
  // pruned position in the other buffer; a trajectory without an update still has its covariance moved there
  const bool fused_prune = d.Pout != nullptr;
  const bool upd = mrows_ != 0;
  if (!upd && !fused_prune) return;
  int nd_ = 0;
  if (fused_prune) { nd_ = d.fuse_drop[blockIdx.x]; nd_ = nd_ < 0 ? 0 : (nd_ > N ? N : nd_); }
  const int n = 6 * N, n1 = n + 1, D = 15 + n, ld = d.ld, ldR = d.ldR, f_cap = d.f_cap;
  const int LL = (n_max + 1) | 1, LS = n_max | 1, Dm = 15 + n_max, Dp = Dm | 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* sL = reinterpret_cast<double*>(smem_raw);                 // [(n_max+1)][LL] lower: Lam^ -> L (T = L^T, row n = r_n^T), upper zero after B
  double* sD0 = sL + (size_t)(n_max + 1) * LL;                      // [n_max+2] original diagonal (pivot tolerance)
  double* sColD = sD0 + n_max + 2;                                  // [2][cstride] published columns (doubles in B, S in E)
  const int cstride = 2 * n_max + 20;
  S* sPT = reinterpret_cast<S*>(sColD + 2 * cstride);               // [n_max][Dp] P(r, 15 + c) at [c][r]; later S at [i][LS]
  S* sPH = sPT + update_small_pt_elems(n_max, f_cap, sizeof(S));                      // [Dm][LS] PHt -> W   (phase A: staging of B^ rows, doubles)
  S* sZ = sPH + update_small_ph_elems(n_max, sizeof(S));            // [n_max+2] z = L^-1 r_n
  S* sdx = sZ + n_max + 2;                                          // [Dm]
  int* sFlag = reinterpret_cast<int*>(sdx + Dm + 1);                // skipped pivots, bad pivot
#ifdef MSCKF_ABLATE
  long long us_t = clock64();
  if (tid == 0 && blockIdx.x == 0) atomicAdd(&g_us_cycles[15], 1ull);
#endif
  if (upd) {
    if (tid < 2) sFlag[tid] = 0;
    for (int e = tid; e < (n_max + 1) * LL; e += 1024) sL[e] = 0.0;   // (its upper triangle is never written: C and D read zeros there)
    // P[:, 15:] on its way into registers (consecutive threads along a row of P: P is symmetric, row 15 + c = column 15 + c)
    const S* P = d.P + (long)b * ld * ld;
    constexpr int EPP = 7;                                            // 1024 * 7 >= 87 * 72
    S pst[EPP];
#pragma unroll
    for (int q = 0; q < EPP; ++q) {
      const int e = tid + 1024 * q, c = e / D, r = e - c * D;
      pst[q] = P[(long)(15 + min(c, n - 1)) * ld + r];
    }
    // ---- A: Lam^ (lower tiles incl. row n): a wavefront per 16 x 16 tile, the B^ rows of the included tracks staged eight tracks at a time
    const double* Dg = d.Dg + (long)b * d.n_cap * DG_STRIDE;
    const int* order = d.trk_order + (long)b * f_cap;
    const int nt1 = (n1 + 15) >> 4;                                   // tiles per side of Lam^ (<= 5: 15 lower tiles)
    int ati = 0, atj = 0; bool atile = false;
    { int t = 0; for (int i = 0; i < nt1; ++i) for (int j = 0; j <= i; ++j, ++t) if (t == w) { ati = i; atj = j; atile = true; } }
    VD lacc = {0, 0, 0, 0};
    double dgt[4];                                                    // the tile's block-diagonal terms: loads issued now, used at the end of the phase
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 16 * ati + UsM<double>::row(lane, r), j = 16 * atj + (lane & 15);
      dgt[r] = (atile && i <= n && j <= i && !(i == n && j == n)) ? lam_diag_term(Dg, n, d.n_cap, i, j) : 0.0;
    }
    {
      double* sB = reinterpret_cast<double*>(sPH);                    // [24][16 nt1 + 1]
      const int nb1 = 16 * nt1 + 1;
      // the included tracks' row addresses and slot ranges once, through LDS (the P staging block is free until the end of this
      // phase): the staging loop below then is ONE level of global loads per round instead of three dependent ones
      int* sTrk = reinterpret_cast<int*>(sPT);                        // [npass][2]: track, first | last << 8
      for (int e = tid; e < npass; e += 1024) { const int t = order[e]; sTrk[2 * e] = t; sTrk[2 * e + 1] = d.trk_first[(long)b * f_cap + t]; }
      for (int t0 = 0; t0 < npass; t0 += 8) {
        __syncthreads();
        const int nt = min(8, npass - t0);
        // (24 x 81 entries at most: two per thread, both loads in flight before the first is stored)
        double sv[2]; int se[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int e = tid + 1024 * h, row = e / nb1, c = e - row * nb1, tl = row / 3, q = row - 3 * tl;
          double v = 0.0;
          if (e < 24 * nb1 && tl < nt && c < n1) {
            const int t = sTrk[2 * (t0 + tl)], fl = sTrk[2 * (t0 + tl) + 1], first = fl & 63, last = (fl >> 8) & 63;
            // k_feature writes a track's rows only inside its slot range and at column n: everything else reads as zero
            if (c == n || (c >= 6 * first && c < 6 * (last + 1))) v = d.trk_B[((long)b * f_cap + t) * 3 * ldR + (long)q * ldR + c];
          }
          sv[h] = v; se[h] = e;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) if (se[h] < 24 * nb1) sB[se[h]] = sv[h];
        __syncthreads();
        if (atile) {
          const int m = lane & 15, g = lane >> 4;
#pragma unroll
          for (int k = 0; k < 24; k += 4) lacc = UsM<double>::mma(sB[(k + g) * nb1 + 16 * ati + m], sB[(k + g) * nb1 + 16 * atj + m], lacc);
        }
      }
    }
    __syncthreads();
    if (atile) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * ati + UsM<double>::row(lane, r), j = 16 * atj + (lane & 15);
        if (i <= n && j <= i) {
          const double v = dgt[r] - lacc[r];
          sL[i * LL + j] = v;
          if (i == j) sD0[i] = v;
        }
      }
    }
    US_TICK(0);
    // P into LDS (the loads have had phase A to arrive)
#pragma unroll
    for (int q = 0; q < EPP; ++q) {
      const int e = tid + 1024 * q, c = e / D, r = e - c * D;
      if (c < n) sPT[c * Dp + r] = pst[q];
    }
    __syncthreads();
    // ---- B: [T | r_n] = chol(Lam^): rows 0 .. n (row n rides), in registers; the finished columns go back to sL (whose upper triangle is zero)
    {
      double x[US_UMAX][US_SEG_B];
      const int ncs = (n + US_SEG_B - 1) / US_SEG_B, R = n1, ntask = ncs * ((R + 63) >> 6);
#pragma unroll
      for (int u = 0; u < US_UMAX; ++u) {
        const int t = w + 16 * u, cs = t % ncs, i = 64 * (t / ncs) + lane;
#pragma unroll
        for (int jj = 0; jj < US_SEG_B; ++jj) { const int j = US_SEG_B * cs + jj; x[u][jj] = (t < ntask && i < R && j < n && j <= i) ? sL[i * LL + j] : 0.0; }
      }
      __syncthreads();
      US_TICK(1);
      us_chol_tall<double, US_SEG_B, true>(x, n, R, ncs, ntask, w, lane, sColD, cstride, sD0, sFlag, [&](int i, int k, double v) { sL[i * LL + k] = v; });
    }
    US_TICK(2);
    if (tid == 0) st[STAT_RROWS] = n - sFlag[0];
    // ---- C: PHt = P[:, 15:] T^T = Pc L: tile (ti, tj) sums over c >= 16 tj (L is lower triangular)
    {
      const int ntr = (D + 15) >> 4, ntc = (n + 15) >> 4;
      for (int t = w; t < ntr * ntc; t += 16) {
        const int ti = t / ntc, tj = t - ti * ntc;
        const VS acc = us_tile<S>(ti, tj, 16 * tj, n, lane,
                                  [&](int r, int c) -> S { return (r < D && c < n) ? sPT[c * Dp + r] : S(0); },
                                  [&](int c, int k) -> S { return (c < n && k < n) ? (S)sL[c * LL + k] : S(0); });
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int i = 16 * ti + UsM<S>::row(lane, r), k = 16 * tj + (lane & 15); if (i < D && k < n) sPH[i * LS + k] = acc[r]; }
      }
    }
    __syncthreads();
    US_TICK(3);
    // ---- D: S = T PHt[15:, :] + sigma^2 I, lower tiles, into the P staging block as [i][LS]
    const S sig2 = d.prm[(long)b * PRM_STRIDE + PRM_SIG2];
    S* sS = sPT;
    {
      const int ntc = (n + 15) >> 4;
      int t = 0;
      for (int ti = 0; ti < ntc; ++ti)
        for (int tj = 0; tj <= ti; ++tj, ++t) {
          if ((t & 15) != w) continue;
          const VS acc = us_tile<S>(ti, tj, 16 * ti, n, lane,
                                    [&](int i, int c) -> S { return (i < n && c < n) ? (S)sL[c * LL + i] : S(0); },
                                    [&](int c, int j) -> S { return (c < n && j < n) ? sPH[(15 + c) * LS + j] : S(0); });
          // (sS aliases the P staging block, which phase C has finished reading; this phase reads sL and sPH only)
#pragma unroll
          for (int r = 0; r < 4; ++r) { const int i = 16 * ti + UsM<S>::row(lane, r), j = 16 * tj + (lane & 15); if (i < n && j <= i) sS[i * LS + j] = acc[r] + (i == j ? sig2 : S(0)); }
        }
    }
    __syncthreads();
    US_TICK(4);
    // ---- E: S = L L^T in registers, rows n .. n + D - 1 = PHt and row n + D = r_n^T riding: W replaces PHt in place, z in sZ
    {
      S x[US_UMAX][US_SEG_E];
      const int ncs = (n + US_SEG_E - 1) / US_SEG_E, R = n + D + 1, ntask = ncs * ((R + 63) >> 6);
#pragma unroll
      for (int u = 0; u < US_UMAX; ++u) {
        const int t = w + 16 * u, cs = t % ncs, i = 64 * (t / ncs) + lane;
#pragma unroll
        for (int jj = 0; jj < US_SEG_E; ++jj) {
          const int j = US_SEG_E * cs + jj;
          S v = S(0);
          if (t < ntask && j < n) {
            if (i < n) v = j <= i ? sS[i * LS + j] : S(0);
            else if (i < n + D) v = sPH[(i - n) * LS + j];
            else if (i == n + D) v = (S)sL[n * LL + j];
          }
          x[u][jj] = v;
        }
      }
      __syncthreads();
      US_TICK(5);
      S* sColS = reinterpret_cast<S*>(sColD);
      us_chol_tall<S, US_SEG_E, false>(x, n, R, ncs, ntask, w, lane, sColS, cstride, sD0, sFlag,
                                       [&](int i, int k, S v) { if (i >= n + D) sZ[k] = v; else if (i >= n) sPH[(i - n) * LS + k] = v; });
    }
    US_TICK(6);
    if (tid == 0 && sFlag[1]) atomicOr(&st[STAT_ERR], STAT_ERR_PIVOT);
    // ---- G: dx = W z, injected into the state
    for (int r = tid; r < D; r += 1024) {
      S a = 0;
      for (int k = 0; k < n; ++k) a += sPH[r * LS + k] * sZ[k];
      sdx[r] = a;
      d.dx[(long)b * ld + r] = a;
    }
  }
  US_TICK(7);
  // ---- H: P <- P - W W^T, lower tiles on the matrix cores, both triangles written from the same value.  With the frame's prune
  // riding along, rows / columns of the nd_ oldest camera states vanish and later ones move up by 6 nd_, into the other buffer
  {
    const S* Pr = d.P + (long)b * ld * ld;
    S* Po = (fused_prune ? d.Pout : d.P) + (long)b * ld * ld;
    const int ntr = (D + 15) >> 4, cut = 15 + 6 * nd_, kend = upd ? n : 0;
    int t = 0;
    for (int ti = 0; ti < ntr; ++ti)
      for (int tj = 0; tj <= ti; ++tj, ++t) {
        if ((t & 15) != w) continue;
        S pv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int i = min(16 * ti + UsM<S>::row(lane, r), D - 1), j = min(16 * tj + (lane & 15), D - 1); pv[r] = Pr[(long)j * ld + i]; }
        const VS acc = us_tile<S>(ti, tj, 0, kend, lane,
                                  [&](int i, int k) -> S { return (i < D && k < n) ? sPH[i * LS + k] : S(0); },
                                  [&](int k, int j) -> S { return (j < D && k < n) ? sPH[j * LS + k] : S(0); });
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 16 * ti + UsM<S>::row(lane, r), j = 16 * tj + (lane & 15);
          const bool keep = (i < 15 || i >= cut) && (j < 15 || j >= cut);
          if (i < D && j <= i && keep) {
            const int di = i < 15 ? i : i - 6 * nd_, dj = j < 15 ? j : j - 6 * nd_;
            const S v = pv[r] - acc[r];
            Po[(long)dj * ld + di] = v; Po[(long)di * ld + dj] = v;
          }
        }
      }
  }
  __syncthreads();
  US_TICK(8);
  if (upd && tid < 256) inject_from_dx<S>(d, b, tid, 256, sdx);
  if (fused_prune) {
    __syncthreads();   // the camera states just corrected are compacted by other threads
    prune_bookkeeping<S>(d, b, tid, N, N - nd_, nd_, 0);
  }
  US_TICK(9);
}

static bool update_small_ok = false;   // the 160 KB attribute was granted (kalman_device_setup)
size_t update_small_lds_bytes(int n_max, int f_cap, size_t scalar) {
  const size_t n1 = (size_t)n_max + 1, LL = n1 | 1, Dm = 15 + (size_t)n_max, cstride = 2 * (size_t)n_max + 20;
  return (n1 * LL + n1 + 1 + 2 * cstride) * sizeof(double) + (update_small_pt_elems(n_max, f_cap, scalar) + update_small_ph_elems(n_max, scalar) + n_max + 2 + Dm + 1) * scalar + 64;
}
template <class S>
bool launch_update_small(const Dev<S>& d, int b0, int nb, hipStream_t st, int n_max) {
  if (nb <= 0) return true;
  const size_t lds = update_small_lds_bytes(n_max, d.f_cap, sizeof(S));
  // tasks of the factorizations: ceil(n / 12) x ceil((2 n + 16) / 64) and ceil(n / 4) x ceil((n + 1) / 64) <= 32; tiles of Lam^: <= 15 of the 16 wavefronts
  const int ncs = (n_max + US_SEG_E - 1) / US_SEG_E, nrc = (2 * n_max + 16 + 63) / 64, nt1 = (n_max + 1 + 15) / 16;
  const int nrcb = (n_max + 1 + 63) / 64;
  const bool b4 = ((n_max + 3) / 4) * nrcb <= 16 * US_UMAX, b8 = ((n_max + 7) / 8) * nrcb <= 16 * US_UMAX;
  if (!update_small_ok || lds > 156 * 1024 || ncs * nrc > 16 * US_UMAX || !(b4 || b8) || nt1 * (nt1 + 1) / 2 > 16 || (15 + n_max) * n_max > 1024 * 7) return false;
  if (b4) hipLaunchKernelGGL((k_update_small<S, 4>), dim3(nb), dim3(1024), lds, st, d, b0, n_max);
  else hipLaunchKernelGGL((k_update_small<S, 8>), dim3(nb), dim3(1024), lds, st, d, b0, n_max);
  return true;
}
template bool launch_update_small<float>(const Dev<float>&, int, int, hipStream_t, int);
template bool launch_update_small<double>(const Dev<double>&, int, int, hipStream_t, int);

template <class S>
void launch_kalman(const Dev<S>& d, int b0, int nb, hipStream_t st) {
  if (nb <= 0) return;
  const int n = d.n6cap, D = 15 + n;
  // S: formed inside the blocked gain solve where that runs (float, square-root gain form, windows up to 32 cameras); nothing
  // then reads the column-major PHt, so only its row-major copy is written
  const bool s_fused = sizeof(S) == 4 && d.joseph == 0 && d.gain_fused_s && (n + 15) / 16 <= 12;
  if (s_fused) gemm_pht_rowmajor(d, b0, nb, D, n, st); else gemm<S, OP_PHT>(d, b0, nb, D, n, st);
  if (!s_fused) gemm<S, OP_S>(d, b0, nb, n, n, st);
  const int nbn = (n + 15) / 16;
  const int nbn_max = sizeof(S) == 4 ? 12 : 8;
  const size_t lds_chol = (size_t)n * (n + 1) * sizeof(S);
  auto chol_inv = [&]() {
    if (lds_chol <= 150 * 1024) hipLaunchKernelGGL((k_chol_inv<S, true>), dim3(nb), dim3(256), lds_chol, st, d, b0);
    else hipLaunchKernelGGL((k_chol_inv<S, false>), dim3(nb), dim3(256), 0, st, d, b0);
  };
  if (d.joseph != 1) {
    // ---- square-root gain form: W = PHt L^-T, dx = W L^-1 r_n, P <- P - W W^T
    if (d.joseph == 0 && gain_blocked(d, b0, nb, st)) {}   // float: blocked matrix-core Cholesky with appended rows
    else if (nbn <= 4) launch_gain_w<S, 4>(d, b0, nb, st);
    else if (nbn <= 8) launch_gain_w<S, 8>(d, b0, nb, st);
    else if (nbn <= nbn_max) launch_gain_w<S, (sizeof(S) == 4 ? 12 : 8)>(d, b0, nb, st);
    else if (d.joseph == 0 && nbn > 12 && launch_chol_gain_large<S>(d, b0, nb, st)) {}   // two-level blocked factorization + row solves
    else {
      chol_inv();
      gemm<S, OP_W>(d, b0, nb, D, n, st);
      hipLaunchKernelGGL((k_dx_w<S>), dim3(nb), dim3(256), (size_t)2 * d.n6cap * sizeof(S), st, d, b0);
    }
    gemm<S, OP_DOWN>(d, b0, nb, D, D, st);   // its first tile's workgroup also applies dx to the state (inject_from_dx)
    return;
  }
  // ---- Joseph form (the reference's literal sequence)
  // K = PHt S^-1: register-resident factor/solve when the window fits the 16x16 thread grid, otherwise
  // Cholesky + triangular inverse (LDS or global) followed by two GEMMs.
  if (nbn <= nbn_max) {
    if (nbn <= 4) hipLaunchKernelGGL((k_gain<S, 4>), dim3(nb), dim3(256), 0, st, d, b0);
    else if (nbn <= 8) hipLaunchKernelGGL((k_gain<S, 8>), dim3(nb), dim3(256), 0, st, d, b0);
    else hipLaunchKernelGGL((k_gain<S, 12>), dim3(nb), dim3(256), 0, st, d, b0);
  } else {
    chol_inv();
    gemm<S, OP_W>(d, b0, nb, D, n, st);
    gemm<S, OP_K>(d, b0, nb, D, n, st);
  }
  if (nbn <= nbn_max) hipLaunchKernelGGL((k_inject<S, true>), dim3(nb), dim3(256), (size_t)d.ld * sizeof(S), st, d, b0);
  else hipLaunchKernelGGL((k_inject<S, false>), dim3(nb), dim3(256), (size_t)(d.ld + d.n6cap) * sizeof(S), st, d, b0);
  gemm<S, OP_A>(d, b0, nb, D, D, st);
  gemm<S, OP_AP>(d, b0, nb, D, D, st);
  gemm<S, OP_X>(d, b0, nb, D, D, st);
  if (sizeof(S) != 4) hipLaunchKernelGGL((k_symmetrize<S, false>), dim3(16, nb), dim3(256), 0, st, d, b0);   // float: fused into the X product
}

void kalman_device_setup() {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_chol_inv<float, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_chol_inv<double, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  update_small_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(k_update_small<float, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess &&
                    hipFuncSetAttribute(reinterpret_cast<const void*>(k_update_small<double, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess &&
                    hipFuncSetAttribute(reinterpret_cast<const void*>(k_update_small<float, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess &&
                    hipFuncSetAttribute(reinterpret_cast<const void*>(k_update_small<double, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
  (void)hipGetLastError();
}

template void launch_kalman<float>(const Dev<float>&, int, int, hipStream_t);
template void launch_kalman<double>(const Dev<double>&, int, int, hipStream_t);

}  // namespace msckf
