// kernels_chol.hip -- blocked right-looking Cholesky on the matrix cores, one workgroup (16 wavefronts) per matrix.
//
// Two uses, one kernel template:
//   GRAM (T = double): [T_H | r_n] = chol(Lam^), Lam^ = [H_o | r_o]^T [H_o | r_o] accumulated by k_gram -- the
//        compression of the stacked Jacobian (HouseholderQR + Q_1^T r_o of measurementUpdate, msckf.h:1338-1366) in
//        information form; semi-definite pivot skipping (a direction the stack says nothing about gives
//        a zero row of T_H).
//   GAIN (T = float):  S = L L^T for S = T_H P T_H^T + R_n (msckf.h:1369) with the rows [P T_H^T ; r_n^T] appended, which
//        the factorization turns into W = P T_H^T L^-T and z^T = (L^-1 r_n)^T; dx = K r_n = W z (msckf.h:1370-1373 without
//        the explicit inverse).  NPART workgroups per trajectory share the appended rows (each redoes the factorization).
//
// The trailing matrix lives in MFMA accumulators: 16 x 16 blocks, 4 x 4 block-cyclic over the sixteen wavefronts.  Per panel
// of 16 columns:
//   (1) the owners drop the panel's blocks into LDS;
//   (2) ONE wavefront factors the 16 x 16 diagonal block with lane = row, pivots and multipliers broadcast by v_readlane,
//       and runs the SAME eliminations on the rows of the identity: that yields M with  y = x M  for the forward
//       substitution of any row x against the block (zero columns for skipped pivots included);
//   (3) all wavefronts form the panel below the diagonal as L21 = A21 M on the matrix cores (the per-row substitution of
//       round 2's kernel -- 136 dependent FMAs fed by LDS reads per thread -- is gone);
//   (4) finished rows of T_H / columns of W go to global memory, dx accumulates;
//   (5) rank-16 update of the trailing blocks, operands straight from the LDS panel.
// 12 panels x 4 barriers instead of 180 steps x (LDS exchange + barrier); the O(n^3) part runs at MFMA rate.
#include <utility>

#include "dev_common.h"


namespace msckf {

// max(x, lo) as ONE instruction (v_max; NaN -> lo): the library fmax goes through a canonicalizing v_max x, x first, and this
// sits on the pivot chain
__device__ __forceinline__ float clamp_min(float x, float lo) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(lo)); return r; }
__device__ __forceinline__ double clamp_min(double x, double lo) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(x), "v"(lo)); return r; }

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

// CH_GRAM_A / CH_GRAM_B: the two levels of the factorization of a Gram matrix with more than 192 columns (windows of
// more than 31 cameras): A factors the leading 192 x 192 block in place (f64 L11 and the per-panel M kept for k_trsm_l21),
// k_trsm_l21 forms L21 = A21 L11^-T one 16-row block per wavefront, B applies L21 L21^T to its accumulators ("pre-panels")
// and factors the Schur complement.
// CH_S_A / CH_S_B: the same two levels for S = T_H P T_H^T + R_n of a window with more than 32 cameras (n > 192): L (in
// place in Smat's lower triangle) and every panel's M are kept, and W = P T_H^T L^-T, z = L^-1 r_n follow from k_trsm_rows
// over all the panels (rows are independent: one 16-row block per wavefront) instead of riding along as appended rows.
enum { CH_GRAM = 0, CH_GAIN = 1, CH_GRAM_A = 2, CH_GRAM_B = 3, CH_S_A = 4, CH_S_B = 5 };
constexpr int CH_SPLIT = 192;   // columns of level A

#ifdef MSCKF_ABLATE
// phase timers of the -DMSCKF_ABLATE build: shader-clock cycles of workgroup 0's thread 0 per phase, summed over launches
// [mode][0 load, 1 panel->LDS, 2 diagonal block, 3 L21, 4 outputs, 5 trailing update, 6 launches]
__device__ unsigned long long g_chol_cycles[2][8];
// GAIN, the load phase in detail, per part: [part][0 own S blocks formed, 1 published, 2 rendezvous, 3 siblings' blocks read, 4 rest of the set-up, 5 launches]
__device__ unsigned long long g_chol_sub[4][8];
#define CH_SUB(slot) do { if (MODE == CH_GAIN && tid == 0 && bi_ == 0) { const long long t_ = clock64(); atomicAdd(&g_chol_sub[part & 3][slot], (unsigned long long)(t_ - tsub)); tsub = t_; } } while (0)
__device__ int g_chol_dbg = 0;   // ablation: 1 the other wavefronts skip outputs / trailing update while wavefront 0 factors the next diagonal block (wrong results, timing only)
#define CH_TICK(slot) do { if (tid == 0) { const long long t_ = clock64(); cyc[slot] += t_ - tlast; tlast = t_; } } while (0)
#else
#define CH_TICK(slot) do {} while (0)
#define CH_SUB(slot) do {} while (0)
#endif

// NB: 16-column blocks of the factored matrix; NA: appended 16-row blocks held by ONE workgroup (GAIN), 0 for GRAM
template <class T, class SO, int NB, int NA, int MODE, int NPART>
__global__ __launch_bounds__(1024) void k_chol_mfma(Dev<SO> d, int b0, int nb) {
  typedef typename Mf<T>::V V;
  constexpr int NR = NB + NA, HR = (NR + 3) / 4, HC = (NB + 3) / 4, LP = 17;   // sixteen wavefronts: block (i, j) belongs to wavefront 4 (i & 3) + (j & 3)
  // GAIN: the NPART workgroups of a trajectory read the same T and P T_H^T: on one XCD (xcd_item), whose L2 then serves three of the four
  int bi_ = (int)blockIdx.x, part = 0;
  if (MODE == CH_GAIN && !xcd_item(nb, NPART, bi_, part)) return;
  const int b = b0 + bi_;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int pi = w >> 2, pj = w & 3;
  // which part of a trajectory forms the S blocks of register set (ii, jj) when the parts share the product (gain_fused_s == 2):
  // block row set 0 (up to twelve k-blocks per block) alone on part 0, the two sets of row set 1 on parts 1 and 2, row set 2's
  // three (few k-blocks each) on parts 1, 2, 3
  auto s_owner = [](int ii, int jj) -> int { return (ii == 0 ? 0 : (ii == 1 ? 1 + jj : (jj == 2 ? 3 : 1 + jj))) % NPART; };
  // wavefront 0 carries the 16-pivot chains of the diagonal blocks: it wins instruction arbitration against the three
  // wavefronts that share its SIMD (and everybody against co-resident throughput waves of another slice)
  if (w == 0) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(1);
  int* st = d.stats + (long)b * STAT_STRIDE;
  const int mrows_ = st[STAT_MROWS], ncam_ = d.ncam[b];   // independent scalar loads, one wait
  if (mrows_ == 0) return;
  constexpr bool GRAMLIKE = MODE == CH_GRAM || MODE == CH_GRAM_A || MODE == CH_GRAM_B;
  constexpr bool SLIKE = MODE == CH_S_A || MODE == CH_S_B;
  constexpr bool LEVEL_A = MODE == CH_GRAM_A || MODE == CH_S_A, LEVEL_B = MODE == CH_GRAM_B || MODE == CH_S_B;
  constexpr int OFF = LEVEL_B ? CH_SPLIT : 0;                            // first row/column of this launch's block
  const int N = ncam_, nfull = 6 * N, D = 15 + nfull;
  // n: columns this launch factors (local); main_rows: rows of the main block that exist: GRAM n + 1 (row n = H_o^T r_o),
  // GAIN n; appended rows of this part (GAIN): a slice of the D rows of P T_H^T plus, as the LAST row of the slice's
  // blocks, r_n^T
  const int n = LEVEL_A ? min(nfull, CH_SPLIT) : nfull - OFF;
  const int main_rows = !GRAMLIKE ? n : (LEVEL_A ? min(nfull + 1, CH_SPLIT) : nfull + 1 - OFF);
  if (LEVEL_B && n <= 0) return;
  const bool two_level = MODE == CH_GRAM_A && nfull + 1 > CH_SPLIT;      // level A of a two-level factorization
  const int app_per = MODE == CH_GAIN ? 16 * NA - 1 : 0;              // rows of PHt per part (the last appended row is r_n^T)
  const int app_lo = part * app_per, app_hi = min(D, app_lo + app_per);   // [app_lo, app_hi) rows of PHt
  const int zrow = 16 * NR - 1;                                          // panel row of r_n^T
  __shared__ T sPP[2][16 * NR][LP];   // panels p & 1: every row, 16 columns (two, for the lookahead)
  __shared__ T sMM[2][16][LP];        // M: y = x M solves y L11^T = x (zero columns for skipped pivots)
  __shared__ T sD0[16 * NB];      // GRAM: original diagonal (pivot tolerance)

  // ---- sources
  const double* Lam = nullptr; const double* Dg = nullptr;
  SO* Sm = nullptr; const SO* PHtT = nullptr; const SO* R0 = d.Rbuf + ((long)b * d.nchunk) * (long)d.n6cap * d.ldR;
  if (GRAMLIKE) { Lam = d.Lam + (long)b * d.ldR * d.ldR; Dg = d.Dg + (long)b * d.n_cap * DG_STRIDE; }
  else { Sm = d.Smat + (long)b * d.n6cap * d.n6cap; PHtT = d.K + (long)b * d.ld * d.n6cap; }
  // Initial accumulators in two passes: first every block's loads, raw, from clamped (always valid) addresses straight into
  // the accumulator registers -- nothing uses a loaded value, so no wait separates the blocks and the whole matrix is in
  // flight at once (one memory round trip instead of one per block: the load phase was 25 k / 48 k cycles of ~200 k) --
  // then the masks (rows / columns beyond the window read as zero).
  auto el_ptr = [&](int row, int col) -> const T* {
    if (GRAMLIKE) {
      const int I = OFF + row, J = OFF + col, hi = I >= J ? I : J, lo = I >= J ? J : I;
      return reinterpret_cast<const T*>(Lam) + (long)min(hi, d.ldR - 1) * d.ldR + min(lo, d.ldR - 1);
    }
    const int cc = min(OFF + col, d.n6cap - 1);
    if (SLIKE) return reinterpret_cast<const T*>(Sm) + (long)min(OFF + row, d.n6cap - 1) * d.n6cap + cc;
    if (row < 16 * NB) return reinterpret_cast<const T*>(Sm) + (long)min(row, d.n6cap - 1) * d.n6cap + cc;   // S (OP_S wrote both triangles): lanes run along col
    const int ar = min(app_lo + row - 16 * NB, d.ld - 1);
    const T* pw = reinterpret_cast<const T*>(PHtT) + (long)ar * d.n6cap + cc;      // row-major copy of P T_H^T left by OP_PHT
    const T* pz = reinterpret_cast<const T*>(R0) + (long)cc * d.ldR + min(nfull, d.ldR - 1);   // r_n[col]
    return row == zrow ? pz : pw;
  };
  auto el_ok = [&](int row, int col) -> bool {
    if (GRAMLIKE) { const int I = OFF + row, J = OFF + col, hi = I >= J ? I : J, lo = I >= J ? J : I; return hi <= nfull && lo < nfull; }
    if (SLIKE) return row < n && col < n;
    if (col >= n) return false;
    if (row < 16 * NB) return row < n;
    return row == zrow || app_lo + row - 16 * NB < app_hi;
  };
  static_assert(GRAMLIKE ? sizeof(T) == 8 : sizeof(T) == sizeof(SO), "the accumulators are loaded without conversion");
#ifdef MSCKF_ABLATE
  long long cyc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = clock64(), tsub = tlast;
#endif
  // S = T_H (P T_H^T)[15:, :] + sigma^2 I (msckf.h:1369) is formed HERE, straight into the accumulators, instead of being
  // loaded: block (i, j) = sum over the k-blocks kb >= i (T_H is upper triangular) of T(16 i .., k) PHt(15 + k, 16 j ..),
  // four MFMAs per k-block, operands from global memory (T rows as one 16-byte load per lane, PHt through its
  // row-major copy).  The S GEMM was a launch of its own (24 us, bound by its start-up and drain).
  auto s_block = [&](int i, int j) -> V {
    V out = V{0, 0, 0, 0};
    if constexpr (MODE == CH_GAIN) {
      const int rr = lane & 15, gq = lane >> 4;
      const T* Trow = reinterpret_cast<const T*>(R0) + (long)min(16 * i + rr, d.n6cap - 1) * d.ldR;
      const T* Bcol = reinterpret_cast<const T*>(PHtT) + min(16 * j + rr, d.n6cap - 1);
      V sacc = V{0, 0, 0, 0};
      typedef T t4 __attribute__((ext_vector_type(4)));
      const int kbmax = min(NB, (n + 15) >> 4);
      // six k-blocks per pass: their thirty loads are issued together (clamped addresses, masks afterwards), then the MFMAs
      constexpr int KU = 6;
      for (int kb0 = i; kb0 < kbmax; kb0 += KU) {
        t4 a4[KU]; T bv[KU][4];
#pragma unroll
        for (int u = 0; u < KU; ++u) {
          const int k0 = 16 * min(kb0 + u, NB - 1) + 4 * gq;
          a4[u] = *reinterpret_cast<const t4*>(Trow + k0);
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) bv[u][s4] = Bcol[(long)min(15 + k0 + s4, d.ld - 1) * d.n6cap];
        }
#pragma unroll
        for (int u = 0; u < KU; ++u) {
          if (kb0 + u >= kbmax) break;
          const int k0 = 16 * (kb0 + u) + 4 * gq;
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) {
            // T_H is upper triangular: the part of row 16 i + rr left of its diagonal is masked here rather than trusted to be an
            // exact zero in whatever route produced Rbuf (the k-blocks start at kb = i, so only the diagonal block has such entries)
            const bool ok = k0 + s4 < n, upper = k0 + s4 >= 16 * i + rr;   // (the A operand's row is 16 i + rr; B's lane index is a column)
            sacc = Mf<T>::mma(ok && upper ? a4[u][s4] : T(0), ok ? bv[u][s4] : T(0), sacc);
          }
        }
      }
      const T sg2 = (T)d.prm[(long)b * PRM_STRIDE + PRM_SIG2];
#pragma unroll
      for (int r = 0; r < 4; ++r) out[r] = sacc[r] + ((i == j && Mf<T>::row(lane, r) == (lane & 15)) ? sg2 : T(0));
    }
    return out;
  };
  V acc[HR][HC];
#pragma unroll
  for (int ii = 0; ii < HR; ++ii)
#pragma unroll
    for (int jj = 0; jj < HC; ++jj) {
      acc[ii][jj] = V{0, 0, 0, 0};
      if (4 * ii + 3 < 4 * jj && 4 * ii + 3 < NB) continue;   // compile-time: main block row above the column block
      const int i = 4 * ii + pi, j = 4 * jj + pj;
      if (i >= NR || j >= NB || (i < NB && j > i) || 16 * j >= n) continue;
      if (i < NB && 16 * i >= main_rows) continue;
      if (MODE == CH_GAIN && d.gain_fused_s >= 2 && i < NB && s_owner(ii, jj) != part) continue;   // a sibling part forms this block
      if (MODE == CH_GAIN && d.gain_fused_s && i < NB) { acc[ii][jj] = s_block(i, j); continue; }
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[ii][jj][r] = *el_ptr(16 * i + Mf<T>::row(lane, r), 16 * j + (lane & 15));
      // split-K SYRK (kernels_gram.hip): the tiles of block column j / 4 came in min(j / 4 + P - 2, P) partial sums (P =
      // d.gram_parts), lam_part apart; added here in a fixed order (the loads are as unconditional as the ones above: more
      // round trips, no waits between)
      if (MODE == CH_GRAM && d.gram_parts >= 3) {
        const int ncopy = min(jj + d.gram_parts - 2, d.gram_parts);
#pragma unroll
        for (int c = 1; c < 4; ++c)
          if (c < ncopy) {
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[ii][jj][r] += *(el_ptr(16 * i + Mf<T>::row(lane, r), 16 * j + (lane & 15)) + c * d.lam_part);
          }
      }
    }
  // (Measured and rejected, round 4: the 78 blocks dealt out by cost over the 64 wavefronts of the four parts whatever their
  // accumulator layout -- no wavefront with more than two passes of six k-blocks, every part equally loaded: forming + publishing
  // takes the same ~20 k cycles -- 48 KB of operands per pass through one CU's vector memory path and the write-through of the
  // published blocks, not the number of passes, set it; profiles/r04_o_*.)
  // The product S is MFMA-bound on the one CU a part runs on (11.6 MFLOP against 256 FLOP/cycle: ~46 k cycles when every part
  // forms all of it).  Split: a part forms the blocks s_owner() gives it (the costly block rows -- many k-blocks -- spread over
  // the parts), publishes them in Smat with agent-scope stores (write-through: the siblings may sit on another XCD), the parts
  // of the trajectory meet at a counter barrier, and each reads the blocks it did not form.
  CH_SUB(0);
  if constexpr (MODE == CH_GAIN) {
    if (d.gain_fused_s >= 2) {
      T* Sg = reinterpret_cast<T*>(Sm);
#pragma unroll
      for (int ii = 0; ii < HR; ++ii)
#pragma unroll
        for (int jj = 0; jj < HC; ++jj) {
          if (4 * ii + 3 < 4 * jj && 4 * ii + 3 < NB) continue;
          const int i = 4 * ii + pi, j = 4 * jj + pj;
          if (i >= NB || j > i || 16 * j >= n || 16 * i >= main_rows || s_owner(ii, jj) != part) continue;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * i + Mf<T>::row(lane, r), col = 16 * j + (lane & 15);
            if (row < d.n6cap && col < d.n6cap) __hip_atomic_store(Sg + (long)row * d.n6cap + col, acc[ii][jj][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this thread's stores have been acknowledged
      __syncthreads();
      CH_SUB(1);
      // The wait is BOUNDED: nothing guarantees that the siblings are resident (another process or slice may hold the CUs they
      // need while this part holds its own), so after 20 000 polls (agent-scope load + s_sleep: some tens of milliseconds)
      // without them this part forms the missing blocks itself -- the same instruction sequence its sibling would have run, the
      // same bits; the split is a speed-up, never a dependency (tests force the fall-back with gain_fused_s == 3).
      __shared__ int s_all_here;
      if (tid == 0) {
        unsigned* bar = d.gain_bar + (long)b * 32;
        // the counter advances by a power of two per launch (three parts: part 0 counts twice), so that launches stay aligned
        // with the multiples of GP across the 2^32 wrap
        constexpr unsigned GP = NPART == 3 ? 4u : (unsigned)NPART;
        static_assert((GP & (GP - 1)) == 0, "rendezvous group size must divide 2^32");
        const unsigned old = atomicAdd(bar, (NPART == 3 && part == 0) ? 2u : 1u);
        const unsigned target = (old / GP + 1u) * GP;
        int here = 0;
        const int max_spin = d.gain_fused_s == 3 ? 0 : 20000;   // 3: test hook -- never wait, so that the fall-back runs
        for (int spin = 0; spin <= max_spin; ++spin) {
          if ((int)(__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) >= 0) { here = 1; break; }
          __builtin_amdgcn_s_sleep(2);
        }
        s_all_here = here;
      }
      __syncthreads();
      CH_SUB(2);
      const bool all_here = s_all_here != 0;
#pragma unroll
      for (int ii = 0; ii < HR; ++ii)
#pragma unroll
        for (int jj = 0; jj < HC; ++jj) {
          if (4 * ii + 3 < 4 * jj && 4 * ii + 3 < NB) continue;
          const int i = 4 * ii + pi, j = 4 * jj + pj;
          if (i >= NB || j > i || 16 * j >= n || 16 * i >= main_rows || s_owner(ii, jj) == part) continue;
          if (!all_here) { acc[ii][jj] = s_block(i, j); continue; }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = min(16 * i + Mf<T>::row(lane, r), d.n6cap - 1), col = min(16 * j + (lane & 15), d.n6cap - 1);
            acc[ii][jj][r] = __hip_atomic_load(Sg + (long)row * d.n6cap + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
    }
  }
#ifdef MSCKF_ABLATE
  if (MODE == CH_GAIN) { __builtin_amdgcn_s_waitcnt(0); }
#endif
  CH_SUB(3);
#pragma unroll
  for (int ii = 0; ii < HR; ++ii)
#pragma unroll
    for (int jj = 0; jj < HC; ++jj) {
      if (4 * ii + 3 < 4 * jj && 4 * ii + 3 < NB) continue;
      const int i = 4 * ii + pi, j = 4 * jj + pj;
      if (i >= NR || j >= NB || (i < NB && j > i) || 16 * j >= n) continue;
      if (i < NB && 16 * i >= main_rows) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) if (!el_ok(16 * i + Mf<T>::row(lane, r), 16 * j + (lane & 15))) acc[ii][jj][r] = T(0);
    }
  if (GRAMLIKE)
    for (int t = tid; t < 16 * NB; t += 1024) {
      double dv = lam_hat(Lam, Dg, d.ldR, nfull, d.n_cap, OFF + t, OFF + t);
      if (MODE == CH_GRAM && d.gram_parts >= 3) {
        const int ncopy = min(t / 64 + d.gram_parts - 2, d.gram_parts);
        for (int c = 1; c < ncopy; ++c) dv += lam_hat(Lam + c * d.lam_part, Dg, d.ldR, nfull, d.n_cap, t, t);
      }
      sD0[t] = t < n ? (T)dv : T(0);
    }
  const T tol = T(64.0 * 2.220446049250313e-16);
  int nskip = 0;
  bool badpiv = false;   // factorization of S: a pivot that is not positive (P or S lost positive definiteness) -> STAT_ERR bit 2
  T dxacc = 0;   // GAIN: thread t < app rows accumulates dx[app_lo + t] = sum_k W(., k) z_k
  SO* Rt = d.Rbuf + ((long)b * d.nchunk) * (long)d.n6cap * d.ldR;
  SO* Wg = d.W + (long)b * d.ld * d.n6cap;

  // ---- level B: acc -= L21 L21^T, the CH_SPLIT columns of L21 (left in Lam by k_trsm_l21) staged 16 at a time
  if (LEVEL_B) {
    T (*sP)[LP] = sPP[0];
    for (int q = 0; q < CH_SPLIT / 16; ++q) {
      __syncthreads();
      for (int e = tid; e < 16 * NB * 16; e += 1024) {
        const int r = e >> 4, c = e & 15;
        if (SLIKE) sP[r][c] = OFF + r < nfull ? (T)Sm[(long)(OFF + r) * d.n6cap + 16 * q + c] : T(0);
        else {
          const double lv = Lam[(long)min(OFF + r, d.ldR - 1) * d.ldR + 16 * q + c];
          sP[r][c] = OFF + r <= nfull ? (T)lv : T(0);
        }
      }
      __syncthreads();
#pragma unroll
      for (int jj = 0; jj < HC; ++jj) {
        const int j = 4 * jj + pj;
        if (j >= NB || 16 * j >= n) continue;
        T bq[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) bq[s4] = sP[16 * j + (lane & 15)][4 * s4 + (lane >> 4)];
#pragma unroll
        for (int ii = 0; ii < HR; ++ii) {
          if (4 * ii + 3 < 4 * jj && 4 * ii + 3 < NB) continue;
          const int i = 4 * ii + pi;
          if (i >= NR || i < j || 16 * i >= main_rows) continue;
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) {
            const T a = -sP[16 * i + (lane & 15)][4 * s4 + (lane >> 4)];
            acc[ii][jj] = Mf<T>::mma(a, bq[s4], acc[ii][jj]);
          }
        }
      }
    }
  }

  // ---- the panel loop with one panel of lookahead: while the other wavefronts apply panel p to the trailing blocks right of
  // column p + 1, wavefront 0 already factors the diagonal block of panel p + 1 (its column was updated and dropped into
  // the second LDS panel first).  The 16-pivot chain of the diagonal block is the longest phase of a panel; it no longer
  // waits for the matrix-core update, and the update no longer waits for it.
  // (1) blocks (i, p), i >= p, from the accumulators to LDS panel p & 1
  auto drop = [&](auto pc, auto whichc) __attribute__((always_inline)) {
    constexpr int p = decltype(pc)::value;
    constexpr int WHICH = decltype(whichc)::value;        // 0 all blocks (i, p), i >= p; 1 the diagonal block only; 2 the blocks below it
    T (*sP)[LP] = sPP[p & 1];
    if (pj == (p & 3)) {
#pragma unroll
      for (int ii = 0; ii < HR; ++ii) {
        if (4 * ii + 3 < p && 4 * ii + 3 < NB) continue;   // compile-time: block row above the panel
        const int i = 4 * ii + pi;
        if (i < p || i >= NR) continue;
        if (WHICH == 1 && i != p) continue;
        if (WHICH == 2 && i == p) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) sP[16 * i + Mf<T>::row(lane, r)][lane & 15] = acc[ii][p >> 2][r];
      }
    }
  };
  // (2) diagonal block on ONE wavefront, all 64 lanes busy: lane (r = lane & 15, g = lane >> 4) holds row r, columns
  // 4q + g of the block and of the identity image v.  Per pivot: rsqrt on the owner, dinv by v_readlane, the scaled pivot
  // column and the multipliers L(j, k) reach the other lanes through the LDS crossbar (ds_bpermute), 4 + 4 FMAs/lane.
  // Measured and rejected (round 3): lane = row with the whole row in registers, L(j, k) broadcast by v_readlane as an SGPR
  // operand of the FMA, L^-1 built by the same instruction stream in lanes 16..31 -- no LDS inside the chain, 120 FMAs + 240
  // v_readlane per block instead of 12 ds_bpermute per pivot: 60 -> 73 us (f64), 51 -> 60 us (f32).
  // Also rejected (round 3): four ADJACENT columns per lane, a 4-column panel factored with v_readlane multipliers (uniform
  // source lanes), only the finished panel crossing to the other groups by ds_bpermute + a rank-4 update (3 crossbar round
  // trips per block instead of 16): diagonal-block phase 102 k -> 131 k cycles (f64), 66 k -> 75 k (f32).  On this chip a
  // VALU -> v_readlane -> VALU hop on the critical chain costs more than the ds_bpermute it replaces.
  // Also rejected (round 3): the rank-1 update formed from the UNSCALED column, x(r, j) -= x(r, k) x(j, k) / piv, so that the
  // crossbar reads of column k do not wait for the pivot's rsqrt: 100 k -> 106 k cycles (f64), 66 k -> 71 k (f32), and the
  // float result drifts past the square-root-gain vs Joseph tolerance (1 / piv = rsqrt^2 loses a bit per pivot): the step is
  // bound by the number of crossbar operations a wavefront can have in flight, not by the rsqrt in front of them.
  // bound by the number of crossbar operations a wavefront can have in flight, not by the rsqrt in front of them.
  // Also rejected (round 3): a leaner pivot step -- columns left unscaled until the end of the block, no per-element masks
  // (zero multipliers instead), crossbar addresses formed once: ~30 instead of ~50 instructions per pivot in f32, yet 64 k -> 68 k
  // cycles (f32), and the sixteen saved scale factors pushed the f64 instance into spills (load 15 k -> 39 k cycles).  Neither
  // the instruction count nor the number of waits sets the ~335 (f32) / ~490 (f64) cycles per pivot; what remains is the chain
  // rsq -> Newton -> v_readlane -> scale -> ds_bpermute -> FMA itself.
  // Also rejected (round 3): TWO pivots per crossbar round trip in float (raw columns k and k + 1 read together, column k + 1
  // brought up to date locally, d2 = x(k+1, k+1) - L(k+1, k)^2 from three broadcast scalars -- the same products and bits as
  // the one-pivot form): eight dependent crossbar latencies per block instead of sixteen, diagonal blocks 64.4 k -> 65.2 k
  // cycles (profiles/r03_aj_*).  So the crossbar latency is not what a pivot step waits for either.
  // And the hardware v_rsq_f32 without its Newton step (three dependent operations less per pivot): 60.5 k -> 57.9 k cycles, but
  // the float square-root-gain vs Joseph comparison fails its 1e-3 -- the seed is not accurate enough for 180 pivots in a row.
  // Round 4, the pivot step (profiles/r04_n_*): ~80 instructions per pivot in f64 -> ~55 by (i) no masks on the updates -- a
  // multiplier that must not act is a zero: L(r, k) = 0 above the diagonal, hence L(j, k) = 0 for j < k; column k itself (j = k,
  // register qk of group gk) is updated with garbage and then overwritten with its finished values; entries above the diagonal
  // pick up finite garbage that nothing reads (the store at the end masks them); (ii) crossbar addresses formed once, the pivot's
  // group through the instruction's offset field; (iii) the column masked BEFORE it is scaled, i.e. while the reciprocal square
  // root is under way, the diagonal lane scaling its own pivot (no second broadcast); (iv) the clamp of a non-positive pivot as one
  // v_max.  Diagonal blocks 90 k -> 77 k cycles (f64), 61 k -> 55 k (f32); 200 k -> 204 k updates/s in four slices, 182.5 k ->
  // 186.6 k on one stream; same values in every entry that is used.  Measured and rejected on the way: finished columns written
  // straight to the LDS panel inside the loop (the exec-masked stores and their share of lgkmcnt cost more than the select they
  // remove: f32 55 k -> 65 k, and the f64 instance spills); ONE Newton step for the f64 pivot of a float filter (77 k -> 69 k,
  // +2 %, but the 60-camera float filter leaves the Householder route by more than the test's tolerance: Lam^ carries the
  // SQUARED condition of the stack); the rank-1 update from the unscaled column with 1 / piv from v_rcp + Newton, which takes the
  // crossbar round trip out of the chain altogether (same speed as this form
  // and the float square-root-gain vs Joseph comparison at 44 cameras fails its 1e-3); 1 / sqrt(piv) broadcast as its two
  // factors (seed, last Newton factor) with the next pivot's seed started right behind the one FMA it waits for, finished columns
  // in registers of their own, every select beside the chain (eight dependent operations per pivot instead of eleven: f32
  // 55 k -> 58 k, f64 unchanged).  So neither the crossbar round trip nor the number of dependent operations sets the ~290 (f32)
  // / ~400 (f64) cycles of a pivot step on a compute unit whose other fifteen wavefronts are at work (230 / 330 with them idle);
  // what did pay was the number of instructions.
  auto diag = [&](auto pc) __attribute__((always_inline)) {
    constexpr int p = decltype(pc)::value;
    T (*sP)[LP] = sPP[p & 1];
    T (*sM)[LP] = sMM[p & 1];
    if (w == 0) {
      const int kcount = min(16, n - 16 * p);
      const int r = lane & 15, g = lane >> 4;
      T x[4], v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) { x[q] = sP[16 * p + r][4 * q + g]; v[q] = (4 * q + g == r) ? T(1) : T(0); }
      const T told0 = GRAMLIKE ? tol * sD0[16 * p + r] : T(0);
      // crossbar byte addresses, formed once: lane (row, group 0); the pivot's group comes in through the instruction's offset field
      const int a_r = 4 * r;
      int a_q[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) a_q[q] = 4 * ((4 * q + g) & 15);
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int gk = k & 3, qk = k >> 2, src = 16 * gk + k;     // pivot (k, k) lives in lane src, register x[qk]
        const T xk = x[qk];
        bool skip_l;
        T piv = xk;
        if (GRAMLIKE) skip_l = (k >= kcount) || !(piv > told0);
        else { skip_l = k >= kcount; badpiv = badpiv || (lane == src && k < kcount && !(piv > T(0))); piv = clamp_min(piv, Lim<T>::tiny()); }
        // column k before scaling, L(r, k) = 0 above the diagonal: formed while the reciprocal square root is under way (the
        // diagonal lane takes the clamped pivot: piv dinv = sqrt(pivot))
        const T pre = r > k ? xk : (r == k ? piv : T(0));
        const T dinv_l = skip_l ? T(0) : fast_rsqrt(piv);
        const T dinv = wave_bcast(dinv_l, src);
        if (GRAMLIKE && k < kcount && dinv == T(0)) ++nskip;
        // No masks on the updates below -- a multiplier that must not act is a zero: L(r, k) = 0 above the diagonal, hence
        // L(j, k) = 0 for j < k; column k itself (j = k: register qk of group gk) is updated with garbage and then overwritten
        // with its finished values.  Entries above the diagonal of the block pick up finite garbage that nothing reads (the
        // store at the end masks them).
        const T c_src = pre * dinv;
        const T vs = v[qk] * dinv;
        const T c = lane_gather(c_src, a_r + 64 * gk);      // L(r, k) for every group
        const T vk = lane_gather(vs, a_r + 64 * gk);        // v(r, k)
        T ljk[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (4 * q + 3 <= k) continue;                     // compile-time: all of this register's columns are <= k
          ljk[q] = lane_gather(c_src, a_q[q] + 64 * gk);    // L(j, k), j = 4 q + g
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (4 * q + 3 <= k) continue;
          x[q] -= c * ljk[q];
          v[q] -= vk * ljk[q];
        }
        if (g == gk) { x[qk] = c_src; v[qk] = vs; }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) { const int j = 4 * q + g; sP[16 * p + r][j] = j <= r ? x[q] : T(0); sM[r][j] = v[q]; }
    }
  };
  // (3) panel below the diagonal block: L21 = A21 M on the matrix cores, one 16-row block per wavefront at a time
  auto l21 = [&](auto pc) __attribute__((always_inline)) {
    constexpr int p = decltype(pc)::value;
    T (*sP)[LP] = sPP[p & 1];
    T (*sM)[LP] = sMM[p & 1];
    T bq[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) bq[s4] = sM[(lane >> 4) + 4 * s4][lane & 15];
    for (int i = p + 1 + w; i < NR; i += 16) {
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
  };
  // (4) results of panel p, written by wavefronts 1..15 (960 threads) while wavefront 0 factors the next diagonal block
  auto outputs = [&](auto pc) __attribute__((always_inline)) {
    constexpr int p = decltype(pc)::value;
    T (*sP)[LP] = sPP[p & 1];
    T (*sM)[LP] = sMM[p & 1];
    if (w == 0) return;
    const int t = tid - 64;                             // 0 .. 959
    if (GRAMLIKE) {
      // rows 16p .. 16p+15 of T = L^T: T[k][c] = L(c, k), zero left of the diagonal and beyond column n
      for (int c = t; c < 16 * NB; c += 960) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int k = 16 * p + j;
          const T val = (c >= k && c <= n) ? sP[c][j] : T(0);
          if (k < n && OFF + c < d.ldR) Rt[(long)(OFF + k) * d.ldR + OFF + c] = (SO)val;
          if (MODE == CH_GRAM_A && two_level && c >= k) d.Lam[(long)b * d.ldR * d.ldR + (long)c * d.ldR + k] = (double)sP[c][j];   // f64 L11 in place
        }
      }
      if (MODE == CH_GRAM_B) {                          // columns left of this launch's block: zero
        for (int c = t; c < OFF; c += 960) {
#pragma unroll
          for (int j = 0; j < 16; ++j) { const int k = 16 * p + j; if (k < n) Rt[(long)(OFF + k) * d.ldR + c] = SO(0); }
        }
      }
      if (MODE == CH_GRAM_A && two_level)
        for (int e = t; e < 256; e += 960) d.Mp[((long)b * (CH_SPLIT / 16) + p) * 256 + e] = (double)sM[e >> 4][e & 15];
    } else if (SLIKE) {
      // L in place (row-major lower triangle of Smat) and this panel's M, for k_trsm_rows
      for (int c = t; c < 16 * NB; c += 960) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int k = 16 * p + j;
          if (c >= k && c < n && k < n) Sm[(long)(OFF + c) * d.n6cap + OFF + k] = (SO)sP[c][j];
        }
      }
      for (int e = t; e < 256; e += 960) d.Mp2[((long)b * 24 + OFF / 16 + p) * 256 + e] = (SO)sM[e >> 4][e & 15];
    } else {
      // columns 16p .. 16p+15 of W for this part's rows; dx += W(:, k) z_k (thread 64 + a owns appended row a)
      if (t < app_per) {
        const int ar = app_lo + t;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int k = 16 * p + j;
          const T wv = sP[16 * NB + t][j];
          if (k < n && ar < app_hi) { Wg[(long)k * d.ld + ar] = (SO)wv; dxacc += wv * sP[zrow][j]; }
        }
      }
    }
  };
  // (5) rank-16 update of trailing blocks: acc(i, j) -= L(i, p) L(j, p)^T; NEXT: the next diagonal block (p+1, p+1) only --
  // all the next diagonal-block factorization waits for --, else everything but it
  auto trail = [&](auto pc, auto nextc) __attribute__((always_inline)) {
    constexpr int p = decltype(pc)::value;
    constexpr bool NEXT = decltype(nextc)::value;
    T (*sP)[LP] = sPP[p & 1];
#pragma unroll
    for (int jj = 0; jj < HC; ++jj) {
      if (4 * jj + 3 <= p) continue;                    // compile-time: at or left of the panel for every residue
      if (NEXT && 4 * jj > p + 1) continue;             // compile-time: right of column p + 1 for every residue
      const int j = 4 * jj + pj;
      if (j <= p || j >= NB || 16 * j >= n) continue;
      if (NEXT && j != p + 1) continue;
      T bq[4];
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) bq[s4] = sP[16 * j + (lane & 15)][4 * s4 + (lane >> 4)];
#pragma unroll
      for (int ii = 0; ii < HR; ++ii) {
        if (4 * ii + 3 < 4 * jj && 4 * ii + 3 < NB) continue;   // compile-time: main block row above the column block
        const int i = 4 * ii + pi;
        if (i >= NR || (i < NB && (i < j || 16 * i >= main_rows))) continue;
        if (NEXT ? i != p + 1 : (i == p + 1 && j == p + 1)) continue;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          const T a = -sP[16 * i + (lane & 15)][4 * s4 + (lane >> 4)];
          acc[ii][jj] = Mf<T>::mma(a, bq[s4], acc[ii][jj]);
        }
      }
    }
  };
  __syncthreads();
  CH_SUB(4);
  CH_TICK(0);
  if (n > 0) {
    drop(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    __syncthreads();
    CH_TICK(1);
    diag(std::integral_constant<int, 0>{});
    __syncthreads();
    CH_TICK(2);
  }
  auto panel = [&](auto pc) __attribute__((always_inline)) {
    constexpr int p = decltype(pc)::value;
    if (16 * p < n) {
      const bool more = p + 1 < NB && 16 * (p + 1) < n;
      l21(pc);
      __syncthreads();
      CH_TICK(3);
      constexpr int pn = p + 1 < NB ? p + 1 : p;
      if (more) { trail(pc, std::true_type{}); drop(std::integral_constant<int, pn>{}, std::integral_constant<int, 1>{}); }   // next diagonal block only
      __syncthreads();
      CH_TICK(4);
      if (more) diag(std::integral_constant<int, pn>{});
      CH_TICK(2);
#ifdef MSCKF_ABLATE
      if (!(g_chol_dbg & 1)) {
#endif
      outputs(pc);
      trail(pc, std::false_type{});
      if (more) drop(std::integral_constant<int, pn>{}, std::integral_constant<int, 2>{});   // the rest of the next panel (rows disjoint from the diagonal block's)
#ifdef MSCKF_ABLATE
      }
#endif
      __syncthreads();
      CH_TICK(5);
    }
  };
  cstatic_for<NB>(panel);
#ifdef MSCKF_ABLATE
  CH_TICK(5);
  if (tid == 0 && blockIdx.x == 0 && blockIdx.y == 0) {
    for (int q = 0; q < 6; ++q) atomicAdd(&g_chol_cycles[MODE == CH_GAIN ? 1 : 0][q], (unsigned long long)cyc[q]);
    atomicAdd(&g_chol_cycles[MODE == CH_GAIN ? 1 : 0][6], 1ull);
  }
  if (MODE == CH_GAIN && tid == 0 && bi_ == 0) atomicAdd(&g_chol_sub[part & 3][5], 1ull);
#endif
  if (!GRAMLIKE && w == 0 && __any(badpiv ? 1 : 0) && lane == 0) atomicOr(&st[STAT_ERR], STAT_ERR_PIVOT);
  if (MODE == CH_GRAM || MODE == CH_GRAM_A) { if (tid == 0) st[STAT_RROWS] = nfull - nskip; }
  else if (MODE == CH_GRAM_B) { if (tid == 0) st[STAT_RROWS] -= nskip; }
  else if (SLIKE) {}
  else if (tid >= 64 && tid - 64 < app_per && app_lo + tid - 64 < app_hi) d.dx[(long)b * d.ld + app_lo + tid - 64] = (SO)dxacc;
}

// Y = X L^-T for 16-row blocks X that do not take part in the factorization itself, one block per wavefront, independent
// of every other block.  Per 16-column panel p of L: Y_p = X_p M_p on the matrix cores (M_p from the factorization:
// y = x M solves y L_pp^T = x), then X_q -= Y_p L(q, p)^T for the panels right of it.
//   TR_GRAM  X = rows CH_SPLIT .. n of Lam^ (the row n = H_o^T r_o included), columns < CH_SPLIT: L21 of the two-level
//            Gram factorization; Y goes back to Lam (f64, read by level B) and, transposed, into T_H's columns CH_SPLIT..
//   TR_S21   the same for S (rows CH_SPLIT .. n-1 of Smat), in place
//   TR_W     X = [P T_H^T ; r_n^T] (D + 1 rows, all n columns): W = P T_H^T L^-T (msckf.h:1370 without the inverse) and
//            z = L^-1 r_n, left in W and Linv[0..n)
enum { TR_GRAM = 0, TR_S21 = 1, TR_W = 2 };
// LDS hand-over between the lanes of ONE wavefront (write in one layout, read in another): no s_barrier needed, but the
// compiler must not move the reads above the writes -- per-thread addresses differ, so only the fences order them
#define WAVE_LDS_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); } while (0)
template <class T, class SO, int NP, int TMODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 4))) void k_trsm_rows(Dev<SO> d, int b0, int nb, int nwg) {
  typedef typename Mf<T>::V V;
  constexpr int LP = 17;
  // the row blocks of a trajectory all read its factor L: on one XCD (xcd_item), whose private L2 then fetches it once
  int bi_, wg_;
  if (!xcd_item(nb, nwg, bi_, wg_)) return;
  const int b = b0 + bi_, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int* st = d.stats + (long)b * STAT_STRIDE;
  if (st[STAT_MROWS] == 0) return;
  const int i = 4 * wg_ + w;                             // 16-row block
  const int nfull = 6 * d.ncam[b], D = 15 + nfull;
  const int ncols = TMODE == TR_W ? nfull : min(nfull, CH_SPLIT);      // columns of L that exist
  const int R0 = TMODE == TR_W ? 16 * i : CH_SPLIT + 16 * i;           // first row of the block (TR_W: row of [PHt ; r_n^T])
  const int row_end = TMODE == TR_GRAM ? nfull + 1 : (TMODE == TR_S21 ? nfull : D + 1);   // rows that exist
  // (a wavefront without a block keeps taking part in the staging of L and its barriers: `active` instead of a return)
  bool active = TMODE == TR_GRAM ? R0 < d.ldR : R0 < row_end;
  double* Lam = TMODE == TR_GRAM ? d.Lam + (long)b * d.ldR * d.ldR : nullptr;
  SO* Sm = TMODE != TR_GRAM ? d.Smat + (long)b * d.n6cap * d.n6cap : nullptr;
  SO* Rt = d.Rbuf + ((long)b * d.nchunk) * (long)d.n6cap * d.ldR;
  __shared__ T sT[4][16][LP];
  // round 6: the block column L(q, p), q > p, of the panel in LDS, fetched once per WORKGROUP with whole 64-byte rows; every
  // wavefront of the four takes its MFMA operands from there.  (Before: every wavefront fetched every operand itself, a 4-byte
  // load per lane scattered over sixteen rows, one per MFMA: the W solve of a 60-camera window ran at 9 % of the f32 MFMA rate.)
  __shared__ T sLc[(NP - 1) * 16][LP];
  if (TMODE == TR_GRAM && active && R0 >= row_end) {     // nothing below the split in this block: zero columns of T_H
    for (int e = lane; e < 16 * CH_SPLIT; e += 64) { const int k = e >> 4, c = e & 15; if (k < nfull) Rt[(long)k * d.ldR + R0 + c] = SO(0); }
    active = false;
  }
  auto xin = [&](int row, int col) -> T {                // X(row, col), row = absolute row of this mode's operand
    if (TMODE == TR_GRAM) return (T)lam_hat(Lam, nullptr, d.ldR, nfull, d.n_cap, row, col);
    if (col >= ncols) return T(0);
    if (TMODE == TR_S21) return row < nfull ? (T)Sm[(long)row * d.n6cap + col] : T(0);
    if (row < D) return (T)d.K[(long)b * d.ld * d.n6cap + (long)row * d.n6cap + col];   // row-major copy of P T_H^T left by OP_PHT
    return row == D ? (T)Rt[(long)col * d.ldR + nfull] : T(0);                           // r_n[col]
  };
  auto lblk = [&](int q, int p, int r, int c) -> T {     // L(16 q + r, 16 p + c)
    if (TMODE == TR_GRAM) return (T)Lam[(long)(16 * q + r) * d.ldR + 16 * p + c];
    return (T)Sm[(long)(16 * q + r) * d.n6cap + 16 * p + c];
  };
  const int lrow_max = (TMODE == TR_GRAM ? d.ldR : d.n6cap) - 1;   // last row of L's storage (rows past the matrix multiply columns that are never stored)
  V acc[NP];
#pragma unroll
  for (int q = 0; q < NP; ++q)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[q][r] = active ? xin(R0 + Mf<T>::row(lane, r), 16 * q + (lane & 15)) : T(0);
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    if (16 * p >= ncols) continue;
    // the panel's block column into LDS: rows 16 (p + 1) .. of the columns 16 p .. 16 p + 15 (sixteen threads per 64-byte row)
    if (p + 1 < NP) {
      __syncthreads();                                   // the previous panel's operands have been read
      const int NRW = (NP - 1 - p) * 16;
      for (int e0 = 0; e0 < NRW * 16; e0 += 256) {
        const int e = e0 + tid, r = e >> 4, c = e & 15;
        if (e < NRW * 16 && 16 * (p + 1) + (r & ~15) < ncols) sLc[r][c] = lblk(p + 1, p, min(r, lrow_max - 16 * (p + 1)), c);
      }
      __syncthreads();
    }
    if (!active) continue;
    // Y = X_p M_p
#pragma unroll
    for (int r = 0; r < 4; ++r) sT[w][Mf<T>::row(lane, r)][lane & 15] = acc[p][r];
    WAVE_LDS_SYNC();
    V y = V{0, 0, 0, 0};
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const T a = sT[w][lane & 15][(lane >> 4) + 4 * s4];
      const int mi = ((lane >> 4) + 4 * s4) * 16 + (lane & 15);
      const T bq = TMODE == TR_GRAM ? (T)d.Mp[((long)b * (CH_SPLIT / 16) + p) * 256 + mi] : (T)d.Mp2[((long)b * 24 + p) * 256 + mi];
      y = Mf<T>::mma(a, bq, y);
    }
    WAVE_LDS_SYNC();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = Mf<T>::row(lane, r), col = lane & 15;
      sT[w][row][col] = y[r];
      if (TMODE == TR_GRAM) { if (R0 + row <= nfull) Lam[(long)(R0 + row) * d.ldR + 16 * p + col] = (double)y[r]; }
      if (TMODE == TR_S21) { if (R0 + row < nfull && 16 * p + col < ncols) Sm[(long)(R0 + row) * d.n6cap + 16 * p + col] = (SO)y[r]; }
    }
    WAVE_LDS_SYNC();
    // transposed outputs, lanes along the rows of the block
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = 16 * p + (lane >> 4) + 4 * r, c = lane & 15;
      const T yv = sT[w][c][(lane >> 4) + 4 * r];
      if (TMODE == TR_GRAM) { if (k < nfull) Rt[(long)k * d.ldR + R0 + c] = (R0 + c <= nfull) ? (SO)yv : SO(0); }
      if (TMODE == TR_W && k < nfull) {
        if (R0 + c < D) d.W[(long)b * d.ld * d.n6cap + (long)k * d.ld + R0 + c] = (SO)yv;
        else if (R0 + c == D) d.Linv[(long)b * d.n6cap * d.n6cap + k] = (SO)yv;       // z_k
      }
    }
    // X_q -= Y L(q, p)^T, q > p
    T a[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) a[s4] = -sT[w][lane & 15][4 * s4 + (lane >> 4)];
#pragma unroll
    for (int q = p + 1; q < NP; ++q) {
      if (16 * q >= ncols) continue;
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) acc[q] = Mf<T>::mma(a[s4], sLc[(q - p - 1) * 16 + (lane & 15)][4 * s4 + (lane >> 4)], acc[q]);
    }
    WAVE_LDS_SYNC();
    __builtin_amdgcn_sched_barrier(0);   // keep the next panels' loads of L out of this one (the scheduler otherwise hoists them all: spills)
  }
}

// dx = W z (z = L^-1 r_n left in Linv[0..n) by k_trsm_rows<TR_W>); one workgroup per trajectory
template <class SO>
__global__ __launch_bounds__(256) void k_dx_wz(Dev<SO> d, int b0) {
  const int b = b0 + blockIdx.x, tid = threadIdx.x;
  if (d.stats[(long)b * STAT_STRIDE + STAT_MROWS] == 0) return;
  const int n = 6 * d.ncam[b], D = 15 + n;
  __shared__ SO sz[384];
  for (int j = tid; j < n; j += 256) sz[j] = d.Linv[(long)b * d.n6cap * d.n6cap + j];
  __syncthreads();
  const SO* W = d.W + (long)b * d.ld * d.n6cap;
  for (int i = tid; i < D; i += 256) {
    SO s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    int j = 0;
    for (; j + 3 < n; j += 4) { s0 += W[(long)j * d.ld + i] * sz[j]; s1 += W[(long)(j + 1) * d.ld + i] * sz[j + 1]; s2 += W[(long)(j + 2) * d.ld + i] * sz[j + 2]; s3 += W[(long)(j + 3) * d.ld + i] * sz[j + 3]; }
    for (; j < n; ++j) s0 += W[(long)j * d.ld + i] * sz[j];
    d.dx[(long)b * d.ld + i] = (s0 + s1) + (s2 + s3);
  }
}

// S = L L^T in two levels, W = P T_H^T L^-T, z = L^-1 r_n, dx = W z for windows with n > 192 (6 n_cap in {256, 320, 384} padded)
template <class S>
bool launch_chol_gain_large(const Dev<S>& d, int b0, int nb, hipStream_t st) {
  if (nb <= 0) return true;
  const int nblk = (d.n6cap + 15) / 16;
  if (!d.Mp2 || nblk <= 12 || nblk > 24) return false;
  hipLaunchKernelGGL((k_chol_mfma<S, S, 12, 0, CH_S_A, 1>), dim3(nb), dim3(1024), 0, st, d, b0, nb);
  hipLaunchKernelGGL((k_trsm_rows<S, S, 12, TR_S21>), dim3(xcd_grid(nb, (nblk - 12 + 3) / 4)), dim3(256), 0, st, d, b0, nb, (nblk - 12 + 3) / 4);
  if (nblk <= 16) hipLaunchKernelGGL((k_chol_mfma<S, S, 4, 0, CH_S_B, 1>), dim3(nb), dim3(1024), 0, st, d, b0, nb);
  else if (nblk <= 20) hipLaunchKernelGGL((k_chol_mfma<S, S, 8, 0, CH_S_B, 1>), dim3(nb), dim3(1024), 0, st, d, b0, nb);
  else hipLaunchKernelGGL((k_chol_mfma<S, S, 12, 0, CH_S_B, 1>), dim3(nb), dim3(1024), 0, st, d, b0, nb);
  const int rblk = (15 + d.n6cap + 1 + 15) / 16;          // 16-row blocks of [P T_H^T ; r_n^T]
  if (nblk <= 16) hipLaunchKernelGGL((k_trsm_rows<S, S, 16, TR_W>), dim3(xcd_grid(nb, (rblk + 3) / 4)), dim3(256), 0, st, d, b0, nb, (rblk + 3) / 4);
  else if (nblk <= 20) hipLaunchKernelGGL((k_trsm_rows<S, S, 20, TR_W>), dim3(xcd_grid(nb, (rblk + 3) / 4)), dim3(256), 0, st, d, b0, nb, (rblk + 3) / 4);
  else hipLaunchKernelGGL((k_trsm_rows<S, S, 24, TR_W>), dim3(xcd_grid(nb, (rblk + 3) / 4)), dim3(256), 0, st, d, b0, nb, (rblk + 3) / 4);
  hipLaunchKernelGGL((k_dx_wz<S>), dim3(nb), dim3(256), 0, st, d, b0);
  return true;
}

// (Measured and rejected, round 4: the factorizations padded with unused dynamic LDS so that they have their compute unit to
// themselves -- the gain factorization leaves 128 registers per SIMD free, room for one k_feature wavefront of another slice
// each, sharing the LDS pipe its pivot chain runs on: 200.9 k / 199.4 k -> 202.3 k updates/s in four slices, inside the noise.)
template <class S>
bool launch_chol_gram(const Dev<S>& d, int b0, int nb, hipStream_t st) {
  if (nb <= 0) return true;
  const int nblk = d.ldR / 16;
  switch (nblk) {
    case 4: hipLaunchKernelGGL((k_chol_mfma<double, S, 4, 0, CH_GRAM, 1>), dim3(nb), dim3(1024), 0, st, d, b0, nb); return true;
    case 8: hipLaunchKernelGGL((k_chol_mfma<double, S, 8, 0, CH_GRAM, 1>), dim3(nb), dim3(1024), 0, st, d, b0, nb); return true;
    case 12: hipLaunchKernelGGL((k_chol_mfma<double, S, 12, 0, CH_GRAM, 1>), dim3(nb), dim3(1024), 0, st, d, b0, nb); return true;
    default: break;
  }
  if (!d.Mp || (nblk != 16 && nblk != 20 && nblk != 24)) return false;
  // two levels: columns [0, 192), L21, Schur complement
  hipLaunchKernelGGL((k_chol_mfma<double, S, 12, 0, CH_GRAM_A, 1>), dim3(nb), dim3(1024), 0, st, d, b0, nb);
  hipLaunchKernelGGL((k_trsm_rows<double, S, 12, TR_GRAM>), dim3(xcd_grid(nb, (nblk - 12 + 3) / 4)), dim3(256), 0, st, d, b0, nb, (nblk - 12 + 3) / 4);
  if (nblk == 16) hipLaunchKernelGGL((k_chol_mfma<double, S, 4, 0, CH_GRAM_B, 1>), dim3(nb), dim3(1024), 0, st, d, b0, nb);
  else if (nblk == 20) hipLaunchKernelGGL((k_chol_mfma<double, S, 8, 0, CH_GRAM_B, 1>), dim3(nb), dim3(1024), 0, st, d, b0, nb);
  else hipLaunchKernelGGL((k_chol_mfma<double, S, 12, 0, CH_GRAM_B, 1>), dim3(nb), dim3(1024), 0, st, d, b0, nb);
  return true;
}

// GAIN: parts x NA blocks of 16 rows must cover the D + 1 appended rows (each part also carries r_n^T)
template <int NB, int NA, int NPART>
static void gain_launch(const Dev<float>& d, int b0, int nb, hipStream_t st) {
  hipLaunchKernelGGL((k_chol_mfma<float, float, NB, NA, CH_GAIN, NPART>), dim3(xcd_grid(nb, NPART)), dim3(1024), 0, st, d, b0, nb);
}
bool launch_chol_gain(const Dev<float>& d, int b0, int nb, hipStream_t st) {
  if (nb <= 0) return true;
  const int nbn = (d.n6cap + 15) / 16;
  // rows per part = 16 NA - 1; D = 15 + 6 n_cap <= NPART (16 NA - 1)
  if (nbn <= 4) { gain_launch<4, 2, 3>(d, b0, nb, st); return true; }          // D <= 79 <= 3 * 31
  if (nbn <= 8) { gain_launch<8, 3, 3>(d, b0, nb, st); return true; }          // D <= 143 <= 3 * 47
  if (nbn <= 12) {                                                             // D <= 207 <= 4 * 63 = 2 * 111 - ...
    // four parts per trajectory fill the chip at 64 trajectories; from a batch of 96 on, two parts (each redoes the factorization: half the
    // workgroups, one round of them instead of two).  Rows are independent and S's blocks are formed by the same instruction
    // sequence whichever part owns them: same bits either way.  MSCKF_HIP_GAIN_PARTS=2|4 forces one (A/B runs).
    static const int force = [] { const char* e = getenv("MSCKF_HIP_GAIN_PARTS"); return e ? atoi(e) : 0; }();
    const bool two = force ? force == 2 : d.B >= 96;   // (the whole batch: its slices run these launches side by side)
    if (two && 15 + 6 * d.n_cap <= 2 * 111) gain_launch<12, 7, 2>(d, b0, nb, st); else gain_launch<12, 4, 4>(d, b0, nb, st);
    return true;
  }
  return false;
}

#ifdef MSCKF_ABLATE
void chol_debug_set(int v) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_chol_dbg), &v, sizeof(int)); }
void chol_sub_read(unsigned long long* out32, int reset) {
  (void)hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_chol_sub), sizeof(unsigned long long) * 32);
  if (reset) { unsigned long long z[32] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_chol_sub), z, sizeof(z)); }
}
void chol_cycles_read(unsigned long long* out16, int reset) {
  (void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_chol_cycles), sizeof(unsigned long long) * 16);
  if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_chol_cycles), z, sizeof(z)); }
}
#endif

template bool launch_chol_gram<float>(const Dev<float>&, int, int, hipStream_t);
template bool launch_chol_gain_large<float>(const Dev<float>&, int, int, hipStream_t);
template bool launch_chol_gain_large<double>(const Dev<double>&, int, int, hipStream_t);
template bool launch_chol_gram<double>(const Dev<double>&, int, int, hipStream_t);

}  // namespace msckf
