// Information-form compression of the stacked, null-space-projected Jacobian.
//
// The reference stacks H_o (m x D, m ~ 5 800 at a 30-camera window) and reduces it with a Householder QR to
// T_H (n x n), r_n = Q_1^T r_o  (msckf.h:1338-1366).  The update that follows depends on the stack only
// through  H_o^T H_o  and  H_o^T r_o  (T_H^T T_H and T_H^T r_n), and for one track
//     H_o_j^T H_o_j = H_x_j^T (I - Q_f Q_f^T) H_x_j = H_x_j^T H_x_j - B_j^T B_j ,   B_j = Q_f^T H_x_j  (3 x 6 M_j)
// where H_x_j^T H_x_j is block diagonal (one 6 x 6 block per observing camera).  So
//     Lam^ = [H_o | r_o]^T [H_o | r_o] = blockdiag(sum h^T h | sum h^T r) - sum_j [B_j | c_j]^T [B_j | c_j]
// costs 3 n^2 flop per track instead of 2 rho_j n^2, and [T_H | r_n] is its Cholesky factor (R of the QR up to
// the signs of its rows; rows of unobservable directions come out as zero rows, as they do in the QR up to rounding).
// The subtraction cancels, so everything in this file runs in f64 whatever the filter's scalar type is: with the
// reflectors of k_feature orthonormal to f64 rounding Lam^ is positive semi-definite to f64 rounding, and the float
// filter loses nothing against the float QR (scripts/experiments/gram_vs_qr.py: same error vs the f64 result).
//
//   k_gram    SYRK sum B^T B on the f64 matrix cores (v_mfma_f64_16x16x4_f64), one workgroup per 64-column panel
//             (strip of upper 64 x 64 tiles) over the tracks that reach the panel, staged through LDS 8 at a time; extra workgroups reduce the block-diagonal part (one wavefront
//             per camera slot, lanes over tracks)
//   (the block-diagonal part is reduced in k_select's launch, kernels_feature.hip: k_select_diag; the factorization is the
//   blocked matrix-core Cholesky of kernels_chol.hip -- the register-resident k_chol_T / k_chol_blk of rounds 1-2 are gone)
#include <utility>
#include "dev_common.h"

namespace msckf {

// Ablation bits of the -DMSCKF_ABLATE build only (scripts/gram_ablate.py; the product library passes 0): 1 no
// block-diagonal reduction, 2 no prefetch after the first chunks, 4 no MFMA, 8 half the MFMAs.  Results are garbage with any of
// them set.
#ifdef MSCKF_ABLATE
int g_gram_dbg = 0;
// phase timers of the SYRK launch (shader-clock cycles of thread 0 of trajectory 0's workgroups): [strip][0 start-up + track
// count, 1 K loop, 2 group sum, 3 epilogue, 4 launches]
__device__ unsigned long long g_gram_cycles[8][5];
#define GR_TICK(slot) do { if (threadIdx.x == 0 && bi == 0) { const long long t_ = clock64(); atomicAdd(&g_gram_cycles[bx & 7][slot], (unsigned long long)(t_ - gr_t)); gr_t = t_; } } while (0)
#else
#define GR_TICK(slot) do {} while (0)
#endif

typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int GK = 24;   // rows of B per staged chunk = 8 tracks
constexpr int GT_MAX = 1; // 64 x 64 tiles of one block row held by one workgroup (1: every tile its own workgroup -- the MFMA work of a
                          // trajectory spreads over 6 CUs instead of 3 at a 30-camera window; measured against 3)
constexpr int GORD = 1024; // track order staged in LDS (f_cap <= GORD on this route)

// SYRK launch: 512 threads = two groups of four wavefronts; the groups take alternate chunks of the K loop (each with its
// own accumulators, LDS stage and two-deep register prefetch, sharing the barriers) and are summed through LDS at the end:
// twice the loads in flight and two wavefronts per SIMD on a loop that is bound by load latency, with a fixed summation
// order.
template <class S>
__global__ __launch_bounds__(512) void k_gram(Dev<S> d, int b0, int nb, int npairs, int dbg, int xoff) {
  // tiles of one trajectory on the trajectory's XCD (xcd_item): every strip reads the trajectory's B^ rows, and an XCD's L2 is
  // private -- fetch 54 -> 22 MB per launch.  The time does not change (the loop is bound by load latency and the matrix
  // cores, not by L2 misses); the fabric traffic is what the other slices' kernels get back
  int bi, bxi;
  if (!xcd_item(nb, npairs, bi, bxi)) return;
  const int b = b0 + bi, grp = threadIdx.x >> 8, tid = threadIdx.x & 255, lane = tid & 63, w = tid >> 6;
  const int* st = d.stats + (long)b * STAT_STRIDE;
  const int mrows_ = st[STAT_MROWS], P = st[STAT_PASSED], N = d.ncam[b];   // independent scalar loads, one wait
  if (mrows_ == 0) return;
  const int n = 6 * N, ldL = d.ldR, f_cap = d.f_cap, m_cap = d.m_cap;
  const int* order = d.trk_order + (long)b * f_cap;

  const int bx = bxi + xoff;
#ifdef MSCKF_ABLATE
  long long gr_t = clock64();
#endif
  // ---- SYRK strip: a workgroup owns up to GT_MAX tiles (ti, tj0 .. tj0+GT_MAX-1) of block row ti (windows with more
  // than GT_MAX panels split a block row over several workgroups).  The A panel (columns 64 ti ..) is staged once for
  // all of them, and only the tracks whose FIRST camera slot lies at or before the panel take part: the tracks are
  // sorted by first slot (k_select), every column of the panel is zero for the others, so the K loop runs over a
  // prefix of the sorted order (tracks end at the newest camera, so the stack is a staircase: ~35 % of the tracks
  // reach the first panel, ~75 % the second at a 30-camera window).
  // SPLIT K: the stack is a staircase, so a tile of block row ti sums over a prefix of the sorted tracks that grows with ti
  // (~35 % / 75 % / 100 % of them at a 30-camera window) and the launch used to wait for the one workgroup of the last
  // diagonal tile.  With d.gram_parts == P (3 or 4) a tile of block row ti is cut into min(ti + P - 2, P) workgroups over
  // contiguous chunk ranges of its K loop (P = 3: 1 + 1 + 1, 2 + 2, 3 -- ten workgroups of about equal length per trajectory
  // instead of six; P = 4: 2 + 2 + 2, 3 + 3, 4 -- sixteen);
  // each writes its partial sum to its own copy of Lam^ (d.lam_part apart) and the blocked Cholesky adds the copies, in a
  // fixed order, while it loads (kernels_chol.hip) -- no atomics, no extra pass, bit-reproducible.
  const int np_cap = ldL / 64;                       // panels the buffers hold
  int ti = 0, tj0 = 0, part = 0, nparts = 1;
  {
    int rem = bx;
    for (ti = 0; ti < np_cap; ++ti) {
      nparts = d.gram_parts > 1 ? min(ti + d.gram_parts - 2, d.gram_parts) : 1;
      const int ng = (np_cap - ti + GT_MAX - 1) / GT_MAX * nparts;
      if (rem < ng) { tj0 = ti + GT_MAX * (rem / nparts); part = rem % nparts; break; }
      rem -= ng;
    }
  }
  const int nt = min(np_cap, n / 64 + 1);            // tiles that hold a column <= n
  if (ti >= nt || tj0 >= nt) return;
  const bool has_diag = tj0 == ti;                   // local tile 0 is the diagonal tile (its B operand is the A panel)
  extern __shared__ __attribute__((aligned(16))) unsigned char gram_smem[];
  typedef double (*PanelA)[64];
  typedef double (*PanelB)[GK][64];
  double* sbase = reinterpret_cast<double*>(gram_smem);
  constexpr int GRP_DOUBLES = (1 + GT_MAX) * GK * 64;          // one group's stage: A panel + GT_MAX B panels
  PanelA sA = reinterpret_cast<PanelA>(sbase + grp * GRP_DOUBLES);
  PanelB sB = reinterpret_cast<PanelB>(sbase + grp * GRP_DOUBLES + GK * 64);
  int* sOrd = reinterpret_cast<int*>(sbase + 2 * GRP_DOUBLES);
  __shared__ int sCnt;
  if (threadIdx.x == 0) sCnt = 0;
  __syncthreads();
  {
    const int hi_slot = (64 * ti + 63) / 6;          // last slot with a column in the panel
    int c = 0;
    for (int e = threadIdx.x; e < P && e < GORD; e += 512) {
      const int t = order[e];
      const int fl = d.trk_first[(long)b * f_cap + t];      // first | last << 8 camera slot of the track
      sOrd[e] = t | ((fl & 63) << 10) | ((((fl >> 8) & 63) - (fl & 63) + 1) << 16);   // track, first slot, slots in the range
      c += ((fl & 255) <= hi_slot) ? 1 : 0;
    }
    c = (int)wave_sum((float)c);
    if (lane == 0 && c) atomicAdd(&sCnt, c);
  }
  __syncthreads();
  const int KT = 3 * sCnt;
  // this workgroup's share of the K loop: whole chunks [k_lo, KE)
  const int nchunk_k = (KT + GK - 1) / GK;
  const int k_lo = GK * (nchunk_k * part / nparts), KE = min(KT, GK * (nchunk_k * (part + 1) / nparts));
  GR_TICK(0);
  const int lr = tid >> 6, lc = tid & 63;
  const int wi = w & 1, wj = w >> 1;
  // two register stages: the loads of chunk c+2 are issued as soon as chunk c has been staged to LDS, so two
  // chunks of global latency are in flight behind the MFMAs.  Branch-free: rows past the end are clamped to the
  // last row and masked when staged (a conditional per row makes the compiler drain vmcnt between the loads).
  // k_feature writes a track's rows of B^ only inside the track's slot range [first, last] and at column n (Q_f^T r): a row's
  // loads stay unconditional (whatever an older frame left outside the range), `ok` masks them when they are staged
  struct Stage { double a[GK / 4]; double b[GT_MAX][GK / 4]; unsigned ok; };
  const int ca_col = 64 * ti + lc;
  const bool ca_isn = ca_col == n;
  int cb_col[GT_MAX]; bool cb_isn[GT_MAX];
#pragma unroll
  for (int u = 0; u < GT_MAX; ++u) { cb_col[u] = 64 * (tj0 + u) + lc; cb_isn[u] = cb_col[u] == n; }
  Stage r0, r1;
  auto fetch = [&](Stage& r, int kc) {
    const double* rows[GK / 4];
    unsigned okm = 0;
#pragma unroll
    for (int it = 0; it < GK / 4; ++it) {
      const int kk = min(kc + lr + 4 * it, KT - 1);
      const int p = kk / 3, q = kk - 3 * p;
      const int e = sOrd[min(p, GORD - 1)];
      const int t = e & 1023, c_lo = 6 * ((e >> 10) & 63);
      const unsigned c_w = 6u * (unsigned)(e >> 16);
      rows[it] = d.trk_B + (((long)b * f_cap + t) * 3 + q) * (long)ldL;
      okm |= (((unsigned)(ca_col - c_lo) < c_w) || ca_isn) ? (1u << it) : 0u;
#pragma unroll
      for (int u = 0; u < GT_MAX; ++u) okm |= (((unsigned)(cb_col[u] - c_lo) < c_w) || cb_isn[u]) ? (1u << (8 * (u + 1) + it)) : 0u;
    }
    r.ok = okm;
    // every lane loads (no branch, no wait between the loads), but a lane whose column lies outside the track's slot range
    // reads one fixed, always-cached word instead of a row element nobody wrote: ~45 % of the row elements at cfg3
    const double* dummy = d.Dg;
#pragma unroll
    for (int it = 0; it < GK / 4; ++it) {
      r.a[it] = *(((okm >> it) & 1u) ? rows[it] + 64 * ti + lc : dummy);
#pragma unroll
      for (int u = 0; u < GT_MAX; ++u)
        if (tj0 + u > ti && tj0 + u < nt) r.b[u][it] = *(((okm >> (8 * (u + 1) + it)) & 1u) ? rows[it] + 64 * (tj0 + u) + lc : dummy);
    }
  };
  auto stage = [&](const Stage& r, int kc) {
#pragma unroll
    for (int it = 0; it < GK / 4; ++it) {
      const bool ok = kc + lr + 4 * it < KE;
      sA[lr + 4 * it][lc] = (ok && ((r.ok >> it) & 1u)) ? r.a[it] : 0.0;
#pragma unroll
      for (int u = 0; u < GT_MAX; ++u)
        if (tj0 + u > ti && tj0 + u < nt) sB[u][lr + 4 * it][lc] = (ok && ((r.ok >> (8 * (u + 1) + it)) & 1u)) ? r.b[u][it] : 0.0;
    }
  };
  v4d acc[GT_MAX][2][2];
#pragma unroll
  for (int u = 0; u < GT_MAX; ++u)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[u][i][j] = v4d{0.0, 0.0, 0.0, 0.0};
  auto compute = [&]() {
#pragma unroll
    for (int k4 = 0; k4 < GK; k4 += 4) {
      const int kr = k4 + (lane >> 4), cc = lane & 15;
      const double a0 = sA[kr][wi * 32 + cc], a1 = sA[kr][wi * 32 + 16 + cc];
#pragma unroll
      for (int u = 0; u < GT_MAX; ++u) {
        if (tj0 + u >= nt) continue;
        const bool dg = has_diag && u == 0;
        const double c0 = dg ? sA[kr][wj * 32 + cc] : sB[u][kr][wj * 32 + cc];
        const double c1 = dg ? sA[kr][wj * 32 + 16 + cc] : sB[u][kr][wj * 32 + 16 + cc];
        // operands swapped: the accumulator holds the TRANSPOSED block (rows = columns of the tj panel), which is the
        // block's position in the lower triangle of Lam^ -- the epilogue stores it with the lanes along a row
        acc[u][0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(c0, a0, acc[u][0][0], 0, 0, 0);
        acc[u][0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(c1, a0, acc[u][0][1], 0, 0, 0);
        if (dbg & 8) continue;
        acc[u][1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(c0, a1, acc[u][1][0], 0, 0, 0);
        acc[u][1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(c1, a1, acc[u][1][1], 0, 0, 0);
      }
    }
  };
  // group g takes the chunks g, g + 2, g + 4, ... (chunk = GK rows); the loop control runs on group 0's chunk (uniform for
  // the workgroup: both groups pass the same barriers), a group whose chunk lies past the end stages zeros and skips the MFMAs
  const int goff = k_lo + grp * GK;
  if (goff < KE) fetch(r0, goff);
  if (2 * GK + goff < KE) fetch(r1, 2 * GK + goff);
  for (int kc = 0; k_lo + kc < KE; kc += 4 * GK) {
    __syncthreads();
    stage(r0, kc + goff);
    __syncthreads();
    if (kc + 4 * GK + goff < KE && !(dbg & 2)) fetch(r0, kc + 4 * GK + goff);
    if (!(dbg & 4) && kc + goff < KE) compute();
    if (k_lo + kc + 2 * GK >= KE) break;
    __syncthreads();
    stage(r1, kc + 2 * GK + goff);
    __syncthreads();
    if (kc + 6 * GK + goff < KE && !(dbg & 2)) fetch(r1, kc + 6 * GK + goff);
    if (!(dbg & 4) && kc + 2 * GK + goff < KE) compute();
  }
  // sum of the two groups through the (now free) stage area: group 1 stores, group 0 adds in a fixed order
  __syncthreads();
  GR_TICK(1);
  {
    double* red = sbase;                                        // [GT_MAX][2][2][4][256]
    if (grp == 1) {
#pragma unroll
      for (int u = 0; u < GT_MAX; ++u)
#pragma unroll
        for (int ib = 0; ib < 2; ++ib)
#pragma unroll
          for (int jb = 0; jb < 2; ++jb)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(((u * 2 + ib) * 2 + jb) * 4 + r) * 256 + tid] = acc[u][ib][jb][r];
    }
    __syncthreads();
    if (grp == 1) return;
#pragma unroll
    for (int u = 0; u < GT_MAX; ++u)
#pragma unroll
      for (int ib = 0; ib < 2; ++ib)
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[u][ib][jb][r] += red[(((u * 2 + ib) * 2 + jb) * 4 + r) * 256 + tid];
  }
  GR_TICK(2);
  // epilogue: Lam^ = (block-diagonal part, reduced by the launch that precedes this one on the stream) - sum B^T B.  Only the
  // lower triangle is read downstream (lam_hat): acc[u][ib][jb] holds rows 64 tj + 32 wj + 16 jb .., columns 64 ti + 32 wi +
  // 16 ib .. (transposed accumulation), one coalesced store per element; a diagonal tile is stored whole
  double* Lam = d.Lam + (long)part * d.lam_part + (long)b * ldL * ldL;   // part 0 carries the block-diagonal term
  const double* Dgb = d.Dg + (long)b * d.n_cap * DG_STRIDE;
  // two passes: first every block-diagonal term is fetched and folded into the accumulator (all loads in flight together),
  // then the stores.  In one pass (load, subtract, store per element) each load has to wait for the previous element's store
  // to be acknowledged -- stores count in vmcnt and the compiler cannot rule out that Lam aliases Dg.
#pragma unroll
  for (int u = 0; u < GT_MAX; ++u) {
    const int tj = tj0 + u;
#pragma unroll
    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
      for (int jb = 0; jb < 2; ++jb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = 64 * tj + wj * 32 + jb * 16 + (lane >> 4) + 4 * r;     // row of Lam^ (>= the column, except inside a diagonal tile)
          const int i = 64 * ti + wi * 32 + ib * 16 + (lane & 15);            // column
          const double term = (part == 0 && tj < nt) ? lam_diag_term(Dgb, n, d.n_cap, i, j) : 0.0;
          acc[u][ib][jb][r] = term - acc[u][ib][jb][r];
        }
  }
#pragma unroll
  for (int u = 0; u < GT_MAX; ++u) {
    const int tj = tj0 + u;
    if (tj >= nt) continue;
#pragma unroll
    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
      for (int jb = 0; jb < 2; ++jb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = 64 * tj + wj * 32 + jb * 16 + (lane >> 4) + 4 * r;
          const int i = 64 * ti + wi * 32 + ib * 16 + (lane & 15);
          Lam[(long)j * ldL + i] = acc[u][ib][jb][r];
        }
  }
#ifdef MSCKF_ABLATE
  __builtin_amdgcn_s_waitcnt(0);
  GR_TICK(3);
  if (threadIdx.x == 0 && bi == 0) atomicAdd(&g_gram_cycles[bx & 7][4], 1ull);
#endif
}

static size_t gram_lds_bytes() { return (size_t)2 * (1 + GT_MAX) * GK * 64 * sizeof(double) + GORD * sizeof(int); }
void gram_device_setup() {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_gram<float>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)gram_lds_bytes());
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_gram<double>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)gram_lds_bytes());
}

#ifdef MSCKF_ABLATE
void gram_cycles_read(unsigned long long* out40, int reset) {
  (void)hipMemcpyFromSymbol(out40, HIP_SYMBOL(g_gram_cycles), sizeof(unsigned long long) * 40);
  if (reset) { unsigned long long z[40] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_gram_cycles), z, sizeof(z)); }
}
#endif

template <class S>
void launch_gram(const Dev<S>& din, int b0, int nb, hipStream_t st, int phase) {
  if (nb <= 0) return;
  Dev<S> d = din;
  // split-K partial sums only where the consumer adds them up: the single-level blocked Cholesky (k_chol_mfma, CH_GRAM)
  d.gram_parts = (d.compress && d.ldR <= 192 && d.lam_part > 0) ? (din.gram_parts >= 3 ? din.gram_parts : 3) : 1;
#ifdef MSCKF_ABLATE
  const int g_dbg = g_gram_dbg;
#else
  const int g_dbg = 0;
#endif
  int npairs = 0;                                             // SYRK workgroups: <= GT_MAX tiles of one block row each
  for (int ti = 0; ti < d.ldR / 64; ++ti) npairs += (d.ldR / 64 - ti + GT_MAX - 1) / GT_MAX * (d.gram_parts > 1 ? std::min(ti + d.gram_parts - 2, d.gram_parts) : 1);
  if (phase != 2)   // the block-diagonal reduction already ran in k_select's launch (launch_select_diag)
    hipLaunchKernelGGL(k_gram<S>, dim3(xcd_grid(nb, npairs)), dim3(512), gram_lds_bytes(), st, d, b0, nb, npairs, g_dbg, 0);
  if (phase == 3) return;
  (void)launch_chol_gram<S>(d, b0, nb, st);   // blocked matrix-core Cholesky, kernels_chol.hip (two levels beyond 192 columns)
}

template void launch_gram<float>(const Dev<float>&, int, int, hipStream_t, int);
template void launch_gram<double>(const Dev<double>&, int, int, hipStream_t, int);

}  // namespace msckf
