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
//   k_chol_T  register-resident right-looking Cholesky (16 x 16 thread grid, 2-D block-cyclic, one LDS exchange
//             and one barrier per step) with semi-definite pivot skipping; writes [T | r_n] in the layout the
//             Kalman stage reads
#include <utility>
#include "dev_common.h"

namespace msckf {

// Ablation bits of the -DMSCKF_ABLATE build only (scripts/gram_ablate.py; the product library passes 0): 1 no
// block-diagonal reduction, 2 no prefetch after the first chunks, 4 no MFMA, 8 half the MFMAs, 32/64/128/256 skip phase
// b/c/d/e of k_chol_blk.  Results are garbage with any of them set.
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

// Block-diagonal part of Lam^ (sum h^T h per camera slot, sum h^T r): one wavefront per camera slot, lanes over the
// gated-in tracks.  Its own kernel (it was a branch of the SYRK kernel: 3 us slower there).  Measured and rejected: four
// tracks per lane with the three dependent loads batched (18.0 vs 16.2 us -- the 27 f64 wave reductions dominate).
template <class S>
__global__ __launch_bounds__(256) void k_gram_diag(Dev<S> d, int b0) {
  const int b = b0 + blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int* st = d.stats + (long)b * STAT_STRIDE;
  const int mrows_ = st[STAT_MROWS], P = st[STAT_PASSED], N = d.ncam[b];   // independent scalar loads, one wait
  if (mrows_ == 0) return;
  const int f_cap = d.f_cap, m_cap = d.m_cap;
  const int* order = d.trk_order + (long)b * f_cap;
  const int s = 4 * (int)blockIdx.x + w;
  if (s >= N) return;
  double acc[27];
#pragma unroll
  for (int e = 0; e < 27; ++e) acc[e] = 0.0;
  for (int p = lane; p < P; p += 64) {
    const long tb = (long)b * f_cap + order[p];
    const int i = d.trk_inv[tb * d.n_cap + s];
    if (i < 0) continue;
    const long h0i = (tb * m_cap + i) * 12;
    const S* rw = d.trk_rw + tb * 2 * m_cap + 2 * i;
    double h0[6], h1[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) { h0[k] = (double)ld_hx(d, h0i + k); h1[k] = (double)ld_hx(d, h0i + 6 + k); }
    const double r0 = (double)rw[0], r1 = (double)rw[1];
    int e = 0;
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
      for (int c = a; c < 6; ++c) acc[e++] += h0[a] * h0[c] + h1[a] * h1[c];
#pragma unroll
    for (int a = 0; a < 6; ++a) acc[21 + a] += h0[a] * r0 + h1[a] * r1;
  }
  double* out = d.Dg + ((long)b * d.n_cap + s) * DG_STRIDE;
#pragma unroll
  for (int e = 0; e < 27; ++e) {
    const double v = wave_sum(acc[e]);
    if (lane == 0) out[e] = v;
  }
}

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

// [T | r_n] = chol(Lam^) with Lam^ = Dg - sum B^T B; element (i, j), i >= j, of the lower factor lives in thread
// (i % 16, j % 16).  A pivot below 64 eps times its original diagonal belongs to a direction the stack carries no
// information about (the gauge freedoms of the window): its row of T is set to zero.
template <class S, int NBN>
__global__ __launch_bounds__(256) void k_chol_T(Dev<S> d, int b0) {
  constexpr int G = 16;
  const int b = b0 + blockIdx.x, tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  __builtin_amdgcn_s_setprio(3);   // latency-bound chain: win instruction arbitration against co-resident throughput waves of the other slice
  int* st = d.stats + (long)b * STAT_STRIDE;
  if (st[STAT_MROWS] == 0) return;
  const int N = d.ncam[b], n = 6 * N, ldL = d.ldR;
  const double* Lam = d.Lam + (long)b * ldL * ldL;
  const double* Dg = d.Dg + (long)b * d.n_cap * DG_STRIDE;
  __shared__ double sCol[2][G * NBN];
  __shared__ double sD0[G * NBN];
  double A[NBN][NBN];
#pragma unroll
  for (int a = 0; a < NBN; ++a)
#pragma unroll
    for (int bb = 0; bb < NBN; ++bb) {
      if (a < bb) continue;
      const int i = G * a + tx, j = G * bb + ty;
      const double lv = lam_hat(Lam, Dg, ldL, n, d.n_cap, i, j);
      const double val = i >= j ? lv : 0.0;
      if (i == j && i < n) sD0[i] = val;
      A[a][bb] = val;
    }
  __syncthreads();
  const double tol = 64.0 * 2.220446049250313e-16;
  int nskip = 0, buf = 0;
#pragma unroll
  for (int kb = 0; kb < NBN; ++kb) {
    const int kk_hi = min(G, n - G * kb);
    for (int kk = 0; kk < kk_hi; ++kk) {
      const int k = G * kb + kk;
      if (ty == kk) {
#pragma unroll
        for (int a = kb; a < NBN; ++a) sCol[buf][G * a + tx] = A[a][kb];
      }
      __syncthreads();
      const double dkk = sCol[buf][k];
      const bool skip = !(dkk > tol * sD0[k]);
      const double dinv = skip ? 0.0 : fast_rsqrt(dkk);
      const double dd = dkk * dinv;
      nskip += skip ? 1 : 0;
      double li[NBN], lj[NBN];
#pragma unroll
      for (int a = kb; a < NBN; ++a) li[a] = (a > kb || tx > kk) ? sCol[buf][G * a + tx] * dinv : 0.0;
#pragma unroll
      for (int bb = kb; bb < NBN; ++bb) lj[bb] = (bb > kb || ty > kk) ? sCol[buf][G * bb + ty] * dinv : 0.0;
#pragma unroll
      for (int a = kb; a < NBN; ++a)
#pragma unroll
        for (int bb = kb; bb <= a; ++bb) A[a][bb] -= li[a] * lj[bb];
      if (ty == kk) {
#pragma unroll
        for (int a = kb; a < NBN; ++a) {
          if (a > kb || tx > kk) A[a][kb] = li[a];
          else if (a == kb && tx == kk) A[a][kb] = dd;
        }
      }
      buf ^= 1;
    }
  }
  // T[k][c] = L[c][k]: rows k < n, columns c <= n; zeros below the diagonal and in the padding columns
  S* Rt = d.Rbuf + ((long)b * d.nchunk) * (long)d.n6cap * d.ldR;
#pragma unroll
  for (int a = 0; a < NBN; ++a)
#pragma unroll
    for (int bb = 0; bb < NBN; ++bb) {
      const int i = G * a + tx, j = G * bb + ty;
      if (j >= n || i >= d.ldR) continue;
      double val = 0.0;
      if (a >= bb) { if (i >= j && i <= n) val = A[a][bb]; }
      Rt[(long)j * d.ldR + i] = (S)val;
    }
  if (tid == 0) st[STAT_RROWS] = n - nskip;
}

// Blocked right-looking Cholesky of Lam^ on the f64 matrix cores.  The trailing matrix lives in MFMA accumulator
// registers: 16 x 16 blocks, 2 x 2 block-cyclic over the four wavefronts (block (i, j) belongs to wave 2 (i & 1) +
// (j & 1), <= 21 blocks = 168 registers per lane).  Per panel of 16 columns: the owners drop the panel's blocks into
// LDS, wave 0 factors the 16 x 16 diagonal block (lanes = rows, pivot/column broadcasts by v_readlane, semi-definite
// pivot skipping as in k_chol_T), all threads solve their row of the panel against it (thread = matrix row), the
// finished rows of T = L^T go to global memory, and every wave applies the rank-16 update to its blocks with
// v_mfma_f64_16x16x4_f64 (operands straight from the LDS panel).  12 panels x 4 barriers instead of 180 x 1, and
// the O(n^3) part runs at MFMA rate.
template <class F, int... Ps>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Ps...>) { (f(std::integral_constant<int, Ps>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

template <class S, int NBLK>
__global__ __launch_bounds__(256) void k_chol_blk(Dev<S> d, int b0, int dbg) {
  constexpr int H = NBLK / 2, NR = 16 * NBLK, LP = 17;
  const int b = b0 + blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int pi = w >> 1, pj = w & 1;
  int* st = d.stats + (long)b * STAT_STRIDE;
  if (st[STAT_MROWS] == 0) return;
  const int N = d.ncam[b], n = 6 * N, ldL = d.ldR;
  const double* Lam = d.Lam + (long)b * ldL * ldL;
  const double* Dg = d.Dg + (long)b * d.n_cap * DG_STRIDE;
  S* Rt = d.Rbuf + ((long)b * d.nchunk) * (long)d.n6cap * d.ldR;
  __shared__ double sP[NR][LP];     // current panel: rows 0 .. NR-1, 16 columns
  __shared__ double sL[16][LP];     // factored diagonal block
  __shared__ double sDinv[16];
  __shared__ double sD0[NR];
  v4d acc[H][H];
#pragma unroll
  for (int ii = 0; ii < H; ++ii)
#pragma unroll
    for (int jj = 0; jj <= ii; ++jj) {                 // jj > ii is never a lower block, whatever the wave's parity
      acc[ii][jj] = v4d{0.0, 0.0, 0.0, 0.0};
      const int i = 2 * ii + pi, j = 2 * jj + pj;
      if (i < j || 16 * j >= n || 16 * i > n) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[ii][jj][r] = lam_hat(Lam, Dg, ldL, n, d.n_cap, 16 * i + (lane >> 4) + 4 * r, 16 * j + (lane & 15));
    }
  for (int t = tid; t < NR; t += 256) { const double dv = lam_hat(Lam, Dg, ldL, n, d.n_cap, t, t); sD0[t] = t < n ? dv : 0.0; }
  const double tol = 64.0 * 2.220446049250313e-16;
  int nskip = 0;
  // the panel index must be a compile-time constant (it selects accumulator registers): static_for, not a loop
  auto panel = [&](auto pc) __attribute__((always_inline)) {
    constexpr int p = decltype(pc)::value;
    if (16 * p < n) {
    __syncthreads();                                    // previous panel's operands are no longer read
    // ---- (a) the panel's blocks (i, p), i >= p, from the accumulators to LDS
    if (pj == (p & 1)) {
#pragma unroll
      for (int ii = 0; ii < H; ++ii) {
        const int i = 2 * ii + pi;
        if (i < p || 16 * i > n || (p >> 1) > ii) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) sP[16 * i + (lane >> 4) + 4 * r][lane & 15] = acc[ii][(p >> 1) <= ii ? (p >> 1) : 0][r];
      }
    }
    __syncthreads();
    // ---- (b) diagonal block: lanes 0..15 of wave 0 hold one row each
    if (w == 0 && !(dbg & 32)) {
      const int kcount = min(16, n - 16 * p);
      const int lr = lane & 15;
      double x[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) x[j] = sP[16 * p + lr][j];
      const double d0 = sD0[16 * p + lr];
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        // every lane takes the rsqrt of its own x[k]; lane k's is the pivot's
        const bool skip_l = (k >= kcount) || !(x[k] > tol * d0);
        const double dinv_l = skip_l ? 0.0 : fast_rsqrt(x[k]);
        const double dinv = wave_bcast(dinv_l, k);
        const double pv = wave_bcast(x[k], k);
        if (k < kcount && dinv == 0.0) ++nskip;
        x[k] = lr == k ? pv * dinv : (lr > k ? x[k] * dinv : 0.0);
        if (lane == 0) sDinv[k] = dinv;
#pragma unroll
        for (int j = k + 1; j < 16; ++j) {
          const double ljk = wave_bcast(x[k], j);       // L(j, k)
          if (lr >= j) x[j] -= x[k] * ljk;
        }
      }
      if (lane < 16) {
#pragma unroll
        for (int j = 0; j < 16; ++j) { sL[lane][j] = x[j]; sP[16 * p + lane][j] = x[j]; }
      }
    }
    __syncthreads();
    // ---- (c) rows below the diagonal block: thread = matrix row, forward substitution against L_pp
    {
      const int R = 16 * p + 16 + tid;
      if (tid < NR - 16 * p - 16 && R <= n && !(dbg & 64)) {
        double x[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) x[j] = sP[R][j];
        // column-oriented: after x[k] is final the 15-k updates are independent (dependency depth 16, not 120)
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          x[k] *= sDinv[k];
#pragma unroll
          for (int j = k + 1; j < 16; ++j) x[j] -= x[k] * sL[j][k];
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) sP[R][j] = x[j];
      }
    }
    __syncthreads();
    // ---- (d) rows 16p .. 16p+15 of T = L^T are final: T[k][c] = L(c, k), zero left of the diagonal and beyond column n
    if (tid < NR && !(dbg & 128)) {
      const int c = tid;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int k = 16 * p + j;
        const double val = (c >= k && c <= n) ? sP[c][j] : 0.0;
        if (k < n) Rt[(long)k * d.ldR + c] = (S)val;
      }
    }
    // ---- (e) rank-16 update of the trailing blocks on the matrix cores: acc(i, j) -= L(i, p) L(j, p)^T
#pragma unroll
    for (int ii = 0; ii < H; ++ii)
#pragma unroll
      for (int jj = 0; jj <= ii; ++jj) {
        const int i = 2 * ii + pi, j = 2 * jj + pj;
        if (2 * jj + 1 <= p) continue;                  // compile-time: at or left of the panel for either parity
        if (i < j || j <= p || 16 * i > n || 16 * j >= n || (dbg & 256)) continue;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          const double a = -sP[16 * i + (lane & 15)][4 * s4 + (lane >> 4)];
          const double bq = sP[16 * j + (lane & 15)][4 * s4 + (lane >> 4)];
          acc[ii][jj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bq, acc[ii][jj], 0, 0, 0);
        }
      }
    }
  };
  static_for<NBLK>(panel);
  if (tid == 0) st[STAT_RROWS] = n - nskip;
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
  d.gram_parts = (d.compress == 3 && d.ldR <= 192 && d.lam_part > 0) ? (din.gram_parts >= 3 ? din.gram_parts : 3) : 1;
#ifdef MSCKF_ABLATE
  const int g_dbg = g_gram_dbg;
#else
  const int g_dbg = 0;
#endif
  int npairs = 0;                                             // SYRK workgroups: <= GT_MAX tiles of one block row each
  for (int ti = 0; ti < d.ldR / 64; ++ti) npairs += (d.ldR / 64 - ti + GT_MAX - 1) / GT_MAX * (d.gram_parts > 1 ? std::min(ti + d.gram_parts - 2, d.gram_parts) : 1);
  const int ndiag = (d.n_cap + 3) / 4;
  if (phase != 2) {
    // (phase 3: the block-diagonal reduction already ran in k_select's launch, launch_select_diag)
    // two launches of the same kernel: the block-diagonal reduction (short, many small workgroups) and the SYRK strips
    // (MFMA-bound, <= 192 workgroups).  In ONE launch the dispatcher packs strips two to a CU behind the reduction
    // workgroups and they share the matrix cores (measured: MFMA phase 2x longer).
    if (!(g_dbg & 1) && phase != 3) hipLaunchKernelGGL(k_gram_diag<S>, dim3(ndiag, nb), dim3(256), 0, st, d, b0);
    hipLaunchKernelGGL(k_gram<S>, dim3(xcd_grid(nb, npairs)), dim3(512), gram_lds_bytes(), st, d, b0, nb, npairs, g_dbg, 0);
  }
  if (phase == 1 || phase == 3) return;
  // k_chol_blk (blocked, trailing update on the f64 matrix cores) is 10 % faster than k_chol_T in isolation (104 vs
  // 114 us) but holds the whole register file of its CU (256 VGPR + 188 AGPR), so nothing of the other slice's stream
  // co-schedules with it: 3 % slower end to end with two streams.  Kept selectable (msckf_hip_set_compression(h, 2)).
  if ((d.compress == 3 || d.ldR > 192) && launch_chol_gram<S>(d, b0, nb, st)) return;   // blocked matrix-core Cholesky, kernels_chol.hip (the only one for > 192 columns)
  if (d.compress == 2) {
    switch (d.ldR / 16) {
      case 4: hipLaunchKernelGGL((k_chol_blk<S, 4>), dim3(nb), dim3(256), 0, st, d, b0, g_dbg); break;
      case 8: hipLaunchKernelGGL((k_chol_blk<S, 8>), dim3(nb), dim3(256), 0, st, d, b0, g_dbg); break;
      default: hipLaunchKernelGGL((k_chol_blk<S, 12>), dim3(nb), dim3(256), 0, st, d, b0, g_dbg); break;
    }
    return;
  }
  switch (d.ldR / 16) {
    case 4: hipLaunchKernelGGL((k_chol_T<S, 4>), dim3(nb), dim3(256), 0, st, d, b0); break;
    case 8: hipLaunchKernelGGL((k_chol_T<S, 8>), dim3(nb), dim3(256), 0, st, d, b0); break;
    default: hipLaunchKernelGGL((k_chol_T<S, 12>), dim3(nb), dim3(256), 0, st, d, b0); break;
  }
}

template void launch_gram<float>(const Dev<float>&, int, int, hipStream_t, int);
template void launch_gram<double>(const Dev<double>&, int, int, hipStream_t, int);

}  // namespace msckf
