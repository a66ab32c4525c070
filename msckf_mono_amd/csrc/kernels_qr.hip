// kernels_qr.hip -- Householder compression of the stacked projected Jacobian (msckf.h:1338-1366) as a streaming
// TSQR on gfx950.  Route (b) of DESIGN.md section 4.4: used for windows with n + 1 > 192 and as the A/B reference of
// the information-form route (kernels_gram.hip), which is the default below that size.
//
// The reference stacks H_o (m x D, m ~ 5 600 for 200 tracks in a 30-camera window), runs a dense
// HouseholderQR, forms the FULL m x m Q and a dense m x m R_o (msckf.h:1343-1366).  Here H_o is never
// materialised: every workgroup owns an n x (n+1) upper-triangular work matrix [R | Q^T r], packed in LDS when it
// fits (n = 6 N camera columns -- the 15 IMU columns of H_o are identically zero, msckf.h:949) and folds blocks of
// RW rows into it with structured Householder reflectors ("QR update" of [R; B]).  Rows are regenerated on the fly
// from the per-track compact form written by k_feature:
//       H_o_j[i, col] = [col in the 6 columns of obs (3+i)/2] Hx[(3+i)&1][.] - V[3+i,:] . Zf[:, col]
// Layout inside a workgroup (QR_NW = 12 wavefronts): ONE wavefront owns one block of RW rows at a time, lane l holds
// columns l, l+64, ... (NC per lane) of all RW rows in registers.  Step k needs no LDS exchange and no barrier inside
// the wave: the pivot column comes from its owner lane with v_readlane, the dot products are lane-local over the RW
// rows, the column norm is the pivot column's self product.  The wavefronts form a software pipeline over R's rows:
// block q may run step k as soon as block q-1 has finished step k; R's row k is handed from wave to wave through LDS
// with a workgroup-scope release/acquire progress word per block.
// TSQR: stage 1 = nchunk workgroups per trajectory (the sorted blocks are dealt round-robin, so every chunk sees the
// same mix of leading-zero counts), stage 2 = one merge of all chunk triangles into chunk 0 (same kernel, rows sourced
// from the other chunks' R, interleaved by row index).  With isotropic pixel noise (u_var' == v_var', the
// configuration BASELINE.json is quoted on) the Kalman update depends on the stack only through H_o^T H_o and
// H_o^T r_o, so any orthogonal compression gives the reference's result; R_n = sigma^2 I exactly (SURVEY.md 8a
// Q1/Q1b/Q2); anisotropic noise is handled by pre-whitening the rows.
#include "dev_common.h"

namespace msckf {

#ifdef MSCKF_ABLATE
__device__ int g_qr_dbg[4] = {0, 0, 0, 0};   // ablation knobs of the -DMSCKF_ABLATE build (scripts/qr_ablate.py)
#endif

// Reflector scalars: beta = sqrt(s), g = 1/(beta*u).  Double: IEEE sqrt / divide.  Float: hardware rsq / rcp
// (~1 ulp) refined by one Newton step each -- within 1 ulp of the correctly rounded values at a third of the
// instructions of the IEEE expansions, which sit on the per-step critical path of the elimination.
template <class S> __device__ __forceinline__ S qr_rcp(S x) { return S(1) / x; }
template <> __device__ __forceinline__ float qr_rcp<float>(float x) {
  const float g0 = __builtin_amdgcn_rcpf(x);
  return g0 * (2.0f - x * g0);
}
template <class S> __device__ __forceinline__ S fast_sqrt(S s) { return dsqrt(s); }
template <> __device__ __forceinline__ float fast_sqrt<float>(float s) {
  const float rs = __builtin_amdgcn_rsqf(s);
  const float b0 = s * rs;
  return b0 + 0.5f * rs * (s - b0 * b0);
}

// Work triangle access: packed upper triangle in LDS (row k holds columns k..n) or ldR-strided rows in global.
template <class S, bool RLDS>
struct Tri {
  S* base; int n, ldR;
  __device__ __forceinline__ S get(int k, int col) const {
    if (RLDS) return (col >= k && col <= n) ? base[k * (n + 1) - k * (k - 1) / 2 + col - k] : S(0);
    return base[(long)k * ldR + col];
  }
  __device__ __forceinline__ void put(int k, int col, S v) const {
    if (RLDS) { if (col >= k && col <= n) base[k * (n + 1) - k * (k - 1) / 2 + col - k] = v; }
    else if (col >= k) base[(long)k * ldR + col] = v;
  }
};

// One workgroup = QR_NW wavefronts working as a software pipeline over the rows of R: wavefront h owns the
// row blocks h, h+QR_NW, h+2 QR_NW, ... of its chunk (RW rows x all columns in registers, lane l = columns l, l+64, ..),
// and block q may run Householder step k as soon as block q-1 has finished step k (R's row k is handed
// from wave to wave through LDS with a release/acquire progress word).  Inside a wavefront a step needs no
// LDS exchange and no barrier: the pivot column is read from its owner lane with v_readlane, the dot
// products are complete within the wave, the column norm is the pivot column's self product.
constexpr int QR_NW = 12;        // wavefronts per workgroup = depth of the software pipeline
constexpr int QR_MAXBLK = 512;   // ring of per-block progress words in LDS

template <class S, int NC, int RW, bool RLDS>
__global__ __launch_bounds__(64 * QR_NW) void k_qr_update(Dev<S> d, int b0, int stage, int level) {
  const int b = b0 + blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, h = tid >> 6;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  int* sProg = reinterpret_cast<int*>(smem_raw);          // [QR_MAXBLK] per-block progress (elimination steps done)
  S* sR = reinterpret_cast<S*>(smem_raw + 4 * QR_MAXBLK); // packed triangle (RLDS)
  const int n = 6 * d.ncam[b];
  const int ldR = d.ldR;
  const int m_cap = d.m_cap, f_cap = d.f_cap;
  const int* rs = d.row_start + (long)b * (f_cap + 1);
  const int* order = d.trk_order + (long)b * f_cap;
  const int P = d.stats[(long)b * STAT_STRIDE + STAT_PASSED];     // gated-in tracks (sorted positions)
  const int m_total = d.stats[(long)b * STAT_STRIDE + STAT_MROWS];
  if (m_total == 0) return;

  S* Rt;                  // target triangle (global)
  const S* Rs = nullptr;  // stage 2: source triangle
  int row_begin, row_end;
  if (stage == 1) {
    const int c = blockIdx.x;
    Rt = d.Rbuf + ((long)b * d.nchunk + c) * (long)d.n6cap * ldR;
    row_begin = 0; row_end = m_total;   // blocks of RW rows are dealt round-robin to the chunks (see below)
    if (RLDS) { for (int e = tid; e < (n + 1) * (n + 2) / 2; e += 64 * QR_NW) sR[e] = 0; }
    else { for (int e = tid; e < n * ldR; e += 64 * QR_NW) Rt[e] = 0; }   // fresh triangle
  } else {
    // stage 2: ONE workgroup per trajectory folds the triangles of chunks 1..C-1 into chunk 0.  The source
    // rows are interleaved by row index (row i of every source before row i+1 of any), so the blocks'
    // leading-zero counts ascend and the pipeline never waits on a later block.
    if (d.nchunk < 2) return;
    Rt = d.Rbuf + ((long)b * d.nchunk) * (long)d.n6cap * ldR;
    Rs = Rt;   // source c lives at Rs + c * n6cap * ldR
    row_begin = 0; row_end = n * (d.nchunk - 1);
    if (RLDS) {
      for (int e = tid; e < n * (n + 1); e += 64 * QR_NW) {
        const int k = e / (n + 1), col = e - k * (n + 1);
        if (col >= k) sR[k * (n + 1) - k * (k - 1) / 2 + col - k] = Rt[(long)k * ldR + col];
      }
    }
  }
  for (int e = tid; e < QR_MAXBLK; e += 64 * QR_NW) sProg[e] = -1;
  __syncthreads();
  Tri<S, RLDS> R;
  R.base = RLDS ? sR : Rt; R.n = n; R.ldR = ldR;
  // Stage 1: the sorted stack is cut into blocks of RW rows; global block g belongs to chunk g % nchunk (so
  // every chunk gets the same mix of long and short tracks) and is that chunk's block q = g / nchunk.
  const int cstride = (stage == 1) ? d.nchunk : 1, coff = (stage == 1) ? (int)blockIdx.x : 0;
  const int gblk = (row_end - row_begin + RW - 1) / RW;
  const int nblk = (gblk - coff + cstride - 1) / cstride;
#ifdef MSCKF_ABLATE
  const int dbg = g_qr_dbg[0];
#else
  constexpr int dbg = 0;
#endif
  // blocks are dealt to the wavefronts round-robin (a boustrophedon order balances the step totals better but
  // stalls the pipeline at every turn: measured 9 % slower)
  for (int q = h; q < nblk; q += QR_NW) {
    const int blk0 = row_begin + (q * cstride + coff) * RW;
    // ---------------- load this wave's block: RW rows x NC columns per lane
    // lane r resolves stacked row blk0+r to (track, row-in-track) -- one parallel binary search per block
    // instead of RW serial ones -- and the results are handed out with v_readlane
    S Bv[RW][NC];
    int kmin = n;
    int my_t = -1, my_row = 0;
    if (dbg & 2) {   // ablation: skip the loader
#pragma unroll
      for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int j = 0; j < NC; ++j) Bv[r][j] = S(0.001) * S((lane * 7 + r * 13 + j * 5 + q) % 17 - 8);
      kmin = 0;
    } else if (stage == 1) {
      // lane r resolves stacked row blk0+r to (track, row-in-track) with one parallel binary search and
      // fetches that row's scalars (V row, r_o, first column of its own observation); the per-row values
      // are then handed out with v_readlane, so the coalesced Z / H_x loads of all rows issue back to back
      S my_v0 = 0, my_v1 = 0, my_v2 = 0, my_ro = 0;
      int my_c0 = 0;
      const int grl = blk0 + lane;
      if (lane < RW && grl < row_end) {
        int lo = 0, hi = P;   // invariant rs[lo] <= grl < rs[hi]
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (rs[mid] <= grl) lo = mid; else hi = mid; }
        my_t = order[lo];
        my_row = 3 + (grl - rs[lo]);              // row of Q^T [H_x | r]
        const long tb = (long)b * f_cap + my_t;
        kmin = 6 * (d.trk_first[tb] & 255);
        const S* Vr = d.trk_V + (tb * 2 * m_cap + my_row) * 4;
        my_v0 = Vr[0]; my_v1 = Vr[1]; my_v2 = Vr[2];
        my_ro = d.trk_ro[tb * 2 * m_cap + my_row];
        my_c0 = 6 * d.trk_slots[wl_first(d, b - b0, my_t) + (my_row >> 1)];
      }
      kmin = wave_min_i(kmin);
      S zf[3][NC];                                   // Z of the current track at this lane's columns
      int t_prev = -2;
#pragma unroll
      for (int r = 0; r < RW; ++r) {
        const int t_raw = wave_bcast(my_t, r);
        const bool live = t_raw >= 0;               // false past the end of the stack (last block only)
        const int t = live ? t_raw : 0;             // every load below stays in bounds
        const int row = wave_bcast(my_row, r);
        const S v0 = wave_bcast(my_v0, r), v1 = wave_bcast(my_v1, r), v2 = wave_bcast(my_v2, r);
        const S ro = wave_bcast(my_ro, r);
        const int c0 = wave_bcast(my_c0, r);
        const long tb = (long)b * f_cap + t;
        if (t != t_prev) {                          // wave-uniform: consecutive rows mostly share their track
          const S* Zf = d.trk_Zf + tb * 3 * (long)ldR;                      // [3][ldR], coalesced over columns
#pragma unroll
          for (int j = 0; j < NC; ++j) {
            const int col = lane + 64 * j;
            zf[0][j] = Zf[col]; zf[1][j] = Zf[ldR + col]; zf[2][j] = Zf[2 * ldR + col];
          }
          t_prev = t;
        }
        const long hx0 = (tb * m_cap + (row >> 1)) * 12 + (row & 1) * 6;
#pragma unroll
        for (int j = 0; j < NC; ++j) {
          const int col = lane + 64 * j;
          S val = -(v0 * zf[0][j] + v1 * zf[1][j] + v2 * zf[2][j]);
          const unsigned dcol = (unsigned)(col - c0);
          if (dcol < 6u) val += ld_hx(d, hx0 + dcol);
          if (col == n) val = ro;
          Bv[r][j] = live ? val : S(0);
        }
      }
    } else {
      kmin = min(blk0 / (d.nchunk - 1), n);
#pragma unroll
      for (int r = 0; r < RW; ++r) {
        const int gr = blk0 + r;
        const int si = gr / (d.nchunk - 1), sc = 1 + gr % (d.nchunk - 1);
#pragma unroll
        for (int j = 0; j < NC; ++j) {
          const int col = lane + 64 * j;
          Bv[r][j] = (gr < row_end && col <= n) ? Rs[((long)sc * d.n6cap + si) * ldR + col] : S(0);
        }
      }
    }
    if (dbg & 1) kmin = n;   // ablation: skip the elimination steps
    if (dbg & 4) kmin = 0;   // ablation: no leading-zero skipping
    if (dbg & 8) {           // ablation: keep the loader's timing but eliminate dense pseudo-random data
#pragma unroll
      for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int j = 0; j < NC; ++j) Bv[r][j] = Bv[r][j] * S(1e-30) + S(0.001) * S((lane * 7 + r * 13 + j * 5 + q) % 17 - 8);
    }
    kmin = __builtin_amdgcn_readfirstlane(kmin);
    // steps below kmin are no-ops for this block
    // progress word of block q lives in ring slot q % QR_MAXBLK and carries the block index in its high bits
    if (lane == 0) __hip_atomic_store(&sProg[q & (QR_MAXBLK - 1)], (q << 10) | kmin, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);

    // ---------------- fold the block into R: Householder steps k = kmin .. n-1
#pragma unroll
    for (int jk = 0; jk < NC; ++jk) {
      const int l_lo = max(kmin - 64 * jk, 0);
      const int l_hi = min(64, n - 64 * jk);
      for (int lk = l_lo; lk < l_hi; ++lk) {
        const int k = 64 * jk + lk;
        if (q > 0) {   // wait until block q-1 has finished step k (its R row k is final for us)
          const int need = ((q - 1) << 10) | (k + 1);
          while (__hip_atomic_load(&sProg[(q - 1) & (QR_MAXBLK - 1)], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < need) __builtin_amdgcn_s_sleep(1);
        }
        S rk[NC];
#pragma unroll
        for (int j = jk; j < NC; ++j) rk[j] = R.get(k, lane + 64 * j);
        // pivot column of this block, from the owner lane
        S xr[RW];
#pragma unroll
        for (int r = 0; r < RW; ++r) xr[r] = wave_bcast(Bv[r][jk], lk);
        S tot[NC];
#pragma unroll
        for (int j = jk; j < NC; ++j) {
          S s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
          for (int r = 0; r < RW; r += 4) { s0 += xr[r] * Bv[r][j]; s1 += xr[r + 1] * Bv[r + 1][j]; s2 += xr[r + 2] * Bv[r + 2][j]; s3 += xr[r + 3] * Bv[r + 3][j]; }
          tot[j] = (s0 + s1) + (s2 + s3);
        }
        const S sigma = wave_bcast(tot[jk], lk);
        if (sigma > Lim<S>::tiny()) {
          // reflector H = I - tau v v^T, v = [1; x'/u], u = x0 - beta, tau = -u/beta.  With g = 1/(beta u):
          //   R[k][j] += e_j g u,   B[:,j] += e_j g x',   e_j = u R[k][j] + x'^T b_j      (one reciprocal per step)
          const S x0 = wave_bcast(rk[jk], lk);
          S beta = fast_sqrt(x0 * x0 + sigma);
          if (x0 >= S(0)) beta = -beta;
          const S u = x0 - beta;
          const S g = qr_rcp(beta * u);
          const S gu = g * u;
#pragma unroll
          for (int j = jk; j < NC; ++j) {
            const int col = lane + 64 * j;
            if (col > k) {
              const S e = u * rk[j] + tot[j];
              rk[j] += e * gu;
              const S cj = e * g;
#pragma unroll
              for (int r = 0; r < RW; ++r) Bv[r][j] += cj * xr[r];
            } else if (col == k) {
              rk[j] = beta;      // the eliminated column of B is never read again (its lane is masked from now on)
            }
          }
#pragma unroll
          for (int j = jk; j < NC; ++j) R.put(k, lane + 64 * j, rk[j]);
        }
        if (lane == 0) __hip_atomic_store(&sProg[q & (QR_MAXBLK - 1)], (q << 10) | (k + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    if (lane == 0) __hip_atomic_store(&sProg[q & (QR_MAXBLK - 1)], (q << 10) | 1023, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __syncthreads();
  if (RLDS) {   // publish the triangle (zeros below the diagonal and in the padding columns)
    for (int e = tid; e < n * ldR; e += 64 * QR_NW) {
      const int k = e / ldR, col = e - k * ldR;
      Rt[e] = (col >= k && col <= n) ? sR[k * (n + 1) - k * (k - 1) / 2 + col - k] : S(0);
    }
  }
}

template <class S, int NC, bool RLDS>
static void launch_compress_impl(const Dev<S>& d, int b0, int nb, hipStream_t st, int phase, size_t lds) {
  constexpr int RW = (sizeof(S) == 4 && NC <= 3) ? 32 : 16;
  auto kern = k_qr_update<S, NC, RW, RLDS>;
  if (phase != 2) hipLaunchKernelGGL(kern, dim3(d.nchunk, nb), dim3(64 * QR_NW), lds, st, d, b0, 1, 0);
  if (phase == 1) return;
  if (d.nchunk > 1) hipLaunchKernelGGL(kern, dim3(1, nb), dim3(64 * QR_NW), lds, st, d, b0, 2, 0);
}
template <class S, int NC>
static void launch_compress_nc(const Dev<S>& d, int b0, int nb, hipStream_t st, int phase) {
  const size_t base = 4 * QR_MAXBLK;
  const size_t tri = sizeof(S) * (size_t)(d.n6cap + 1) * (d.n6cap + 2) / 2;
  if (base + tri <= 150 * 1024) launch_compress_impl<S, NC, true>(d, b0, nb, st, phase, base + tri);
  else launch_compress_impl<S, NC, false>(d, b0, nb, st, phase, base);
}

template <class S>
void launch_compress(const Dev<S>& d, int b0, int nb, hipStream_t st, int phase) {
  if (nb <= 0) return;
  const int nc = d.ldR / 64;
  switch (nc) {
    case 1: launch_compress_nc<S, 1>(d, b0, nb, st, phase); break;
    case 2: launch_compress_nc<S, 2>(d, b0, nb, st, phase); break;
    case 3: launch_compress_nc<S, 3>(d, b0, nb, st, phase); break;
    case 4: launch_compress_nc<S, 4>(d, b0, nb, st, phase); break;
    case 5: launch_compress_nc<S, 5>(d, b0, nb, st, phase); break;
    default: launch_compress_nc<S, 6>(d, b0, nb, st, phase); break;
  }
}

#ifdef MSCKF_ABLATE
void qr_debug_set(int idx, int val) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_qr_dbg), &val, sizeof(int), idx * sizeof(int)); }
#endif

// one-time, per-device setup (called from msckf_hip_create after hipSetDevice): LDS limit of every instantiation
template <class S, int NC> static void qr_setup_nc() {
  constexpr int RW = (sizeof(S) == 4 && NC <= 3) ? 32 : 16;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_qr_update<S, NC, RW, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_qr_update<S, NC, RW, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}
template <class S> static void qr_setup_s() { qr_setup_nc<S, 1>(); qr_setup_nc<S, 2>(); qr_setup_nc<S, 3>(); qr_setup_nc<S, 4>(); qr_setup_nc<S, 5>(); qr_setup_nc<S, 6>(); }
void qr_device_setup() { qr_setup_s<float>(); qr_setup_s<double>(); }

template void launch_compress<float>(const Dev<float>&, int, int, hipStream_t, int);
template void launch_compress<double>(const Dev<double>&, int, int, hipStream_t, int);

}  // namespace msckf
