// kernels_qr.hip -- QR compression of the stacked projected Jacobian (msckf.h:1338-1366) as a streaming
// TSQR on gfx950.
//
// The reference stacks H_o (m x D, m ~ 5 600 for 200 tracks in a 30-camera window), runs a dense
// HouseholderQR, forms the FULL m x m Q and a dense m x m R_o (msckf.h:1343-1366).  Here H_o is never
// materialised: every workgroup owns an n x (n+1) upper-triangular work matrix [R | Q^T r] in HBM/L2
// (n = 6 N camera columns -- the 15 IMU columns of H_o are identically zero, msckf.h:949) and folds blocks
// of 4*RW rows into it with structured Householder reflectors ("QR update" of [R; B]).  Rows are
// regenerated on the fly from the per-track compact form written by k_feature:
//       H_o_j[i, 6c+d] = [row 3+i belongs to obs c] Hx_c[(3+i)&1][d] - V[3+i,:] . Z_c[:, d]
// Block layout inside a workgroup (256 threads = 4 wavefronts):
//   wave h owns rows h*RW .. h*RW+RW-1 of the block, lane l owns columns l, l+64, ... (NC per lane);
//   the block lives in registers (RW*NC per lane).  Step k: the wave that needs column k reads it from
//   the owner lane with v_readlane (no LDS), every wave forms partial dot products over its rows, one
//   LDS exchange + one barrier combines them (the norm of column k comes out of the same exchange as
//   its self-product), then R's row k and the block are updated.  R's row k is only touched in step k, so
//   it is streamed from/to L2 with coalesced row accesses.
// TSQR: stage 1 = nchunk independent row chunks per trajectory, stage 2 = binary-tree merge of the chunk
// triangles (same kernel, rows sourced from another chunk's R).  With isotropic pixel noise
// (u_var' == v_var', the configuration BASELINE.json is quoted on) the Kalman update depends on the
// stack only through H_o^T H_o and H_o^T r_o, so any orthogonal compression gives the reference's result;
// R_n = sigma^2 I exactly.
#include "dev_common.h"

namespace msckf {

template <class S, int NC, int RW>
__global__ __launch_bounds__(256) void k_qr_update(Dev<S> d, int b0, int stage, int level) {
  const int b = b0 + blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, h = tid >> 6;
  __shared__ S sPart[2][4][64 * NC];
  __shared__ int sKmin[4];
  const int n = 6 * d.ncam[b];
  const int ldR = d.ldR;
  const int m_cap = d.m_cap, f_cap = d.f_cap;
  const int* rs = d.row_start + (long)b * (f_cap + 1);
  const int F = d.stats[(long)b * STAT_STRIDE + STAT_NTRACKS];
  const int m_total = d.stats[(long)b * STAT_STRIDE + STAT_MROWS];
  if (m_total == 0) return;

  S* Rt;            // target triangle
  const S* Rs = nullptr;  // stage 2: source triangle
  int row_begin, row_end;
  if (stage == 1) {
    const int c = blockIdx.x;
    Rt = d.Rbuf + ((long)b * d.nchunk + c) * (long)d.n6cap * ldR;
    row_begin = (int)((long)m_total * c / d.nchunk);
    row_end = (int)((long)m_total * (c + 1) / d.nchunk);
    for (int e = tid; e < n * ldR; e += 256) Rt[e] = 0;   // fresh triangle
    __syncthreads();
  } else {
    const int tgt = (2 * blockIdx.x) << level, src = tgt + (1 << level);
    if (src >= d.nchunk) return;
    Rt = d.Rbuf + ((long)b * d.nchunk + tgt) * (long)d.n6cap * ldR;
    Rs = d.Rbuf + ((long)b * d.nchunk + src) * (long)d.n6cap * ldR;
    row_begin = 0; row_end = n;
  }
  const int BR = 4 * RW;
  int buf = 0;
  for (int blk0 = row_begin; blk0 < row_end; blk0 += BR) {
    // ---------------- load the block: RW rows x NC columns per lane
    S Bv[RW][NC];
    int kmin = n;
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      const int gr = blk0 + h * RW + r;
      if (gr >= row_end) {
#pragma unroll
        for (int j = 0; j < NC; ++j) Bv[r][j] = 0;
        continue;
      }
      if (stage == 1) {
        // track owning stacked row gr: last t with rs[t] <= gr
        int lo = 0, hi = F;   // invariant rs[lo] <= gr < rs[hi]
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (rs[mid] <= gr) lo = mid; else hi = mid; }
        const int t = lo;
        const long tb = (long)b * f_cap + t;
        const int row = 3 + (gr - rs[t]);            // row of Q^T [H_x | r]
        const int cobs = row >> 1, sub = row & 1;
        const S* Vr = d.trk_V + (tb * 2 * m_cap + row) * 4;
        const S v0 = Vr[0], v1 = Vr[1], v2 = Vr[2];
        const signed char* inv = d.trk_inv + tb * d.n_cap;
        const S* Hx = d.trk_Hx + (tb * m_cap) * 12;
        const S* Z = d.trk_Z + (tb * m_cap) * 18;
        const int first = 6 * d.trk_first[tb];
        kmin = first < kmin ? first : kmin;
#pragma unroll
        for (int j = 0; j < NC; ++j) {
          const int col = lane + 64 * j;
          S val = 0;
          if (col < n) {
            const int s = col / 6, dd = col - 6 * s;
            const int c = inv[s];
            if (c >= 0) {
              const S* z = Z + c * 18 + dd;
              val = -(v0 * z[0] + v1 * z[6] + v2 * z[12]);
              if (c == cobs) val += Hx[c * 12 + sub * 6 + dd];
            }
          } else if (col == n) {
            val = d.trk_ro[tb * 2 * m_cap + row];
          }
          Bv[r][j] = val;
        }
      } else {
        kmin = gr < kmin ? gr : kmin;
#pragma unroll
        for (int j = 0; j < NC; ++j) {
          const int col = lane + 64 * j;
          Bv[r][j] = (col <= n) ? Rs[(long)gr * ldR + col] : S(0);
        }
      }
    }
    if (lane == 0) sKmin[h] = kmin;
    __syncthreads();
    kmin = min(min(sKmin[0], sKmin[1]), min(sKmin[2], sKmin[3]));
    __syncthreads();

    // ---------------- fold the block into R: Householder steps k = kmin .. n-1
#pragma unroll
    for (int jk = 0; jk < NC; ++jk) {
      const int l_lo = max(kmin - 64 * jk, 0);
      const int l_hi = min(64, n - 64 * jk);
      for (int lk = l_lo; lk < l_hi; ++lk) {
        const int k = 64 * jk + lk;
        // R row k (coalesced; only slots >= jk can hold columns >= k)
        S rk[NC];
#pragma unroll
        for (int j = jk; j < NC; ++j) rk[j] = Rt[(long)k * ldR + lane + 64 * j];
        // pivot column of this wave's rows, from the owner lane
        S xr[RW];
#pragma unroll
        for (int r = 0; r < RW; ++r) xr[r] = wave_bcast(Bv[r][jk], lk);
        // partial dot products x'^T b_j over this wave's rows
        S part[NC];
#pragma unroll
        for (int j = jk; j < NC; ++j) {
          S s = 0;
#pragma unroll
          for (int r = 0; r < RW; ++r) s += xr[r] * Bv[r][j];
          part[j] = s;
          sPart[buf][h][64 * j + lane] = s;
        }
        __syncthreads();
        S tot[NC];
#pragma unroll
        for (int j = jk; j < NC; ++j)
          tot[j] = (sPart[buf][0][64 * j + lane] + sPart[buf][1][64 * j + lane]) + (sPart[buf][2][64 * j + lane] + sPart[buf][3][64 * j + lane]);
        const S sigma = (sPart[buf][0][k] + sPart[buf][1][k]) + (sPart[buf][2][k] + sPart[buf][3][k]);
        buf ^= 1;
        if (sigma <= Lim<S>::tiny()) continue;          // zero tail: the step is the identity
        const S x0 = wave_bcast(rk[jk], lk);
        S beta = dsqrt(x0 * x0 + sigma);
        if (x0 >= S(0)) beta = -beta;
        const S inv = S(1) / (x0 - beta);
        const S tau = (beta - x0) / beta;
#pragma unroll
        for (int j = jk; j < NC; ++j) {
          const int col = lane + 64 * j;
          if (col > k) {
            const S w = rk[j] + tot[j] * inv;
            const S tw = tau * w;
            rk[j] -= tw;
            const S cj = tw * inv;
#pragma unroll
            for (int r = 0; r < RW; ++r) Bv[r][j] -= cj * xr[r];
          } else if (col == k) {
            rk[j] = beta;
#pragma unroll
            for (int r = 0; r < RW; ++r) Bv[r][j] = 0;
          }
        }
        if (h == 0) {
#pragma unroll
          for (int j = jk; j < NC; ++j) {
            const int col = lane + 64 * j;
            if (col >= k) Rt[(long)k * ldR + col] = rk[j];
          }
        }
        (void)part;
      }
    }
    __syncthreads();   // R rows written by wave 0 must be visible to every wave before the next block
  }
}

template <class S, int NC>
static void launch_compress_nc(const Dev<S>& d, int b0, int nb, hipStream_t st, int phase) {
  constexpr int RW = 16;
  if (phase != 2) hipLaunchKernelGGL((k_qr_update<S, NC, RW>), dim3(d.nchunk, nb), dim3(256), 0, st, d, b0, 1, 0);
  if (phase == 1) return;
  for (int level = 0; (1 << level) < d.nchunk; ++level) {
    const int pairs = (d.nchunk + (2 << level) - 1) / (2 << level);
    hipLaunchKernelGGL((k_qr_update<S, NC, RW>), dim3(pairs, nb), dim3(256), 0, st, d, b0, 2, level);
  }
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

template void launch_compress<float>(const Dev<float>&, int, int, hipStream_t, int);
template void launch_compress<double>(const Dev<double>&, int, int, hipStream_t, int);

}  // namespace msckf
