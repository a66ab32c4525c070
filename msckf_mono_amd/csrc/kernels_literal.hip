// kernels_literal.hip -- the reference's compression under anisotropic pixel noise (u_var' != v_var'), built the reference's
// way on the device: R_o_j = A_j^T R_j A_j per track, HouseholderQR of the stack in column order with the zero-tail rule
// (rows 0..14 of H_o verbatim), R_n = Q_1^T R_o Q_1 (msckf.h:423-431, 1343-1366) -- see literal_core.h, which holds the
// algorithm (shared with the host build that tests/test_literal_core.py checks against the oracle on the CPU).
//
// One workgroup of 1024 threads per trajectory, f64.  The reference's sequence of Householder steps runs on a compressed
// representation (literal_core.h, literal_compact): the rows that can become pivot rows (the first 15 + 6N) explicitly, all
// the others through their Gram matrix -- k_gram's f64 H_o^T H_o minus the explicit rows -- so that a step costs O(n^2)
// instead of a pass over the m x n stack, which is never built (the sweep over the dense stack, literal_general, stays
// selectable with MSCKF_HIP_LITERAL_ROUTE=1 for tests and A/B runs: 146 ms against ~10 per update at a 30-camera window).  The output is the
// information matrix Lam^ = [T_H | r_n]^T R_n^-1 [T_H | r_n] in the place where k_gram leaves H_o^T H_o, so that the blocked
// Cholesky and the Kalman stage run unchanged with sigma^2 = 1.
#include "dev_common.h"
#include "literal_core.h"

namespace msckf {

constexpr int LIT_LDS_DOUBLES = 8000;   // 62.5 KB: chunks of rows of G for the G^T G stage (two workgroups per CU still fit)

template <class S>
__global__ __launch_bounds__(1024) void k_literal(Dev<S> d, int b0, int nb) {
  const int bi = blockIdx.x, b = b0 + bi;
  if (bi >= nb) return;
  const S* prm = d.prm + (long)b * PRM_STRIDE;
  if (prm[PRM_LIT] == S(0)) return;                          // isotropic (or pre-whitened) trajectory: k_gram's Lam^ stands
  int* st = d.stats + (long)b * STAT_STRIDE;
  if (st[STAT_MROWS] == 0) return;
  __shared__ double red[40];                                 // two reductions at a time (literal_core.h: wg_sum2): 2 x 16 wavefronts
  extern __shared__ double lit_lds[];
  lit::Ctx c;
  c.tid = threadIdx.x; c.nt = blockDim.x; c.lane = threadIdx.x & 63; c.wave = threadIdx.x >> 6; c.nw = blockDim.x >> 6; c.red = red; c.tim = d.lit.tim ? d.lit.tim + (long)b * 16 : nullptr; c.lds = lit_lds; c.lds_doubles = LIT_LDS_DOUBLES;
  const LitBufs& L = d.lit;
  const int n1 = d.n6cap + 1;
  lit::Args<S> a;
  a.F = d.trk_n[(long)bi * d.wl_stride_n];
  a.m_cap = d.m_cap;
  a.N = d.ncam[b];
  a.status = d.trk_status + (long)b * d.f_cap; a.inc_bit = ST_INCLUDED;
  a.M = d.trk_M + (long)bi * d.wl_stride_f;
  if (d.trk_off) { a.slots = d.trk_slots; a.off = d.trk_off + (long)bi * d.wl_stride_f; }
  else { a.slots = d.trk_slots + (long)bi * d.wl_stride_o; a.off = nullptr; }
  a.Hx = d.trk_Hx + (long)b * d.f_cap * d.m_cap * 12;
  a.rw = d.trk_rw + (long)b * d.f_cap * 2 * d.m_cap;
  a.u_var = (double)prm[PRM_UVAR]; a.v_var = (double)prm[PRM_VVAR]; a.tol = L.tol;
  a.ldx = L.ldx; a.X = L.X ? L.X + (long)b * L.ldx * n1 : nullptr;
  a.tau = L.tau + (long)b * (2 * n1 + 2);
  a.Vf = L.Vf + (long)b * d.f_cap * 2 * d.m_cap * 3; a.Tf = L.Tf + (long)b * d.f_cap * 9;
  a.row0 = L.row0 + (long)b * (d.f_cap + 1); a.obs0 = L.obs0 + (long)b * (d.f_cap + 1); a.otrk = L.otrk + (long)b * L.ldg;
  a.kept = L.kept + (long)b * L.kept_stride;
  a.r_cap = L.r_cap; a.TH = L.TH + (long)b * L.r_cap * n1;
  a.ldg = L.ldg; a.G = L.G ? L.G + (long)b * L.ldg * L.r_cap : nullptr;
  a.ldz = L.ldz; a.Z = L.Z + (long)b * L.ldz * L.ldz;
  a.Lam = d.Lam + (long)b * d.ldR * d.ldR; a.ldL = d.ldR;
  a.info = L.info + (long)b * 8;
  a.LamIn = a.Lam; a.lam_part = d.lam_part; a.gram_parts = (d.compress && d.ldR <= 192 && d.lam_part > 0) ? (d.gram_parts >= 3 ? d.gram_parts : 3) : 1;
  a.W2 = L.W2 + (long)b * L.w2_stride;
  lit::literal_compress(c, a, L.route);
  // the blocked Cholesky adds the split-K copies of Lam^ that k_gram leaves (Dev::lam_part apart): none here
  if (d.lam_part) {
    const int n = 6 * a.N;
    for (int cpy = 1; cpy < 4; ++cpy) {
      double* Lc = d.Lam + cpy * d.lam_part + (long)b * d.ldR * d.ldR;
      for (long e = threadIdx.x; e < (long)(n + 1) * (n + 1); e += blockDim.x) {
        const int hi = (int)(e / (n + 1)), lo = (int)(e - (long)hi * (n + 1));
        if (lo <= hi) Lc[(long)hi * d.ldR + lo] = 0.0;
      }
    }
  }
}

template <class S>
void launch_literal(const Dev<S>& d, int b0, int nb, hipStream_t st) {
  if (nb <= 0 || !d.lit.W2) return;
  hipLaunchKernelGGL(k_literal<S>, dim3(nb), dim3(1024), LIT_LDS_DOUBLES * sizeof(double), st, d, b0, nb);
}
template void launch_literal<float>(const Dev<float>&, int, int, hipStream_t);
template void launch_literal<double>(const Dev<double>&, int, int, hipStream_t);

}  // namespace msckf
