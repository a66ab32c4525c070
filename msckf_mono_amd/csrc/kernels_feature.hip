// kernels_feature.hip -- per-track stage of MSCKF::marginalize, one wavefront (64 lanes) per track.
//
//   checkMotion          msckf.h:980-1025      lane i holds observation i / camera state i of the track
//   initializePosition   msckf.h:1147-1285     (+ generateInitialGuess :1126-1145, jacobian :1287-1323, cost :1027-1047)
//   calcResidual         msckf.h:960-978
//   calcMeasJacobian     msckf.h:905-958       2x6 blocks with the observability projection
//   null-space           msckf.h:954-957       3 Householder reflectors of H_f_j (the reference takes the
//                                              last 2M-3 columns of JacobiSVD's U; only the span matters)
//   gatingTest           msckf.h:1103-1124     5% quantile of chi2(M+1), sigma^2 = u_var'
//
// What is NOT done the reference's way: H_o_j = A_j^T H_x_j is never formed.  With
// Q = I - V T V^T (compact WY of the 3 reflectors) the projected block is
//     H_o_j = (H_x_j)[3:, :] - V[3:, :] * Z,     Z = T^T V^T H_x_j   (3 x 6M, block-local per lane)
// so a track is handed to the TSQR route as {H_x blocks, V, Z scattered to state columns, r_o} instead of
// (2M-3)*(15+6N); for the information-form route (default, kernels_gram.hip) it publishes B = Q_f^T [H_x | r] in f64
// (reflectors redone in f64 so that H_o^T H_o = H_x^T H_x - B^T B holds to f64 rounding), the whitened residual and
// the slot -> observation map.  The gate uses G = H_x P_cc H_x^T assembled from 6x6 blocks of P (192 M^2 flop instead
// of the dense 2 rho D^2) and never projects it: gamma = r_o^T S^-1 r_o is the generalized-least-squares residual
// min_x (r - H_f x)^T N^-1 (r - H_f x), N = G + sigma^2 I, so ONE register-resident Cholesky of N (8 x 8 lane grid,
// specialised on the track length) with [r^T ; H_f^T] riding along yields it from a 4 x 4 Schur corner (gate_chol).
#include "chi2_table.h"
#include "dev_common.h"

namespace msckf {

__constant__ double c_chi2[99];
#ifdef MSCKF_ABLATE
__device__ int g_feat_dbg = 0;   // ablation knob of the -DMSCKF_ABLATE build (scripts/feat_ablate.py); not in the product library
#endif

template <class S> struct Pose { M3<S> R; V3<S> t; };

// Reciprocal / square root inside the Levenberg-Marquardt iteration.  double: IEEE (the 1e-6 parity of the double filter
// follows the reference's iterates).  float: hardware rcp + one Newton step (<= 1 ulp) instead of the ~10-instruction IEEE
// division sequence -- the iteration is 28 % of the kernel's VALU instructions, half of them divisions; the float filter's
// tolerances (tests/helpers.py:check_tracks) are four orders of magnitude above the difference.
template <class S> __device__ __forceinline__ S lm_rcp(S x);
template <> __device__ __forceinline__ double lm_rcp<double>(double x) { return 1.0 / x; }
template <> __device__ __forceinline__ float lm_rcp<float>(float x) { const float r = __builtin_amdgcn_rcpf(x); return r * __builtin_fmaf(-x, r, 2.0f); }
template <class S> __device__ __forceinline__ S lm_sqrt(S x);
template <> __device__ __forceinline__ double lm_sqrt<double>(double x) { return sqrt(x); }
template <> __device__ __forceinline__ float lm_sqrt<float>(float x) { return __builtin_amdgcn_sqrtf(x); }

template <class S>
__device__ __forceinline__ S tri_cost(const Pose<S>& T, S a, S b, S rho, S zx, S zy) {  // :1027-1047
  const V3<S> h = mulv(T.R, mk3(a, b, S(1))) + (rho * T.t);
  if (sizeof(S) == 4) { const S iz = lm_rcp(h.z); const S dx = h.x * iz - zx, dy = h.y * iz - zy; return dx * dx + dy * dy; }
  const S dx = h.x / h.z - zx, dy = h.y / h.z - zy;
  return dx * dx + dy * dy;
}

// 3x3 LDL^T solve without pivoting (restates Eigen's .ldlt().solve() for the SPD system of :1222)
template <class S>
__device__ __forceinline__ void ldlt3(const S A[6], S lam, const S b[3], S x[3]) {
  // A packed: a00 a01 a02 a11 a12 a22
  const S a00 = A[0] + lam, a01 = A[1], a02 = A[2], a11 = A[3] + lam, a12 = A[4], a22 = A[5] + lam;
  const S d0 = a00;
  S y0, y1, y2, l10, l20, l21;
  if (sizeof(S) == 4) {
    const S i0 = lm_rcp(d0);
    l10 = a01 * i0; l20 = a02 * i0;
    const S d1 = a11 - l10 * l10 * d0, i1 = lm_rcp(d1);
    l21 = (a12 - l20 * l10 * d0) * i1;
    const S d2 = a22 - l20 * l20 * d0 - l21 * l21 * d1;
    y0 = b[0]; y1 = b[1] - l10 * y0; y2 = b[2] - l20 * y0 - l21 * y1;
    y0 *= i0; y1 *= i1; y2 *= lm_rcp(d2);
  } else {
    l10 = a01 / d0; l20 = a02 / d0;
    const S d1 = a11 - l10 * l10 * d0;
    l21 = (a12 - l20 * l10 * d0) / d1;
    const S d2 = a22 - l20 * l20 * d0 - l21 * l21 * d1;
    y0 = b[0]; y1 = b[1] - l10 * y0; y2 = b[2] - l20 * y0 - l21 * y1;
    y0 /= d0; y1 /= d1; y2 /= d2;
  }
  x[2] = y2; x[1] = y1 - l21 * x[2]; x[0] = y0 - l10 * x[1] - l20 * x[2];
}

// Householder QR of the 2M x 3 block H_f (rows 2*lane, 2*lane+1 live in this lane) as compact WY:
// Q = I - V T V^T, V unit lower trapezoidal (row-local), T upper triangular.  hf is overwritten.
template <class T>
__device__ __forceinline__ void house3(T hf[2][3], int lane, T v[2][3], T Tm[3][3]) {
  T tau[3];
  const int row0 = 2 * lane;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    T t2 = 0;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) if (row0 + s2 > k) t2 += hf[s2][k] * hf[s2][k];
    t2 = wave_sum(t2);
    const T c0 = wave_bcast(hf[k & 1][k], k >> 1);
    T inv = 0;
    if (t2 <= Lim<T>::tiny()) { tau[k] = 0; }
    else {
      // beta = -sign(c0) |x|, v = x / (c0 - beta), tau = (beta - c0) / beta with the hardware rsqrt / rcp seeds + Newton steps
      // (~1 ulp; the IEEE sqrt and two divisions were 80 of the double instance's instructions per reflector)
      const T nn = c0 * c0 + t2, rs = fast_rsqrt(nn);
      T beta = nn * rs, ibeta = rs;
      if (c0 >= T(0)) { beta = -beta; ibeta = -ibeta; }
      inv = fast_rcp(c0 - beta);
      tau[k] = (beta - c0) * ibeta;
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const int row = row0 + s2;
      v[s2][k] = row > k ? hf[s2][k] * inv : (row == k ? T(1) : T(0));
    }
    // apply to the remaining columns of H_f
#pragma unroll
    for (int j = k + 1; j < 3; ++j) {
      T s = v[0][k] * hf[0][j] + v[1][k] * hf[1][j];
      s = wave_sum(s) * tau[k];
      hf[0][j] -= s * v[0][k];
      hf[1][j] -= s * v[1][k];
    }
  }
  const T d01 = wave_sum(v[0][0] * v[0][1] + v[1][0] * v[1][1]);
  const T d02 = wave_sum(v[0][0] * v[0][2] + v[1][0] * v[1][2]);
  const T d12 = wave_sum(v[0][1] * v[0][2] + v[1][1] * v[1][2]);
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) Tm[i][j] = 0;
  Tm[0][0] = tau[0]; Tm[1][1] = tau[1]; Tm[2][2] = tau[2];
  Tm[0][1] = -tau[1] * Tm[0][0] * d01;
  Tm[0][2] = -tau[2] * (Tm[0][0] * d02 + Tm[0][1] * d12);
  Tm[1][2] = -tau[2] * Tm[1][1] * d12;
}

// packed lower triangle: element (i, j), j <= i
#define TRI(i, j) ((i) * ((i) + 1) / 2 + (j))
#define SYM(i, j) ((i) >= (j) ? TRI(i, j) : TRI(j, i))

// Gate statistic without the projected block.  With N = G + sigma^2 I (2M x 2M, G = H_x P_cc H_x^T) and A any basis of the
// left null space of H_f,
//     gamma = r_o^T (A^T N A)^-1 r_o = min_x (r - H_f x)^T N^-1 (r - H_f x) = y^T y - b^T C^-1 b,
//     y = L^-1 r,  Y = L^-1 H_f,  b = Y^T y,  C = Y^T Y,   N = L L^T
// (the generalized-least-squares identity; gatingTest, msckf.h:1103-1124, computes the left-hand side).  So the gate is ONE
// Cholesky of N with the four rows [r^T ; H_f^T] riding along and a zero 4 x 4 corner: after the 2M pivots the corner holds
// -[y^T y, b^T ; b, C] (the Schur complement).  No null-space reflectors in working precision, no G V product, no rank-6
// correction E of G, no gamma accumulation per pivot -- a third of the kernel's instructions in the form this replaces
// (S = (Q^T G Q)[3:,3:] assembled from G, V and E = (G V) T - 1/2 V (T^T V^T G V T)).
//
// Register-resident: the wavefront is an 8 x 8 lane grid, lane (tx,ty) owns elements (8a+tx, 8b+ty) of the lower triangle of
// the (2M + 4)-square matrix, NB = number of 8 x 8 blocks in use (compile-time: a short track runs a proportionally
// shorter instruction stream).  sG: packed lower triangle, rows 0 .. 2M-1
// = G, rows 2M .. 2M+3 = [r ; H_f columns] (their corner entries are ignored).  Returns false when a pivot is not
// positive; the corner is left in sC[qi * 4 + qj], qj <= qi.
template <class S, int NB>
__device__ __forceinline__ bool gate_chol(const S* sG, S* sC, int lane, int R2, S sig2) {
  bool spd = true;
  const int tx = lane & 7, ty = lane >> 3, nr = R2 + 4;
  S A[NB][NB];
  int rbase[NB];   // packed-triangle offset of row 8 a2 + tx (a block below the diagonal block never needs SYM's swap)
#pragma unroll
  for (int a2 = 0; a2 < NB; ++a2) rbase[a2] = TRI(8 * a2 + tx, 0);
#pragma unroll
  for (int b2 = 0; b2 < NB; ++b2) {
    const int j = 8 * b2 + ty;
#pragma unroll
    for (int a2 = b2; a2 < NB; ++a2) {
      const int i = 8 * a2 + tx;
      S val = 0;
      if (i < nr && j < nr && !(i >= R2 && j >= R2)) {
        val = sG[a2 > b2 ? rbase[a2] + j : SYM(i, j)];
        if (i == j) val += sig2;
      }
      A[a2][b2] = val;
    }
  }
  // Pivot k = 8 kb + kk: the pivot itself is read from its owner lane (kk, kk) by v_readlane (an SGPR: the reciprocal starts at
  // once), and the pivot column reaches the lanes through the LDS crossbar WITHOUT touching LDS memory (ds_bpermute): row
  // 8 a + tx of the column lives in lane (tx, kk), row 8 b + ty in lane (ty, kk).  One crossbar round trip per pivot, no
  // barrier, no exec-masked branches (the write column / barrier / read pivot / read column exchange this replaces was
  // three to four serialized LDS round trips per pivot).
#pragma unroll
  for (int kb = 0; kb < NB; ++kb) {
    const int kk_hi = min(8, R2 - 8 * kb);
    for (int kk = 0; kk < kk_hi; ++kk) {
      const S dkk = wave_bcast(A[kb][kb], kk * 9);
      spd = spd && dkk > S(0);                        // (no early exit: a non-positive pivot only poisons entries nobody reads; the caller drops the result)
      // L is never needed itself: the update is A(i, j) -= A(i, k) A(j, k) / d -- one reciprocal (hardware seed; double:
      // + Newton) and one scaled operand instead of rsqrt + Newton and two scaled operands
      const S dinv2 = sizeof(S) == 4 ? (S)__builtin_amdgcn_rcpf((float)dkk) : fast_rcp(dkk);
      const int src_i = (kk * 8 + tx) << 2, src_j = (kk * 8 + ty) << 2;
      S li[NB], lj[NB];
#pragma unroll
      for (int a2 = kb; a2 < NB; ++a2) li[a2] = lane_gather(A[a2][kb], src_i);
#pragma unroll
      for (int b2 = kb; b2 < NB; ++b2) lj[b2] = lane_gather(A[b2][kb], src_j);
      // No masks on the diagonal block's multipliers: a lane with tx <= kk or ty <= kk holds an entry of a row or column that is
      // finished (or of the block's unused upper triangle) -- nothing reads those again, live entries (row, column > k) only
      // ever meet multipliers of live rows (kernels_chol.hip's diagonal block dropped its masks the same way)
#pragma unroll
      for (int a2 = kb; a2 < NB; ++a2) li[a2] *= dinv2;
#pragma unroll
      for (int a2 = kb; a2 < NB; ++a2)
#pragma unroll
        for (int b2 = kb; b2 <= a2; ++b2) A[a2][b2] -= li[a2] * lj[b2];
    }
  }
  __syncthreads();
  // the 4 x 4 corner (rows / columns 2M .. 2M+3) sits in the last two block rows / columns in use; NB may exceed the blocks
  // in use by one (the long-track instantiations come in steps of two), so the last three are searched
#pragma unroll
  for (int a2 = (NB >= 3 ? NB - 3 : 0); a2 < NB; ++a2)
#pragma unroll
    for (int b2 = (NB >= 3 ? NB - 3 : 0); b2 <= a2; ++b2) {
      const int qi = 8 * a2 + tx - R2, qj = 8 * b2 + ty - R2;
      if (qi >= 0 && qi < 4 && qj >= 0 && qj <= qi) sC[qi * 4 + qj] = A[a2][b2];
    }
  __syncthreads();
  return spd;
}

// gate_chol for the long tracks of a float filter (31 .. 62 observations: 2M + 4 up to 128, NB up to 16 blocks of 8), WITHOUT the
// whole packed matrix in LDS.  A 60-observation track's triangle is 31 KB -- four wavefronts per compute unit, and the long
// bin of cfg5's tracks was bound by exactly that (k_feature<LONG> 4.4 of the frame's 12 ms).  The lane grid takes the matrix
// from a staging area anyway, so it takes it sixteen columns at a time: `assemble(c0)` fills rows c0 .. 2M+3 of columns
// c0 .. c0+15 ([row - c0][17]: 8.4 KB at most), the two block columns are picked up into the registers, the next sixteen
// columns reuse the space.  Same entries, same pivot loops as gate_chol.
template <int NB, class Asm>
__device__ __forceinline__ bool gate_chol_staged(float* sCol, float* sC, int lane, int R2, float sig2, Asm&& assemble) {
  typedef float S;
  bool spd = true;
  const int tx = lane & 7, ty = lane >> 3, nr = R2 + 4;
  S A[NB][NB];
#pragma unroll
  for (int cb = 0; cb < NB; cb += 2) {
    const int c0 = 8 * cb;
    if (c0 < nr) assemble(c0);
    __syncthreads();
#pragma unroll
    for (int bo = 0; bo < 2; ++bo) {
      const int b2 = cb + bo;
      if (b2 < NB) {
        const int j = 8 * b2 + ty;
#pragma unroll
        for (int a2 = 0; a2 < NB; ++a2) {
          if (a2 < b2) continue;
          const int i = 8 * a2 + tx;
          S val = 0;
          if (i < nr && j < nr && !(i >= R2 && j >= R2)) {
            const int hi = (a2 > b2 || i >= j) ? i : j, lo = (a2 > b2 || i >= j) ? j : i;
            val = sCol[(hi - c0) * 17 + (lo - c0)];
            if (i == j) val += sig2;
          }
          A[a2][b2] = val;
        }
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int kb = 0; kb < NB; ++kb) {
    const int kk_hi = min(8, R2 - 8 * kb);
    for (int kk = 0; kk < kk_hi; ++kk) {
      const S dkk = wave_bcast(A[kb][kb], kk * 9);
      spd = spd && dkk > S(0);
      const S dinv2 = (S)__builtin_amdgcn_rcpf((float)dkk);
      const int src_i = (kk * 8 + tx) << 2, src_j = (kk * 8 + ty) << 2;
      S li[NB], lj[NB];
#pragma unroll
      for (int a2 = kb; a2 < NB; ++a2) li[a2] = lane_gather(A[a2][kb], src_i);
#pragma unroll
      for (int b2 = kb; b2 < NB; ++b2) lj[b2] = lane_gather(A[b2][kb], src_j);
#pragma unroll
      for (int a2 = kb; a2 < NB; ++a2) li[a2] *= dinv2;
#pragma unroll
      for (int a2 = kb; a2 < NB; ++a2)
#pragma unroll
        for (int b2 = kb; b2 <= a2; ++b2) A[a2][b2] -= li[a2] * lj[b2];
    }
  }
  __syncthreads();
#pragma unroll
  for (int a2 = (NB >= 3 ? NB - 3 : 0); a2 < NB; ++a2)
#pragma unroll
    for (int b2 = (NB >= 3 ? NB - 3 : 0); b2 <= a2; ++b2) {
      const int qi = 8 * a2 + tx - R2, qj = 8 * b2 + ty - R2;
      if (qi >= 0 && qi < 4 && qj >= 0 && qj <= qi) sC[qi * 4 + qj] = A[a2][b2];
    }
  __syncthreads();
  return spd;
}

// gamma = y^T y - b^T C^-1 b from the corner -[y^T y, . ; b, C] left by the factorization (c[qi * 4 + qj], qj <= qi).  C is
// 3 x 3 symmetric positive definite (H_f has full column rank for a triangulable feature); a direction H_f says nothing
// about (pivot below rounding of C's diagonal) contributes nothing.
template <class S>
__device__ __forceinline__ S gate_gamma_from_corner(const S* c) {
  const S yy = -c[0];
  const S b0 = -c[4], b1 = -c[8], b2 = -c[12];
  const S c00 = -c[5], c10 = -c[9], c11 = -c[10], c20 = -c[13], c21 = -c[14], c22 = -c[15];
  const S tol = (sizeof(S) == 4 ? S(1e-6) : S(1e-13)) * (c00 + c11 + c22);
  S g = yy;
  // LDL^T forward elimination of [C | b]
  if (c00 > tol) {
    const S i0 = S(1) / c00;
    const S l10 = c10 * i0, l20 = c20 * i0;
    const S d1 = c11 - l10 * c10, e21 = c21 - l20 * c10, f1 = b1 - l10 * b0, f2 = b2 - l20 * b0;
    g -= b0 * b0 * i0;
    if (d1 > tol) {
      const S i1 = S(1) / d1;
      const S l21 = e21 * i1;
      const S d2 = c22 - l20 * c20 - l21 * e21, h2 = f2 - l21 * f1;
      g -= f1 * f1 * i1;
      if (d2 > tol) g -= h2 * h2 / d2;
    } else {
      const S d2 = c22 - l20 * c20;
      if (d2 > tol) g -= f2 * f2 / d2;
    }
  } else {
    if (c11 > tol) {
      const S i1 = S(1) / c11;
      const S l21 = c21 * i1;
      const S d2 = c22 - l21 * c21, h2 = b2 - l21 * b1;
      g -= b1 * b1 * i1;
      if (d2 > tol) g -= h2 * h2 / d2;
    } else if (c22 > tol) g -= b2 * b2 / c22;
  }
  return g > S(0) ? g : S(0);
}

// gate_chol for float filters and at most 8 x 8 blocks, with HALF the LDS-crossbar traffic.  The crossbar (ds_bpermute: ~4 LDS
// cycles per wavefront instruction, ONE crossbar per compute unit for its sixteen wavefronts) is what bounds gate_chol: 2 (NB - kb)
// exchanges per pivot, 3.5 M per launch of the benchmark = 23 us of the compute units' LDS time.  Here the lane grid is laid out
// the other way round -- lane = 8 tx + ty, element (8 a + tx, 8 b + ty) -- so that the pivot column's entry of a lane's ROW sits in
// the lane's own group of 8 (position kk): the row-side multiplier is a DPP broadcast inside the group (quad broadcast + one
// masked row shift, vector ALU only; kk is a template parameter because DPP controls are immediates), and only the column-side
// multiplier L(8 b + ty, k), which lives in lane 8 ty + kk, still crosses the LDS crossbar.  Same arithmetic, same masks-free
// updates as gate_chol.  The packed triangle is read with consecutive lanes on consecutive words.
template <int KK> __device__ __forceinline__ float grp8_bcast_scaled(float v, float s) {
  constexpr int q = KK & 3, QP = q | (q << 2) | (q << 4) | (q << 6);
  const int t = __float_as_int(__int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), QP, 0xF, 0xF, true)) * s);   // every quad: its lane q, scaled (one v_mul_f32_dpp)
  // the group's other quad takes it over: lanes 4..7 of a group from 0..3 (row_shr:4, banks 1 and 3) or 0..3 from 4..7 (row_shl:4, banks 0 and 2)
  return __int_as_float(KK < 4 ? __builtin_amdgcn_update_dpp(t, t, 0x114, 0xF, 0xA, false) : __builtin_amdgcn_update_dpp(t, t, 0x104, 0xF, 0x5, false));
}
template <int NB, int KB, int KK>
__device__ __forceinline__ void gate_pivot_dpp(float (&A)[NB][NB], bool& spd, const int srcj4) {
  const float dkk = wave_bcast(A[KB][KB], KK * 9);
  spd = spd && dkk > 0.0f;
  const float dinv = __builtin_amdgcn_rcpf(dkk);
  float li[NB], lj[NB];
#pragma unroll
  for (int b2 = KB; b2 < NB; ++b2) lj[b2] = lane_gather(A[b2][KB], srcj4 + 4 * KK);
#pragma unroll
  for (int a2 = KB; a2 < NB; ++a2) li[a2] = grp8_bcast_scaled<KK>(A[a2][KB], dinv);
#pragma unroll
  for (int a2 = KB; a2 < NB; ++a2)
#pragma unroll
    for (int b2 = KB; b2 <= a2; ++b2) A[a2][b2] -= li[a2] * lj[b2];
}
template <int NB, int KB>
__device__ __forceinline__ void gate_block_dpp(float (&A)[NB][NB], bool& spd, const int srcj4, const int R2) {
  const int kk_hi = R2 - 8 * KB;                       // pivots of this diagonal block (wave-uniform)
  if (kk_hi > 0) gate_pivot_dpp<NB, KB, 0>(A, spd, srcj4);
  if (kk_hi > 1) gate_pivot_dpp<NB, KB, 1>(A, spd, srcj4);
  if (kk_hi > 2) gate_pivot_dpp<NB, KB, 2>(A, spd, srcj4);
  if (kk_hi > 3) gate_pivot_dpp<NB, KB, 3>(A, spd, srcj4);
  if (kk_hi > 4) gate_pivot_dpp<NB, KB, 4>(A, spd, srcj4);
  if (kk_hi > 5) gate_pivot_dpp<NB, KB, 5>(A, spd, srcj4);
  if (kk_hi > 6) gate_pivot_dpp<NB, KB, 6>(A, spd, srcj4);
  if (kk_hi > 7) gate_pivot_dpp<NB, KB, 7>(A, spd, srcj4);
  if constexpr (KB + 1 < NB) gate_block_dpp<NB, KB + 1>(A, spd, srcj4, R2);
}
template <int NB>
__device__ __forceinline__ bool gate_chol_dpp(const float* sG, float* sC, int lane, int R2, float sig2) {
  bool spd = true;
  const int tx = lane >> 3, ty = lane & 7, nr = R2 + 4;
  float A[NB][NB];
#pragma unroll
  for (int a2 = 0; a2 < NB; ++a2) {
    const int i = 8 * a2 + tx, rb = TRI(i, 0);
#pragma unroll
    for (int b2 = 0; b2 <= a2; ++b2) {
      const int j = 8 * b2 + ty;
      float val = 0;
      if (i < nr && j < nr && !(i >= R2 && j >= R2)) {
        val = sG[a2 > b2 ? rb + j : SYM(i, j)];
        if (i == j) val += sig2;
      }
      A[a2][b2] = val;
    }
  }
  gate_block_dpp<NB, 0>(A, spd, ty << 5, R2);         // column-side source lane 8 ty + kk, as a byte index
  __syncthreads();
#pragma unroll
  for (int a2 = (NB >= 3 ? NB - 3 : 0); a2 < NB; ++a2)
#pragma unroll
    for (int b2 = (NB >= 3 ? NB - 3 : 0); b2 <= a2; ++b2) {
      const int qi = 8 * a2 + tx - R2, qj = 8 * b2 + ty - R2;
      if (qi >= 0 && qi < 4 && qj >= 0 && qj <= qi) sC[qi * 4 + qj] = A[a2][b2];
    }
  __syncthreads();
  return spd;
}

// LONG: tracks of more than 30 observations (2M + 4 > 64) keep the gate's Cholesky in registers too (up to 16 x 16 blocks
// per lane); a separate instantiation, so that the short-track kernel keeps its register budget.
template <class S, bool LONG>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(LONG ? 1 : (sizeof(S) == 4 ? 4 : 2), sizeof(S) == 4 && !LONG ? 4 : 2))) void k_feature(Dev<S> d, int b0, int nb, int lm, int m_lo, int m_hi, int staged) {
  // all tracks of a trajectory on one XCD (xcd_item): the gate's 6 x 6 blocks of P then come out of an L2 that holds 1/8 of
  // the batch's covariances (the (track, trajectory) grid spread every trajectory over all eight: 68 MB fetched per launch for
  // 9.7 MB of covariance; now 17 MB).  Measured and rejected: a PERSISTENT form of this kernel -- one residency
  // of wavefronts pulling tracks from per-XCD ticket queues, so that a set of compute units could be kept free for the
  // one-workgroup-per-trajectory kernels of the other slices: 134 -> 174 us even with one 128-byte line per ticket counter
  // and prefetched tickets (all wavefronts start in phase and hit the same pipes together), and the units kept free were
  // taken by the next slice's per-track wavefronts.
  int bi, t;
  if (!xcd_item(nb, d.f_cap, bi, t)) return;
  const int b = b0 + bi, lane = threadIdx.x;
  const int F = d.trk_n[(long)bi * d.wl_stride_n];
  if (t >= F) return;
  // tracks residualized before this update, for the block-diagonal reduction that runs in k_select's launch (it must not read
  // n_resid while k_select updates it)
  if (t == 0 && lane == 0) d.nres_upd[b] = (int)(d.n_resid[b] > 1000 ? 1000 : d.n_resid[b]);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int m_cap = d.m_cap;
  // N's source G (symmetric, 2M x 2M) plus the four rows that ride along (r, the columns of H_f) as a packed lower triangle:
  // element (i, j), j <= i, at TRI(i, j).  sHx (G stage only) shares its space with sC, where the factorization leaves the
  // 4 x 4 corner: 9.9 KB per wavefront at 30 observations, sixteen per CU.  The LDS layout is sized for lm <= m_cap
  // observations: with long windows launch_feature bins the tracks by length (this launch takes m_lo < M <= m_hi), so that
  // a short track does not hold the 31 KB a 60-observation track needs (5 wavefronts per CU)
  S* sG = reinterpret_cast<S*>(smem_raw);              // [(2 lm + 4)(2 lm + 5) / 2]
  S* sHx = sG + (staged ? (2 * lm + 4) * 17 : (2 * lm + 4) * (2 * lm + 5) / 2);       // [lm][12]  (staged: sixteen columns of the gate matrix at a time, gate_chol_staged)
  S* sC = sHx;                                         // [16]: the corner (gate_chol_staged writes it after its last assemble: H_x is done with by then)
  const int xlen = lm * 12 > 16 ? lm * 12 : 16;
  int* sSlot = reinterpret_cast<int*>(sHx + xlen);

  const long tb = (long)b * d.f_cap + t;               // per-track output index
  const int M = d.trk_M[(long)bi * d.wl_stride_f + t];
  const long wo = wl_first(d, bi, t);
  const S* prm = d.prm + (long)b * PRM_STRIDE;
  const S* imu = d.imu + (long)b * IMU_STRIDE;
  const int ld = d.ld;
  const bool act = lane < M;
  int status = 0;
#ifdef MSCKF_ABLATE
  const int fdbg = g_feat_dbg;
#else
  constexpr int fdbg = 0;
#endif
  if (M <= m_lo || M > m_hi) return;                   // another launch's bin
  if (M < 2 || M > m_cap || M > 64) {                  // cannot be residualized (checkMotion :982 returns false)
    if (lane == 0) { d.trk_status[tb] = 0; d.trk_gamma[tb] = 0; d.trk_first[tb] = 0; }
    return;
  }
  // ---- load this lane's camera state and observation
  const int slot = act ? d.trk_slots[wo + lane] : 0;
  const S* cs = d.cam + ((long)b * d.n_cap + slot) * CAM_STRIDE;
  const Q4<S> qc = ldq(cs);
  const V3<S> pcg = ld3(cs + 4);
  const S zx = act ? d.trk_obs[2 * (wo + lane)] : S(0), zy = act ? d.trk_obs[2 * (wo + lane) + 1] : S(0);
  const M3<S> C = q2rot(qc);
  const V3<S> g = ld3(imu + IG);
  if (lane < lm) sSlot[lane] = act ? slot : -1;
  const int slot_lo = wave_min_i(act ? slot : 0x7fffffff), slot_hi = -wave_min_i(act ? -slot : 0x7fffffff);
  // first camera of the track (lane 0) broadcast
  M3<S> C0; V3<S> p0;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) C0.m[i][j] = wave_bcast(C.m[i][j], 0);
  p0 = mk3(wave_bcast(pcg.x, 0), wave_bcast(pcg.y, 0), wave_bcast(pcg.z, 0));
  const S z0x = wave_bcast(zx, 0), z0y = wave_bcast(zy, 0);

  // ---- checkMotion :980-1025
  if (!(fdbg & 256)) {
    V3<S> dir = mk3(z0x, z0y, S(1));
    dir = (S(1) / dsqrt(dot3(dir, dir))) * dir;
    dir = multv(C0, dir);
    const V3<S> tr = pcg - p0;
    const S par = dot3(tr, dir);
    const V3<S> orth = tr - (par * dir);
    S nrm = (act && lane > 0) ? dsqrt(dot3(orth, orth)) : S(0);
    nrm = wave_max(nrm);
    if (nrm > prm[PRM_TRANS]) status |= ST_MOTION_OK;
  }

  // ---- initializePosition :1147-1285 (Levenberg-Marquardt on inverse depth, wave-reduced sums)
  Pose<S> T;
  T.R = mulmt(C, C0);
  T.t = mulv(C, p0 - pcg);
  S sa, sb, srho;   // solution (alpha, beta, rho)
  {
    // generateInitialGuess with the LAST camera and the first/last observations :1172-1176
    const int L = M - 1;
    S depth;
    {
      const V3<S> m = mulv(T.R, mk3(z0x, z0y, S(1)));
      const S A0 = m.x - zx * m.z, A1 = m.y - zy * m.z;
      const S b0v = zx * T.t.z - T.t.x, b1v = zy * T.t.z - T.t.y;
      depth = (S(1) / (A0 * A0 + A1 * A1)) * (A0 * b0v + A1 * b1v);
      depth = wave_bcast(depth, L);
    }
    const S ix = z0x * depth, iy = z0y * depth, iz = depth;
    sa = ix / iz; sb = iy / iz; srho = S(1) / iz;
  }
  const bool given = d.mode == 1;   // stored feature position supplied by the host (pruneRedundantStates)
  S lambda = S(1e-3), delta_norm = 0;
  S total_cost = wave_sum(act ? tri_cost(T, sa, sb, srho, zx, zy) : S(0));
  bool reduced = false;
  int inner = 0, outer = 0;
  if (!given && !(fdbg & 1)) do {
    S Ab[9];  // a00 a01 a02 a11 a12 a22 b0 b1 b2
    {
      const V3<S> h = mulv(T.R, mk3(sa, sb, S(1))) + (srho * T.t);
      S W[3][3];
      for (int i = 0; i < 3; ++i) { W[i][0] = T.R.m[i][0]; W[i][1] = T.R.m[i][1]; }
      W[0][2] = T.t.x; W[1][2] = T.t.y; W[2][2] = T.t.z;
      S J[2][3];
      S r0, r1, e, w;
      if (sizeof(S) == 4) {
        const S iz = lm_rcp(h.z), xz = h.x * iz, yz = h.y * iz;
        for (int j = 0; j < 3; ++j) {
          J[0][j] = iz * (W[0][j] - xz * W[2][j]);
          J[1][j] = iz * (W[1][j] - yz * W[2][j]);
        }
        r0 = xz - zx; r1 = yz - zy;
        e = lm_sqrt(r0 * r0 + r1 * r1);
        w = (e <= S(0.01)) ? S(1) : S(0.005) * lm_rcp(e);
      } else {
        for (int j = 0; j < 3; ++j) {
          J[0][j] = S(1) / h.z * W[0][j] - h.x / (h.z * h.z) * W[2][j];
          J[1][j] = S(1) / h.z * W[1][j] - h.y / (h.z * h.z) * W[2][j];
        }
        r0 = h.x / h.z - zx; r1 = h.y / h.z - zy;
        e = dsqrt(r0 * r0 + r1 * r1);
        w = (e <= S(0.01)) ? S(1) : S(0.01) / (S(2) * e);
      }
      const S w2 = (w == S(1)) ? S(1) : w * w;
      const S m = act ? w2 : S(0);
      Ab[0] = m * (J[0][0] * J[0][0] + J[1][0] * J[1][0]);
      Ab[1] = m * (J[0][0] * J[0][1] + J[1][0] * J[1][1]);
      Ab[2] = m * (J[0][0] * J[0][2] + J[1][0] * J[1][2]);
      Ab[3] = m * (J[0][1] * J[0][1] + J[1][1] * J[1][1]);
      Ab[4] = m * (J[0][1] * J[0][2] + J[1][1] * J[1][2]);
      Ab[5] = m * (J[0][2] * J[0][2] + J[1][2] * J[1][2]);
      Ab[6] = m * (J[0][0] * r0 + J[1][0] * r1);
      Ab[7] = m * (J[0][1] * r0 + J[1][1] * r1);
      Ab[8] = m * (J[0][2] * r0 + J[1][2] * r1);
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) Ab[i] = wave_sum(Ab[i]);
    // The damping ladder lambda, 10 lambda, 100 lambda, ... (clamped at 1e12) of the inner loop is known before its first
    // step is tried: lane c solves the 3 x 3 system for the c-th rung, all rungs at once, and the loop only evaluates the
    // cost of one candidate per pass (in float the last outer iteration fails all 11 rungs: 11 solves become one).
    S lam_c = lambda;
#pragma unroll
    for (int t2 = 0; t2 < 11; ++t2) { const S up = lam_c * 10 < S(1e12) ? lam_c * 10 : S(1e12); if (t2 < lane) lam_c = up; }
    S dlc[3];
    ldlt3(Ab, lam_c, Ab + 6, dlc);
    int cand = 0;
    do {
      const S dl[3] = {wave_bcast(dlc[0], cand), wave_bcast(dlc[1], cand), wave_bcast(dlc[2], cand)};
      const S na = sa - dl[0], nb = sb - dl[1], nr = srho - dl[2];
      delta_norm = lm_sqrt(dl[0] * dl[0] + dl[1] * dl[1] + dl[2] * dl[2]);
      // A step that no longer changes any of the three parameters (it is below half an ulp of each: the usual fate of the
      // upper rungs of the ladder in the last outer iteration) gives bit for bit the cost already held in total_cost -- the
      // same function of the same arguments -- so the candidate fails without being evaluated (same control flow, same result)
      S new_cost = total_cost;
      if (!(na == sa && nb == sb && nr == srho)) new_cost = wave_sum(act ? tri_cost(T, na, nb, nr, zx, zy) : S(0));
      if (new_cost < total_cost) {
        reduced = true; sa = na; sb = nb; srho = nr; total_cost = new_cost;
        const S lam_now = wave_bcast(lam_c, cand);
        const S l10th = sizeof(S) == 4 ? lam_now * S(0.1) : lam_now / 10;
        lambda = l10th > S(1e-10) ? l10th : S(1e-10);
      } else {
        reduced = false;
        lambda = wave_bcast(lam_c, cand + 1);       // = min(10 * rung, 1e12)
      }
      ++cand;
    } while (inner++ < 10 && !reduced);
    inner = 0;
  } while (outer++ < 10 && delta_norm > S(5e-7));
  const V3<S> fin = mk3(sa / srho, sb / srho, S(1) / srho);
  {
    const V3<S> pos = mulv(T.R, fin) + T.t;
    const int bad = (act && pos.z <= S(0)) ? 1 : 0;
    const bool any_bad = __any(bad);
    const S ncost = total_cost / (S(2) * S(M) * S(M));
    if (!any_bad && !(ncost > prm[PRM_GN])) status |= ST_TRI_VALID;
  }
  V3<S> pf = multv(C0, fin) + p0;   // :1282
  if (given) {
    pf = ld3(d.trk_pfin + tb * 4);
    status |= ST_MOTION_OK | ST_TRI_VALID;
  }

  // ---- calcResidual :960-978 and calcMeasJacobian :915-950 for this lane's observation
  S hx[2][6], hf[2][3], r[2];
  {
    const V3<S> pc = mulv(C, pf - pcg);
    const S X = pc.x, Y = pc.y, Z = pc.z;
    r[0] = zx - X / Z; r[1] = zy - Y / Z;
    S Ji[2][3];
    Ji[0][0] = S(1) * (S(1) / Z); Ji[0][1] = 0; Ji[0][2] = (-X / Z) * (S(1) / Z);
    Ji[1][0] = 0; Ji[1][1] = S(1) * (S(1) / Z); Ji[1][2] = (-Y / Z) * (S(1) / Z);
    const M3<S> sk = skew3(pc);
    S A[2][6];
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 3; ++j) {
        S a = 0, bb = 0;
        for (int k = 0; k < 3; ++k) { a += Ji[i][k] * sk.m[k][j]; bb += Ji[i][k] * C.m[k][j]; }
        A[i][j] = a; A[i][3 + j] = -bb;
      }
    const V3<S> uh = mulv(C, g);
    const V3<S> ut = mulv(skew3(pf - pcg), g);
    const S u[6] = {uh.x, uh.y, uh.z, ut.x, ut.y, ut.z};
    S uu = 0;
    for (int k = 0; k < 6; ++k) uu += u[k] * u[k];
    for (int i = 0; i < 2; ++i) {
      S Au = 0;
      for (int k = 0; k < 6; ++k) Au += A[i][k] * u[k];
      for (int k = 0; k < 6; ++k) {
        const S h = act ? A[i][k] - Au * (S(1) / uu) * u[k] : S(0);
        hx[i][k] = h;
        if (k >= 3) hf[i][k - 3] = -h;
      }
    }
    if (!act) { r[0] = 0; r[1] = 0; }
    // row weights: (1,1) for isotropic noise and on the literal anisotropic route (kernels_literal.hip), (1/sigma_u, 1/sigma_v)
    // when anisotropic noise is handled by pre-whitening
    const S wu = prm[PRM_WU], wv = prm[PRM_WV];
    r[0] *= wu; r[1] *= wv;
    for (int k = 0; k < 6; ++k) { hx[0][k] *= wu; hx[1][k] *= wv; }
    for (int k = 0; k < 3; ++k) { hf[0][k] *= wu; hf[1][k] *= wv; }
    if (d.h16) {   // fp16 Jacobian (dtype MSCKF_HIP_F16H_F32P): rounded here, every consumer below sees the rounded blocks
      for (int i = 0; i < 2; ++i) for (int k = 0; k < 6; ++k) hx[i][k] = (S)__half2float(__float2half_rn((float)hx[i][k]));
      for (int i = 0; i < 2; ++i) for (int k = 0; k < 3; ++k) hf[i][k] = -hx[i][3 + k];
    }
  }

  // ---- information-form compression (compress == 1): B = Q_f^T [H_x | r] in f64.  H_o^T H_o = H_x^T H_x - B^T B
  // holds to f64 rounding only when Q_f is orthonormal to f64 rounding, so the three reflectors are redone in
  // f64 on the (float-rounded) H_f; the gate below keeps the S-precision reflectors.
  double Bq[3][6], cq[3];
  if (d.compress && !(fdbg & 8)) {
   if constexpr (sizeof(S) == 4) {
    // Float filters: only B^T B = [H_x | r]^T H_f (H_f^T H_f)^-1 H_f^T [H_x | r] matters downstream (k_gram), so any B with that
    // Gram matrix will do: B = L^-1 H_f^T [H_x | r] with H_f^T H_f = L L^T (3 x 3, f64).  H_f^T H_x is LOCAL to the lane (the
    // camera's 6 columns meet only this observation's two rows): nine f64 wave sums (H_f^T H_f, H_f^T r) and no broadcast,
    // where the three reflectors took twelve sums and a dozen broadcasts (17 % of the kernel's instructions).  The squared
    // condition of H_f costs ~1e-10 in the projector -- below the float Jacobian's own rounding; double filters keep the
    // reflectors (a low-parallax window needs the projector to 1e-13).
    const double f0[3] = {(double)hf[0][0], (double)hf[0][1], (double)hf[0][2]}, f1[3] = {(double)hf[1][0], (double)hf[1][1], (double)hf[1][2]};
    const double s00 = wave_sum(f0[0] * f0[0] + f1[0] * f1[0]), s01 = wave_sum(f0[0] * f0[1] + f1[0] * f1[1]), s02 = wave_sum(f0[0] * f0[2] + f1[0] * f1[2]);
    const double s11 = wave_sum(f0[1] * f0[1] + f1[1] * f1[1]), s12 = wave_sum(f0[1] * f0[2] + f1[1] * f1[2]), s22 = wave_sum(f0[2] * f0[2] + f1[2] * f1[2]);
    const double r0 = (double)r[0], r1 = (double)r[1];
    const double t0 = wave_sum(f0[0] * r0 + f1[0] * r1), t1 = wave_sum(f0[1] * r0 + f1[1] * r1), t2 = wave_sum(f0[2] * r0 + f1[2] * r1);
    const double tiny = 1e-300;
    const double i00 = fast_rsqrt(s00 > tiny ? s00 : tiny), l10 = s01 * i00, l20 = s02 * i00;
    const double p11 = s11 - l10 * l10, i11 = fast_rsqrt(p11 > tiny ? p11 : tiny), l21 = (s12 - l20 * l10) * i11;
    const double p22 = s22 - l20 * l20 - l21 * l21, i22 = fast_rsqrt(p22 > tiny ? p22 : tiny);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const double h0 = (double)hx[0][k], h1 = (double)hx[1][k];
      const double g0 = f0[0] * h0 + f1[0] * h1, g1 = f0[1] * h0 + f1[1] * h1, g2 = f0[2] * h0 + f1[2] * h1;
      const double b0v = g0 * i00, b1v = (g1 - l10 * b0v) * i11;
      Bq[0][k] = b0v; Bq[1][k] = b1v; Bq[2][k] = (g2 - l20 * b0v - l21 * b1v) * i22;
    }
    cq[0] = t0 * i00; cq[1] = (t1 - l10 * cq[0]) * i11; cq[2] = (t2 - l20 * cq[0] - l21 * cq[1]) * i22;
   } else {
    double hfd[2][3], vd[2][3], Td[3][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int k = 0; k < 3; ++k) hfd[i][k] = (double)hf[i][k];
    house3<double>(hfd, lane, vd, Td);
    double Zd[3][6], yd[3];
    {
      double Wc[3][6], wr[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) {
#pragma unroll
        for (int k = 0; k < 6; ++k) Wc[q][k] = vd[0][q] * (double)hx[0][k] + vd[1][q] * (double)hx[1][k];
        wr[q] = wave_sum(vd[0][q] * (double)r[0] + vd[1][q] * (double)r[1]);
      }
#pragma unroll
      for (int q = 0; q < 3; ++q) {
#pragma unroll
        for (int k = 0; k < 6; ++k) { double a2 = 0; for (int p2 = 0; p2 <= q; ++p2) a2 += Td[p2][q] * Wc[p2][k]; Zd[q][k] = a2; }
        double a3 = 0;
        for (int p2 = 0; p2 <= q; ++p2) a3 += Td[p2][q] * wr[p2];
        yd[q] = a3;
      }
    }
    // rows 0..2 of Q^T [H_x | r]: row i lives in lane i>>1 (row 0,1 -> lane 0, row 2 -> lane 1)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int src = i >> 1, sr = i & 1;
      double vi[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) vi[q] = wave_bcast(vd[sr][q], src);
      const double ri = wave_bcast((double)r[sr], src);
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const double own = (lane == src) ? (double)hx[sr][k] : 0.0;
        Bq[i][k] = own - (vi[0] * Zd[0][k] + vi[1] * Zd[1][k] + vi[2] * Zd[2][k]);
      }
      cq[i] = ri - (vi[0] * yd[0] + vi[1] * yd[1] + vi[2] * yd[2]);
    }
   }
  }

  // ---- publish H_x and, for the information form, B^ = Q_f^T [H_x | r] right away: nothing below changes them, and the 42
  // registers of B^ (f64) would otherwise stay live through the whole gate (the kernel sits at its register budget)
  if (!(fdbg & 64)) {
    if (act) {
      if (d.h16) { __half* oH = d.trk_Hx16 + (tb * m_cap) * 12; for (int i = 0; i < 2; ++i) for (int k = 0; k < 6; ++k) oH[lane * 12 + i * 6 + k] = __float2half_rn((float)hx[i][k]); }
      else { S* oHx = d.trk_Hx + (tb * m_cap) * 12; for (int i = 0; i < 2; ++i) for (int k = 0; k < 6; ++k) oHx[lane * 12 + i * 6 + k] = hx[i][k]; }
    }
    if (d.compress) {
      // B scattered to state columns ([3][ldR] f64, zero where unobserved, column n = Q_f^T r), the whitened
      // residual and the slot -> observation map for the block-diagonal part of the Gram matrix
      // Only the columns of the track's slot range [first, last] are written (zeros where a slot inside the range is not
      // observed); k_gram masks everything outside the range (trk_first carries both ends).  Zero-filling all 3 x ldR doubles
      // of every track was 59 MB of the launch's 107 MB of writes.
      double* oB = d.trk_B + tb * 3 * (long)d.ldR;
      signed char* oI = d.trk_inv + tb * d.n_cap;
      // (a track that saw every camera of its range -- the usual case -- has no gap to zero: its 3 x 6 M values are the
      // only thing written; slots of a track are distinct, msckf_hip_set_tracks / scenario_set reject repeats)
      const int c_lo = 6 * slot_lo, c_n = 6 * (slot_hi - slot_lo + 1);
      if (slot_hi - slot_lo + 1 != M)
        for (int e = lane; e < 3 * c_n; e += 64) { const int q = e / c_n; oB[(long)q * d.ldR + c_lo + (e - q * c_n)] = 0.0; }
      for (int e = lane; e < d.n_cap; e += 64) oI[e] = -1;
      __syncthreads();
      if (act) {
        for (int q = 0; q < 3; ++q) for (int k = 0; k < 6; ++k) oB[(long)q * d.ldR + 6 * slot + k] = Bq[q][k];
        oI[slot] = (signed char)lane;
        d.trk_rw[tb * 2 * m_cap + 2 * lane] = r[0]; d.trk_rw[tb * 2 * m_cap + 2 * lane + 1] = r[1];
      }
      if (lane < 3) oB[(long)lane * d.ldR + 6 * (d.ncam_bias ? d.ncam_upd[b] : d.ncam[b])] = lane == 0 ? cq[0] : (lane == 1 ? cq[1] : cq[2]);
    }
  }

  // ---- TSQR route only: Householder QR of H_f_j (2M x 3) in working precision as compact WY, Z_c = T^T (V_rows^T Hx_c)
  // (3 x 6, local to the lane), Q^T r = r - V (T^T (V^T r)); published right away (nothing below changes them).  The
  // information-form route needs none of this: its reflectors were the f64 ones above, and the gate below works on H_f itself.
  const int row0 = 2 * lane;
  S rr_ro = 0;                                         // |r_o|^2, for the optional early accept of the gate only
  if (!d.compress) {
    S hfw[2][3], v[2][3], Tm[3][3];
    for (int i = 0; i < 2; ++i) for (int k = 0; k < 3; ++k) hfw[i][k] = hf[i][k];
    house3<S>(hfw, lane, v, Tm);
    S Zc[3][6];
    {
      S Wc[3][6];
      for (int q = 0; q < 3; ++q) for (int k = 0; k < 6; ++k) Wc[q][k] = v[0][q] * hx[0][k] + v[1][q] * hx[1][k];
      for (int q = 0; q < 3; ++q) for (int k = 0; k < 6; ++k) {
        S s = 0;
        for (int p = 0; p <= q; ++p) s += Tm[p][q] * Wc[p][k];
        Zc[q][k] = s;
      }
    }
    S qr[2];
    {
      S wr[3], y[3];
      for (int q = 0; q < 3; ++q) wr[q] = wave_sum(v[0][q] * r[0] + v[1][q] * r[1]);
      for (int q = 0; q < 3; ++q) { S s = 0; for (int p = 0; p <= q; ++p) s += Tm[p][q] * wr[p]; y[q] = s; }
      for (int s2 = 0; s2 < 2; ++s2) qr[s2] = r[s2] - (v[s2][0] * y[0] + v[s2][1] * y[1] + v[s2][2] * y[2]);
    }
    if (d.gate_early) {
      S rr = 0;
      for (int s2 = 0; s2 < 2; ++s2) { const int row = row0 + s2; if (row >= 3 && row < 2 * M) rr += qr[s2] * qr[s2]; }
      rr_ro = wave_sum(rr);
    }
    if (!(fdbg & 64)) {
      S* oV = d.trk_V + (tb * 2 * m_cap) * 4;
      S* oZ = d.trk_Zf + tb * 3 * (long)d.ldR;   // Z scattered to state columns: [3][ldR], zero where unobserved
      S* oR = d.trk_ro + tb * 2 * m_cap;
      if (act) {
        for (int s2 = 0; s2 < 2; ++s2) {
          const int row = row0 + s2;
          oV[row * 4 + 0] = v[s2][0]; oV[row * 4 + 1] = v[s2][1]; oV[row * 4 + 2] = v[s2][2]; oV[row * 4 + 3] = 0;
          oR[row] = qr[s2];   // (Q^T r)[row]; rows >= 3 are r_o
        }
      }
      for (int e = lane; e < 3 * d.ldR; e += 64) oZ[e] = 0;
      __syncthreads();
      if (act)
        for (int q = 0; q < 3; ++q) for (int k = 0; k < 6; ++k) oZ[(long)q * d.ldR + 6 * slot + k] = Zc[q][k];
    }
  } else if (d.gate_early) {
    // |r_o|^2 = |r|^2 - |first three entries of Q_f^T r|^2 (the f64 reflectors above)
    const S rn = wave_sum(r[0] * r[0] + r[1] * r[1]);
    rr_ro = rn - (S)(cq[0] * cq[0] + cq[1] * cq[1] + cq[2] * cq[2]);
  }

  // ---- optional exact early accept of the gate (msckf_hip_set_gate_early_accept, off by default): S >= sigma^2 I, so
  // gamma = r_o^T S^-1 r_o <= |r_o|^2 / sigma^2.  If that bound is already below the chi-square threshold (with a
  // factor 2 in hand for the rounding of P's smallest eigenvalues) the track passes whatever G is: no G, no Cholesky.
  // The decision is the reference's; trk_gamma then holds the bound and the status carries ST_GATE_BOUND.
  const S thresh = S(c_chi2[M < 98 ? M : 98]);   // table[dof+1], dof = M-1   (:433, :1117)
  bool spd = true;
  S gamma = 0;
  bool early = false;
  if (d.gate_early) {
    const S ub = rr_ro / prm[PRM_SIG2G];
    if (ub < S(0.5) * thresh) { early = true; gamma = ub; status |= ST_GATE_BOUND; }
  }
  if (!early) {
  // ---- stage H_x in LDS; G = H_x P_cc H_x^T from 6x6 blocks of P (upper block-triangle + mirror)
  if (lane < lm) {
    for (int i = 0; i < 2; ++i) for (int k = 0; k < 6; ++k) sHx[lane * 12 + i * 6 + k] = hx[i][k];
  }
  __syncthreads();
  const int R2 = 2 * M;
  const S* P = d.P + (long)b * ld * ld;
  bool done_staged = false;
  if constexpr (LONG && sizeof(S) == 4) {
    if (staged) {
      done_staged = true;
      S* sCol = sG;
      auto assemble = [&](int c0) {
        const int a0 = c0 >> 1, a1 = min(M, a0 + 8);       // observations whose two columns fall into [c0, c0 + 16)
        if (a0 < M) {
          const int na = a1 - a0, cnt = na * M - (a0 + a1 - 1) * na / 2;   // pairs (a, bq): a in [a0, a1), bq in [a, M)
          for (int p = lane; p < cnt; p += 64) {
            int a = a0, q = p;
#pragma unroll
            for (int s8 = 0; s8 < 7; ++s8) { const int len = M - a; if (q >= len) { q -= len; ++a; } }
            const int bq = a + q;
            const int sa2 = sSlot[a], sb2 = sSlot[bq];
            const S* Pt = P + (long)(15 + 6 * sa2) * ld + 15 + 6 * sb2;   // P through its mirror image (k_feature's packed form above)
            S pv[6][6];
            typedef S s2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int i = 0; i < 6; ++i) {
              const S* col = Pt + (long)i * ld;
              const s2 m12 = *reinterpret_cast<const s2*>(col + 1), m34 = *reinterpret_cast<const s2*>(col + 3);
              pv[0][i] = col[0]; pv[1][i] = m12.x; pv[2][i] = m12.y; pv[3][i] = m34.x; pv[4][i] = m34.y; pv[5][i] = col[5];
            }
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 ha[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) ha[i] = f2{(float)sHx[a * 12 + i], (float)sHx[a * 12 + 6 + i]};
            f2 Tt[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) {
              f2 t = f2{0.0f, 0.0f};
#pragma unroll
              for (int i = 0; i < 6; ++i) t += ha[i] * (float)pv[j][i];
              Tt[j] = t;
            }
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
              f2 g = f2{0.0f, 0.0f};
#pragma unroll
              for (int j = 0; j < 6; ++j) g += Tt[j] * (float)sHx[bq * 12 + cc * 6 + j];
              S* dst = sCol + (2 * bq + cc - c0) * 17 + (2 * a - c0);
              dst[0] = (S)g.x;
              if (a != bq || 1 <= cc) dst[1] = (S)g.y;
            }
          }
        }
        if (act) {                                       // the four riding rows of this lane's two columns
#pragma unroll
          for (int s2i = 0; s2i < 2; ++s2i) {
            const int row = row0 + s2i;
            if (row >= c0 && row < c0 + 16) {
              sCol[(R2 - c0) * 17 + row - c0] = r[s2i];
#pragma unroll
              for (int q = 0; q < 3; ++q) sCol[(R2 + 1 + q - c0) * 17 + row - c0] = hf[s2i][q];
            }
          }
        }
      };
      const S sig2s = prm[PRM_SIG2G];
      const int nbr = (R2 + 4 + 7) >> 3;
      if (nbr <= 10) spd = gate_chol_staged<10>(sCol, sC, lane, R2, sig2s, assemble);
      else if (nbr <= 12) spd = gate_chol_staged<12>(sCol, sC, lane, R2, sig2s, assemble);
      else if (nbr <= 14) spd = gate_chol_staged<14>(sCol, sC, lane, R2, sig2s, assemble);
      else spd = gate_chol_staged<16>(sCol, sC, lane, R2, sig2s, assemble);
      if (spd) gamma = gate_gamma_from_corner<S>(sC);
    }
  }
  if (!done_staged) {
  {
    // pairs (a, bq), a <= bq, enumerated as a rectangle of M/2 (rounded up) rows of width M | 1: row k of the rectangle
    // holds row k of the triangle (M - k pairs) followed by row M - 1 - k (M even) or M - k (M odd) -- two integer
    // operations per pair instead of a square root and a search
    const int npair = (fdbg & 32) ? 0 : M * (M + 1) / 2;
    const int Wd = M | 1;
    const float invW = 1.0f / (float)Wd;
    for (int p = lane; p < npair; p += 64) {
      const int k = (int)(((float)p + 0.5f) * invW), c = p - k * Wd;
      const bool first = c < M - k;
      const int a = first ? k : ((M & 1) ? M - k : M - 1 - k);
      const int bq = a + (first ? c : c - (M - k));
      const int sa2 = sSlot[a], sb2 = sSlot[bq];
      // The block P(6 s_a + i, 6 s_b + j) is read through its mirror image P(6 s_b + j, 6 s_a + i) (P is kept bit-symmetric):
      // consecutive lanes are consecutive bq for one a, so with the b block along the ROWS of column-major P a wavefront's
      // addresses are consecutive 24-byte runs of a few columns (~12 cache lines per load instead of 64).  Row 15 + 6 s is
      // odd, so rows j = 1..4 of a run are two aligned pairs: 4 loads per column instead of 6.
      const S* Pt = P + (long)(15 + 6 * sa2) * ld + 15 + 6 * sb2;   // element (i, j) at Pt[i * ld + j]
      S pv[6][6];
      typedef S s2 __attribute__((ext_vector_type(2)));
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        if (fdbg & 2) { for (int j = 0; j < 6; ++j) pv[j][i] = S(i == j ? 1e-4 : 0); continue; }
        const S* col = Pt + (long)i * ld;
        const s2 m12 = *reinterpret_cast<const s2*>(col + 1), m34 = *reinterpret_cast<const s2*>(col + 3);
        pv[0][i] = col[0]; pv[1][i] = m12.x; pv[2][i] = m12.y; pv[3][i] = m34.x; pv[4][i] = m34.y; pv[5][i] = col[5];
      }
      if constexpr (sizeof(S) == 4) {
        // both rows of the 2 x 6 block at once on the packed-f32 pipe: (T0j, T1j) += (h0i, h1i) * P(i, j)
        typedef float f2 __attribute__((ext_vector_type(2)));
        f2 ha[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) ha[i] = f2{(float)sHx[a * 12 + i], (float)sHx[a * 12 + 6 + i]};
        f2 Tt[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          f2 t = f2{0.0f, 0.0f};
#pragma unroll
          for (int i = 0; i < 6; ++i) t += ha[i] * (float)pv[j][i];
          Tt[j] = t;
        }
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
          f2 g = f2{0.0f, 0.0f};
#pragma unroll
          for (int j = 0; j < 6; ++j) g += Tt[j] * (float)sHx[bq * 12 + cc * 6 + j];
          if (a != bq || 0 <= cc) sG[TRI(2 * bq + cc, 2 * a)] = (S)g.x;          // a <= bq: row 2bq+cc >= column 2a+rr
          if (a != bq || 1 <= cc) sG[TRI(2 * bq + cc, 2 * a + 1)] = (S)g.y;
        }
      } else {
        S Tt[2][6];
        for (int j = 0; j < 6; ++j) {
          S s0 = 0, s1 = 0;
          for (int i = 0; i < 6; ++i) { s0 += sHx[a * 12 + i] * pv[j][i]; s1 += sHx[a * 12 + 6 + i] * pv[j][i]; }
          Tt[0][j] = s0; Tt[1][j] = s1;
        }
        for (int rr = 0; rr < 2; ++rr) for (int cc = 0; cc < 2; ++cc) {
          S s = 0;
          for (int j = 0; j < 6; ++j) s += Tt[rr][j] * sHx[bq * 12 + cc * 6 + j];
          if (a != bq || rr <= cc) sG[TRI(2 * bq + cc, 2 * a + rr)] = s;     // a <= bq: row 2bq+cc >= column 2a+rr
        }
      }
    }
  }
  // ---- the four rows that ride along: r^T and the three columns of H_f (rows 2M .. 2M+3 of the packed triangle)
  if (act) {
    for (int s2 = 0; s2 < 2; ++s2) {
      const int row = row0 + s2;
      sG[TRI(R2, row)] = r[s2];
      for (int q = 0; q < 3; ++q) sG[TRI(R2 + 1 + q, row)] = hf[s2][q];
    }
  }
  __syncthreads();
  // ---- gamma = r_o^T S^-1 r_o through N = G + sigma^2 I = L L^T (gate_chol): y = L^-1 r, Y = L^-1 H_f, gamma = y^T y - b^T C^-1 b
  const S sig2 = prm[PRM_SIG2G];
  const int nr = R2 + 4;
  if (fdbg & 4) { gamma = 0; }
  else if (LONG && nr > 64 && nr <= (sizeof(S) == 4 ? 128 : 96)) {   // double: 12 blocks fit the register file, longer tracks take the LDS path
    const int nbr = (nr + 7) >> 3;
    if (nbr <= 10) spd = gate_chol<S, LONG ? 10 : 1>(sG, sC, lane, R2, sig2);
    else if (nbr <= 12) spd = gate_chol<S, LONG ? 12 : 1>(sG, sC, lane, R2, sig2);
    else if (nbr <= 14) spd = gate_chol<S, (LONG && sizeof(S) == 4) ? 14 : 1>(sG, sC, lane, R2, sig2);
    else spd = gate_chol<S, (LONG && sizeof(S) == 4) ? 16 : 1>(sG, sC, lane, R2, sig2);
  }
  else if (nr <= 64) {
    const int nbr = (nr + 7) >> 3;   // 8 x 8 blocks in use (wave-uniform)
    switch (nbr) {
      case 1: spd = gate_chol<S, 1>(sG, sC, lane, R2, sig2); break;
      case 2: spd = gate_chol<S, 2>(sG, sC, lane, R2, sig2); break;
      case 3: spd = gate_chol<S, 3>(sG, sC, lane, R2, sig2); break;
      case 4: spd = gate_chol<S, 4>(sG, sC, lane, R2, sig2); break;
      case 5: spd = gate_chol<S, 5>(sG, sC, lane, R2, sig2); break;
      case 6: spd = gate_chol<S, 6>(sG, sC, lane, R2, sig2); break;
      case 7: spd = gate_chol<S, 7>(sG, sC, lane, R2, sig2); break;
      default: spd = gate_chol<S, 8>(sG, sC, lane, R2, sig2); break;
    }
  } else {
    // in place in the packed triangle (tracks too long for the register file): sigma^2 on the diagonal, zero corner, 2M pivots
    for (int i = lane; i < R2; i += 64) sG[TRI(i, i)] += sig2;
    if (lane < 16) { const int qi = lane >> 2, qj = lane & 3; if (qj <= qi) sG[TRI(R2 + qi, R2 + qj)] = 0; }
    __syncthreads();
    for (int k = 0; k < R2; ++k) {
      const S dkk = sG[TRI(k, k)];
      if (!(dkk > S(0))) { spd = false; break; }
      const S dinv = S(1) / dsqrt(dkk);
      for (int i = k + 1 + lane; i < nr; i += 64) sG[TRI(i, k)] *= dinv;
      __syncthreads();
      for (int i = k + 1 + lane; i < nr; i += 64) {
        const S lik = sG[TRI(i, k)];
        for (int j = k + 1; j <= i; ++j) sG[TRI(i, j)] -= lik * sG[TRI(j, k)];
      }
      __syncthreads();
    }
    if (lane < 16) { const int qi = lane >> 2, qj = lane & 3; if (qj <= qi) sC[qi * 4 + qj] = sG[TRI(R2 + qi, R2 + qj)]; }
    __syncthreads();
  }
  if (spd && !(fdbg & 4)) gamma = gate_gamma_from_corner<S>(sC);
  }   // !done_staged
  }   // !early
  if (spd && gamma < thresh) status |= ST_GATE_PASS;

  // ---- publish the compact representation of the projected block
  if (!(fdbg & 64)) {
    if (lane == 0) {
      d.trk_status[tb] = status;
      d.trk_gamma[tb] = gamma;
      d.trk_first[tb] = slot_lo | (slot_hi << 8);   // first and last camera slot of the track (TRK_FIRST / TRK_LAST)
      S* opf = d.trk_pf + tb * 4;
      opf[0] = pf.x; opf[1] = pf.y; opf[2] = pf.z; opf[3] = 0;
    }
  }
}

// ---------------------------------------------------------------- k_feature_pair: TWO tracks per wavefront
// Float filters on the information-form route (the headline configuration; cfg5's tracks of up to 30 observations).
// k_feature gives a track a whole wavefront with lane = observation: at the benchmark's track lengths (3 .. 29, 16 on
// average) three lanes in four idle through triangulation, Jacobian and B^ -- the kernel is bound by VALU issue, and that
// part is 58 % of its instructions.  Here a wavefront takes two tracks, one per half (lanes 0-31 / 32-63, lane = observation
// within the half), and every reduction / broadcast of that part runs inside a half (DPP inside rows of 16 + one
// v_permlane16_swap across the two rows; broadcasts through ds_bpermute), the Levenberg-Marquardt loops predicated per half.
// Which two: the trajectory's tracks are ranked by length (counting sort redone by every wavefront from the 4 lengths a
// lane loads: ~100 instructions) and wavefront w takes ranks w and n-1-w -- the shortest with the longest -- so that
//   * every wavefront of the launch has the same amount of work (M_a + M_b ~ const: no tail of long-track wavefronts),
//   * both gate matrices fit the LDS one long track needs (tri(2 M_a + 4) + tri(2 M_b + 4) <= S_cap; a pair that does not
//     fit -- a frame of maximum-length tracks -- assembles and factors its two matrices one after the other in the same space).
// The gate products G = H_x P_cc H_x^T of both tracks are enumerated together over the 64 lanes (pairs of observations),
// the two Cholesky factorizations run one after the other on the full 8 x 8 lane grid (gate_chol, as in k_feature).
// Same arithmetic per track as k_feature<float> (sums over a half instead of the wavefront: identical terms, the zeros
// of the idle lanes fall elsewhere in the tree); decisions and gamma agree to rounding (tests/test_gpu_parity.py A/B).
constexpr int GS = 32;   // lanes per track
// LDS hand-over inside ONE wavefront (its LDS operations complete in order; the fences stop the compiler from moving reads above writes)
__device__ __forceinline__ void wave_lds_sync() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
#ifdef MSCKF_ABLATE
// phase timers of the -DMSCKF_ABLATE build (scripts/feat_phases.py): shader-clock cycles per phase summed over ALL wavefronts
// [0 pairing, 1 loads + checkMotion + first cost, 2 Levenberg-Marquardt, 3 Jacobian + B^ + publish, 4 G assembly, 5 factorizations, 6 tail, 7 wavefronts]
__device__ unsigned long long g_featp_cycles[8];
// g_feat_dbg & 0x100000: timers on (they cost: eight device-scope atomics per wavefront); g_feat_dbg & (0x400 << slot): the
// wavefront returns after phase `slot` (timing of a truncated kernel: results are garbage)
#define FP_TICK(slot) do { if ((fdbg & 0x100000) && lane == 0) { const long long t_ = clock64(); atomicAdd(&g_featp_cycles[slot], (unsigned long long)(t_ - tlast)); tlast = clock64(); } \
                           if (fdbg & (0x400 << (slot))) return; } while (0)
#else
#define FP_TICK(slot) do { } while (0)
#endif

__device__ __forceinline__ float x16_sum(float v) {   // v[lane] + v[lane ^ 16]
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ double x16_sum(double v) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = (unsigned)(b & 0xffffffffull), hi = (unsigned)(b >> 32);
  const auto rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  const double a = __longlong_as_double((long long)(((unsigned long long)rh[0] << 32) | rl[0]));
  const double c = __longlong_as_double((long long)(((unsigned long long)rh[1] << 32) | rl[1]));
  return a + c;
}
// sum over the lane's half of the wavefront; every lane of the half gets the same bits
template <class T> __device__ __forceinline__ T half_sum(T v) {
  v += dpp_x<DPP_QUAD_X1>(v); v += dpp_x<DPP_QUAD_X2>(v); v += dpp_x<DPP_HALF_MIRROR>(v); v += dpp_x<DPP_ROW_MIRROR>(v);
  return x16_sum(v);
}
__device__ __forceinline__ float half_max(float v) {
  float t;
  t = dpp_x<DPP_QUAD_X1>(v); v = t > v ? t : v; t = dpp_x<DPP_QUAD_X2>(v); v = t > v ? t : v;
  t = dpp_x<DPP_HALF_MIRROR>(v); v = t > v ? t : v; t = dpp_x<DPP_ROW_MIRROR>(v); v = t > v ? t : v;
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  const float a = __uint_as_float(r[0]), c = __uint_as_float(r[1]);
  return a > c ? a : c;
}
__device__ __forceinline__ int half_min_i(int v) {
  int t;
  t = dpp_x<DPP_QUAD_X1>(v); v = t < v ? t : v; t = dpp_x<DPP_QUAD_X2>(v); v = t < v ? t : v;
  t = dpp_x<DPP_HALF_MIRROR>(v); v = t < v ? t : v; t = dpp_x<DPP_ROW_MIRROR>(v); v = t < v ? t : v;
  const auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
  return min((int)r[0], (int)r[1]);
}

// Register budget: three wavefronts per SIMD (168 registers, 5 values per lane in scratch).  At four (128 registers) 34 live values
// per lane sit in scratch: 56 MB of the launch's 96 MB of writes at 64 trajectories (166 MB of traffic per launch instead of 64),
// ~70 scratch instructions per wavefront.  Throughput: the three-wavefront build was ahead while the slices ran the one-track form
// (185-189 k -> 192-193 k updates/s), with pairs in the slices the four-wavefront build is (medians 203.8 / 204.6 k vs 207.2 /
// 208.6 k, alternating runs on one lease: 1.5 %, inside what leases differ by) -- the build that moves a third of the bytes ships.
// (Both side by side, picked by launch size: the 128-register one then reads 95 us for the large launches, warm or not.)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_feature_pair(Dev<float> d, int b0, int nb, int lm, int m_lo, int m_hi, int s_cap, int items, int single) {
  typedef float S;
  int bi, w;
  if (!xcd_item(nb, items, bi, w)) return;
  const int b = b0 + bi, lane = threadIdx.x;
#ifdef MSCKF_ABLATE
  const int fdbg = g_feat_dbg;
  long long tlast = clock64();
  if (fdbg & 0x200) return;                            // launch floor
#endif
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  S* sG = reinterpret_cast<S*>(smem_raw);              // [s_cap]: the packed gate matrices of the two tracks
  S* sC = sG + s_cap;                                  // [16]: the corner left by gate_chol
  int* sSlot = reinterpret_cast<int*>(sC + 16);        // [2][lm]
  int* sHist = reinterpret_cast<int*>(sG);             // [64], prologue only
  const int m_cap = d.m_cap, f_cap = d.f_cap;
  const int* Mlist = d.trk_M + (long)bi * d.wl_stride_f;

  // ---- which two tracks: ranks w and nv - 1 - w of the trajectory's tracks of this launch's bin, by length (stable).
  // Everything the ranking needs is requested at once -- the track count, the lengths and the list offsets of up to 512
  // tracks (clamped addresses: nothing waits for the count) -- one memory round trip instead of four dependent ones
  // (count -> lengths -> chosen track's length -> its offset).
  constexpr int NC = 8;                                // F <= 512 (launch_feature checks)
  int Mv[NC], Ov[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    Mv[c] = -1; Ov[c] = 0;
    if (64 * c < f_cap) {
      const int tt = min(64 * c + lane, f_cap - 1);
      Mv[c] = Mlist[tt];
      if (d.trk_off) Ov[c] = d.trk_off[(long)bi * d.wl_stride_f + tt];
    }
  }
  const int F = d.trk_n[(long)bi * d.wl_stride_n];
  if (w == 0 && lane == 0) d.nres_upd[b] = (int)(d.n_resid[b] > 1000 ? 1000 : d.n_resid[b]);
  sHist[lane] = 0;
  wave_lds_sync();
  if ((single ? w : 2 * w) >= F) return;
#ifdef MSCKF_ABLATE
  if (fdbg & 0x80000) { if (Mv[0] + Ov[0] == -12345) d.trk_gamma[0] = 0; return; }   // the first round trip only
#endif
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int t = 64 * c + lane;
    int M = -1;
    if (64 * c < F) {
      if (t < F) {
        M = Mv[c];
        if (M <= m_lo || M > m_hi) M = -1;             // another launch's bin
        else if (M < 2 || M > m_cap || M > GS) {       // cannot be residualized (checkMotion :982 returns false)
          if (w == 0) { const long tb = (long)b * f_cap + t; d.trk_status[tb] = 0; d.trk_gamma[tb] = 0; d.trk_first[tb] = 0; }
          M = -1;
        }
      }
      if (M >= 0) atomicAdd(&sHist[M], 1);
    }
    Mv[c] = M;
  }
  wave_lds_sync();
  int tA, tB, nv, MA, MB, woA, woB;
  {
    const int c0 = sHist[lane];
    // inclusive prefix over the 64 bins on the DPP network (four shifts inside the rows of 16, two row broadcasts)
    int scan = c0;
    scan += __builtin_amdgcn_update_dpp(0, scan, 0x111, 0xF, 0xF, true);
    scan += __builtin_amdgcn_update_dpp(0, scan, 0x112, 0xF, 0xF, true);
    scan += __builtin_amdgcn_update_dpp(0, scan, 0x114, 0xF, 0xF, true);
    scan += __builtin_amdgcn_update_dpp(0, scan, 0x118, 0xF, 0xF, true);
    scan += dppm_i<DPP_ROW_BCAST15, 0xA>(scan);
    scan += dppm_i<DPP_ROW_BCAST31, 0xC>(scan);
    nv = wave_bcast(scan, 63);
    if ((single ? w : 2 * w) >= nv) return;
    int tsel[2], msel[2], osel[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int p = single ? (nv - 1 - w) : (s ? nv - 1 - w : w);   // single: one track per wavefront, longest first
      const unsigned long long mb = __ballot(scan > p);
      const int Mb = __builtin_ctzll(mb);              // the length whose bin holds rank p
      int r = p - (wave_bcast(scan, Mb) - wave_bcast(c0, Mb));
      int found = -1, off = 0;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        if (64 * c < F && found < 0) {
          const unsigned long long m = __ballot(Mv[c] == Mb);
          const int n = __popcll(m);
          if (r < n) {
            const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
            const int hl = __builtin_ctzll(__ballot(Mv[c] == Mb && rank == r));
            found = 64 * c + hl; off = wave_bcast(Ov[c], hl);
          } else r -= n;
        }
      }
      tsel[s] = found; msel[s] = Mb; osel[s] = off;
    }
    tA = tsel[0]; tB = tsel[1]; MA = msel[0]; MB = msel[1]; woA = osel[0]; woB = osel[1];
  }
  wave_lds_sync();                                     // sHist is sG
  FP_TICK(0);
  const bool hasB = !single && (nv - 1 - w) != w;
  if (!hasB) MB = 0;
  const int g = lane >> 5, gl = lane & (GS - 1), gbase4 = (lane & GS) << 2;
  const bool has = g == 0 || hasB;
  const int t = g ? tB : tA;
  const long tb = (long)b * f_cap + t;                 // per-track output index
  const int M = g ? MB : MA;
  const long wo = d.trk_off ? (long)(g ? woB : woA) : (long)bi * d.wl_stride_o + (long)t * m_cap;
  const S* prm = d.prm + (long)b * PRM_STRIDE;
  const S* imu = d.imu + (long)b * IMU_STRIDE;
  const int ld = d.ld;
  const bool act = gl < M;
  int status = 0;
  // (requested here, long before their use: a load issued behind the output stores at the end would wait for all of them --
  // loads and stores share one in-order counter)
  const S thresh = S(c_chi2[M < 98 ? M : 98]);   // table[dof+1], dof = M-1   (:433, :1117)
  const int ncam_now = d.ncam_bias ? d.ncam_upd[b] : d.ncam[b];
  // value of the half's lane i (i may differ between the halves)
  auto hget = [&](S v, int i) -> S { return __int_as_float(__builtin_amdgcn_ds_bpermute(gbase4 + (i << 2), __float_as_int(v))); };

  // ---- load this lane's camera state and observation
  const int slot = act ? d.trk_slots[wo + gl] : 0;
  const S* cs = d.cam + ((long)b * d.n_cap + slot) * CAM_STRIDE;
  const Q4<S> qc = ldq(cs);
  const V3<S> pcg = ld3(cs + 4);
  const S zx = act ? d.trk_obs[2 * (wo + gl)] : S(0), zy = act ? d.trk_obs[2 * (wo + gl) + 1] : S(0);
  const M3<S> C = q2rot(qc);
  const V3<S> gv = ld3(imu + IG);
  if (gl < lm) sSlot[g * lm + gl] = act ? slot : -1;
  const int slot_lo = half_min_i(act ? slot : 0x7fffffff), slot_hi = -half_min_i(act ? -slot : 0x7fffffff);
  // first camera of the track (lane 0 of the half) broadcast
  M3<S> C0; V3<S> p0;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) C0.m[i][j] = hget(C.m[i][j], 0);
  p0 = mk3(hget(pcg.x, 0), hget(pcg.y, 0), hget(pcg.z, 0));
  const S z0x = hget(zx, 0), z0y = hget(zy, 0);

  // ---- checkMotion :980-1025
  {
    V3<S> dir = mk3(z0x, z0y, S(1));
    dir = (S(1) / dsqrt(dot3(dir, dir))) * dir;
    dir = multv(C0, dir);
    const V3<S> tr = pcg - p0;
    const S par = dot3(tr, dir);
    const V3<S> orth = tr - (par * dir);
    S nrm = (act && gl > 0) ? dsqrt(dot3(orth, orth)) : S(0);
    nrm = half_max(nrm);
    if (nrm > prm[PRM_TRANS]) status |= ST_MOTION_OK;
  }

  // ---- initializePosition :1147-1285 (Levenberg-Marquardt on inverse depth, sums over the half), predicated per half
  Pose<S> T;
  T.R = mulmt(C, C0);
  T.t = mulv(C, p0 - pcg);
  S sa, sb, srho;   // solution (alpha, beta, rho)
  {
    S depth;
    {
      const V3<S> m = mulv(T.R, mk3(z0x, z0y, S(1)));
      const S A0 = m.x - zx * m.z, A1 = m.y - zy * m.z;
      const S b0v = zx * T.t.z - T.t.x, b1v = zy * T.t.z - T.t.y;
      depth = (S(1) / (A0 * A0 + A1 * A1)) * (A0 * b0v + A1 * b1v);
      depth = hget(depth, M > 0 ? M - 1 : 0);
    }
    const S ix = z0x * depth, iy = z0y * depth, iz = depth;
    sa = ix / iz; sb = iy / iz; srho = S(1) / iz;
  }
  const bool given = d.mode == 1;   // stored feature position supplied by the host (pruneRedundantStates)
  S lambda = S(1e-3), delta_norm = 0;
  S total_cost = half_sum(act ? tri_cost(T, sa, sb, srho, zx, zy) : S(0));
  FP_TICK(1);
  bool reduced = false;
  int inner = 0, outer = 0;
  bool oact = has && !given;                           // this half's outer loop is still running
  while (__any(oact)) {
    S Ab[9];  // a00 a01 a02 a11 a12 a22 b0 b1 b2
    {
      const V3<S> h = mulv(T.R, mk3(sa, sb, S(1))) + (srho * T.t);
      S W[3][3];
      for (int i = 0; i < 3; ++i) { W[i][0] = T.R.m[i][0]; W[i][1] = T.R.m[i][1]; }
      W[0][2] = T.t.x; W[1][2] = T.t.y; W[2][2] = T.t.z;
      S J[2][3];
      const S iz = lm_rcp(h.z), xz = h.x * iz, yz = h.y * iz;
      for (int j = 0; j < 3; ++j) {
        J[0][j] = iz * (W[0][j] - xz * W[2][j]);
        J[1][j] = iz * (W[1][j] - yz * W[2][j]);
      }
      const S r0 = xz - zx, r1 = yz - zy;
      const S e = lm_sqrt(r0 * r0 + r1 * r1);
      const S wgt = (e <= S(0.01)) ? S(1) : S(0.005) * lm_rcp(e);
      const S w2 = (wgt == S(1)) ? S(1) : wgt * wgt;
      const S m = act ? w2 : S(0);
      Ab[0] = m * (J[0][0] * J[0][0] + J[1][0] * J[1][0]);
      Ab[1] = m * (J[0][0] * J[0][1] + J[1][0] * J[1][1]);
      Ab[2] = m * (J[0][0] * J[0][2] + J[1][0] * J[1][2]);
      Ab[3] = m * (J[0][1] * J[0][1] + J[1][1] * J[1][1]);
      Ab[4] = m * (J[0][1] * J[0][2] + J[1][1] * J[1][2]);
      Ab[5] = m * (J[0][2] * J[0][2] + J[1][2] * J[1][2]);
      Ab[6] = m * (J[0][0] * r0 + J[1][0] * r1);
      Ab[7] = m * (J[0][1] * r0 + J[1][1] * r1);
      Ab[8] = m * (J[0][2] * r0 + J[1][2] * r1);
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) Ab[i] = half_sum(Ab[i]);
    // the damping ladder lambda, 10 lambda, ... (clamped at 1e12): lane c of the half solves the c-th rung (k_feature)
    S lam_c = lambda;
#pragma unroll
    for (int t2 = 0; t2 < 11; ++t2) { const S up = lam_c * 10 < S(1e12) ? lam_c * 10 : S(1e12); if (t2 < gl) lam_c = up; }
    S dlc[3];
    ldlt3(Ab, lam_c, Ab + 6, dlc);
    int cand = 0;
    bool iact = oact;                                  // this half's inner loop is still running
    while (__any(iact)) {
      const S dl0 = hget(dlc[0], cand), dl1 = hget(dlc[1], cand), dl2 = hget(dlc[2], cand);
      const S lam_now = hget(lam_c, cand), lam_next = hget(lam_c, cand + 1);
      const S na = sa - dl0, nb2 = sb - dl1, nr = srho - dl2;
      const S dn = lm_sqrt(dl0 * dl0 + dl1 * dl1 + dl2 * dl2);
      // a step that changes none of the three parameters gives bit for bit the cost already held: not evaluated (k_feature)
      const bool changed = iact && !(na == sa && nb2 == sb && nr == srho);
      S new_cost = total_cost;
      if (__any(changed)) {
        const S sc = half_sum(act ? tri_cost(T, na, nb2, nr, zx, zy) : S(0));
        new_cost = changed ? sc : total_cost;
      }
      const bool better = iact && new_cost < total_cost;
      const S l10th = lam_now * S(0.1);
      const S lam_acc = l10th > S(1e-10) ? l10th : S(1e-10);
      sa = better ? na : sa; sb = better ? nb2 : sb; srho = better ? nr : srho; total_cost = better ? new_cost : total_cost;
      if (iact) {
        delta_norm = dn;
        reduced = better;
        lambda = better ? lam_acc : lam_next;          // lam_next = min(10 * rung, 1e12)
        ++cand;
        iact = inner < 10 && !reduced;
        ++inner;
      }
    }
    if (oact) {
      inner = 0;
      oact = outer < 10 && delta_norm > S(5e-7);
      ++outer;
    }
  }
  FP_TICK(2);
  const V3<S> fin = mk3(sa / srho, sb / srho, S(1) / srho);
  {
    const V3<S> pos = mulv(T.R, fin) + T.t;
    const unsigned long long badm = __ballot(act && pos.z <= S(0));
    const bool any_bad = ((g ? (unsigned)(badm >> 32) : (unsigned)badm)) != 0u;
    const S ncost = total_cost / (S(2) * S(M) * S(M));
    if (!any_bad && !(ncost > prm[PRM_GN])) status |= ST_TRI_VALID;
  }
  V3<S> pf = multv(C0, fin) + p0;   // :1282
  if (given) {
    if (has) pf = ld3(d.trk_pfin + tb * 4);
    status |= ST_MOTION_OK | ST_TRI_VALID;
  }

  // ---- calcResidual :960-978 and calcMeasJacobian :915-950 for this lane's observation
  S hx[2][6], r[2];   // H_f = -H_x[:, 3:6] (:949), not kept separately
  {
    const V3<S> pc = mulv(C, pf - pcg);
    const S X = pc.x, Y = pc.y, Z = pc.z;
    r[0] = zx - X / Z; r[1] = zy - Y / Z;
    S Ji[2][3];
    Ji[0][0] = S(1) * (S(1) / Z); Ji[0][1] = 0; Ji[0][2] = (-X / Z) * (S(1) / Z);
    Ji[1][0] = 0; Ji[1][1] = S(1) * (S(1) / Z); Ji[1][2] = (-Y / Z) * (S(1) / Z);
    const M3<S> sk = skew3(pc);
    S A[2][6];
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 3; ++j) {
        S a = 0, bb = 0;
        for (int k = 0; k < 3; ++k) { a += Ji[i][k] * sk.m[k][j]; bb += Ji[i][k] * C.m[k][j]; }
        A[i][j] = a; A[i][3 + j] = -bb;
      }
    const V3<S> uh = mulv(C, gv);
    const V3<S> ut = mulv(skew3(pf - pcg), gv);
    const S u[6] = {uh.x, uh.y, uh.z, ut.x, ut.y, ut.z};
    S uu = 0;
    for (int k = 0; k < 6; ++k) uu += u[k] * u[k];
    for (int i = 0; i < 2; ++i) {
      S Au = 0;
      for (int k = 0; k < 6; ++k) Au += A[i][k] * u[k];
      for (int k = 0; k < 6; ++k) {
        const S h = act ? A[i][k] - Au * (S(1) / uu) * u[k] : S(0);
        hx[i][k] = h;
      }
    }
    if (!act) { r[0] = 0; r[1] = 0; }
    const S wu = prm[PRM_WU], wv = prm[PRM_WV];
    r[0] *= wu; r[1] *= wv;
    for (int k = 0; k < 6; ++k) { hx[0][k] *= wu; hx[1][k] *= wv; }
    if (d.h16) {   // fp16 Jacobian (dtype MSCKF_HIP_F16H_F32P): rounded here, every consumer below sees the rounded blocks
      for (int i = 0; i < 2; ++i) for (int k = 0; k < 6; ++k) hx[i][k] = (S)__half2float(__float2half_rn((float)hx[i][k]));
    }
  }

  // H_f^T H_f = L L^T (3 x 3, f64) and c = L^-1 H_f^T r: nine f64 sums over the half (k_feature's float-filter form of B^)
  struct Lq { double i00, l10, l20, i11, l21, i22, cq[3]; };
  auto normal_eq = [&]() -> Lq {
    Lq q;
    const double f0[3] = {-(double)hx[0][3], -(double)hx[0][4], -(double)hx[0][5]}, f1[3] = {-(double)hx[1][3], -(double)hx[1][4], -(double)hx[1][5]};
    const double s00 = half_sum(f0[0] * f0[0] + f1[0] * f1[0]), s01 = half_sum(f0[0] * f0[1] + f1[0] * f1[1]), s02 = half_sum(f0[0] * f0[2] + f1[0] * f1[2]);
    const double s11 = half_sum(f0[1] * f0[1] + f1[1] * f1[1]), s12 = half_sum(f0[1] * f0[2] + f1[1] * f1[2]), s22 = half_sum(f0[2] * f0[2] + f1[2] * f1[2]);
    const double r0 = (double)r[0], r1 = (double)r[1];
    const double t0 = half_sum(f0[0] * r0 + f1[0] * r1), t1 = half_sum(f0[1] * r0 + f1[1] * r1), t2 = half_sum(f0[2] * r0 + f1[2] * r1);
    const double tiny = 1e-300;
    q.i00 = fast_rsqrt(s00 > tiny ? s00 : tiny); q.l10 = s01 * q.i00; q.l20 = s02 * q.i00;
    const double p11 = s11 - q.l10 * q.l10; q.i11 = fast_rsqrt(p11 > tiny ? p11 : tiny); q.l21 = (s12 - q.l20 * q.l10) * q.i11;
    const double p22 = s22 - q.l20 * q.l20 - q.l21 * q.l21; q.i22 = fast_rsqrt(p22 > tiny ? p22 : tiny);
    q.cq[0] = t0 * q.i00; q.cq[1] = (t1 - q.l10 * q.cq[0]) * q.i11; q.cq[2] = (t2 - q.l20 * q.cq[0] - q.l21 * q.cq[1]) * q.i22;
    return q;
  };
  FP_TICK(3);
  // ---- optional exact early accept of the gate (k_feature): gamma <= |r_o|^2 / sigma^2
  const int row0 = 2 * gl;
  S gamma = 0;
  bool early = false, spd = true;
  if (d.gate_early) {
    const Lq q = normal_eq();
    const S rn = half_sum(r[0] * r[0] + r[1] * r[1]);
    const S rr_ro = rn - (S)(q.cq[0] * q.cq[0] + q.cq[1] * q.cq[1] + q.cq[2] * q.cq[2]);
    const S ub = rr_ro / prm[PRM_SIG2G];
    if (ub < S(0.5) * thresh) { early = true; gamma = ub; status |= ST_GATE_BOUND; }
  }

  // ---- the gate: G = H_x P_cc H_x^T of both tracks from 6 x 6 blocks of P, the four riding rows, gate_chol per track
  // (the 2 x 6 Jacobian blocks stay in their lanes' registers: a pair's two blocks come through the LDS crossbar,
  // ds_bpermute -- 2.9 KB of LDS per wavefront less, which is what holds sixteen wavefronts on a compute unit)
  const bool needA = !wave_bcast((int)early, 0), needB = hasB && !wave_bcast((int)early, GS);
  const int szA = (2 * MA + 4) * (2 * MA + 5) / 2, szB = (2 * MB + 4) * (2 * MB + 5) / 2;
#ifdef MSCKF_ABLATE
  const int npass = (needA && needB && (szA + szB > s_cap || (fdbg & 0x40000))) ? 2 : 1;
#else
  const int npass = (needA && needB && szA + szB > s_cap) ? 2 : 1;
#endif
  const S* P = d.P + (long)b * ld * ld;
  const S sig2 = prm[PRM_SIG2G];
  for (int pass = 0; pass < npass; ++pass) {
    const bool selA = needA && (npass == 1 || pass == 0), selB = needB && (npass == 1 || pass == 1);
    const int offB = selA ? szA : 0;
    const int npA = selA ? MA * (MA + 1) / 2 : 0, npB = selB ? MB * (MB + 1) / 2 : 0;
    const float invWA = 1.0f / (float)(MA | 1), invWB = 1.0f / (float)(MB | 1);
    for (int p0 = 0; p0 < npA + npB; p0 += 64) {
      // pairs (a, bq), a <= bq, of the track `sel`, enumerated as a folded rectangle (k_feature).  Every lane runs every
      // round (ds_bpermute returns zero for a source lane that is switched off); a lane without a pair redoes the last one
      // and does not store
      const bool pvalid = p0 + lane < npA + npB;
      const int p = pvalid ? p0 + lane : npA + npB - 1;
      const bool sel = p >= npA;
      const int pq = sel ? p - npA : p, Mq = sel ? MB : MA, Wd = Mq | 1;
      const int k = (int)(((float)pq + 0.5f) * (sel ? invWB : invWA)), c = pq - k * Wd;
      const bool first = c < Mq - k;
      const int a = first ? k : ((Mq & 1) ? Mq - k : Mq - 1 - k);
      const int bq = a + (first ? c : c - (Mq - k));
      const int* sl = sSlot + (sel ? lm : 0);
      const int la4 = ((sel ? GS : 0) + a) << 2, lb4 = ((sel ? GS : 0) + bq) << 2;
      S* G = sG + (sel ? offB : 0);
      const int sa2 = sl[a], sb2 = sl[bq];
      // P(6 s_a + i, 6 s_b + j) through its mirror image (P is bit-symmetric): consecutive lanes read consecutive 24-byte runs
      const S* Pt = P + (long)(15 + 6 * sa2) * ld + 15 + 6 * sb2;   // element (i, j) at Pt[i * ld + j]
      S pv[6][6];
      typedef S s2 __attribute__((ext_vector_type(2)));
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const S* col = Pt + (long)i * ld;
        const s2 m12 = *reinterpret_cast<const s2*>(col + 1), m34 = *reinterpret_cast<const s2*>(col + 3);
        pv[0][i] = col[0]; pv[1][i] = m12.x; pv[2][i] = m12.y; pv[3][i] = m34.x; pv[4][i] = m34.y; pv[5][i] = col[5];
      }
      typedef float f2 __attribute__((ext_vector_type(2)));
      f2 ha[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) ha[i] = f2{lane_gather(hx[0][i], la4), lane_gather(hx[1][i], la4)};
      S hb[2][6];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) hb[i][j] = lane_gather(hx[i][j], lb4);
      f2 Tt[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        f2 tt = f2{0.0f, 0.0f};
#pragma unroll
        for (int i = 0; i < 6; ++i) tt += ha[i] * pv[j][i];
        Tt[j] = tt;
      }
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) {
        f2 gg = f2{0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < 6; ++j) gg += Tt[j] * hb[cc][j];
        if (pvalid) G[TRI(2 * bq + cc, 2 * a)] = gg.x;                  // a <= bq: row 2bq+cc >= column 2a+rr
        if (pvalid && (a != bq || 1 <= cc)) G[TRI(2 * bq + cc, 2 * a + 1)] = gg.y;
      }
    }
    // the four rows that ride along: r^T and the three columns of H_f (rows 2M .. 2M+3 of the packed triangle)
    if (act && (g ? selB : selA)) {
      S* G = sG + (g ? offB : 0);
      const int R2 = 2 * M;
      for (int s2i = 0; s2i < 2; ++s2i) {
        const int row = row0 + s2i;
        G[TRI(R2, row)] = r[s2i];
        for (int q = 0; q < 3; ++q) G[TRI(R2 + 1 + q, row)] = -hx[s2i][3 + q];
      }
    }
    __syncthreads();
    FP_TICK(4);
    for (int s = 0; s < 2; ++s) {
      if (!(s ? selB : selA)) continue;
      const S* G = sG + (s ? offB : 0);
      const int R2 = 2 * (s ? MB : MA), nbr = (R2 + 4 + 7) >> 3;   // 8 x 8 blocks in use (wave-uniform)
      bool ok;
      switch (nbr) {
        case 1: ok = gate_chol_dpp<1>(G, sC, lane, R2, sig2); break;
        case 2: ok = gate_chol_dpp<2>(G, sC, lane, R2, sig2); break;
        case 3: ok = gate_chol_dpp<3>(G, sC, lane, R2, sig2); break;
        case 4: ok = gate_chol_dpp<4>(G, sC, lane, R2, sig2); break;
        case 5: ok = gate_chol_dpp<5>(G, sC, lane, R2, sig2); break;
        case 6: ok = gate_chol_dpp<6>(G, sC, lane, R2, sig2); break;
        case 7: ok = gate_chol_dpp<7>(G, sC, lane, R2, sig2); break;
        default: ok = gate_chol_dpp<8>(G, sC, lane, R2, sig2); break;
      }
      const S gm = ok ? gate_gamma_from_corner<S>(sC) : S(0);
      if (g == s) { spd = ok; gamma = gm; }
      __syncthreads();                                 // sC is rewritten by the other track's factorization
    }
  }
  FP_TICK(5);
  if (spd && gamma < thresh) status |= ST_GATE_PASS;

  // ---- publish status, gamma, slot range and the triangulated point
  if (has && gl == 0) {
    d.trk_status[tb] = status;
    d.trk_gamma[tb] = gamma;
    d.trk_first[tb] = slot_lo | (slot_hi << 8);   // first and last camera slot of the track (TRK_FIRST / TRK_LAST)
    S* opf = d.trk_pf + tb * 4;
    opf[0] = pf.x; opf[1] = pf.y; opf[2] = pf.z; opf[3] = 0;
  }

  // ---- B = L^-1 H_f^T [H_x | r] (f64, k_feature's float-filter form) and every per-track output, at the very END: the output
  // stores (~35 per lane) share the loads' in-order counter, and anything loaded behind them -- the gate's blocks of P, a
  // spilled register -- waited for all of them to drain (the kernel published H_x and B^ before the gate to shorten B^'s 36
  // registers' lives: 14 us of the launch went into those waits)
  {
    const Lq q = normal_eq();
    double Bq[3][6];
    const double f0[3] = {-(double)hx[0][3], -(double)hx[0][4], -(double)hx[0][5]}, f1[3] = {-(double)hx[1][3], -(double)hx[1][4], -(double)hx[1][5]};
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const double h0 = (double)hx[0][k], h1 = (double)hx[1][k];
      const double g0 = f0[0] * h0 + f1[0] * h1, g1 = f0[1] * h0 + f1[1] * h1, g2 = f0[2] * h0 + f1[2] * h1;
      const double b0v = g0 * q.i00, b1v = (g1 - q.l10 * b0v) * q.i11;
      Bq[0][k] = b0v; Bq[1][k] = b1v; Bq[2][k] = (g2 - q.l20 * b0v - q.l21 * b1v) * q.i22;
    }
    // slot -> observation map of the half through LDS (the gate matrices are done with): one pass of stores, nothing is
    // written twice, no wait between a fill and a scatter
    int* sInv = reinterpret_cast<int*>(sG) + g * 64;
    sInv[gl] = -1; sInv[gl + GS] = -1;
    wave_lds_sync();
    if (act) sInv[slot] = gl;
    wave_lds_sync();
    double* oB = d.trk_B + tb * 3 * (long)d.ldR;
    signed char* oI = d.trk_inv + tb * d.n_cap;
    if (has) {
      for (int e = gl; e < d.n_cap; e += GS) oI[e] = (signed char)sInv[e];
      if (slot_hi - slot_lo + 1 != M) {                // zeros where a slot inside the track's range is unobserved
        const int c_n = 6 * (slot_hi - slot_lo + 1);
        for (int e = gl; e < 3 * c_n; e += GS) { const int qr = e / c_n, cc = e - qr * c_n; if (sInv[slot_lo + cc / 6] < 0) oB[(long)qr * d.ldR + 6 * slot_lo + cc] = 0.0; }
      }
    }
    if (act) {
      // 16-byte stores (a lane's 48 bytes of H_x / of a B^ row are contiguous and 16-byte aligned: ldR and m_cap * 12 are multiples
      // of 4 elements): 3 + 9 + 1 store instructions per lane instead of 12 + 18 + 2, whole 16-byte sectors reach the L2
      typedef float f4v __attribute__((ext_vector_type(4)));
      typedef double d2v __attribute__((ext_vector_type(2)));
      typedef float f2v __attribute__((ext_vector_type(2)));
      if (d.h16) { __half* oH = d.trk_Hx16 + (tb * m_cap) * 12; for (int i = 0; i < 2; ++i) for (int k = 0; k < 6; ++k) oH[gl * 12 + i * 6 + k] = __float2half_rn((float)hx[i][k]); }
      else {
        f4v* oHx = reinterpret_cast<f4v*>(d.trk_Hx + (tb * m_cap) * 12 + gl * 12);
        oHx[0] = f4v{hx[0][0], hx[0][1], hx[0][2], hx[0][3]}; oHx[1] = f4v{hx[0][4], hx[0][5], hx[1][0], hx[1][1]}; oHx[2] = f4v{hx[1][2], hx[1][3], hx[1][4], hx[1][5]};
      }
#pragma unroll
      for (int qr = 0; qr < 3; ++qr) {
        d2v* ob = reinterpret_cast<d2v*>(oB + (long)qr * d.ldR + 6 * slot);
        ob[0] = d2v{Bq[qr][0], Bq[qr][1]}; ob[1] = d2v{Bq[qr][2], Bq[qr][3]}; ob[2] = d2v{Bq[qr][4], Bq[qr][5]};
      }
      *reinterpret_cast<f2v*>(d.trk_rw + tb * 2 * m_cap + 2 * gl) = f2v{r[0], r[1]};
    }
    if (has && gl < 3) oB[(long)gl * d.ldR + 6 * ncam_now] = gl == 0 ? q.cq[0] : (gl == 1 ? q.cq[1] : q.cq[2]);
  }
  FP_TICK(6);
#ifdef MSCKF_ABLATE
  if ((fdbg & 0x100000) && lane == 0) atomicAdd(&g_featp_cycles[7], 1ull);
#endif
}

size_t feature_pair_lds_bytes(int lm, int& s_cap) {
  const int r2 = 2 * lm + 4;
  s_cap = r2 * (r2 + 1) / 2 + 12 * 13 / 2;            // the longest track + a 4-observation partner
  if (s_cap < 64) s_cap = 64;                          // the prologue's histogram lives there
  return ((size_t)s_cap + 16) * sizeof(float) + 2 * (size_t)lm * sizeof(int) + 16;
}

// One wavefront per trajectory: resolves the order-dependent part of marginalize (:352-399) -- checkMotion is
// skipped while fewer than 4 tracks have ever been residualized (Q4, msckf.h:354) -- and lays the gated-in
// tracks' rows out for the compression stage: tracks are counting-sorted by their first camera slot (stable,
// deterministic), so that consecutive row blocks of the TSQR share their leading zero columns and can start
// their elimination late.  order[p] = track id of sorted position p, row_start[p] = first stacked row.
// Once more than 3 tracks have been residualized the decisions are independent per track and run
// lane-parallel; the first few frames of a run take the serial path.
template <class S>
__device__ __forceinline__ void select_body(const Dev<S>& d, const int b0, const int nb, const int i, const int lane) {
  if (i >= nb) return;
  const int b = b0 + i;
  const int F = d.trk_n[(long)i * d.wl_stride_n];
  long long nres = d.n_resid[b];
  int* st = d.stats + (long)b * STAT_STRIDE;
  int mrej = 0, trej = 0, grej = 0, pass = 0;
  int* rs = d.row_start + (long)b * (d.f_cap + 1);
  int* order = d.trk_order + (long)b * d.f_cap;
  __shared__ int sCnt[64], sBase[64], sRows[64];
  sCnt[lane] = 0; sRows[lane] = 0;
  __syncthreads();
  // ---- fast path (steady state, F <= 256): every per-track word is loaded ONCE, four tracks per lane, all loads in
  // flight together; decisions, stable counting sort by first camera slot and the row prefix then run from registers
  // and LDS.  (The generic path below re-reads per pass: ~12 dependent global round trips for a 64-thread kernel.)
  if (!(nres <= 3 && d.mode == 0) && F <= 256) {
    __shared__ short sMrow[256];
    __shared__ short sOrd[256];
    int sv[4], Mv[4], fv[4];
    bool inc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int t = 64 * c + lane, tc = min(t, max(F - 1, 0));
      const long tb = (long)b * d.f_cap + tc;
      sv[c] = d.trk_status[tb]; Mv[c] = d.trk_M[(long)i * d.wl_stride_f + tc]; fv[c] = d.trk_first[tb] & 63;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int t = 64 * c + lane;
      const bool in = t < F;
      int sx = in ? sv[c] : 0;
      const int M = in ? Mv[c] : 0;
      bool m_rej = false, t_rej = false, g_rej = false, valid = false, incl = false;
      if (in) {
        if (M < 2 || !(sx & ST_MOTION_OK)) { m_rej = true; sx = (M < 2) ? 0 : (sx & ~(ST_TRI_VALID | ST_GATE_PASS)); }
        else if (sx & ST_TRI_VALID) valid = true;
        else { t_rej = true; sx &= ~ST_GATE_PASS; }
        if (valid) { if (sx & ST_GATE_PASS) { incl = true; sx |= ST_INCLUDED; } else g_rej = true; }
        d.trk_status[(long)b * d.f_cap + t] = sx;
        sMrow[t] = (short)(2 * M - 3);
      }
      inc[c] = incl;
      if (incl) { atomicAdd(&sCnt[fv[c]], 1); atomicAdd(&sRows[fv[c]], 2 * M - 3); }
      mrej += __popcll(__ballot(m_rej)); trej += __popcll(__ballot(t_rej)); grej += __popcll(__ballot(g_rej));
      pass += __popcll(__ballot(incl)); if (d.mode == 0) nres += __popcll(__ballot(valid));
    }
    __syncthreads();
    {
      const int c0 = sCnt[lane];
      int scan = c0;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const int up = __shfl_up(scan, o, 64); if (lane >= o) scan += up; }
      sBase[lane] = scan - c0;
    }
    __syncthreads();
    const int nbin = min(d.n_cap, 64);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int bin = inc[c] ? fv[c] : -1;
      int rank = 0, cnt = 0;
      for (int sbin = 0; sbin < nbin; ++sbin) {
        const unsigned long long m = __ballot(bin == sbin);
        if (bin == sbin) { rank = __popcll(m & ((1ull << lane) - 1ull)); cnt = __popcll(m); }
      }
      int pos = 0;
      if (inc[c]) { pos = sBase[bin] + rank; sOrd[pos] = (short)(64 * c + lane); order[pos] = 64 * c + lane; }
      __syncthreads();
      if (inc[c] && rank == 0) sBase[bin] += cnt;     // one writer per bin
      __syncthreads();
    }
    int rows = 0;
    for (int p0 = 0; p0 < pass; p0 += 64) {
      const int p = p0 + lane;
      const int r = p < pass ? (int)sMrow[sOrd[p]] : 0;
      int scan = r;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const int up = __shfl_up(scan, o, 64); if (lane >= o) scan += up; }
      if (p < pass) rs[p] = rows + scan - r;
      rows += __shfl(scan, 63, 64);
    }
    if (lane == 0) {
      rs[pass] = rows;
      d.n_resid[b] = nres;
      st[STAT_NTRACKS] = F; st[STAT_MOTION_REJ] = mrej; st[STAT_TRI_REJ] = trej; st[STAT_GATE_REJ] = grej;
      st[STAT_PASSED] = pass; st[STAT_MROWS] = rows; st[STAT_RROWS] = rows > 0 ? 6 * d.ncam[b] : 0;
    }
    return;
  }
  // ---- pass 1: decisions (status bits), per-bin counts
  if (nres <= 3 && d.mode == 0) {
    if (lane == 0) {
      for (int t = 0; t < F; ++t) {
        const long tb = (long)b * d.f_cap + t;
        int s = d.trk_status[tb];
        const int M = d.trk_M[(long)i * d.wl_stride_f + t];
        bool valid = false;
        if (M < 2) { if (nres > 3) mrej++; else trej++; s = 0; }
        else if (nres > 3 && !(s & ST_MOTION_OK)) { mrej++; s &= ~(ST_TRI_VALID | ST_GATE_PASS); }
        else {
          if (nres <= 3) s |= ST_MOTION_SKIPPED;
          if (s & ST_TRI_VALID) { valid = true; nres++; } else { trej++; s &= ~ST_GATE_PASS; }
        }
        if (valid) {
          if (s & ST_GATE_PASS) { s |= ST_INCLUDED; pass++; const int f0 = d.trk_first[tb] & 63; sCnt[f0]++; sRows[f0] += 2 * M - 3; }
          else grej++;
        }
        d.trk_status[tb] = s;
      }
    }
    mrej = __shfl(mrej, 0, 64); trej = __shfl(trej, 0, 64); grej = __shfl(grej, 0, 64); pass = __shfl(pass, 0, 64);
    nres = __shfl((int)nres, 0, 64);
  } else {
    for (int t0 = 0; t0 < F; t0 += 64) {
      const int t = t0 + lane;
      const bool in = t < F;
      const long tb = (long)b * d.f_cap + t;
      int s = in ? d.trk_status[tb] : 0;
      const int M = in ? d.trk_M[(long)i * d.wl_stride_f + t] : 0;
      bool m_rej = false, t_rej = false, g_rej = false, valid = false, incl = false;
      if (in) {
        if (M < 2 || !(s & ST_MOTION_OK)) { m_rej = true; s = (M < 2) ? 0 : (s & ~(ST_TRI_VALID | ST_GATE_PASS)); }
        else if (s & ST_TRI_VALID) valid = true;
        else { t_rej = true; s &= ~ST_GATE_PASS; }
        if (valid) { if (s & ST_GATE_PASS) { incl = true; s |= ST_INCLUDED; } else g_rej = true; }
        d.trk_status[tb] = s;
      }
      if (incl) { const int f0 = d.trk_first[tb] & 63; atomicAdd(&sCnt[f0], 1); atomicAdd(&sRows[f0], 2 * M - 3); }
      mrej += __popcll(__ballot(m_rej)); trej += __popcll(__ballot(t_rej)); grej += __popcll(__ballot(g_rej));
      pass += __popcll(__ballot(incl)); if (d.mode == 0) nres += __popcll(__ballot(valid));
    }
  }
  __syncthreads();
  // ---- exclusive prefix of the per-bin track counts (bins = first camera slot, ascending)
  {
    const int c = sCnt[lane];
    int scan = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int up = __shfl_up(scan, o, 64); if (lane >= o) scan += up; }
    sBase[lane] = scan - c;
  }
  __syncthreads();
  // ---- pass 2: stable placement (track index order inside a bin), then row prefix over sorted positions
  for (int t0 = 0; t0 < F; t0 += 64) {
    const int t = t0 + lane;
    const long tb = (long)b * d.f_cap + t;
    const bool incl = (t < F) && (d.trk_status[tb] & ST_INCLUDED);
    const int bin = incl ? (d.trk_first[tb] & 63) : -1;
    int pos = -1;
    for (int sbin = 0; sbin < d.n_cap && sbin < 64; ++sbin) {
      const unsigned long long m = __ballot(bin == sbin);
      if (bin == sbin) pos = sBase[sbin] + __popcll(m & ((1ull << lane) - 1ull));
      __syncthreads();
      if (lane == 0) sBase[sbin] += __popcll(m);
      __syncthreads();
    }
    if (incl) order[pos] = t;
  }
  __syncthreads();
  int rows = 0;
  for (int p0 = 0; p0 < pass; p0 += 64) {
    const int p = p0 + lane;
    int r = 0;
    if (p < pass) { const int t = order[p]; r = 2 * d.trk_M[(long)i * d.wl_stride_f + t] - 3; }
    int scan = r;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int up = __shfl_up(scan, o, 64); if (lane >= o) scan += up; }
    if (p < pass) rs[p] = rows + scan - r;
    rows += __shfl(scan, 63, 64);
  }
  if (lane == 0) {
    rs[pass] = rows;
    d.n_resid[b] = nres;
    st[STAT_NTRACKS] = F; st[STAT_MOTION_REJ] = mrej; st[STAT_TRI_REJ] = trej; st[STAT_GATE_REJ] = grej;
    st[STAT_PASSED] = pass; st[STAT_MROWS] = rows; st[STAT_RROWS] = rows > 0 ? 6 * d.ncam[b] : 0;
  }
}

template <class S>
__global__ __launch_bounds__(64) void k_select(Dev<S> d, int b0, int nb) { select_body<S>(d, b0, nb, (int)blockIdx.x, (int)threadIdx.x); }

// k_select and the block-diagonal part of Lam^ (sum h^T h per camera slot, sum h^T r; information-form route) in ONE launch:
// both only read what k_feature left behind, both are latency-bound (one wavefront per trajectory: 11 us; one per camera
// slot: 16 us), and as two launches they ran one after the other.  Workgroup ndiag of a trajectory is k_select (its first
// wavefront); workgroups 0 .. ndiag - 1 reduce four camera slots each, one wavefront per slot, lanes over ALL tracks of the
// update with the inclusion decision re-derived from the raw status bits (k_select's rule, msckf.h:352-399):
//   steady state: M >= 2, motion ok, triangulation valid, gate passed;
//   while fewer than 4 tracks have ever been residualized (Q4, msckf.h:354) the motion check is skipped up to and including
//   the track that brings the count to 4 -- found by scanning the tracks in order from the count k_feature recorded at the
//   start of the update (nres_upd: n_resid itself is being updated by the k_select workgroup of this very launch).
// Two dependent load levels (status | slot map -> Jacobian block) instead of three (sorted order -> slot map -> block).
// Measured and rejected (round 3): the two load levels batched over four rounds of 64 tracks (clamped addresses, masked
// afterwards, no `continue`): 16.5 -> 22.0 us -- 190 registers, and the Jacobian blocks of the ~45 % of (track, slot) pairs
// that do not exist are fetched too.
// Also rejected: one workgroup per slot with its four wavefronts splitting the rounds (16.5 -> 23.0 us: the 27 f64 butterfly
// reductions, ~1 600 instructions per wavefront, are then done four times over), and the 27 sums through an LDS transpose
// instead of butterflies (~160 instructions, but 58 KB of LDS per workgroup: 18.6 us, and the changed summation order moves
// the float TSQR-vs-information-form comparison at the 60-camera geometry past its tolerance).
template <class S>
__global__ __launch_bounds__(256) void k_select_diag(Dev<S> d, int b0, int nb, int ndiag) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, i = blockIdx.y;
  if ((int)blockIdx.x == ndiag) { if (tid < 64) select_body<S>(d, b0, nb, i, tid); return; }
  const int b = b0 + i;
  const int N = d.ncam[b], F = d.trk_n[(long)i * d.wl_stride_n];
  const int s = 4 * (int)blockIdx.x + w;
  if (s >= N) return;
  const int f_cap = d.f_cap, m_cap = d.m_cap;
  const int* stt = d.trk_status + (long)b * f_cap;
  const int* Mv = d.trk_M + (long)i * d.wl_stride_f;
  constexpr int ALL = ST_MOTION_OK | ST_TRI_VALID | ST_GATE_PASS;
  int tcut = 0;                                      // tracks below tcut skip the motion check (Q4)
  if (d.mode == 0 && F > 0) {
    int need = 4 - d.nres_upd[b];                    // valid tracks still to come before the motion check applies
    if (need > 0) {
      tcut = F;
      for (int t0 = 0; t0 < F && need > 0; t0 += 64) {
        const int t = t0 + lane;
        const bool v = t < F && Mv[t] >= 2 && (stt[t] & ST_TRI_VALID);
        unsigned long long m = __ballot(v);
        const int c = __popcll(m);
        if (c < need) { need -= c; continue; }
        int pos = 0;
        for (int k = 0; k < need; ++k) { pos = __builtin_ctzll(m); m &= m - 1; }
        tcut = t0 + pos + 1; need = 0;
      }
    }
  }
  double acc[27];
#pragma unroll
  for (int e = 0; e < 27; ++e) acc[e] = 0.0;
  for (int t = lane; t < F; t += 64) {
    const long tb = (long)b * f_cap + t;
    const int sx = stt[t], M = Mv[t];
    const int oi = d.trk_inv[tb * d.n_cap + s];      // issued with the status loads; meaningful only for residualized tracks
    const int need_bits = t < tcut ? (ST_TRI_VALID | ST_GATE_PASS) : ALL;
    if (M < 2 || (sx & need_bits) != need_bits || oi < 0) continue;
    const long h0i = (tb * m_cap + oi) * 12;
    const S* rw = d.trk_rw + tb * 2 * m_cap + 2 * oi;
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

size_t feature_lds_bytes(int m_cap, size_t scalar, bool staged) {
  const size_t r2 = 2 * (size_t)m_cap + 4;
  const size_t x = std::max<size_t>((size_t)m_cap * 12, 16);
  return ((staged ? r2 * 17 : r2 * (r2 + 1) / 2) + x) * scalar + (size_t)m_cap * sizeof(int) + 16;
}

// one-time, per-device setup (called from msckf_hip_create after hipSetDevice): chi-square table, LDS limits
void feature_device_setup() {
  (void)hipMemcpyToSymbol(HIP_SYMBOL(c_chi2), kChi2Q05, sizeof(double) * 99);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_feature<float, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_feature<double, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_feature<float, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_feature<double, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_feature_pair), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

template <class S>
void launch_feature(const Dev<S>& d, int b0, int nb, hipStream_t st) {
  if (nb <= 0) return;
  const dim3 grid(xcd_grid(nb, d.f_cap));
  constexpr int ALL_LO = -0x7fffffff, ALL_HI = 0x7fffffff, M_REG = 30;   // 2 * 30 + 4 = 64: the register-resident gate factorization
  // float filters on the information-form route: tracks of up to 30 observations two per wavefront (k_feature_pair)
  bool pair = false;
  if constexpr (sizeof(S) == 4) {
    if (d.feat_pair && d.compress && d.f_cap <= 512) {
      pair = true;
      // a small launch (a filter or a few: at most 1 024 tracks) takes ONE track per wavefront -- the kernel's latency is then its
      // slowest wavefront's, and a wavefront with a single track has half the gate work.  A slice of run_frames (16 trajectories, 3 200
      // tracks) keeps the pairs: alone the single form is 1.5 us faster (40 -> 38.5 us), but with the other slices' kernels beside it
      // twice the wavefronts cost more than that -- 191-194 k -> 201-202 k updates/s (medians 193-195 k -> 203-205 k), three
      // alternating runs on one lease.  feat_pair 2 forces pairs, 3 forces the single form.
      const int single = (d.feat_pair == 3 || (d.feat_pair == 1 && (long)nb * d.f_cap <= 1024)) ? 1 : 0;
      const int lm = d.m_cap < M_REG ? d.m_cap : M_REG, items = single ? d.f_cap : (d.f_cap + 1) / 2;
      int s_cap = 0;
      const size_t lds = feature_pair_lds_bytes(lm, s_cap);
      hipLaunchKernelGGL(k_feature_pair, dim3(xcd_grid(nb, items)), dim3(64), lds, st, d, b0, nb, lm, ALL_LO, d.m_cap <= M_REG ? ALL_HI : M_REG, s_cap, items, single);
      if (d.m_cap <= M_REG) return;
    }
  }
  if (d.m_cap <= M_REG) {
    // (measured and rejected, round 4: the tracks of at most 12 / 16 / 20 observations in a launch of their own with the LDS sized
    // for them -- five wavefronts per SIMD instead of four for those: 183 k -> 172 k updates/s on one stream, 201 k -> 199 k in four
    // slices; the second launch's tail costs more than the occupancy gives.  cfg5's windows are another matter, below)
    hipLaunchKernelGGL((k_feature<S, false>), grid, dim3(64), feature_lds_bytes(d.m_cap, sizeof(S)), st, d, b0, nb, d.m_cap, ALL_LO, ALL_HI, 0);
    return;
  }
  // long windows: tracks binned by length, one launch per bin over the same grid (a workgroup whose track belongs to another
  // bin returns at once).  Occupancy is set by the LDS of the bin's longest track, not of the window's
  if (!pair) hipLaunchKernelGGL((k_feature<S, false>), grid, dim3(64), feature_lds_bytes(M_REG, sizeof(S)), st, d, b0, nb, M_REG, ALL_LO, M_REG, 0);
  // float filters, tracks of 31 .. 62 observations: the gate matrix goes through LDS sixteen columns at a time (gate_chol_staged) -- the
  // launch's LDS no longer depends on the square of the longest track, and ONE launch takes every long track (the two bins by length
  // existed for the LDS of the packed triangle: 31 KB at 60 observations)
  const bool staged = sizeof(S) == 4 && d.m_cap <= 62 && d.feat_pair;
  // (measured and rejected: the tracks of 31 .. 46 observations in a launch of their own whose instance holds 12 x 12 blocks at most --
  // 168 registers, three wavefronts per SIMD instead of two: 4.70 -> 4.67 ms at cfg5, the second launch's tail takes what the occupancy gives)
  if (staged) { hipLaunchKernelGGL((k_feature<S, true>), grid, dim3(64), feature_lds_bytes(d.m_cap, sizeof(S), true), st, d, b0, nb, d.m_cap, M_REG, ALL_HI, 1); return; }
  const int mid = (M_REG + d.m_cap + 1) / 2;
  if (d.m_cap - M_REG >= 16) {
    hipLaunchKernelGGL((k_feature<S, true>), grid, dim3(64), feature_lds_bytes(mid, sizeof(S)), st, d, b0, nb, mid, M_REG, mid, 0);
    hipLaunchKernelGGL((k_feature<S, true>), grid, dim3(64), feature_lds_bytes(d.m_cap, sizeof(S)), st, d, b0, nb, d.m_cap, mid, ALL_HI, 0);
  } else hipLaunchKernelGGL((k_feature<S, true>), grid, dim3(64), feature_lds_bytes(d.m_cap, sizeof(S)), st, d, b0, nb, d.m_cap, M_REG, ALL_HI, 0);
}
#ifdef MSCKF_ABLATE
void feat_debug_set(int val) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_feat_dbg), &val, sizeof(int)); }
void featp_cycles_read(unsigned long long* out8, int reset) {
  (void)hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_featp_cycles), sizeof(unsigned long long) * 8);
  if (reset) { unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_featp_cycles), z, sizeof(z)); }
}
#endif

template <class S>
void launch_select(const Dev<S>& d, int b0, int nb, hipStream_t st) {
  if (nb <= 0) return;
  hipLaunchKernelGGL(k_select<S>, dim3(nb), dim3(64), 0, st, d, b0, nb);
}
template <class S>
void launch_select_diag(const Dev<S>& d, int b0, int nb, hipStream_t st) {
  if (nb <= 0) return;
  const int ndiag = (d.n_cap + 3) / 4;
  hipLaunchKernelGGL(k_select_diag<S>, dim3(ndiag + 1, nb), dim3(256), 0, st, d, b0, nb, ndiag);
}

template void launch_feature<float>(const Dev<float>&, int, int, hipStream_t);
template void launch_feature<double>(const Dev<double>&, int, int, hipStream_t);
template void launch_select<float>(const Dev<float>&, int, int, hipStream_t);
template void launch_select<double>(const Dev<double>&, int, int, hipStream_t);
template void launch_select_diag<float>(const Dev<float>&, int, int, hipStream_t);
template void launch_select_diag<double>(const Dev<double>&, int, int, hipStream_t);

}  // namespace msckf
