// dev_common.h -- shared device-side layout and math helpers for the gfx950 MSCKF filter core.
//
// Data layout in HBM (one `Dev<S>` per batch of B independent trajectories, S = float | double):
//   imu  [B][IMU_STRIDE]           q_IG(w,x,y,z) b_g v_I_G b_a p_I_G g q_IG_null v_I_G_null p_I_G_null
//                                  (reference: types.h:69-76 imuState)
//   cam  [B][n_cap][CAM_STRIDE]    q_CG(w,x,y,z) p_C_G                       (types.h:57-67 camState)
//   prm  [B][PRM_STRIDE]           Camera / noiseParams / MSCKFParams scalars (types.h:48-99)
//   P    [B][ld*ld]                full symmetric covariance, column-major, IMU block first, camera
//                                  slot s at rows/cols 15+6s..  (reference keeps three blocks,
//                                  msckf.h:52-54, and re-assembles them on every use :166-174)
// Error-state order (msckf.h:1376-1391): [dtheta(0:3) db_g(3:6) dv(6:9) db_a(9:12) dp(12:15) | per cam dtheta_C dp_C].
#ifndef MSCKF_DEV_COMMON_H
#define MSCKF_DEV_COMMON_H

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

namespace msckf {

constexpr int IMU_STRIDE = 32;
constexpr int CAM_STRIDE = 8;
constexpr int PRM_STRIDE = 48;
constexpr int DG_STRIDE = 28; // per-slot Gram block: 21 (upper 6x6) + 6 (h^T r) + pad
constexpr int RD_STRIDE = 7;  // imuReading: omega(3) a(3) dT   (types.h:78-84)

enum {  // offsets into imu[]
  IQ = 0, IBG = 4, IV = 7, IBA = 10, IP = 13, IG = 16, IQN = 19, IVN = 23, IPN = 26
};
enum {  // offsets into prm[]
  PRM_CU = 0, PRM_CV = 1, PRM_FU = 2, PRM_FV = 3, PRM_B = 4, PRM_QCI = 5, PRM_PCI = 9,
  PRM_UVAR = 12, PRM_VVAR = 13, PRM_Q = 14,  // 12 diagonal entries of Q_imu
  PRM_GN = 26, PRM_RCOND = 27, PRM_TRANS = 28, PRM_RANG = 29, PRM_RDIST = 30,
  PRM_MINTL = 31, PRM_MAXTL = 32, PRM_MAXCS = 33,
  // derived at initialize: row weights and the noise variance the kernels run with.  Isotropic noise: (1, 1, u_var');
  // anisotropic noise: rows are pre-whitened by (1/sigma_u, 1/sigma_v) and the filter runs with unit variance.
  // PRM_SIG2: noise variance of the UPDATE (S = T_H P T_H^T + sigma^2 I); PRM_SIG2G: of the chi-square gate (msckf.h:1117 uses
  // u_var' whatever v_var' is).  They differ only on the literal anisotropic route (PRM_LIT = 1: rows unweighted, the gate
  // with u_var' as in the reference, the update with unit noise on the information matrix of kernels_literal.hip).
  PRM_WU = 34, PRM_WV = 35, PRM_SIG2 = 36, PRM_SIG2G = 37, PRM_LIT = 38
};
enum {  // per-track status bits written by k_feature / k_select
  ST_MOTION_OK = 1, ST_TRI_VALID = 2, ST_GATE_PASS = 4, ST_INCLUDED = 8, ST_MOTION_SKIPPED = 16, ST_GATE_BOUND = 32
};
enum { STAT_NTRACKS = 0, STAT_MOTION_REJ, STAT_TRI_REJ, STAT_GATE_REJ, STAT_PASSED, STAT_MROWS, STAT_RROWS, STAT_ERR, STAT_STRIDE = 8 };
// STAT_ERR is sticky (never cleared by an update) and a bit field: capacity overflow in augmentState; a non-positive pivot
// in the factorization of S = T_H P T_H^T + R_n (the covariance lost positive definiteness: the square-root gain form
// P <- P - W W^T has no PSD guarantee under rounding; the pivot is clamped so that the run continues, but it is reported)
enum { STAT_ERR_NCAP = 1, STAT_ERR_PIVOT = 2 };

// work space of the literal anisotropic compression (kernels_literal.hip / literal_core.h), per trajectory; null when no
// trajectory of the batch uses it
constexpr int LIT_TIM_SLOTS = 24;   // 0..11 phase stamps (7: basis staged), 12..15 the sweep's panels (sums), 16..22 the handed-through rows' steps, 23: Z's main pass done
struct LitBufs {
  double* X = nullptr; double* tau = nullptr; double* Vf = nullptr; double* Tf = nullptr; double* TH = nullptr; double* G = nullptr; double* Z = nullptr; double* W2 = nullptr;
  int* row0 = nullptr; int* obs0 = nullptr; int* otrk = nullptr; int* kept = nullptr; int* info = nullptr;
  double* BD = nullptr;       // [B][f_cap][6][ldR] per track: Q_f^T H_x (3 rows) and D (3 rows), inside the track's slot range (k_lit_pre -> k_lit_gamma)
  double* Gam = nullptr;      // [B][ldR][ldR] lower triangle, (hi, lo) at hi * ldR + lo: sum over the stacked tracks of (u-rows of the projected Jacobian)^T (the same)
  double* Du = nullptr;       // [B][n_cap][24] per camera slot: upper triangle of sum h_u^T h_u (21)
  int serial = 0;             // MSCKF_HIP_LITERAL_SERIAL=1: k_lit_pre's per-track part on one lane with literal_core.h's serial reference (A/B runs)
  long long* tim = nullptr;   // [B][LIT_TIM_SLOTS] phase stamps of k_literal (100 MHz wall clock), only with MSCKF_HIP_LITERAL_TIMERS=1
  int ldx = 0, r_cap = 0, ldg = 0, ldz = 0, kept_stride = 0;
  long w2_stride = 0;
  int route = 0;     // 0: the compact route; 1: the sweep over the dense stack (X, G allocated only then)
  double tol = 0;
};

template <class S>
struct Dev {
  int B, n_cap, f_cap, m_cap;
  int ld;      // leading dimension of P and of every D x * work matrix (multiple of 16)
  int n6cap;   // 6*n_cap
  int ldR;     // row stride of the QR work matrices: 64*NC >= 6*n_cap + 1
  int nchunk;  // TSQR chunks per trajectory
  // filter state
  S* imu; S* cam; S* prm; S* P; int* ncam; long long* n_resid;
  // run_frames, square-root gain form: the frame's prune ("drop the fuse_drop[i] oldest camera states") rides on the covariance
  // downdate -- P - W W^T is written straight to its pruned position in the OTHER covariance buffer, Pout (no in-place hazard,
  // no prune launch); the host swaps P and the spare buffer after such a frame.  Pout == null: plain downdate in place.
  // ncam cannot change while the downdate's other workgroups may still read it: it stays at the old window size, the new
  // one waits in ncam_upd (as "size after the next augment") for the next frame's k_propagate (ncam_defer = 1) to commit.
  S* Pout; const int* fuse_drop; int ncam_defer;
  // current work-list (may point into a resident scenario)
  const int* trk_n; const int* trk_M; const int* trk_slots; const S* trk_obs;
  // trk_off == null: padded single-call lists, track t of the launch's i-th trajectory starts at i * wl_stride_o + t * m_cap;
  // else the COMPACT per-frame lists of a scenario: sum M_j entries, track t starts at trk_off[i * wl_stride_f + t] (wl_first)
  const int* trk_off;
  long wl_stride_n;   // stride between trajectories in trk_n   (ints)
  long wl_stride_f;   // stride between trajectories in trk_M / trk_off (ints)
  long wl_stride_o;   // stride between trajectories in trk_slots (ints) / trk_obs (2 scalars each), padded lists only
  // mode 1 (second update of pruneRedundantStates, msckf.h:545-614): every track comes with its stored feature
  // position trk_pfin[b*f_cap+t][4] -- no checkMotion / triangulation, no Q4 bookkeeping
  int mode; const S* trk_pfin;
  // k_feature launched BEFORE this frame's augmentState on a side stream (run_frames overlaps it with propagate + augment
  // when no track of the frame observes the newest camera): it then takes the window size from ncam_upd (left by the previous
  // frame's prune: ncam after prune + 1) instead of ncam, which augmentState is incrementing meanwhile
  int ncam_bias; int* ncam_upd;
  int* nres_upd;   // [B] n_resid at the start of the update in flight (k_feature -> k_select_diag)
  int gate_early;   // exact early accept of the chi-square gate by the bound |r_o|^2 / sigma^2 (k_feature), off by default
  int feat_pair;    // float filters, information form: k_feature_pair (two tracks per wavefront) instead of k_feature; MSCKF_HIP_FEATURE_PAIR=0 for A/B runs
  // per-track products of k_feature
  int* trk_status; S* trk_pf; S* trk_gamma; S* trk_Hx; S* trk_V; S* trk_Zf; S* trk_ro; int* trk_first;
  // dtype MSCKF_HIP_F16H_F32P (BASELINE.json configs[4]: fp16 Jacobian / fp32 covariance): the 2 x 6 measurement Jacobian
  // blocks are rounded to fp16 where they are formed and stored as fp16 (trk_Hx16 replaces trk_Hx); every consumer sees
  // the rounded values, all accumulation stays in f32 / f64
  int h16; __half* trk_Hx16;
  // k_select
  int* row_start; int* trk_order; int* stats;
  // TSQR
  S* Rbuf;
  // information-form compression (compress == 1, kernels_gram.hip): [T | r_n] = chol(H_o^T H_o) accumulated in f64
  //   trk_B  [B*f_cap][3][ldR]  f64   Q_f^T [H_x | r] of the track scattered to state columns (column n = Q_f^T r)
  //   trk_rw [B*f_cap][2 m_cap]       whitened residual r of the track (before the null-space projection)
  //   trk_inv[B*f_cap][n_cap]   i8    observation index of camera slot s in the track, -1 = not observed
  //   Dg     [B][n_cap][28]     f64   per camera slot: upper triangle of sum h^T h (21) and sum h^T r (6)
  //   Lam    [B][ldR][ldR]      f64   sum B^T B, upper 64x64 tiles
  int compress;   // 0 Householder TSQR; != 0 (3) information form: k_gram + blocked matrix-core Cholesky (kernels_chol.hip)
  double* trk_B; S* trk_rw; signed char* trk_inv; double* Dg; double* Lam;
  // split-K SYRK (kernels_gram.hip): gram_parts == 3: tiles of block row ti are summed by min(ti + 1, 3) workgroups into copies
  // of Lam^ that lie lam_part doubles apart; the blocked Cholesky adds the copies while loading.  lam_part == 0: one copy only
  int gram_parts; long lam_part;
  S* Mp2;       // [B][24][16][16] per-panel M of the two-level factorization of S (6 n_cap > 192), else null
  double* Mp;   // [B][12][16][16] per-panel M of level A of the two-level Gram factorization (windows with 6 n_cap + 1 > 192), else null
  // covariance update: 0 = square-root gain form P <- P - W W^T, S factored by the blocked matrix-core Cholesky (float) /
  // the register-resident one (double); 1 = the reference's Joseph sequence; 2 = square-root gain form, register-resident solve
  int joseph;
  int gain_fused_s;   // float blocked gain solve: S = T_H (P T_H^T)[15:, :] + sigma^2 I is formed inside k_chol_mfma (no S GEMM launch)
  // gain_fused_s == 2 (default; 3 = the same with a zero wait, so that tests reach the fall-back): the parts of a trajectory share the product -- each forms a quarter of S's blocks, writes them to Smat
  // with agent-scope stores, the parts meet at a counter barrier (gain_bar, one 128-byte line per trajectory, only grows) and
  // read the rest back -- a part that does not see its siblings within ~1 ms forms the missing blocks itself; 1: every part forms all of S
  unsigned* gain_bar;
  // Kalman work matrices
  S* PHt; S* Smat; S* Linv; S* W; S* K; S* A; S* AP; S* X; S* dx;
  // prune
  int* keep; int* nkeep;
  LitBufs lit;
};

// first observation of track t of the launch's i-th trajectory in trk_slots / trk_obs
template <class S> __device__ __forceinline__ long wl_first(const Dev<S>& d, int i, int t) {
  return d.trk_off ? (long)d.trk_off[(long)i * d.wl_stride_f + t] : (long)i * d.wl_stride_o + (long)t * d.m_cap;
}

// XCD-aware decomposition of a 1-D grid of 8 * ceil(nb / 8) * items workgroups into (trajectory i < nb, item): MI355X places
// workgroup w on XCD w % 8 (MI355X_MICROARCH.md, workgroup dispatch), each XCD has a private 4 MiB L2, and every item of a
// trajectory reads the same covariance / per-track blocks.  Trajectories i, i + 8, ... share XCD i % 8, all their items
// with them: one XCD's L2 holds 1/8 of the batch instead of all of it.  A pure speed choice (placement is not a contract);
// returns false for the padding workgroups of an XCD that has fewer trajectories than the fullest one.
__device__ __forceinline__ bool xcd_item(int nb, int items, int& i, int& item) {
  const int w = (int)blockIdx.x, x = w & 7, j = w >> 3;
  const int q = j / items;
  i = x + 8 * q; item = j - q * items;
  return i < nb;
}
inline int xcd_grid(int nb, int items) { return 8 * ((nb + 7) / 8) * items; }

// Bookkeeping of a prune, by one workgroup of the trajectory (first 64 threads; nthreads >= 64): publishes the keep list when
// it is "drop the nd oldest", the window size after the next augment (ncam_upd), and compacts cam[] (keep[] ascending =>
// source slot >= destination slot).  n = window size before, nk = after.  Does not touch ncam.
template <class S>
__device__ __forceinline__ void prune_bookkeeping(const Dev<S>& d, int b, int tid, int n, int nk, int nd, int use_keep) {
  int* keep = d.keep + (long)b * d.n_cap;
  if (tid == 0) { d.ncam_upd[b] = min(nk, n) + 1; if (!use_keep) d.nkeep[b] = nk; }
  if (!use_keep && tid < 64) for (int k = tid; k < nk; k += 64) keep[k] = nd + k;
  if (nk < n && tid < 64) {
    S* cam = d.cam + (long)b * d.n_cap * CAM_STRIDE;
    for (int base = 0; base < nk; base += 64) {
      const int i = base + tid;
      S v[CAM_STRIDE];
      if (i < nk) { const int src = use_keep ? keep[i] : nd + i; for (int k = 0; k < CAM_STRIDE; ++k) v[k] = cam[(long)src * CAM_STRIDE + k]; }
      __builtin_amdgcn_wave_barrier();
      __threadfence_block();
      if (i < nk) for (int k = 0; k < CAM_STRIDE; ++k) cam[(long)i * CAM_STRIDE + k] = v[k];
      __builtin_amdgcn_wave_barrier();
      __threadfence_block();
    }
  }
}

// measurement Jacobian element idx of the per-track blocks [track][m_cap][12]
template <class S> __device__ __forceinline__ S ld_hx(const Dev<S>& d, long idx) {
  return d.h16 ? (S)__half2float(d.trk_Hx16[idx]) : d.trk_Hx[idx];
}

// ---------------------------------------------------------------- small math (all S-templated)
template <class S> struct V3 { S x, y, z; };
template <class S> __device__ __forceinline__ V3<S> mk3(S x, S y, S z) { V3<S> r; r.x = x; r.y = y; r.z = z; return r; }
template <class S> __device__ __forceinline__ V3<S> ld3(const S* p) { return mk3(p[0], p[1], p[2]); }
template <class S> __device__ __forceinline__ void st3(S* p, V3<S> v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
template <class S> __device__ __forceinline__ V3<S> operator+(V3<S> a, V3<S> b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
template <class S> __device__ __forceinline__ V3<S> operator-(V3<S> a, V3<S> b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
template <class S> __device__ __forceinline__ V3<S> operator*(S s, V3<S> a) { return mk3(s * a.x, s * a.y, s * a.z); }
template <class S> __device__ __forceinline__ S dot3(V3<S> a, V3<S> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <class S> __device__ __forceinline__ V3<S> cross3(V3<S> a, V3<S> b) {
  return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
template <class S> struct M3 { S m[3][3]; };
template <class S> __device__ __forceinline__ V3<S> mulv(const M3<S>& a, V3<S> v) {
  return mk3(a.m[0][0] * v.x + a.m[0][1] * v.y + a.m[0][2] * v.z,
             a.m[1][0] * v.x + a.m[1][1] * v.y + a.m[1][2] * v.z,
             a.m[2][0] * v.x + a.m[2][1] * v.y + a.m[2][2] * v.z);
}
template <class S> __device__ __forceinline__ V3<S> multv(const M3<S>& a, V3<S> v) {  // a^T v
  return mk3(a.m[0][0] * v.x + a.m[1][0] * v.y + a.m[2][0] * v.z,
             a.m[0][1] * v.x + a.m[1][1] * v.y + a.m[2][1] * v.z,
             a.m[0][2] * v.x + a.m[1][2] * v.y + a.m[2][2] * v.z);
}
template <class S> __device__ __forceinline__ M3<S> mulm(const M3<S>& a, const M3<S>& b) {
  M3<S> r;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j];
  return r;
}
template <class S> __device__ __forceinline__ M3<S> mulmt(const M3<S>& a, const M3<S>& b) {  // a * b^T
  M3<S> r;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[i][0] * b.m[j][0] + a.m[i][1] * b.m[j][1] + a.m[i][2] * b.m[j][2];
  return r;
}
template <class S> __device__ __forceinline__ M3<S> skew3(V3<S> v) {  // matrix_utils.h:8-17
  M3<S> r;
  r.m[0][0] = 0; r.m[0][1] = -v.z; r.m[0][2] = v.y;
  r.m[1][0] = v.z; r.m[1][1] = 0; r.m[1][2] = -v.x;
  r.m[2][0] = -v.y; r.m[2][1] = v.x; r.m[2][2] = 0;
  return r;
}
template <class S> struct Q4 { S w, x, y, z; };
template <class S> __device__ __forceinline__ Q4<S> ldq(const S* p) { Q4<S> q; q.w = p[0]; q.x = p[1]; q.y = p[2]; q.z = p[3]; return q; }
template <class S> __device__ __forceinline__ void stq(S* p, Q4<S> q) { p[0] = q.w; p[1] = q.x; p[2] = q.y; p[3] = q.z; }
template <class S> __device__ __forceinline__ M3<S> q2rot(Q4<S> q) {  // Eigen toRotationMatrix
  const S tx = S(2) * q.x, ty = S(2) * q.y, tz = S(2) * q.z;
  const S twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const S txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const S tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  M3<S> r;
  r.m[0][0] = S(1) - (tyy + tzz); r.m[0][1] = txy - twz; r.m[0][2] = txz + twy;
  r.m[1][0] = txy + twz; r.m[1][1] = S(1) - (txx + tzz); r.m[1][2] = tyz - twx;
  r.m[2][0] = txz - twy; r.m[2][1] = tyz + twx; r.m[2][2] = S(1) - (txx + tyy);
  return r;
}
template <class S> __device__ __forceinline__ Q4<S> qmul(Q4<S> a, Q4<S> b) {
  Q4<S> r;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
  r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
  return r;
}
template <class S> __device__ __forceinline__ S dsqrt(S x);
template <> __device__ __forceinline__ float dsqrt<float>(float x) { return sqrtf(x); }
template <> __device__ __forceinline__ double dsqrt<double>(double x) { return sqrt(x); }
template <class S> __device__ __forceinline__ Q4<S> qnormalized(Q4<S> q) {
  S n = dsqrt(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
  q.w /= n; q.x /= n; q.y /= n; q.z /= n;
  return q;
}
template <class S> __device__ __forceinline__ Q4<S> qinverse(Q4<S> q) {  // Eigen: conj / squaredNorm
  S n2 = q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z;
  Q4<S> r; r.w = q.w / n2; r.x = -q.x / n2; r.y = -q.y / n2; r.z = -q.z / n2;
  return r;
}
template <class S> __device__ __forceinline__ V3<S> qrotate(Q4<S> q, V3<S> v) {  // Eigen _transformVector
  V3<S> u = mk3(q.x, q.y, q.z);
  V3<S> uv = cross3(u, v);
  uv = uv + uv;
  return v + (q.w * uv) + cross3(u, uv);
}
// buildUpdateQuat, msckf.h:851-872
template <class S> __device__ __forceinline__ Q4<S> update_quat(V3<S> dtheta) {
  V3<S> dq = S(0.5) * dtheta;
  S cs = dot3(dq, dq);
  Q4<S> q;
  q.w = (cs > S(1)) ? S(1) : dsqrt(S(1) - cs);
  q.x = -dq.x; q.y = -dq.y; q.z = -dq.z;
  return qnormalized(q);
}

// ---------------------------------------------------------------- wave64 helpers
__device__ __forceinline__ float wave_bcast(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ double wave_bcast(double v, int lane) {
  long long b = __double_as_longlong(v);
  int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), lane);
  int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ int wave_bcast(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
// value of `v` in the lane whose byte index (lane << 2) is `src4`: ds_bpermute, the LDS crossbar without LDS memory
__device__ __forceinline__ float lane_gather(float v, int src4) { return __int_as_float(__builtin_amdgcn_ds_bpermute(src4, __float_as_int(v))); }
__device__ __forceinline__ double lane_gather(double v, int src4) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_ds_bpermute(src4, (int)(b & 0xffffffffLL)), hi = __builtin_amdgcn_ds_bpermute(src4, (int)(b >> 32));
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// Wave-wide reductions on the DPP network instead of ds_bpermute shuffles: a reduction is four DPP steps inside the
// rows of 16 lanes (quad swaps, half-row mirror, row mirror: every lane of a row ends up with the row's result)
// plus one v_readlane per row.  A dependent chain of 6 LDS-crossbar round trips becomes ~10 short VALU ops; the
// summation order is fixed, every lane gets the same value.
template <int CTRL> __device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
template <int CTRL> __device__ __forceinline__ float dpp_x(float v) { return __int_as_float(dpp_i<CTRL>(__float_as_int(v))); }
template <int CTRL> __device__ __forceinline__ double dpp_x(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = dpp_i<CTRL>((int)(b & 0xffffffffLL)), hi = dpp_i<CTRL>((int)(b >> 32));
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
template <int CTRL> __device__ __forceinline__ int dpp_x(int v) { return dpp_i<CTRL>(v); }
constexpr int DPP_QUAD_X1 = 0xB1, DPP_QUAD_X2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_ROW_MIRROR = 0x140;

// masked DPP move: lanes of the rows selected by ROWMASK receive the source lane's value, the others 0
template <int CTRL, int ROWMASK> __device__ __forceinline__ int dppm_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, ROWMASK, 0xF, false); }
template <int CTRL, int ROWMASK> __device__ __forceinline__ float dppm_x(float v) { return __int_as_float(dppm_i<CTRL, ROWMASK>(__float_as_int(v))); }
template <int CTRL, int ROWMASK> __device__ __forceinline__ double dppm_x(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = dppm_i<CTRL, ROWMASK>((int)(b & 0xffffffffLL)), hi = dppm_i<CTRL, ROWMASK>((int)(b >> 32));
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
constexpr int DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143;

template <class S> __device__ __forceinline__ S wave_sum(S v) {
  v += dpp_x<DPP_QUAD_X1>(v); v += dpp_x<DPP_QUAD_X2>(v); v += dpp_x<DPP_HALF_MIRROR>(v); v += dpp_x<DPP_ROW_MIRROR>(v);
  // every lane holds its row's sum: fold row 0 into 1 and 2 into 3, then rows {0,1} into 3; lane 63 has the total
  v += dppm_x<DPP_ROW_BCAST15, 0xA>(v);
  v += dppm_x<DPP_ROW_BCAST31, 0xC>(v);
  return wave_bcast(v, 63);
}
template <class S> __device__ __forceinline__ S wave_max(S v) {
  S t;
  t = dpp_x<DPP_QUAD_X1>(v); v = t > v ? t : v; t = dpp_x<DPP_QUAD_X2>(v); v = t > v ? t : v;
  t = dpp_x<DPP_HALF_MIRROR>(v); v = t > v ? t : v; t = dpp_x<DPP_ROW_MIRROR>(v); v = t > v ? t : v;
  const S a = wave_bcast(v, 0), b = wave_bcast(v, 16), c = wave_bcast(v, 32), e = wave_bcast(v, 48);
  const S ab = a > b ? a : b, ce = c > e ? c : e;
  return ab > ce ? ab : ce;
}
__device__ __forceinline__ int wave_min_i(int v) {
  int t;
  t = dpp_x<DPP_QUAD_X1>(v); v = t < v ? t : v; t = dpp_x<DPP_QUAD_X2>(v); v = t < v ? t : v;
  t = dpp_x<DPP_HALF_MIRROR>(v); v = t < v ? t : v; t = dpp_x<DPP_ROW_MIRROR>(v); v = t < v ? t : v;
  return min(min(wave_bcast(v, 0), wave_bcast(v, 16)), min(wave_bcast(v, 32), wave_bcast(v, 48)));
}

// 1/sqrt(x) for the pivots of the register-resident factorizations: hardware rsq seed + Newton steps instead of the
// IEEE sqrt and division sequences (the pivot sits on the critical path of every elimination step)
template <class S> __device__ __forceinline__ S fast_rsqrt(S x);
template <> __device__ __forceinline__ float fast_rsqrt<float>(float x) {
  const float r = __builtin_amdgcn_rsqf(x);
  return r * (1.5f - 0.5f * x * r * r);
}
template <> __device__ __forceinline__ double fast_rsqrt<double>(double x) {
  double r = __builtin_amdgcn_rsq(x);
  r = r + 0.5 * r * (1.0 - x * r * r);
  r = r + 0.5 * r * (1.0 - x * r * r);
  return r;
}

// 1/x from the hardware seed + Newton steps (full precision of the type to ~1 ulp): the IEEE division sequence is ~10 (f32)
// / ~25 (f64) instructions
template <class S> __device__ __forceinline__ S fast_rcp(S x);
template <> __device__ __forceinline__ float fast_rcp<float>(float x) {
  const float r = __builtin_amdgcn_rcpf(x);
  return __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
}
template <> __device__ __forceinline__ double fast_rcp<double>(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
  return r;
}

template <class S> struct Lim;
template <> struct Lim<float> { static __device__ __forceinline__ float tiny() { return 1.17549435e-38f; } };
template <> struct Lim<double> { static __device__ __forceinline__ double tiny() { return 2.2250738585072014e-308; } };

// Block-diagonal part of Lam^ = [H_o | r_o]^T [H_o | r_o] at (I, J): sum h^T h inside a camera's 6 x 6 block, sum h^T r
// in row/column n, zero elsewhere (Dg: per camera slot 21 upper-triangle entries + 6 entries of h^T r).
__device__ __forceinline__ double lam_diag_term(const double* Dg, int n, int n_cap, int I, int J) {
  const int hi = I >= J ? I : J, lo = I >= J ? J : I;
  const bool isy = hi == n && lo < n, isd = hi < n && (hi / 6 == lo / 6);
  const int a6 = lo % 6, c6 = hi % 6;
  const int idx = isy ? 21 + a6 : (isd ? a6 * 6 - a6 * (a6 - 1) / 2 + (c6 - a6) : 0);
  const double dg = Dg[min(lo / 6, n_cap - 1) * DG_STRIDE + idx];
  return (isy || isd) ? dg : 0.0;
}
// Lam^(I, J) as k_gram left it (block-diagonal part minus sum B^T B, both triangles stored): lower triangle incl. row n
// (= H_o^T r_o); rows beyond n and the (n, n) corner read as zero.  Branch-free: the load is issued unconditionally from
// a clamped address and masked afterwards, so that a thread's loads are all in flight together.
__device__ __forceinline__ double lam_hat(const double* Lam, const double* /*Dg*/, int ldL, int n, int /*n_cap*/, int I, int J) {
  const int hi = I >= J ? I : J, lo = I >= J ? J : I;
  const bool ok = hi <= n && lo < n;
  const double lv = Lam[(long)min(hi, ldL - 1) * ldL + min(lo, ldL - 1)];
  return ok ? lv : 0.0;
}

// ---------------------------------------------------------------- launch entry points (one per .hip file)
template <class S> void launch_propagate(const Dev<S>& d, int b0, int nb, const S* readings, long rd_stride, int K, hipStream_t st, bool then_augment = false);
template <class S> void launch_augment(const Dev<S>& d, int b0, int nb, hipStream_t st);
template <class S> void launch_prune(const Dev<S>& d, int b0, int nb, hipStream_t st, const int* drop = nullptr, int drop_const = -1);
template <class S> void launch_feature(const Dev<S>& d, int b0, int nb, hipStream_t st);
template <class S> void launch_select(const Dev<S>& d, int b0, int nb, hipStream_t st);
template <class S> void launch_select_diag(const Dev<S>& d, int b0, int nb, hipStream_t st);   // k_select + block-diagonal part of Lam^ in one launch
// phase: 0 = stage 1 + merges, 1 = stage 1 only (chunk-local QR updates), 2 = merge tree only
template <class S> void launch_compress(const Dev<S>& d, int b0, int nb, hipStream_t st, int phase);
// phase: 0 = both, 1 = Gram accumulation only, 2 = Cholesky only, 3 = SYRK only (the block-diagonal part came with launch_select_diag)
template <class S> void launch_gram(const Dev<S>& d, int b0, int nb, hipStream_t st, int phase);
template <class S> void launch_kalman(const Dev<S>& d, int b0, int nb, hipStream_t st);
// kernels_kalman.hip: Gram matrix, both factorizations, gain, state injection and covariance downdate of a SMALL window (n_max = 6 x cameras) in one
// launch, one workgroup per trajectory; false when the window does not fit its LDS (the caller then takes the usual chain)
template <class S> bool launch_update_small(const Dev<S>& d, int b0, int nb, hipStream_t st, int n_max);
size_t update_small_lds_bytes(int n_max, int f_cap, size_t scalar);
template <class S> void launch_literal(const Dev<S>& d, int b0, int nb, hipStream_t st, int part = 0);   // kernels_literal.hip: Lam^ of the literal anisotropic compression (part: 0 all, 1 k_lit_pre, 2 k_lit_gamma, 3 k_literal)
// blocked matrix-core Cholesky (kernels_chol.hip): [T | r_n] = chol(Lam^) for the information form; S = L L^T with
// [PHt ; r_n^T] appended (W, dx) for the float Kalman stage.  Return false when the window does not fit the kernel.
template <class S> bool launch_chol_gram(const Dev<S>& d, int b0, int nb, hipStream_t st);
bool launch_chol_gain(const Dev<float>& d, int b0, int nb, hipStream_t st);
template <class S> bool launch_chol_gain_large(const Dev<S>& d, int b0, int nb, hipStream_t st);
size_t feature_lds_bytes(int m_cap, size_t scalar, bool staged = false);
// one-time per-device setup of each kernel file (constant tables, dynamic-LDS limits); msckf_hip_create calls them
void feature_device_setup();
void qr_device_setup();
void kalman_device_setup();
void gram_device_setup();
void literal_device_setup();
bool literal_lds_available();   // false when the device refused the phase kernels' dynamic LDS (94 KB per workgroup)
long lit_ws_doubles(int n6, int m_cap, int r_cap);   // per-trajectory scratch of k_literal (doubles)

}  // namespace msckf
#endif
