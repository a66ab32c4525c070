// kernels_state.hip -- covariance propagation, state augmentation and camera-state pruning (gfx950).
//
//   k_propagate  <- MSCKF::propagate            msckf.h:101-145 (+ calcF :874-890, calcG :892-903,
//                                                propogateImuStateRK :1425-1467, OC patch :116-132)
//   k_augment    <- MSCKF::augmentState         msckf.h:148-212
//   k_prune      <- pruneEmptyStates' gather    msckf.h:719-757, matrix_utils.h:58-87
//
// Design (not a translation): K queued IMU samples are fused into one launch; the 15x15 blocks live in
// LDS, Phi_total = Phi_K...Phi_1 is accumulated so the only O(N) part, P_IC <- Phi P_IC, touches HBM once
// per image instead of once per IMU sample; G Q G^T is applied in its closed block-diagonal form;
// augmentation uses the 6 non-zero 3x3 blocks of J instead of two dense (D+6) x D GEMMs and never
// computes the unused determinant of msckf.h:176.
#include "dev_common.h"

namespace msckf {

// One wavefront multiplies two 15x15 row-major LDS matrices (no aliasing between C and A/B)
template <class S>
__device__ __forceinline__ void mm15(S* C, const S* A, const S* B, int lane) {
  for (int e = lane; e < 225; e += 64) {
    const int i = e / 15, j = e % 15;
    S s = 0;
#pragma unroll
    for (int k = 0; k < 15; ++k) s += A[i * 15 + k] * B[k * 15 + j];
    C[e] = s;
  }
}
template <class S>
__device__ __forceinline__ void mm15_abt(S* C, const S* A, const S* B, int lane) {  // C = A*B^T
  for (int e = lane; e < 225; e += 64) {
    const int i = e / 15, j = e % 15;
    S s = 0;
#pragma unroll
    for (int k = 0; k < 15; ++k) s += A[i * 15 + k] * B[j * 15 + k];
    C[e] = s;
  }
}
__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }

constexpr int PG = 16;      // IMU samples per group
constexpr int SST = 20;     // LDS stride of one IMU state: q(4) b_g(3) v(3) b_a(3) p(3) dT(1) pad

// propogateImuStateRK, msckf.h:1425-1467: RK on the JPL-ordered quaternion (-x,-y,-z,w) with 0.5*Omega(w),
// Euler on v and p.  `st` holds q b_g v b_a p; returns the propagated q, v, p in `out`.
template <class S>
__device__ __forceinline__ void imu_rk(const S* st, V3<S> g, V3<S> om, V3<S> ac, S dT, S* out) {
  const Q4<S> q = ldq(st);
  const V3<S> bg = ld3(st + 4), v = ld3(st + 7), ba = ld3(st + 10), p = ld3(st + 13);
  const V3<S> wh = om - bg, ah = ac - ba;
  const M3<S> C = q2rot(q);
  S y0[4] = {-q.x, -q.y, -q.z, q.w};
  auto omul = [&](const S* y, S* o) {  // o = 0.5*omegaMat(wh) * y   (matrix_utils.h:20-30)
    o[0] = S(0.5) * (wh.z * y[1] - wh.y * y[2] + wh.x * y[3]);
    o[1] = S(0.5) * (-wh.z * y[0] + wh.x * y[2] + wh.y * y[3]);
    o[2] = S(0.5) * (wh.y * y[0] - wh.x * y[1] + wh.z * y[3]);
    o[3] = S(0.5) * (-wh.x * y[0] - wh.y * y[1] - wh.z * y[2]);
  };
  S k0[4], k1[4], k2[4], k3[4], k4[4], k5[4], t[4];
  omul(y0, k0);
  for (int i = 0; i < 4; ++i) t[i] = y0[i] + (k0[i] / S(4)) * dT;
  omul(t, k1);
  for (int i = 0; i < 4; ++i) t[i] = y0[i] + (k0[i] / S(8) + k1[i] / S(8)) * dT;
  omul(t, k2);
  for (int i = 0; i < 4; ++i) t[i] = y0[i] + (-k1[i] / S(2) + k2[i]) * dT;
  omul(t, k3);
  for (int i = 0; i < 4; ++i) t[i] = y0[i] + (k0[i] * S(3) / S(16) + k3[i] * S(9) / S(16)) * dT;
  omul(t, k4);
  for (int i = 0; i < 4; ++i)
    t[i] = y0[i] + (-k0[i] * S(3) / S(7) + k1[i] * S(2) / S(7) + k2[i] * S(12) / S(7) - k3[i] * S(12) / S(7) + k4[i] * S(8) / S(7)) * dT;
  omul(t, k5);
  S yt[4];
  for (int i = 0; i < 4; ++i) yt[i] = y0[i] + (S(7) * k0[i] + S(32) * k2[i] + S(12) * k3[i] + S(32) * k4[i] + S(7) * k5[i]) * dT / S(90);
  Q4<S> qn; qn.w = yt[3]; qn.x = -yt[0]; qn.y = -yt[1]; qn.z = -yt[2];
  qn = qnormalized(qn);
  stq(out, qn);
  st3(out + 4, bg);
  st3(out + 7, v + (dT * (multv(C, ah) + g)));
  st3(out + 10, ba);
  st3(out + 13, p + (dT * v));
}

// K queued IMU samples in one launch, one workgroup (4 wavefronts) per trajectory, samples in groups of PG:
//   A  IMU *state* chain (:105, :1425-1467): the RK step is a linear map of the quaternion that depends on the reading only
//      (the biases are constant while propagating), so the maps M_s are built in parallel and one thread walks
//      q_{s+1} = normalize(M_s q_s); velocity and position follow from the per-sample rotations
//   B  Phi_s = expm(F_s dT), one thread per sample: F dT is assembled from its five 3x3 blocks (calcF :885-889) and its
//      exponential reduces to three series in the 3x3 matrix -[w^ x] dT (replaces Eigen's Pade, :111 -- the same Taylor
//      series, equal to working precision), then OC-patched (:116-132)
//   C  the sequential chains P_II <- sym(Phi (P_II + G Q G^T dT) Phi^T) (:134,143) and Phi_total = Phi_K ... Phi_1 advance
//      together, one 15 x 15 element per thread
//   D  P_IC <- Phi_total P_IC for all camera columns, both halves of the symmetric storage    (:144)
#ifdef MSCKF_ABLATE
// phase timers of the -DMSCKF_ABLATE build (scripts/chol_phases.py): shader-clock cycles of workgroup 0, thread 0:
// 0 load, 1 state chain, 2 Phi series, 3 P_II / Phi_total chains, 4 write back + P_IC, 5 launches
__device__ unsigned long long g_prop_cycles[8];
#define PR_TICK(slot) do { if (tid == 0) { const long long t_ = clock64(); pcyc[slot] += t_ - ptl; ptl = t_; } } while (0)
void prop_cycles_read(unsigned long long* out8, int reset) {
  (void)hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_prop_cycles), sizeof(unsigned long long) * 8);
  if (reset) { unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_prop_cycles), z, sizeof(z)); }
}
#else
#define PR_TICK(slot) do {} while (0)
#endif

// augmentState (msckf.h:148-212) for one trajectory by one workgroup of 256 threads; sJP: [6][ld] scratch in LDS.  Called by
// k_augment and, fused, at the end of k_propagate (run_frames: the two are always back to back, one launch less per frame).
template <class S>
__device__ __forceinline__ void augment_body(const Dev<S>& d, int b, int tid, S* sJP) {
  const int n = d.ncam[b];
  if (n >= d.n_cap) { if (tid == 0) atomicOr(&d.stats[(long)b * STAT_STRIDE + STAT_ERR], STAT_ERR_NCAP); return; }
  const int D = 15 + 6 * n, ld = d.ld;
  const S* imu = d.imu + (long)b * IMU_STRIDE;
  const S* prm = d.prm + (long)b * PRM_STRIDE;
  S* P = d.P + (long)b * ld * ld;
  const Q4<S> q = ldq(imu + IQ), qci = ldq(prm + PRM_QCI);
  const V3<S> pci = ld3(prm + PRM_PCI);
  const V3<S> lever = qrotate(qinverse(q), pci);   // q_IG^-1 * p_C_I   :160,183
  const M3<S> Jtt = q2rot(qci), Jpt = skew3(lever);
  if (tid == 0) {
    S* cs = d.cam + ((long)b * d.n_cap + n) * CAM_STRIDE;
    stq(cs, qnormalized(qmul(qci, q)));             // :152-154
    st3(cs + 4, ld3(imu + IP) + lever);             // :159-160
  }
  for (int c = tid; c < D; c += 256) {              // J P, J non-zero only in cols 0-2 and 12-14 (:180-184)
    const S* pc = P + (long)c * ld;
    const V3<S> th = mk3(pc[0], pc[1], pc[2]);
    const V3<S> a = mulv(Jtt, th), bb = mulv(Jpt, th);
    const S jp[6] = {a.x, a.y, a.z, bb.x + pc[12], bb.y + pc[13], bb.z + pc[14]};
#pragma unroll
    for (int i = 0; i < 6; ++i) { sJP[i * ld + c] = jp[i]; P[(long)c * ld + D + i] = jp[i]; P[(long)(D + i) * ld + c] = jp[i]; }
  }
  __syncthreads();
  if (tid < 36) {                                   // corner J P J^T, symmetrised (:195-197)
    const int i = tid / 6, j = tid % 6;
    auto corner = [&](int r, int cc) {
      const S* jp = sJP + r * ld;
      if (cc < 3) return jp[0] * Jtt.m[cc][0] + jp[1] * Jtt.m[cc][1] + jp[2] * Jtt.m[cc][2];
      const int c3 = cc - 3;
      return jp[0] * Jpt.m[c3][0] + jp[1] * Jpt.m[c3][1] + jp[2] * Jpt.m[c3][2] + jp[12 + c3];
    };
    P[(long)(D + j) * ld + D + i] = (corner(i, j) + corner(j, i)) / S(2);
  }
  if (tid == 0) d.ncam[b] = n + 1;
}

template <class S>
__global__ __launch_bounds__(256) void k_augment(Dev<S> d, int b0) {
  const int b = b0 + blockIdx.x, tid = threadIdx.x;
  __builtin_amdgcn_s_setprio(3);   // latency-bound chain: win instruction arbitration against co-resident throughput waves of the other slice
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  augment_body<S>(d, b, tid, reinterpret_cast<S*>(smem_raw));
}

// 16x16x4 MFMA of the scalar type, with the row a lane group's accumulator register r belongs to
typedef double pd4 __attribute__((ext_vector_type(4)));
typedef float pf4 __attribute__((ext_vector_type(4)));
template <class T> struct Mfs;
template <> struct Mfs<double> {   // v_mfma_f64_16x16x4_f64: C/D row = g + 4 r
  typedef pd4 V;
  static __device__ __forceinline__ V mma(double a, double b, V c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ int crow(int g, int r) { return g + 4 * r; }
};
template <> struct Mfs<float> {    // v_mfma_f32_16x16x4_f32: C/D row = 4 g + r
  typedef pf4 V;
  static __device__ __forceinline__ V mma(float a, float b, V c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ int crow(int g, int r) { return 4 * g + r; }
};

template <class S, bool AUGMENT>
__global__ __launch_bounds__(256) void k_propagate(Dev<S> d, int b0, const S* readings, long rd_stride, int K) {
  const int b = b0 + blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  __builtin_amdgcn_s_setprio(3);   // latency-bound chain: win instruction arbitration against co-resident throughput waves of the other slice
#ifdef MSCKF_ABLATE
  long long pcyc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ptl = clock64();
#endif
  __shared__ S sState[(PG + 1) * SST];
  __shared__ S sPhi[PG * 225];
  __shared__ S sQt[PG * 32];        // G Q G^T dT of the group's samples, compact: [0,15) diagonal, [15,24) the C^T Qa C block, 31 = 0
  __shared__ S sTot[225];
  __shared__ S sPii[225];
  __shared__ S sNull[12];          // q_null v_null p_null used by the next sample
  __shared__ S sG[4];
  __shared__ S sRd[PG * RD_STRIDE];  // the group's IMU samples, staged once (the state chain is one thread: a global
                                      // read per sample would put a memory round trip on every step of the chain)
  __shared__ S sQ[12];
  __shared__ S sRk[PG * 16];         // RK maps M_s of the group's samples (column-major 4 x 4)
  __shared__ S sDv[PG * 3];          // velocity increments of the group's samples
  S* imu = d.imu + (long)b * IMU_STRIDE;
  const S* prm = d.prm + (long)b * PRM_STRIDE;
  S* P = d.P + (long)b * d.ld * d.ld;
  const int ld = d.ld;
  // a prune that rode on the previous frame's covariance downdate left the new window size in ncam_upd (+1): commit it here,
  // where no other workgroup of the trajectory can be reading ncam (augment_body re-reads it behind barriers of this workgroup)
  const int ncams = d.ncam_defer ? d.ncam_upd[b] - 1 : d.ncam[b];
  if (d.ncam_defer && tid == 0) d.ncam[b] = ncams;
  const int n = 6 * ncams;
  const S* rd = readings + (long)(b - b0) * rd_stride;
  // the IMU-camera block of P is only multiplied by Phi_total at the very end: fetch this wave's tiles of it (16 camera
  // columns each, B-operand layout of the 16x16x4 MFMA) now, so that the loads fly under the whole chain
  constexpr int PIC_T = 6;   // 4 waves x 6 tiles x 16 columns >= 6 * 63
  S pic[PIC_T][4];
  {
    const int g = lane >> 4, c = lane & 15;
#pragma unroll
    for (int t = 0; t < PIC_T; ++t) {
      const int col = (w + 4 * t) * 16 + c;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) { const int k = Mfs<S>::crow(g, kk); pic[t][kk] = P[(long)(15 + min(col, d.n6cap - 1)) * ld + min(k, 14)]; }   // unconditional (masked where used): no wait for the window size in front of the kernel's first loads
    }
  }
  if (tid < 16) sState[tid] = imu[tid];                 // q b_g v b_a p
  if (tid < 3) sG[tid] = imu[IG + tid];
  if (tid >= 64 && tid < 64 + 12) sQ[tid - 64] = prm[PRM_Q + tid - 64];
  if (tid >= 32 && tid < 32 + 10) sNull[tid - 32] = imu[IQN + (tid - 32)];
  for (int e = tid; e < 225; e += 256) {
    const int i = e / 15, j = e % 15;
    sPii[e] = P[(long)j * ld + i];
  }
  for (int e = tid; e < min(PG, K) * RD_STRIDE; e += 256) sRd[e] = rd[e];   // the first group's IMU samples ride on the same round trip
  __syncthreads();
  // chain matrices in MFMA accumulator layout (phase C): wave 0 P_II, wave 1 Phi_total
  typename Mfs<S>::V Mreg = {0, 0, 0, 0};
  {
    const int g = lane >> 4, c = lane & 15;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = Mfs<S>::crow(g, r);
      if (i < 15 && c < 15) Mreg[r] = w == 0 ? sPii[i * 15 + c] : (i == c ? S(1) : S(0));
    }
  }
  for (int k0 = 0; k0 < K; k0 += PG) {
    const int G = min(PG, K - k0);
    if (k0 > 0) { for (int e = tid; e < G * RD_STRIDE; e += 256) sRd[e] = rd[(long)k0 * RD_STRIDE + e]; __syncthreads(); }
    PR_TICK(0);
    const V3<S> g = mk3(sG[0], sG[1], sG[2]);
    // ---- A: state chain.  The gyro bias does not change during propagation, so the RK step of sample s is a LINEAR map
    // y -> M_s y of the (JPL-ordered) quaternion that depends on the reading only: thread (s, c) runs the reference's
    // six stages (propogateImuStateRK, msckf.h:1437-1453) on the unit vector e_c and leaves column c of M_s in LDS; one
    // thread then walks the chain q_{s+1} = normalize(M_s q_s) (16 FMAs + a normalisation per sample instead of ~300
    // dependent operations), and the velocity / position chains follow from the per-sample rotations.
    if (tid < 4 * G) {
      const int s = tid >> 2, c = tid & 3;
      const S* r = sRd + s * RD_STRIDE;
      const S dT = r[6];
      const V3<S> wh = ld3(r) - mk3(sState[4], sState[5], sState[6]);
      auto omul = [&](const S* y, S* o) {  // o = 0.5*omegaMat(wh) * y   (matrix_utils.h:20-30)
        o[0] = S(0.5) * (wh.z * y[1] - wh.y * y[2] + wh.x * y[3]);
        o[1] = S(0.5) * (-wh.z * y[0] + wh.x * y[2] + wh.y * y[3]);
        o[2] = S(0.5) * (wh.y * y[0] - wh.x * y[1] + wh.z * y[3]);
        o[3] = S(0.5) * (-wh.x * y[0] - wh.y * y[1] - wh.z * y[2]);
      };
      S y0[4] = {c == 0 ? S(1) : S(0), c == 1 ? S(1) : S(0), c == 2 ? S(1) : S(0), c == 3 ? S(1) : S(0)};
      S k0[4], k1[4], k2[4], k3[4], k4[4], k5[4], t[4];
      omul(y0, k0);
      for (int i = 0; i < 4; ++i) t[i] = y0[i] + (k0[i] / S(4)) * dT;
      omul(t, k1);
      for (int i = 0; i < 4; ++i) t[i] = y0[i] + (k0[i] / S(8) + k1[i] / S(8)) * dT;
      omul(t, k2);
      for (int i = 0; i < 4; ++i) t[i] = y0[i] + (-k1[i] / S(2) + k2[i]) * dT;
      omul(t, k3);
      for (int i = 0; i < 4; ++i) t[i] = y0[i] + (k0[i] * S(3) / S(16) + k3[i] * S(9) / S(16)) * dT;
      omul(t, k4);
      for (int i = 0; i < 4; ++i)
        t[i] = y0[i] + (-k0[i] * S(3) / S(7) + k1[i] * S(2) / S(7) + k2[i] * S(12) / S(7) - k3[i] * S(12) / S(7) + k4[i] * S(8) / S(7)) * dT;
      omul(t, k5);
      for (int i = 0; i < 4; ++i) sRk[(s * 4 + c) * 4 + i] = y0[i] + (S(7) * k0[i] + S(32) * k2[i] + S(12) * k3[i] + S(32) * k4[i] + S(7) * k5[i]) * dT / S(90);
    }
    __syncthreads();
    if (tid == 0) {
      Q4<S> q = ldq(sState);
      for (int s = 0; s < G; ++s) {
        const S y0[4] = {-q.x, -q.y, -q.z, q.w};
        const S* M = sRk + s * 16;                       // column c at M[c*4 ..]
        S yt[4];
        for (int i = 0; i < 4; ++i) yt[i] = M[i] * y0[0] + M[4 + i] * y0[1] + M[8 + i] * y0[2] + M[12 + i] * y0[3];
        Q4<S> qn; qn.w = yt[3]; qn.x = -yt[0]; qn.y = -yt[1]; qn.z = -yt[2];
        q = qnormalized(qn);
        stq(sState + (s + 1) * SST, q);
      }
    }
    __syncthreads();
    if (tid < G) {          // velocity increment of sample s from ITS rotation: (C_IG^T (a - b_a) + g) dT   (:1459-1461)
      const int s = tid;
      const S* r = sRd + s * RD_STRIDE;
      const M3<S> C = q2rot(ldq(sState + s * SST));
      const V3<S> ah = ld3(r + 3) - mk3(sState[10], sState[11], sState[12]);
      st3(sDv + 3 * s, r[6] * (multv(C, ah) + g));
    }
    __syncthreads();
    if (tid == 0) {
      V3<S> v = ld3(sState + 7), p = ld3(sState + 13);
      const V3<S> bg = ld3(sState + 4), ba = ld3(sState + 10);
      for (int s = 0; s < G; ++s) {
        const S dT = sRd[s * RD_STRIDE + 6];
        const V3<S> vn = v + ld3(sDv + 3 * s), pn = p + (dT * v);      // p uses the OLD velocity (:1465)
        S* o = sState + (s + 1) * SST;
        st3(o + 4, bg); st3(o + 7, vn); st3(o + 10, ba); st3(o + 13, pn);
        v = vn; p = pn;
      }
    }
    __syncthreads();
    PR_TICK(1);
    // ---- B: Phi_s = expm(F_s dT) of every sample, one thread per sample.  F dT has five non-zero 3x3 blocks (calcF
    // :885-889): W = -[w^ x] dT at (th,th), -I dT at (th,bg), X = -C^T [a^ x] dT at (v,th), -C^T dT at (v,ba), I dT at (p,v),
    // and its powers reduce to powers of the 3x3 matrix W:  with E_m = sum_j W^j / (j+m)!
    //   Phi(th,bg) = -dT E_1   Phi(v,th) = X E_1   Phi(v,bg) = -dT X E_2   Phi(v,ba) = -C^T dT
    //   Phi(p,th) = dT X E_2   Phi(p,bg) = -dT^2 X E_3   Phi(p,v) = dT I    Phi(p,ba) = -C^T dT^2 / 2,   identity diagonal
    // -- the same Taylor series as expm of the 15x15 matrix (Eigen's Pade, msckf.h:111, to working precision), summed until
    // the terms vanish; Phi(th,th) is overwritten by the observability patch anyway (:117-118).
    for (int e = tid; e < G * 225; e += 256) { const int q = e % 225; sPhi[e] = (q / 15 == q % 15) ? S(1) : S(0); }
    __syncthreads();
    if (tid < G) {
      const int s = tid;
      S* Phi = sPhi + s * 225;
      const S* st = sState + s * SST;
      const S* r = sRd + s * RD_STRIDE;
      const S dT = r[6];
      const V3<S> wh = ld3(r) - ld3(st + 4), ah = ld3(r + 3) - ld3(st + 10);
      const M3<S> C = q2rot(ldq(st));
      const M3<S> sw = skew3(wh), sa = skew3(ah);
      M3<S> W, X, Pw, E1, E2, E3;
      for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
        W.m[i][j] = -sw.m[i][j] * dT;
        S sm = 0;
        for (int kk = 0; kk < 3; ++kk) sm += C.m[kk][i] * sa.m[kk][j];
        X.m[i][j] = -sm * dT;
        const S id = (i == j) ? S(1) : S(0);
        Pw.m[i][j] = id; E1.m[i][j] = id; E2.m[i][j] = id / S(2); E3.m[i][j] = id / S(6);
      }
      S r1 = 1, r2 = S(0.5), r3 = S(1) / S(6);        // 1/(j+1)!, 1/(j+2)!, 1/(j+3)! for j = 0
      for (int j = 1; j <= 40; ++j) {
        Pw = mulm(Pw, W);                              // W^j
        r1 /= S(j + 1); r2 /= S(j + 2); r3 /= S(j + 3);
        S mx = 0;
        for (int a2 = 0; a2 < 3; ++a2) for (int c2 = 0; c2 < 3; ++c2) {
          const S t1 = Pw.m[a2][c2] * r1;
          E1.m[a2][c2] += t1; E2.m[a2][c2] += Pw.m[a2][c2] * r2; E3.m[a2][c2] += Pw.m[a2][c2] * r3;
          const S am = t1 < 0 ? -t1 : t1;
          mx = am > mx ? am : mx;
        }
        if (mx < (sizeof(S) == 4 ? S(1e-11) : S(1e-21))) break;
      }
      const M3<S> XE1 = mulm(X, E1), XE2 = mulm(X, E2), XE3 = mulm(X, E3);
      for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
        Phi[i * 15 + 3 + j] = -dT * E1.m[i][j];
        Phi[(6 + i) * 15 + j] = XE1.m[i][j];
        Phi[(6 + i) * 15 + 3 + j] = -dT * XE2.m[i][j];
        Phi[(6 + i) * 15 + 9 + j] = -C.m[j][i] * dT;
        Phi[(12 + i) * 15 + j] = dT * XE2.m[i][j];
        Phi[(12 + i) * 15 + 3 + j] = -dT * dT * XE3.m[i][j];
        Phi[(12 + i) * 15 + 6 + j] = (i == j) ? dT : S(0);
        Phi[(12 + i) * 15 + 9 + j] = -C.m[j][i] * dT * dT / S(2);
      }
      // observability-constraint patch of Phi blocks (0,0),(6,0),(12,0)   msckf.h:116-132
      {
        const S* nx = sState + (s + 1) * SST;
        const Q4<S> qn = ldq(nx);
        const V3<S> vn = ld3(nx + 7), pn = ld3(nx + 13);
        Q4<S> qn0; V3<S> vn0, pn0;
        if (k0 + s == 0) { qn0 = ldq(sNull); vn0 = ld3(sNull + 4); pn0 = ld3(sNull + 7); }
        else { qn0 = ldq(st); vn0 = ld3(st + 7); pn0 = ld3(st + 13); }   // re-anchored after every propagate :139-141
        const M3<S> Rk = q2rot(qn0);
        const M3<S> R00 = mulmt(q2rot(qn), Rk);
        const V3<S> u = mulv(Rk, g);
        const V3<S> sv = (S(1) / dot3(u, u)) * u;
        M3<S> A1, A2;
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { A1.m[i][j] = Phi[(6 + i) * 15 + j]; A2.m[i][j] = Phi[(12 + i) * 15 + j]; }
        const V3<S> w1 = mulv(skew3(vn0 - vn), g);
        const V3<S> w2 = mulv(skew3((dT * vn0) + pn0 - pn), g);
        const V3<S> e1 = mulv(A1, u) - w1, e2 = mulv(A2, u) - w2;
        const S e1v[3] = {e1.x, e1.y, e1.z}, e2v[3] = {e2.x, e2.y, e2.z}, s3[3] = {sv.x, sv.y, sv.z};
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
          Phi[i * 15 + j] = R00.m[i][j];
          Phi[(6 + i) * 15 + j] = A1.m[i][j] - e1v[i] * s3[j];
          Phi[(12 + i) * 15 + j] = A2.m[i][j] - e2v[i] * s3[j];
        }
      }
    } else if (tid >= 64) {
      // waves 1-3 meanwhile: G Q G^T dT = diag(Qw, Qbg, C^T Qa C, Qba, 0) dT of every sample of the group   (calcG :899-902, Q diagonal)
      for (int e = tid - 64; e < G * 32; e += 192) {
        const int s = e >> 5, q = e & 31;
        const S dT = sRd[s * RD_STRIDE + 6];
        S val = 0;
        if (q < 12 && q / 3 != 2) val = sQ[q] * dT;
        else if (q >= 15 && q < 24) {
          const M3<S> C = q2rot(ldq(sState + s * SST));
          const int i = (q - 15) / 3, j = (q - 15) % 3;
          S sm = 0;
          for (int kk = 0; kk < 3; ++kk) sm += C.m[kk][i] * sQ[6 + kk] * C.m[kk][j];
          val = sm * dT;
        }
        sQt[e] = val;
      }
    }
    __syncthreads();
    PR_TICK(2);
    // ---- C: sequential chains over the group on the matrix cores, in registers.  Wave 0 carries P_II <- sym(Phi (P_II +
    // G Q G^T dT) Phi^T) (:134,143), wave 1 carries Phi_total <- Phi Phi_total; both hold their 16 x 16 (zero-padded) matrix
    // in the accumulator layout of the 16x16x4 MFMA, lane (g, c) register r = M[crow(g, r)][c].  Taking the k-slot of MFMA kk,
    // lane group g, to mean k = crow(g, kk), that same register file IS the B operand of M (B[k][c]) and the A operand of M^T
    // (A[c][k]), and one set of four values per lane, aPhi[kk] = Phi[c][crow(g, kk)], is both the A operand of Phi and the B
    // operand of Phi^T.  So with P symmetric:  Y = P Phi^T = mma(A = P regs, B = aPhi),  Phi Y = mma(A = aPhi, B = Y regs),
    // (Phi Y)^T = Y^T Phi^T = mma(A = Y regs, B = aPhi) -- the transpose comes out of the matrix core with the same products
    // summed in the same order, so the symmetrisation of :143 is a register add and both halves get the same bits.  No
    // barrier and no LDS traffic inside the chain besides the four reads of Phi and the process-noise term (table sQt,
    // filled by waves 1-3 during phase B).
    {
      typedef Mfs<S> MF;
      const int g = lane >> 4, c = lane & 15;
      if (w < 2) {
        const S* qt = sQt;
        int qidx[4];   // where this lane's four elements find their process-noise term in a sample's compact table
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = MF::crow(g, r), bi = i / 3;
          qidx[r] = (bi != c / 3 || bi >= 4) ? 31 : (bi == 2 ? 15 + (i - 6) * 3 + (c - 6) : (i == c ? i : 31));
        }
        if (w == 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r) Mreg[r] += qt[qidx[r]];
        }
        for (int s = 0; s < G; ++s) {
          const S* Ph = sPhi + s * 225;
          S aPhi[4];
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) { const int k = MF::crow(g, kk); aPhi[kk] = (k < 15 && c < 15) ? Ph[c * 15 + k] : S(0); }
          typename MF::V zero = {0, 0, 0, 0};
          if (w == 0) {
            typename MF::V Y = zero, Pn = zero, Pt = zero;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) Y = MF::mma(Mreg[kk], aPhi[kk], Y);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) { Pn = MF::mma(aPhi[kk], Y[kk], Pn); Pt = MF::mma(Y[kk], aPhi[kk], Pt); }
            qt += 32;
#pragma unroll
            for (int r = 0; r < 4; ++r) Mreg[r] = (Pn[r] + Pt[r]) / S(2) + (s + 1 < G ? qt[qidx[r]] : S(0));
          } else {
            typename MF::V Tn = zero;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) Tn = MF::mma(aPhi[kk], Mreg[kk], Tn);
            Mreg = Tn;
          }
        }
      }
    }
    __syncthreads();
    if (tid < 16) sState[tid] = sState[G * SST + tid];   // carry the last state of the group to slot 0
    __syncthreads();
    PR_TICK(3);
  }
  // chain results back to LDS for the write-back
  if (w < 2) {
    const int g = lane >> 4, c = lane & 15;
    S* dst = w == 0 ? sPii : sTot;
#pragma unroll
    for (int r = 0; r < 4; ++r) { const int i = Mfs<S>::crow(g, r); if (i < 15 && c < 15) dst[i * 15 + c] = Mreg[r]; }
  }
  __syncthreads();
  // ---- write back: state, nulls (re-anchored to the final state), P_II, P_IC
  if (tid < 16) imu[tid] = sState[tid];
  if (tid < 4) imu[IQN + tid] = sState[tid];
  if (tid >= 4 && tid < 7) imu[IVN + tid - 4] = sState[7 + tid - 4];
  if (tid >= 8 && tid < 11) imu[IPN + tid - 8] = sState[13 + tid - 8];
  for (int e = tid; e < 225; e += 256) { const int i = e / 15, j = e % 15; P[(long)j * ld + i] = sPii[e]; }
  // P_IC <- Phi_total P_IC (and its mirror), 16 camera columns per MFMA tile; the operand columns were fetched at kernel entry
  {
    const int g = lane >> 4, c = lane & 15;
    S aT[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) { const int k = Mfs<S>::crow(g, kk); aT[kk] = (k < 15 && c < 15) ? sTot[c * 15 + k] : S(0); }
#pragma unroll
    for (int t = 0; t < PIC_T; ++t) {
      const int col = (w + 4 * t) * 16 + c;
      if ((w + 4 * t) * 16 >= n) break;
      typename Mfs<S>::V o = {0, 0, 0, 0};
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) o = Mfs<S>::mma(aT[kk], (col < n && Mfs<S>::crow(g, kk) < 15) ? pic[t][kk] : S(0), o);
      if (col < n) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = Mfs<S>::crow(g, r);
          if (i < 15) { P[(long)(15 + col) * ld + i] = o[r]; P[(long)i * ld + 15 + col] = o[r]; }
        }
      }
    }
  }
#ifdef MSCKF_ABLATE
  __syncthreads();
  PR_TICK(4);
  if (tid == 0 && blockIdx.x == 0) { for (int q = 0; q < 5; ++q) atomicAdd(&g_prop_cycles[q], (unsigned long long)pcyc[q]); atomicAdd(&g_prop_cycles[5], 1ull); }
#endif
  if (AUGMENT) {
    // the state and the IMU rows of P just written by this workgroup are read back by other threads of it: work-group scope
    // (one CU, one vector L1: the barrier's fence is enough; a device-scope fence waits for the write-back of all of P_IC)
    __syncthreads();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_aug[];
    augment_body<S>(d, b, tid, reinterpret_cast<S*>(smem_aug));
  }
}

// Drop camera states in place: P <- P[keep, keep] (square_slice / column_slice of the covariance, matrix_utils.h:58-87;
// keep[] ascending).  ONE workgroup of 1024 threads per trajectory sweeps the destination columns in ascending chunks of
// 4 CMAX: all sources of a chunk are loaded into registers (and have returned: s_waitcnt + workgroup barrier), then stored.
// keep[] ascending means source index >= destination index in both directions, so the columns a chunk overwrites are
// sources only of destinations at or before that chunk -- already moved, or held in registers.  No workgroup ever waits for
// another one: the earlier form (16 workgroups per trajectory meeting at a counter barrier before their stores) depended on
// the co-residency of its workgroups, which HIP does not promise, and stored anyway when the bounded wait ran out.
// The keep list is either the host's (d.keep / d.nkeep: pruneEmptyStates, pruneRedundantStates) or "drop the n_drop oldest"
// (drop array of the resident scenario, or a constant); the workgroup also publishes it, compacts cam[] and sets ncam.
template <class S, int RMAX, int CMAX>
__global__ __launch_bounds__(1024) void k_prune_inplace(Dev<S> d, int b0, const int* drop, int drop_const, int use_keep) {
  const int b = b0 + blockIdx.y, tid = threadIdx.x;
  const int n = d.ncam[b];
  int nk, nd = 0;
  if (use_keep) nk = d.nkeep[b];
  else { nd = drop ? drop[blockIdx.y] : drop_const; nd = nd < 0 ? 0 : (nd > n ? n : nd); nk = n - nd; }
  const int* keep = d.keep + (long)b * d.n_cap;
  prune_bookkeeping<S>(d, b, tid, n, nk, nd, use_keep);
  if (nk >= n) return;
  const int Dn = 15 + 6 * nk, ld = d.ld;
  S* P = d.P + (long)b * ld * ld;
  __shared__ int sSrc[256 * RMAX];   // Dn <= ld <= 256 * RMAX (launch_prune picks RMAX from ld)
  for (int i = tid; i < Dn; i += 1024) sSrc[i] = i < 15 ? i : 15 + 6 * (use_keep ? keep[(i - 15) / 6] : nd + (i - 15) / 6) + (i - 15) % 6;
  __syncthreads();
  const int ri = tid & 255, cg = tid >> 8;   // thread = row ri (+ 256 r) of the columns j0 + cg + 4 c
  for (int j0 = 0; j0 < Dn; j0 += 4 * CMAX) {
    S v[CMAX][RMAX];
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
      const int j = j0 + cg + 4 * c;
      if (j < Dn) {
        const S* src = P + (long)sSrc[j] * ld;
#pragma unroll
        for (int r = 0; r < RMAX; ++r) { const int i = ri + r * 256; v[c][r] = i < Dn ? src[sSrc[i]] : S(0); }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every load of the chunk has returned ...
    __syncthreads();                                     // ... in every wavefront, before any of its destinations is written
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
      const int j = j0 + cg + 4 * c;
      if (j < Dn) {
        S* dst = P + (long)j * ld;
#pragma unroll
        for (int r = 0; r < RMAX; ++r) { const int i = ri + r * 256; if (i < Dn) dst[i] = v[c][r]; }
      }
    }
    // the next chunk's sources are columns beyond this chunk's destinations (source >= destination): no barrier needed here
  }
  if (tid == 0) d.ncam[b] = nk;
}

template <class S>
void launch_propagate(const Dev<S>& d, int b0, int nb, const S* readings, long rd_stride, int K, hipStream_t st, bool then_augment) {
  if (nb <= 0) return;
  if (K <= 0) { if (then_augment) launch_augment<S>(d, b0, nb, st); return; }
  if (then_augment) hipLaunchKernelGGL((k_propagate<S, true>), dim3(nb), dim3(256), (size_t)6 * d.ld * sizeof(S), st, d, b0, readings, rd_stride, K);
  else hipLaunchKernelGGL((k_propagate<S, false>), dim3(nb), dim3(256), 0, st, d, b0, readings, rd_stride, K);
}
template <class S>
void launch_augment(const Dev<S>& d, int b0, int nb, hipStream_t st) {
  if (nb <= 0) return;
  hipLaunchKernelGGL(k_augment<S>, dim3(nb), dim3(256), (size_t)6 * d.ld * sizeof(S), st, d, b0);
}
// drop == nullptr && drop_const < 0: the host's keep list (d.keep, d.nkeep); otherwise drop the oldest drop[i] (i = trajectory
// index relative to b0) or drop_const camera states
template <class S>
void launch_prune(const Dev<S>& d, int b0, int nb, hipStream_t st, const int* drop, int drop_const) {
  if (nb <= 0) return;
  const int use_keep = (!drop && drop_const < 0) ? 1 : 0;
  const dim3 grid(1, nb);
  // a chunk holds 4 CMAX columns x ceil(ld / 256) rows per thread in registers (ld <= 400: n_cap <= 63)
  if (d.ld <= 256) hipLaunchKernelGGL((k_prune_inplace<S, 1, 16>), grid, dim3(1024), 0, st, d, b0, drop, drop_const, use_keep);
  else hipLaunchKernelGGL((k_prune_inplace<S, 2, 12>), grid, dim3(1024), 0, st, d, b0, drop, drop_const, use_keep);
}

template void launch_propagate<float>(const Dev<float>&, int, int, const float*, long, int, hipStream_t, bool);
template void launch_propagate<double>(const Dev<double>&, int, int, const double*, long, int, hipStream_t, bool);
template void launch_augment<float>(const Dev<float>&, int, int, hipStream_t);
template void launch_augment<double>(const Dev<double>&, int, int, hipStream_t);
template void launch_prune<float>(const Dev<float>&, int, int, hipStream_t, const int*, int);
template void launch_prune<double>(const Dev<double>&, int, int, hipStream_t, const int*, int);

}  // namespace msckf
