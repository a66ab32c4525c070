// include/msckf_mono/msckf.h -- drop-in shim: msckf_mono::MSCKF<_S> on top of libmsckf_hip.so.
//
// Replaces /root/reference/include/msckf_mono/msckf.h (the header-only Eigen EKF, msckf.h:31-1512) for its
// two callers, src/ros_interface.cpp:80-124 and datasets/asl_msckf.cpp:57-298, without touching them: same
// class name, same public member names and argument types (msckf.h:72-848), by-value getters, no
// exceptions.  All numerics run on the GPU through the C-ABI of include/msckf_hip.h; this file only
// converts argument types.  With Eigen present the argument types are the reference's own
// <msckf_mono/types.h> (kept as is in the reference tree); without Eigen (this build image) the Eigen-free
// stand-ins of pod_types.h are used so that the shim can be compiled and tested here.
//
// Differences a caller can observe (all documented in INTEGRATION.md):
//   * one filter = one batch handle with B = 1; capacities come from MSCKFParams (override with
//     MSCKF_SHIM_N_CAP / MSCKF_SHIM_F_CAP / MSCKF_SHIM_M_CAP at compile time; the track capacity per update also at run
//     time, before the filter is initialized: environment MSCKF_SHIM_F_CAP or MSCKF<S>::setTrackCapacity(n));
//   * Q_imu / initial_imu_covar are read through their diagonals (every caller passes .asDiagonal());
//   * u_var_prime != v_var_prime (EuRoC intrinsics): the reference's R_o_j = A_j^T R_j A_j / R_n = Q_1^T R_o Q_1 construction
//     runs on the device, see include/msckf_hip.h (msckf_hip_set_anisotropic_noise);
//   * capacities: a window or a track list beyond the handle's capacity does not abort the caller (the reference has no
//     error channel): the call is refused, lastError() returns -EOVERFLOW / -E2BIG from then on until initialize();
//   * getCamStates()[i].tracked_feature_ids / getPrunedStates() carry the reference's payloads (time, poses, ids,
//     last_correlated_id) -- asl_msckf.cpp:379-424 reads them;
//   * additive: getCovariance(), lastError().
#ifndef MSCKF_MONO_SHIM_MSCKF_H_
#define MSCKF_MONO_SHIM_MSCKF_H_

#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../msckf_hip.h"

#if defined(__has_include)
#if __has_include(<Eigen/Dense>) && !defined(MSCKF_SHIM_NO_EIGEN)
#define MSCKF_SHIM_EIGEN 1
#endif
#endif

#ifdef MSCKF_SHIM_EIGEN
#include <Eigen/Dense>
#include <Eigen/Geometry>
#include <Eigen/StdVector>
#include <msckf_mono/types.h>   // the reference's own types (unchanged)
#else
#include "pod_types.h"
#endif

#ifndef MSCKF_SHIM_N_CAP
#define MSCKF_SHIM_N_CAP 0
#endif
#ifndef MSCKF_SHIM_F_CAP
#define MSCKF_SHIM_F_CAP 512
#endif
#ifndef MSCKF_SHIM_M_CAP
#define MSCKF_SHIM_M_CAP 0
#endif

namespace msckf_mono {

template <typename _S>
class MSCKF {
 public:
#ifdef MSCKF_SHIM_EIGEN
  using Vec2List = std::vector<Vector2<_S>, Eigen::aligned_allocator<Vector2<_S>>>;
  using Vec3List = std::vector<Vector3<_S>, Eigen::aligned_allocator<Vector3<_S>>>;
  EIGEN_MAKE_ALIGNED_OPERATOR_NEW
#else
  using Vec2List = std::vector<Vector2<_S>>;
  using Vec3List = std::vector<Vector3<_S>>;
#endif

  MSCKF() {}
  ~MSCKF() { if (h_) msckf_hip_destroy(h_); }
  // tracks one update() may hand to marginalize() (the reference has no such limit: its vectors grow): the compile-time
  // default MSCKF_SHIM_F_CAP, the environment variable of the same name, or this setter, read when a filter is initialized
  static int& trackCapacity() {
    static int cap = [] { const char* e = std::getenv("MSCKF_SHIM_F_CAP"); const int v = e ? std::atoi(e) : 0; return v > 0 ? v : (int)MSCKF_SHIM_F_CAP; }();
    return cap;
  }
  static void setTrackCapacity(int n) { if (n > 0) trackCapacity() = n; }
  // value semantics, as the reference object (msckf.h:31-67): a copy owns its own device-side filter
  MSCKF(const MSCKF& o) { copy_from(o); }
  MSCKF& operator=(const MSCKF& o) { if (this != &o) copy_from(o); return *this; }

  // msckf.h:72
  void initialize(const Camera<_S>& camera, const noiseParams<_S>& noise_params,
                  const MSCKFParams<_S>& msckf_params, const imuState<_S>& imu_state) {
    camera_ = camera;
    if (h_) { msckf_hip_destroy(h_); h_ = nullptr; }
    int n_cap = MSCKF_SHIM_N_CAP, m_cap = MSCKF_SHIM_M_CAP;
    // the window may grow past max_cam_states up to ~max_track_length; 63 is the build's limit
    if (n_cap <= 0) n_cap = std::min(63, std::max(msckf_params.max_cam_states, std::min(msckf_params.max_track_length, 60)) + 3);
    if (m_cap <= 0) m_cap = std::min(64, std::max(4, std::min(msckf_params.max_track_length, n_cap)));
    n_cap_ = n_cap; m_cap_ = m_cap; sticky_ = 0;
    f_cap_ = trackCapacity();
    rc_ = msckf_hip_create(1, n_cap, f_cap_, m_cap, sizeof(_S) == 4 ? MSCKF_HIP_F32 : MSCKF_HIP_F64, 0, &h_);
    if (report("create")) return;
    double cam[12] = {(double)camera.c_u, (double)camera.c_v, (double)camera.f_u, (double)camera.f_v, (double)camera.b,
                      (double)camera.q_CI.w(), (double)camera.q_CI.x(), (double)camera.q_CI.y(), (double)camera.q_CI.z(),
                      (double)camera.p_C_I(0), (double)camera.p_C_I(1), (double)camera.p_C_I(2)};
    double noise[29];
    noise[0] = (double)noise_params.u_var_prime; noise[1] = (double)noise_params.v_var_prime;
#ifdef MSCKF_SHIM_EIGEN
    for (int i = 0; i < 12; ++i) noise[2 + i] = (double)noise_params.Q_imu(i, i);
    for (int i = 0; i < 15; ++i) noise[14 + i] = (double)noise_params.initial_imu_covar(i, i);
#else
    for (int i = 0; i < 12; ++i) noise[2 + i] = (double)noise_params.Q_imu_diag[i];
    for (int i = 0; i < 15; ++i) noise[14 + i] = (double)noise_params.initial_imu_covar_diag[i];
#endif
    double prm[8] = {(double)msckf_params.max_gn_cost_norm, (double)msckf_params.min_rcond, (double)msckf_params.translation_threshold,
                     (double)msckf_params.redundancy_angle_thresh, (double)msckf_params.redundancy_distance_thresh,
                     (double)msckf_params.min_track_length, (double)msckf_params.max_track_length, (double)msckf_params.max_cam_states};
    double imu[29];
    pack_imu(imu_state, imu);
    rc_ = msckf_hip_initialize(h_, 0, cam, noise, prm, imu);
    report("initialize");
  }
  // msckf.h:101
  void propagate(imuReading<_S>& m) {
    double rd[7] = {(double)m.omega(0), (double)m.omega(1), (double)m.omega(2), (double)m.a(0), (double)m.a(1), (double)m.a(2), (double)m.dT};
    rc_ = msckf_hip_propagate(h_, 0, rd, 1);
    report("propagate");
  }
  // msckf.h:148
  void augmentState(const int& state_id, const _S& time) { rc_ = msckf_hip_augment_state(h_, 0, state_id, (double)time); report("augmentState"); }
  // msckf.h:215
  void update(const Vec2List& measurements, const std::vector<size_t>& feature_ids) {
    flatten(measurements, feature_ids);
    rc_ = msckf_hip_update(h_, 0, buf_.data(), ids_.data(), (int)ids_.size());
    report("update");
  }
  // msckf.h:302
  void addFeatures(const Vec2List& features, const std::vector<size_t>& feature_ids) {
    flatten(features, feature_ids);
    rc_ = msckf_hip_add_features(h_, 0, buf_.data(), ids_.data(), (int)ids_.size());
    report("addFeatures");
  }
  void marginalize() { rc_ = msckf_hip_marginalize(h_, 0); report("marginalize"); }                       // :336
  void pruneRedundantStates() { rc_ = msckf_hip_prune_redundant_states(h_, 0); report("pruneRedundantStates"); }                 // :453
  void pruneEmptyStates() { rc_ = msckf_hip_prune_empty_states(h_, 0); report("pruneEmptyStates"); }      // :685
  void finish() { rc_ = msckf_hip_finish(h_, 0); report("finish"); }                                      // :765

  // getters, by value (msckf.h:810-848)
  inline size_t getNumCamStates() { int n = msckf_hip_get_num_cam_states(h_, 0); return n < 0 ? 0 : (size_t)n; }
  inline imuState<_S> getImuState() {
    double x[29] = {0};
    rc_ = msckf_hip_get_imu_state(h_, 0, x);
    imuState<_S> s;
    setq(s.q_IG, x); set3(s.b_g, x + 4); set3(s.v_I_G, x + 7); set3(s.b_a, x + 10); set3(s.p_I_G, x + 13); set3(s.g, x + 16);
    setq(s.q_IG_null, x + 19); set3(s.v_I_G_null, x + 23); set3(s.p_I_G_null, x + 26);
    return s;
  }
  inline Vec3List getMap() {
    std::vector<double> xyz(3 * (size_t)f_cap_);
    int n = msckf_hip_get_map(h_, 0, xyz.data(), f_cap_);
    Vec3List out;
    for (int i = 0; i < n; ++i) { Vector3<_S> p; set3(p, xyz.data() + 3 * i); out.push_back(p); }
    return out;
  }
  inline Camera<_S> getCamera() { return camera_; }
  inline std::vector<camState<_S>> getCamStates() const {
    const int cap = std::max(msckf_hip_get_num_cam_states(h_, 0), 1);     // the count first: any configured n_cap fits
    std::vector<double> c(7 * (size_t)cap), tm((size_t)cap); std::vector<int> ids((size_t)cap), lc((size_t)cap), nt((size_t)cap);
    const int n = msckf_hip_get_cam_states(h_, 0, c.data(), ids.data(), cap);
    const int nm = msckf_hip_get_cam_meta(h_, 0, tm.data(), nt.data(), lc.data(), cap);
    if (n < 0 || nm < 0) std::fprintf(stderr, "msckf_mono shim: getCamStates failed: %s\n", msckf_hip_last_error());
    std::vector<camState<_S>> out;
    std::vector<uint64_t> fid;
    for (int i = 0; i < n; ++i) {
      camState<_S> s;
      setq(s.q_CG, c.data() + 7 * i); set3(s.p_C_G, c.data() + 7 * i + 4);
      s.state_id = ids[i];
      s.time = i < nm ? (_S)tm[i] : (_S)0;
      s.last_correlated_id = i < nm ? lc[i] : -1;
      if (i < nm && nt[i] > 0) {                     // tracked_feature_ids (types.h:66), read at asl_msckf.cpp:388
        fid.resize((size_t)nt[i]);
        const int k = msckf_hip_get_tracked_feature_ids(h_, 0, i, fid.data(), nt[i]);
        for (int j = 0; j < k; ++j) s.tracked_feature_ids.push_back((size_t)fid[(size_t)j]);
      }
      out.push_back(s);
    }
    return out;
  }
  inline camState<_S> getCamState(size_t i) { return getCamStates()[i]; }
  inline std::vector<camState<_S>> getPrunedStates() {   // sorted by state_id, poses as they were when pruned (:840-848)
    int n = msckf_hip_get_pruned_states(h_, 0, nullptr, nullptr, nullptr, nullptr, 1 << 30);
    if (n < 0) n = 0;
    std::vector<double> c(7 * (size_t)n + 1), tm((size_t)n + 1); std::vector<int> ids((size_t)n + 1), lc((size_t)n + 1);
    n = msckf_hip_get_pruned_states(h_, 0, c.data(), tm.data(), ids.data(), lc.data(), n);
    std::vector<camState<_S>> out;
    for (int i = 0; i < n; ++i) {
      camState<_S> s;
      setq(s.q_CG, c.data() + 7 * i); set3(s.p_C_G, c.data() + 7 * i + 4);
      s.state_id = ids[(size_t)i]; s.time = (_S)tm[(size_t)i]; s.last_correlated_id = lc[(size_t)i];
      out.push_back(s);
    }
    return out;
  }
  // additive accessors
  std::vector<double> getCovariance() {
    const int D = 15 + 6 * (int)getNumCamStates();
    std::vector<double> P((size_t)D * D);
    rc_ = msckf_hip_get_covariance(h_, 0, P.data(), D);
    return P;
  }
  // 0, or the -errno code of the last failed call; a refused call (capacity: -EOVERFLOW, -E2BIG) and the device-side sticky
  // flags (window beyond n_cap, covariance no longer positive definite: msckf_hip_last_stats) stay reported until initialize()
  int lastError() {
    if (sticky_) return sticky_;
    if (h_) {
      int flags = 0;
      if (msckf_hip_get_error_flags(h_, 0, &flags) == 0 && flags) { int st[7]; sticky_ = msckf_hip_last_stats(h_, 0, st); }
    }
    return sticky_ ? sticky_ : rc_;
  }

 private:
  msckf_hip_handle h_ = nullptr;
  Camera<_S> camera_;
  int rc_ = 0, sticky_ = 0, n_cap_ = 0, m_cap_ = 0, f_cap_ = MSCKF_SHIM_F_CAP;
  std::vector<double> buf_;
  std::vector<uint64_t> ids_;

  bool report(const char* what) {
    if (rc_ < 0) {
      std::fprintf(stderr, "[msckf_hip] %s failed (%d): %s\n", what, rc_, msckf_hip_last_error());
      if (!sticky_) sticky_ = rc_;     // the filter is no longer what the caller thinks it is: keep saying so
      return true;
    }
    return false;
  }
  void copy_from(const MSCKF& o) {
    if (h_) { msckf_hip_destroy(h_); h_ = nullptr; }
    camera_ = o.camera_; rc_ = o.rc_; sticky_ = o.sticky_; n_cap_ = o.n_cap_; m_cap_ = o.m_cap_; f_cap_ = o.f_cap_;
    if (!o.h_) return;
    rc_ = msckf_hip_create(1, n_cap_, f_cap_, m_cap_, sizeof(_S) == 4 ? MSCKF_HIP_F32 : MSCKF_HIP_F64, 0, &h_);
    if (report("create (copy)")) return;
    rc_ = msckf_hip_copy_state(h_, o.h_);
    report("copy_state");
  }
  void flatten(const Vec2List& m, const std::vector<size_t>& ids) {
    buf_.resize(2 * m.size()); ids_.resize(ids.size());
    for (size_t i = 0; i < m.size(); ++i) { buf_[2 * i] = (double)m[i](0); buf_[2 * i + 1] = (double)m[i](1); }
    for (size_t i = 0; i < ids.size(); ++i) ids_[i] = (uint64_t)ids[i];
  }
  template <class V> static void set3(V& v, const double* p) { v(0) = (_S)p[0]; v(1) = (_S)p[1]; v(2) = (_S)p[2]; }
  template <class Q> static void setq(Q& q, const double* p) { q.w() = (_S)p[0]; q.x() = (_S)p[1]; q.y() = (_S)p[2]; q.z() = (_S)p[3]; }
  static void pack_imu(const imuState<_S>& s, double* x) {
    x[0] = s.q_IG.w(); x[1] = s.q_IG.x(); x[2] = s.q_IG.y(); x[3] = s.q_IG.z();
    for (int i = 0; i < 3; ++i) { x[4 + i] = s.b_g(i); x[7 + i] = s.v_I_G(i); x[10 + i] = s.b_a(i); x[13 + i] = s.p_I_G(i); x[16 + i] = s.g(i); }
    for (int i = 0; i < 4; ++i) x[19 + i] = x[i];
    for (int i = 0; i < 3; ++i) { x[23 + i] = x[7 + i]; x[26 + i] = x[13 + i]; }
  }
};

}  // namespace msckf_mono
#endif
