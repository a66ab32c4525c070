// pod_types.h -- Eigen-free stand-ins for the argument types of msckf_mono::MSCKF<_S>.
//
// Only used when <Eigen/Dense> is NOT available (this build image has no Eigen): the shim in
// include/msckf_mono/msckf.h then takes these structs instead of the reference's Eigen-based ones
// (/root/reference/include/msckf_mono/types.h:48-126).  Member names and accessors (x(), y(), z(), w(),
// operator()(i)) match what the reference's callers use, so code written against either compiles.
#ifndef MSCKF_MONO_POD_TYPES_H_
#define MSCKF_MONO_POD_TYPES_H_

#include <cstddef>
#include <vector>

namespace msckf_mono {

template <typename S, int N>
struct VecN {
  S v[N] = {};
  S& operator()(int i) { return v[i]; }
  const S& operator()(int i) const { return v[i]; }
  S& operator[](int i) { return v[i]; }
  const S& operator[](int i) const { return v[i]; }
  S& x() { return v[0]; }
  S& y() { return v[1]; }
  S& z() { static_assert(N > 2, "no z"); return v[2]; }
  const S& x() const { return v[0]; }
  const S& y() const { return v[1]; }
  const S& z() const { static_assert(N > 2, "no z"); return v[2]; }
};
template <typename S> using Vector2 = VecN<S, 2>;
template <typename S> using Vector3 = VecN<S, 3>;
template <typename S> using Point = Vector3<S>;
template <typename S> using GyroscopeReading = Vector3<S>;
template <typename S> using AccelerometerReading = Vector3<S>;

template <typename S>
struct Quaternion {  // Hamilton, same accessor names as Eigen::Quaternion
  S w_ = 1, x_ = 0, y_ = 0, z_ = 0;
  Quaternion() {}
  Quaternion(S w, S x, S y, S z) : w_(w), x_(x), y_(y), z_(z) {}
  S& w() { return w_; } S& x() { return x_; } S& y() { return y_; } S& z() { return z_; }
  const S& w() const { return w_; } const S& x() const { return x_; } const S& y() const { return y_; } const S& z() const { return z_; }
};

template <typename S> struct DiagN { std::vector<S> d; };   // diagonal of Q_imu / initial_imu_covar

template <typename S> struct Camera { S c_u = 0, c_v = 0, f_u = 0, f_v = 0, b = 0; Quaternion<S> q_CI; Point<S> p_C_I; };
template <typename S> struct camState {
  Point<S> p_C_G; Quaternion<S> q_CG; S time = 0; int state_id = 0; int last_correlated_id = -1;
  std::vector<size_t> tracked_feature_ids;
};
template <typename S> struct imuState {
  Point<S> p_I_G, p_I_G_null; Vector3<S> v_I_G, b_g, b_a, g, v_I_G_null; Quaternion<S> q_IG, q_IG_null;
};
template <typename S> struct imuReading { GyroscopeReading<S> omega; AccelerometerReading<S> a; S dT = 0; };
template <typename S> struct noiseParams {
  S u_var_prime = 0, v_var_prime = 0;
  S Q_imu_diag[12] = {};            // the reference's 12x12 Q_imu is diagonal in every caller (asl_msckf.cpp:86-90)
  S initial_imu_covar_diag[15] = {};
};
template <typename S> struct MSCKFParams {
  S max_gn_cost_norm = 0, min_rcond = 0, translation_threshold = 0;
  S redundancy_angle_thresh = 0, redundancy_distance_thresh = 0;
  int min_track_length = 0, max_track_length = 0, max_cam_states = 0;
};

}  // namespace msckf_mono
#endif
