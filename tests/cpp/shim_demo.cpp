// tests/cpp/shim_demo.cpp -- compiles the drop-in shim (include/msckf_mono/msckf.h) and drives it with the call
// sequence of datasets/asl_msckf.cpp:227-294 on a small scene; prints the state so the pytest wrapper can compare it
// with the oracle fed the same numbers (read from stdin).  Two builds: WITHOUT Eigen (pod_types.h stand-ins, compiled by
// the test itself) and the branch a maintainer builds -- <Eigen/Dense> + the reference's own <msckf_mono/types.h>, against
// oracle/ref_shim's Eigen surface (tests/cpp/Makefile, built where /root/reference exists, the binary travels).  A copy of
// the filter taken mid-run (MSCKF is copyable, msckf.h:31-67) finishes the run as well and must print the same state.
//   input : cam12 noise29 params8 imu29, then frames: "F K" K*7 readings, n_cur (x y id)*, n_new (x y id)*
#include <cstdio>
#include <iostream>
#include <vector>

#include "msckf_mono/msckf.h"

using namespace msckf_mono;
typedef double S;

int main() {
  double cam[12], noise[29], prm[8], imu[29];
  for (double& v : cam) std::cin >> v;
  for (double& v : noise) std::cin >> v;
  for (double& v : prm) std::cin >> v;
  for (double& v : imu) std::cin >> v;
  Camera<S> camera;
  camera.c_u = cam[0]; camera.c_v = cam[1]; camera.f_u = cam[2]; camera.f_v = cam[3]; camera.b = cam[4];
  camera.q_CI = Quaternion<S>(cam[5], cam[6], cam[7], cam[8]);
  for (int i = 0; i < 3; ++i) camera.p_C_I(i) = cam[9 + i];
  noiseParams<S> np;
  np.u_var_prime = noise[0]; np.v_var_prime = noise[1];
#ifdef MSCKF_SHIM_EIGEN   // the reference's own noiseParams (types.h:86-92): dense matrices, diagonal in every caller
  np.Q_imu.setZero(); np.initial_imu_covar.setZero();
  for (int i = 0; i < 12; ++i) np.Q_imu(i, i) = noise[2 + i];
  for (int i = 0; i < 15; ++i) np.initial_imu_covar(i, i) = noise[14 + i];
#else
  for (int i = 0; i < 12; ++i) np.Q_imu_diag[i] = noise[2 + i];
  for (int i = 0; i < 15; ++i) np.initial_imu_covar_diag[i] = noise[14 + i];
#endif
  MSCKFParams<S> mp;
  mp.max_gn_cost_norm = prm[0]; mp.min_rcond = prm[1]; mp.translation_threshold = prm[2];
  mp.redundancy_angle_thresh = prm[3]; mp.redundancy_distance_thresh = prm[4];
  mp.min_track_length = (int)prm[5]; mp.max_track_length = (int)prm[6]; mp.max_cam_states = (int)prm[7];
  imuState<S> st;
  st.q_IG = Quaternion<S>(imu[0], imu[1], imu[2], imu[3]);
  for (int i = 0; i < 3; ++i) { st.b_g(i) = imu[4 + i]; st.v_I_G(i) = imu[7 + i]; st.b_a(i) = imu[10 + i]; st.p_I_G(i) = imu[13 + i]; st.g(i) = imu[16 + i]; }

  MSCKF<S> msckf;
  msckf.initialize(camera, np, mp, st);
  if (msckf.lastError()) return 2;
  int nframes;
  std::cin >> nframes;
  int state_k = 0;
  MSCKF<S> twin;                                   // assigned from msckf half-way, then fed the same calls
  bool have_twin = false;
  for (int f = 0; f < nframes; ++f) {
    if (f == nframes / 2) { twin = msckf; have_twin = true; if (twin.lastError()) return 4; }
    int K; std::cin >> K;
    for (int k = 0; k < K; ++k) {
      imuReading<S> rd;
      for (int i = 0; i < 3; ++i) std::cin >> rd.omega(i);
      for (int i = 0; i < 3; ++i) std::cin >> rd.a(i);
      std::cin >> rd.dT;
      state_k++;                                   // asl_msckf.cpp:227
      msckf.propagate(rd);                         // :233
      if (have_twin) twin.propagate(rd);
    }
    MSCKF<S>::Vec2List cur, fresh; std::vector<size_t> cur_ids, new_ids;
    int n; std::cin >> n;
    for (int i = 0; i < n; ++i) { Vector2<S> z; size_t id; std::cin >> z(0) >> z(1) >> id; cur.push_back(z); cur_ids.push_back(id); }
    std::cin >> n;
    for (int i = 0; i < n; ++i) { Vector2<S> z; size_t id; std::cin >> z(0) >> z(1) >> id; fresh.push_back(z); new_ids.push_back(id); }
    msckf.augmentState(state_k, (S)f);             // :269
    msckf.update(cur, cur_ids);                    // :274
    msckf.addFeatures(fresh, new_ids);             // :279
    msckf.marginalize();                           // :284
    msckf.pruneEmptyStates();                      // :294
    if (msckf.lastError()) return 3;
    if (have_twin) {
      twin.augmentState(state_k, (S)f); twin.update(cur, cur_ids); twin.addFeatures(fresh, new_ids); twin.marginalize(); twin.pruneEmptyStates();
      if (twin.lastError()) return 5;
    }
  }
  {   // the copy ran the second half on its own device-side filter: same state, bit for bit
    imuState<S> a = msckf.getImuState(), b = twin.getImuState();
    bool same = a.q_IG.w() == b.q_IG.w() && a.q_IG.x() == b.q_IG.x() && msckf.getNumCamStates() == twin.getNumCamStates();
    for (int i = 0; i < 3; ++i) same = same && a.p_I_G(i) == b.p_I_G(i) && a.v_I_G(i) == b.v_I_G(i) && a.b_g(i) == b.b_g(i);
    std::vector<double> Pa = msckf.getCovariance(), Pb = twin.getCovariance();
    same = same && Pa == Pb;
    if (!same) { std::fprintf(stderr, "copy of the filter diverged from the original\n"); return 6; }
  }
  imuState<S> out = msckf.getImuState();
  std::printf("%.17g %.17g %.17g %.17g ", out.q_IG.w(), out.q_IG.x(), out.q_IG.y(), out.q_IG.z());
  for (int i = 0; i < 3; ++i) std::printf("%.17g ", out.b_g(i));
  for (int i = 0; i < 3; ++i) std::printf("%.17g ", out.v_I_G(i));
  for (int i = 0; i < 3; ++i) std::printf("%.17g ", out.b_a(i));
  for (int i = 0; i < 3; ++i) std::printf("%.17g ", out.p_I_G(i));
  std::printf("\n%zu %zu\n", msckf.getNumCamStates(), msckf.getMap().size());
  std::vector<double> P = msckf.getCovariance();
  double tr = 0; const int D = 15 + 6 * (int)msckf.getNumCamStates();
  for (int i = 0; i < D; ++i) tr += P[(size_t)i * D + i];
  std::printf("%.17g\n", tr);
  // what asl_msckf.cpp:379-424 reads from the getters: cam-state ids/times/tracked counts, pruned states' poses
  std::vector<camState<S>> cs = msckf.getCamStates();
  std::printf("%zu", cs.size());
  for (const auto& c : cs) std::printf(" %d %.17g %zu %d", c.state_id, (double)c.time, c.tracked_feature_ids.size(), c.last_correlated_id);
  std::printf("\n");
  std::vector<camState<S>> ps = msckf.getPrunedStates();
  std::printf("%zu", ps.size());
  for (const auto& c : ps)
    std::printf(" %d %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g", c.state_id, (double)c.time, (double)c.q_CG.w(), (double)c.q_CG.x(),
                (double)c.q_CG.y(), (double)c.q_CG.z(), (double)c.p_C_G(0), (double)c.p_C_G(1), (double)c.p_C_G(2));
  std::printf("\n");
  return 0;
}
