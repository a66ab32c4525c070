// tests/cpp/shim_eigen_check.cpp -- compile-only check of the EIGEN branch of the drop-in shim
// (include/msckf_mono/msckf.h with the reference's own <msckf_mono/types.h>): every public member of
// msckf_mono::MSCKF<float> is instantiated with the argument types the reference's callers pass
// (src/ros_interface.cpp:80-124, datasets/asl_msckf.cpp:57-298).  Eigen is not installed in this image, so the test
// compiles against the minimal Eigen surface of oracle/ref_shim (test infrastructure).
#include <vector>

#include "msckf_mono/msckf.h"

#ifndef MSCKF_SHIM_EIGEN
#error "Eigen branch not selected"
#endif

using namespace msckf_mono;

int use_all_members() {
  Camera<float> camera; noiseParams<float> noise; MSCKFParams<float> params; imuState<float> st;
  camera.q_CI = Quaternion<float>(1, 0, 0, 0);
  noise.Q_imu.setZero(); noise.initial_imu_covar.setZero();
  params.max_cam_states = 30; params.max_track_length = 50; params.min_track_length = 3;
  MSCKF<float> msckf;
  msckf.initialize(camera, noise, params, st);
  imuReading<float> rd; rd.dT = 0.005f;
  msckf.propagate(rd);
  msckf.augmentState(1, 0.05f);
  std::vector<Vector2<float>, Eigen::aligned_allocator<Vector2<float>>> feats;
  std::vector<size_t> ids;
  msckf.update(feats, ids);
  msckf.addFeatures(feats, ids);
  msckf.marginalize();
  msckf.pruneRedundantStates();
  msckf.pruneEmptyStates();
  imuState<float> out = msckf.getImuState();
  auto map = msckf.getMap();
  auto cams = msckf.getCamStates();
  auto pruned = msckf.getPrunedStates();
  Camera<float> c2 = msckf.getCamera();
  size_t n = msckf.getNumCamStates();
  for (const auto& ci : pruned) {                       // asl_msckf.cpp:409-424
    Quaternion<float> q = ci.q_CG.inverse();
    (void)q; (void)ci.p_C_G[0]; (void)ci.time;
  }
  for (const auto& cs : cams) { (void)cs.time; (void)cs.tracked_feature_ids.size(); }   // asl_msckf.cpp:384-388
  msckf.finish();
  (void)out; (void)map; (void)c2;
  return (int)n;
}
