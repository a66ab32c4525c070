// bench_src/cfg2_driver.cpp -- bench.py --config cfg2 (BASELINE.json configs[1]: 10-camera window, 50 tracks, double, ONE
// trajectory): the filter driven exactly as the reference's callers drive it (datasets/asl_msckf.cpp:227-296,
// src/ros_interface.cpp:93,111-116) through the drop-in shim include/msckf_mono/msckf.h -- one propagate() per IMU sample with
// the by-value getImuState() after it (asl_msckf.cpp:231-233), then augmentState / update / addFeatures / marginalize /
// pruneEmptyStates per image and the getters the runner reads -- with the reference's own host wall-clock stage timer
// (asl_msckf.cpp:229-296 StageTiming) around each stage.  Compiled by bench.py against libmsckf_hip.so (Eigen-free branch).
//   stdin : cam12 noise29 params8 imu29, n_frames, n_warm, then per frame: K, K*7 readings, n_cur (x y id)*, n_new (x y id)*
//   stdout: one JSON object: per-stage mean microseconds over the timed frames, per-frame microseconds, final state
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <iostream>
#include <vector>

#include "msckf_mono/msckf.h"

using namespace msckf_mono;
typedef double S;
typedef std::chrono::steady_clock Clock;
static double us(Clock::time_point a, Clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); }

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
  for (int i = 0; i < 12; ++i) np.Q_imu_diag[i] = noise[2 + i];
  for (int i = 0; i < 15; ++i) np.initial_imu_covar_diag[i] = noise[14 + i];
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
  int nframes, nwarm;
  std::cin >> nframes >> nwarm;
  // read everything first: parsing is not part of what is timed
  struct Frame { std::vector<imuReading<S>> rd; MSCKF<S>::Vec2List cur, fresh; std::vector<size_t> cur_ids, new_ids; };
  std::vector<Frame> frames(nframes);
  for (auto& f : frames) {
    int K; std::cin >> K; f.rd.resize(K);
    for (auto& r : f.rd) { for (int i = 0; i < 3; ++i) std::cin >> r.omega(i); for (int i = 0; i < 3; ++i) std::cin >> r.a(i); std::cin >> r.dT; }
    int n; std::cin >> n;
    for (int i = 0; i < n; ++i) { Vector2<S> z; size_t id; std::cin >> z(0) >> z(1) >> id; f.cur.push_back(z); f.cur_ids.push_back(id); }
    std::cin >> n;
    for (int i = 0; i < n; ++i) { Vector2<S> z; size_t id; std::cin >> z(0) >> z(1) >> id; f.fresh.push_back(z); f.new_ids.push_back(id); }
  }
  const char* names[7] = {"imu_prop", "msckf_augment_state", "msckf_update", "msckf_add_features", "msckf_marginalize", "msckf_prune_empty_states", "read_state"};
  double acc[7] = {0, 0, 0, 0, 0, 0, 0};
  std::vector<double> per_frame;
  int state_k = 0, timed = 0;
  imuState<S> out;
  for (int fi = 0; fi < nframes; ++fi) {
    Frame& f = frames[fi];
    const bool t = fi >= nwarm;
    double s[7] = {0, 0, 0, 0, 0, 0, 0};
    Clock::time_point f0 = Clock::now(), a, b;
    a = Clock::now();
    for (auto& r : f.rd) { state_k++; msckf.propagate(r); out = msckf.getImuState(); }     // asl_msckf.cpp:227-233
    b = Clock::now(); s[0] = us(a, b); a = b;
    msckf.augmentState(state_k, (S)fi);  b = Clock::now(); s[1] = us(a, b); a = b;          // :269
    msckf.update(f.cur, f.cur_ids);      b = Clock::now(); s[2] = us(a, b); a = b;          // :274
    msckf.addFeatures(f.fresh, f.new_ids); b = Clock::now(); s[3] = us(a, b); a = b;        // :279
    msckf.marginalize();                 b = Clock::now(); s[4] = us(a, b); a = b;          // :284
    msckf.pruneEmptyStates();            b = Clock::now(); s[5] = us(a, b); a = b;          // :294
    out = msckf.getImuState();                                                             // the runner publishes the state per image (:339-357)
    const size_t ncs = msckf.getNumCamStates(); (void)ncs;
    b = Clock::now(); s[6] = us(a, b);
    if (msckf.lastError()) return 3;
    if (t) { for (int i = 0; i < 7; ++i) acc[i] += s[i]; per_frame.push_back(us(f0, b)); ++timed; }
  }
  std::printf("{\"timed_frames\": %d, \"stage_us\": {", timed);
  for (int i = 0; i < 7; ++i) std::printf("%s\"%s\": %.3f", i ? ", " : "", names[i], acc[i] / std::max(timed, 1));
  std::printf("}, \"frame_us\": [");
  for (size_t i = 0; i < per_frame.size(); ++i) std::printf("%s%.3f", i ? ", " : "", per_frame[i]);
  std::printf("], \"imu\": [%.17g, %.17g, %.17g, %.17g", out.q_IG.w(), out.q_IG.x(), out.q_IG.y(), out.q_IG.z());
  for (int i = 0; i < 3; ++i) std::printf(", %.17g", out.b_g(i));
  for (int i = 0; i < 3; ++i) std::printf(", %.17g", out.v_I_G(i));
  for (int i = 0; i < 3; ++i) std::printf(", %.17g", out.b_a(i));
  for (int i = 0; i < 3; ++i) std::printf(", %.17g", out.p_I_G(i));
  std::vector<double> P = msckf.getCovariance();
  double tr = 0; const int D = 15 + 6 * (int)msckf.getNumCamStates();
  for (int i = 0; i < D; ++i) tr += P[(size_t)i * D + i];
  std::printf("], \"n_cam\": %zu, \"trace_P\": %.17g}\n", msckf.getNumCamStates(), tr);
  return 0;
}
