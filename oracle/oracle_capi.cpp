// oracle/oracle_capi.cpp -- TEST INFRASTRUCTURE ONLY.
// C-ABI shim around oracle::MSCKF<float|double> so that tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg can drive the CPU restatement through ctypes.  All scalars cross the boundary as
// double and are narrowed to the oracle's scalar type inside.  Built by oracle/Makefile into
// oracle/liboracle.so; the product library never links it.
#include <chrono>
#include <cstdint>
#include <cstring>
#include <thread>

#include "msckf_oracle.hpp"

namespace {
using namespace oracle;

struct Base {
  virtual ~Base() {}
  virtual void initialize(const double* cam, const double* noise, const double* params, const double* imu) = 0;
  virtual void propagate(const double* rd, int K) = 0;
  virtual void augment(int id, double t) = 0;
  virtual void update(const double* meas, const uint64_t* ids, int n) = 0;
  virtual void add_features(const double* meas, const uint64_t* ids, int n) = 0;
  virtual void marginalize() = 0;
  virtual void prune_redundant() = 0;
  virtual void prune_empty() = 0;
  virtual void finish() = 0;
  virtual int num_cam_states() = 0;
  virtual void get_imu_state(double* out) = 0;
  virtual void set_imu_state(const double* in) = 0;
  virtual void get_cam_states(double* out, int* ids) = 0;
  virtual int get_cam_meta(double* time, int* ntracked, int* last_corr, int cap) = 0;
  virtual int pruned_states(double* out9, int cap) = 0;
  virtual void set_cam_pose(int i, const double* qp) = 0;
  virtual void get_covariance(double* P) = 0;
  virtual void set_covariance(const double* P, int D) = 0;
  virtual void set_tracks(int F, const int* M, const int* slots, const double* obs) = 0;
  virtual int get_tracks(int* M, int* slots, double* obs, int cap_f, int cap_m) = 0;
  virtual void drop_oldest(int n) = 0;
  virtual void last_stats(int* out) = 0;
  virtual int last_tracks(double* out, int cap) = 0;
  virtual int last_deltax(double* out, int cap) = 0;
  virtual int map_points(double* out, int cap) = 0;
  virtual int pruned_ids(int* out, int cap) = 0;
  virtual long num_residualized() = 0;
  virtual void set_num_residualized(long n) = 0;
  virtual void set_mode(int m) = 0;
  virtual Base* clone() = 0;
  virtual void set_whiten(int w) = 0;
  virtual void set_colpiv(int w) = 0;
  virtual void set_tiny(double t) = 0;
  virtual void set_capture(int on) = 0;
  virtual int last_matrix(int which, double* out, long cap, int* cols) = 0;
  virtual int last_track_inputs(int* M, int* pass, int* slots, double* hx, double* r, int cap_f, int cap_m) = 0;
};

template <class S>
struct Impl : Base {
  MSCKF<S> f;
  explicit Impl(int mode) { f.mode = (Mode)mode; }
  static V3<S> v3(const double* p) { return {S(p[0]), S(p[1]), S(p[2])}; }
  static Quat<S> q4(const double* p) { return {S(p[0]), S(p[1]), S(p[2]), S(p[3])}; }
  void initialize(const double* cam, const double* noise, const double* params, const double* imu) override {
    Camera<S> c; c.c_u = S(cam[0]); c.c_v = S(cam[1]); c.f_u = S(cam[2]); c.f_v = S(cam[3]); c.b = S(cam[4]);
    c.q_CI = q4(cam + 5); c.p_C_I = v3(cam + 9);
    NoiseParams<S> n; n.u_var_prime = S(noise[0]); n.v_var_prime = S(noise[1]);
    for (int i = 0; i < 12; ++i) n.Q_imu(i, i) = S(noise[2 + i]);
    for (int i = 0; i < 15; ++i) n.initial_imu_covar(i, i) = S(noise[14 + i]);
    MSCKFParams<S> p; p.max_gn_cost_norm = S(params[0]); p.min_rcond = S(params[1]); p.translation_threshold = S(params[2]);
    p.redundancy_angle_thresh = S(params[3]); p.redundancy_distance_thresh = S(params[4]);
    p.min_track_length = (int)params[5]; p.max_track_length = (int)params[6]; p.max_cam_states = (int)params[7];
    ImuState<S> s; unpack_imu(imu, s);
    f.initialize(c, n, p, s);
  }
  // imu layout (29 doubles): q_IG(4) b_g(3) v(3) b_a(3) p(3) g(3) q_null(4) v_null(3) p_null(3)
  static void unpack_imu(const double* x, ImuState<S>& s) {
    s.q_IG = q4(x); s.b_g = v3(x + 4); s.v_I_G = v3(x + 7); s.b_a = v3(x + 10); s.p_I_G = v3(x + 13); s.g = v3(x + 16);
    s.q_IG_null = q4(x + 19); s.v_I_G_null = v3(x + 23); s.p_I_G_null = v3(x + 26);
  }
  static void pack3(double* o, V3<S> v) { o[0] = v.x; o[1] = v.y; o[2] = v.z; }
  static void pack4(double* o, Quat<S> q) { o[0] = q.w; o[1] = q.x; o[2] = q.y; o[3] = q.z; }
  void propagate(const double* rd, int K) override {
    for (int k = 0; k < K; ++k) { ImuReading<S> m; m.omega = v3(rd + 7 * k); m.a = v3(rd + 7 * k + 3); m.dT = S(rd[7 * k + 6]); f.propagate(m); }
  }
  void augment(int id, double t) override { f.augmentState(id, S(t)); }
  static void conv(const double* meas, const uint64_t* ids, int n, std::vector<V2<S>>& m, std::vector<size_t>& i) {
    m.resize(n); i.resize(n);
    for (int k = 0; k < n; ++k) { m[k] = {S(meas[2 * k]), S(meas[2 * k + 1])}; i[k] = (size_t)ids[k]; }
  }
  void update(const double* meas, const uint64_t* ids, int n) override { std::vector<V2<S>> m; std::vector<size_t> i; conv(meas, ids, n, m, i); f.update(m, i); }
  void add_features(const double* meas, const uint64_t* ids, int n) override { std::vector<V2<S>> m; std::vector<size_t> i; conv(meas, ids, n, m, i); f.addFeatures(m, i); }
  void marginalize() override { f.marginalize(); }
  void prune_redundant() override { f.pruneRedundantStates(); }
  void prune_empty() override { f.pruneEmptyStates(); }
  void finish() override { f.finish(); }
  int num_cam_states() override { return (int)f.getNumCamStates(); }
  void get_imu_state(double* o) override {
    ImuState<S> s = f.getImuState();
    pack4(o, s.q_IG); pack3(o + 4, s.b_g); pack3(o + 7, s.v_I_G); pack3(o + 10, s.b_a); pack3(o + 13, s.p_I_G); pack3(o + 16, s.g);
    pack4(o + 19, s.q_IG_null); pack3(o + 23, s.v_I_G_null); pack3(o + 26, s.p_I_G_null);
  }
  void set_imu_state(const double* in) override { ImuState<S> s; unpack_imu(in, s); f.setImuState(s); }
  void get_cam_states(double* out, int* ids) override {
    auto cs = f.getCamStates();
    for (size_t i = 0; i < cs.size(); ++i) { pack4(out + 7 * i, cs[i].q_CG); pack3(out + 7 * i + 4, cs[i].p_C_G); if (ids) ids[i] = cs[i].state_id; }
  }
  int get_cam_meta(double* time, int* ntracked, int* last_corr, int cap) override {
    auto cs = f.getCamStates();
    for (size_t i = 0; i < cs.size() && (int)i < cap; ++i) { time[i] = cs[i].time; ntracked[i] = (int)cs[i].tracked_feature_ids.size(); last_corr[i] = cs[i].last_correlated_id; }
    return (int)cs.size();
  }
  int pruned_states(double* out9, int cap) override {
    auto p = f.getPrunedStates(); int n = (int)p.size();
    for (int i = 0; i < n && i < cap; ++i) { double* o = out9 + 9 * i; pack4(o, p[i].q_CG); pack3(o + 4, p[i].p_C_G); o[7] = p[i].time; o[8] = p[i].state_id; }
    return n;
  }
  void set_cam_pose(int i, const double* qp) override { f.setCamPose((size_t)i, q4(qp), v3(qp + 4)); }
  void get_covariance(double* P) override { Mat<S> m = f.getCovariance(); for (size_t i = 0; i < m.a.size(); ++i) P[i] = m.a[i]; }
  void set_covariance(const double* P, int D) override { Mat<S> m(D, D); for (size_t i = 0; i < m.a.size(); ++i) m.a[i] = S(P[i]); f.setCovariance(m); }
  void set_tracks(int F, const int* M, const int* slots, const double* obs) override {
    std::vector<std::vector<int>> sl(F); std::vector<std::vector<V2<S>>> ob(F);
    int o = 0;
    for (int t = 0; t < F; ++t) { for (int k = 0; k < M[t]; ++k) { sl[t].push_back(slots[o + k]); ob[t].push_back({S(obs[2 * (o + k)]), S(obs[2 * (o + k) + 1])}); } o += M[t]; }
    f.setTracksToResidualize(sl, ob);
  }
  int get_tracks(int* M, int* slots, double* obs, int cap_f, int cap_m) override {
    const auto& tr = f.tracksToResidualize();
    int F = (int)tr.size(); if (F > cap_f) return -F;
    for (int t = 0; t < F; ++t) {
      M[t] = (int)tr[t].observations.size();
      for (int k = 0; k < M[t] && k < cap_m; ++k) {
        slots[t * cap_m + k] = (int)tr[t].cam_state_indices[k];
        obs[2 * (t * cap_m + k)] = tr[t].observations[k].x; obs[2 * (t * cap_m + k) + 1] = tr[t].observations[k].y;
      }
    }
    return F;
  }
  void drop_oldest(int n) override { f.dropOldest(n); }
  void last_stats(int* o) override {
    const UpdateStats& s = f.last_stats;
    o[0] = s.n_tracks; o[1] = s.n_motion_rejected; o[2] = s.n_tri_rejected; o[3] = s.n_gate_rejected; o[4] = s.n_passed; o[5] = s.m_rows; o[6] = s.r_rows;
  }
  int last_tracks(double* out, int cap) override {
    int n = (int)f.last_tracks.size();
    for (int i = 0; i < n && i < cap; ++i) {
      const TrackDebug& d = f.last_tracks[i];
      double* o = out + 8 * i;
      o[0] = d.motion_ok; o[1] = d.tri_valid; o[2] = d.gate_pass; o[3] = d.rows; o[4] = d.gamma; o[5] = d.p_f_G[0]; o[6] = d.p_f_G[1]; o[7] = d.p_f_G[2];
    }
    return n;
  }
  int last_deltax(double* out, int cap) override { int n = f.last_deltaX.r; for (int i = 0; i < n && i < cap; ++i) out[i] = f.last_deltaX(i, 0); return n; }
  int map_points(double* out, int cap) override { auto m = f.getMap(); int n = (int)m.size(); for (int i = 0; i < n && i < cap; ++i) pack3(out + 3 * i, m[i]); return n; }
  int pruned_ids(int* out, int cap) override { auto p = f.getPrunedStates(); int n = (int)p.size(); for (int i = 0; i < n && i < cap; ++i) out[i] = p[i].state_id; return n; }
  long num_residualized() override { return (long)f.numResidualized(); }
  void set_num_residualized(long n) override { f.setNumResidualized((size_t)n); }
  void set_mode(int m) override { f.mode = (Mode)m; }
  Base* clone() override { return new Impl<S>(*this); }
  void set_whiten(int w) override { f.whiten = w != 0; }
  void set_colpiv(int w) override { f.colpiv_null = w != 0; }
  void set_tiny(double t) override { f.tiny_row_tol = t; }
  void set_capture(int on) override { f.capture = on != 0; }
  // per track of the last marginalize() whose Jacobian was formed (set_capture): M, gated in?, slots [cap_m], H_x [cap_m][12], r [2 cap_m]
  int last_track_inputs(int* M, int* pass, int* slots, double* hx, double* r, int cap_f, int cap_m) override {
    const int F = (int)f.cap_slots.size();
    if (F > cap_f) return -F;
    for (int t = 0; t < F; ++t) {
      const int m = (int)f.cap_slots[t].size();
      if (m > cap_m) return -F;
      M[t] = m; pass[t] = f.cap_pass[t];
      for (int k = 0; k < m; ++k) slots[(size_t)t * cap_m + k] = f.cap_slots[t][k];
      for (int k = 0; k < 12 * m; ++k) hx[(size_t)t * cap_m * 12 + k] = f.cap_hx[t][k];
      for (int k = 0; k < 2 * m; ++k) r[(size_t)t * 2 * cap_m + k] = f.cap_r[t][k];
    }
    return F;
  }
  // which: 0 H_o, 1 r_o, 2 T_H, 3 r_n, 4 R_n, 5 S of the last measurementUpdate (column-major); returns rows
  int last_matrix(int which, double* out, long cap, int* cols) override {
    const Mat<S>* m[6] = {&f.last_Ho, &f.last_ro, &f.last_TH, &f.last_rn, &f.last_Rn, &f.last_S};
    if (which < 0 || which > 5) return -1;
    const Mat<S>& a = *m[which];
    *cols = a.c;
    if ((long)a.r * a.c > cap) return -a.r;
    for (int j = 0; j < a.c; ++j) for (int i = 0; i < a.r; ++i) out[(long)j * a.r + i] = (double)a(i, j);
    return a.r;
  }
};
}  // namespace

extern "C" {
void* oracle_create(int dtype, int mode) { return dtype == 0 ? (Base*)new Impl<float>(mode) : (Base*)new Impl<double>(mode); }
void oracle_destroy(void* h) { delete (Base*)h; }
void oracle_initialize(void* h, const double* cam, const double* noise, const double* params, const double* imu) { ((Base*)h)->initialize(cam, noise, params, imu); }
void oracle_propagate(void* h, const double* rd, int K) { ((Base*)h)->propagate(rd, K); }
void oracle_augment(void* h, int id, double t) { ((Base*)h)->augment(id, t); }
void oracle_update(void* h, const double* meas, const uint64_t* ids, int n) { ((Base*)h)->update(meas, ids, n); }
void oracle_add_features(void* h, const double* meas, const uint64_t* ids, int n) { ((Base*)h)->add_features(meas, ids, n); }
void oracle_marginalize(void* h) { ((Base*)h)->marginalize(); }
void oracle_prune_redundant(void* h) { ((Base*)h)->prune_redundant(); }
void oracle_prune_empty(void* h) { ((Base*)h)->prune_empty(); }
void oracle_finish(void* h) { ((Base*)h)->finish(); }
int oracle_num_cam_states(void* h) { return ((Base*)h)->num_cam_states(); }
void oracle_get_imu_state(void* h, double* out) { ((Base*)h)->get_imu_state(out); }
void oracle_set_imu_state(void* h, const double* in) { ((Base*)h)->set_imu_state(in); }
void oracle_get_cam_states(void* h, double* out, int* ids) { ((Base*)h)->get_cam_states(out, ids); }
int oracle_is_reference(void) { return 0; }
int oracle_get_cam_meta(void* h, double* time, int* ntracked, int* last_corr, int cap) { return ((Base*)h)->get_cam_meta(time, ntracked, last_corr, cap); }
int oracle_pruned_states(void* h, double* out9, int cap) { return ((Base*)h)->pruned_states(out9, cap); }
void oracle_set_cam_pose(void* h, int i, const double* qp) { ((Base*)h)->set_cam_pose(i, qp); }
void oracle_get_covariance(void* h, double* P) { ((Base*)h)->get_covariance(P); }
void oracle_set_covariance(void* h, const double* P, int D) { ((Base*)h)->set_covariance(P, D); }
void oracle_set_tracks(void* h, int F, const int* M, const int* slots, const double* obs) { ((Base*)h)->set_tracks(F, M, slots, obs); }
int oracle_get_tracks(void* h, int* M, int* slots, double* obs, int cap_f, int cap_m) { return ((Base*)h)->get_tracks(M, slots, obs, cap_f, cap_m); }
void oracle_drop_oldest(void* h, int n) { ((Base*)h)->drop_oldest(n); }
void oracle_last_stats(void* h, int* out) { ((Base*)h)->last_stats(out); }
int oracle_last_tracks(void* h, double* out, int cap) { return ((Base*)h)->last_tracks(out, cap); }
int oracle_last_deltax(void* h, double* out, int cap) { return ((Base*)h)->last_deltax(out, cap); }
int oracle_map_points(void* h, double* out, int cap) { return ((Base*)h)->map_points(out, cap); }
int oracle_pruned_ids(void* h, int* out, int cap) { return ((Base*)h)->pruned_ids(out, cap); }
long oracle_num_residualized(void* h) { return ((Base*)h)->num_residualized(); }
void oracle_set_num_residualized(void* h, long n) { ((Base*)h)->set_num_residualized(n); }
void oracle_set_mode(void* h, int mode) { ((Base*)h)->set_mode(mode); }
void* oracle_clone(void* h) { return ((Base*)h)->clone(); }
void oracle_set_whiten(void* h, int w) { ((Base*)h)->set_whiten(w); }
void oracle_set_colpiv_null(void* h, int w) { ((Base*)h)->set_colpiv(w); }
void oracle_set_tiny_row_tol(void* h, double t) { ((Base*)h)->set_tiny(t); }
void oracle_set_capture(void* h, int on) { ((Base*)h)->set_capture(on); }
int oracle_last_track_inputs(void* h, int* M, int* pass, int* slots, double* hx, double* r, int cap_f, int cap_m) { return ((Base*)h)->last_track_inputs(M, pass, slots, hx, r, cap_f, cap_m); }
int oracle_last_matrix(void* h, int which, double* out, long cap, int* cols) { return ((Base*)h)->last_matrix(which, out, cap, cols); }

// Timed CPU baseline: run `n_filters` independent filters over the same pre-built per-frame call
// sequence on `n_threads` std::threads (one filter per thread at a time, as the reference is
// single-threaded per trajectory -- nodes/msckf_mono_node.cpp:8).  The caller passes opaque handles that
// are already initialised and warmed up, plus one frame of input replicated per filter:
//   readings [K x 7], tracks (F, M[], slots[], obs[]), n_drop oldest states after the update.
// Returns wall seconds for all filters to finish `reps` filter updates each.
double oracle_time_updates(void** handles, int n_filters, int n_threads, int reps, const double* readings, int K,
                           int state_id0, int F, const int* M, const int* slots, const double* obs, int n_drop) {
  auto t0 = std::chrono::steady_clock::now();
  std::vector<std::thread> th;
  for (int t = 0; t < n_threads; ++t)
    th.emplace_back([=]() {
      for (int i = t; i < n_filters; i += n_threads) {
        Base* b = (Base*)handles[i];
        for (int r = 0; r < reps; ++r) {
          b->propagate(readings, K);
          b->augment(state_id0 + r, 0.0);
          b->set_tracks(F, M, slots, obs);
          b->marginalize();
          b->drop_oldest(n_drop);
        }
      }
    });
  for (auto& x : th) x.join();
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}
}
