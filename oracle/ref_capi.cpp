// oracle/ref_capi.cpp -- TEST INFRASTRUCTURE ONLY.
//
// C-ABI around the REFERENCE's own class, msckf_mono::MSCKF<float|double>, compiled from the unmodified sources
// where they lie (/root/reference/include/msckf_mono/{msckf.h,types.h,matrix_utils.h}) against the minimal
// Eigen/Boost surface of oracle/ref_shim (neither library exists in this image).  Built by `make -C oracle _ref`
// into oracle/_ref/lib_ref.so (git-ignored; it travels to the GPU box with the snapshot, /root/reference does not).
//
// The exported names and argument layouts are those of oracle/oracle_capi.cpp, so oracle/pyoracle.py drives both
// libraries with the same Python class (pyoracle.Oracle(..., impl="ref")).  The reference keeps its covariance and
// work-list private (msckf.h:33-64); this file is compiled with -fno-access-control to read and, for teacher-forced
// tests, overwrite them -- the reference's code itself is untouched.
#include <chrono>
#include <cstdint>
#include <cstring>
#include <iostream>
#include <thread>
#include <vector>

#include <msckf_mono/msckf.h>

namespace {
using namespace msckf_mono;

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
  virtual void set_cam_pose(int i, const double* qp) = 0;
  virtual void get_covariance(double* P) = 0;
  virtual void set_covariance(const double* P, int D) = 0;
  virtual void set_tracks(int F, const int* M, const int* slots, const double* obs) = 0;
  virtual int get_tracks(int* M, int* slots, double* obs, int cap_f, int cap_m) = 0;
  virtual void drop_oldest(int n) = 0;
  virtual int map_points(double* out, int cap) = 0;
  virtual int pruned_ids(int* out, int cap) = 0;
  virtual int pruned_states(double* out9, int cap) = 0;
  virtual long num_residualized() = 0;
  virtual void set_num_residualized(long n) = 0;
  virtual Base* clone() = 0;
};

template <class S>
struct Impl : Base {
  MSCKF<S> f;
  typedef std::vector<Vector2<S>, Eigen::aligned_allocator<Vector2<S>>> Meas;
  static Vector3<S> v3(const double* p) { return Vector3<S>(S(p[0]), S(p[1]), S(p[2])); }
  static msckf_mono::Quaternion<S> q4(const double* p) { return msckf_mono::Quaternion<S>(S(p[0]), S(p[1]), S(p[2]), S(p[3])); }
  static void pack3(double* o, const Vector3<S>& v) { o[0] = v(0); o[1] = v(1); o[2] = v(2); }
  static void pack4(double* o, const msckf_mono::Quaternion<S>& q) { o[0] = q.w(); o[1] = q.x(); o[2] = q.y(); o[3] = q.z(); }
  static void unpack_imu(const double* x, imuState<S>& s) {
    s.q_IG = q4(x); s.b_g = v3(x + 4); s.v_I_G = v3(x + 7); s.b_a = v3(x + 10); s.p_I_G = v3(x + 13); s.g = v3(x + 16);
    s.q_IG_null = q4(x + 19); s.v_I_G_null = v3(x + 23); s.p_I_G_null = v3(x + 26);
  }
  void initialize(const double* cam, const double* noise, const double* params, const double* imu) override {
    Camera<S> c; c.c_u = S(cam[0]); c.c_v = S(cam[1]); c.f_u = S(cam[2]); c.f_v = S(cam[3]); c.b = S(cam[4]);
    c.q_CI = q4(cam + 5); c.p_C_I = v3(cam + 9);
    noiseParams<S> n; n.u_var_prime = S(noise[0]); n.v_var_prime = S(noise[1]);
    n.Q_imu.setZero(); n.initial_imu_covar.setZero();
    for (int i = 0; i < 12; ++i) n.Q_imu(i, i) = S(noise[2 + i]);
    for (int i = 0; i < 15; ++i) n.initial_imu_covar(i, i) = S(noise[14 + i]);
    MSCKFParams<S> p; p.max_gn_cost_norm = S(params[0]); p.min_rcond = S(params[1]); p.translation_threshold = S(params[2]);
    p.redundancy_angle_thresh = S(params[3]); p.redundancy_distance_thresh = S(params[4]);
    p.min_track_length = (int)params[5]; p.max_track_length = (int)params[6]; p.max_cam_states = (int)params[7];
    imuState<S> s; unpack_imu(imu, s);
    f.initialize(c, n, p, s);
  }
  void propagate(const double* rd, int K) override {
    for (int k = 0; k < K; ++k) { imuReading<S> m; m.omega = v3(rd + 7 * k); m.a = v3(rd + 7 * k + 3); m.dT = S(rd[7 * k + 6]); f.propagate(m); }
  }
  void augment(int id, double t) override { f.augmentState(id, S(t)); }
  static void conv(const double* meas, const uint64_t* ids, int n, Meas& m, std::vector<size_t>& i) {
    m.resize(n); i.resize(n);
    for (int k = 0; k < n; ++k) { m[k] = Vector2<S>(S(meas[2 * k]), S(meas[2 * k + 1])); i[k] = (size_t)ids[k]; }
  }
  void update(const double* meas, const uint64_t* ids, int n) override { Meas m; std::vector<size_t> i; conv(meas, ids, n, m, i); f.update(m, i); }
  void add_features(const double* meas, const uint64_t* ids, int n) override { Meas m; std::vector<size_t> i; conv(meas, ids, n, m, i); f.addFeatures(m, i); }
  void marginalize() override { f.marginalize(); }
  void prune_redundant() override { f.pruneRedundantStates(); }
  void prune_empty() override { f.pruneEmptyStates(); }
  void finish() override { f.finish(); }
  int num_cam_states() override { return (int)f.getNumCamStates(); }
  void get_imu_state(double* o) override {
    imuState<S> s = f.getImuState();
    pack4(o, s.q_IG); pack3(o + 4, s.b_g); pack3(o + 7, s.v_I_G); pack3(o + 10, s.b_a); pack3(o + 13, s.p_I_G); pack3(o + 16, s.g);
    pack4(o + 19, s.q_IG_null); pack3(o + 23, s.v_I_G_null); pack3(o + 26, s.p_I_G_null);
  }
  void set_imu_state(const double* in) override { unpack_imu(in, f.imu_state_); }
  void get_cam_states(double* out, int* ids) override {
    auto cs = f.getCamStates();
    for (size_t i = 0; i < cs.size(); ++i) { pack4(out + 7 * i, cs[i].q_CG); pack3(out + 7 * i + 4, cs[i].p_C_G); if (ids) ids[i] = cs[i].state_id; }
  }
  int get_cam_meta(double* time, int* ntracked, int* last_corr, int cap) override {
    auto cs = f.getCamStates();
    for (size_t i = 0; i < cs.size() && (int)i < cap; ++i) { time[i] = cs[i].time; ntracked[i] = (int)cs[i].tracked_feature_ids.size(); last_corr[i] = cs[i].last_correlated_id; }
    return (int)cs.size();
  }
  void set_cam_pose(int i, const double* qp) override { f.cam_states_[(size_t)i].q_CG = q4(qp); f.cam_states_[(size_t)i].p_C_G = v3(qp + 4); }
  void get_covariance(double* P) override {
    const int n = (int)f.cam_covar_.rows(), D = 15 + n;
    for (int j = 0; j < D; ++j)
      for (int i = 0; i < D; ++i) {
        double v;
        if (i < 15 && j < 15) v = f.imu_covar_(i, j);
        else if (i < 15) v = f.imu_cam_covar_(i, j - 15);
        else if (j < 15) v = f.imu_cam_covar_(j, i - 15);
        else v = f.cam_covar_(i - 15, j - 15);
        P[(size_t)j * D + i] = v;
      }
  }
  void set_covariance(const double* P, int D) override {
    const int n = D - 15;
    f.cam_covar_.resize(n, n); f.imu_cam_covar_.resize(15, n);
    for (int j = 0; j < D; ++j)
      for (int i = 0; i < D; ++i) {
        const S v = S(P[(size_t)j * D + i]);
        if (i < 15 && j < 15) f.imu_covar_(i, j) = v;
        else if (i < 15) f.imu_cam_covar_(i, j - 15) = v;
        else if (j >= 15) f.cam_covar_(i - 15, j - 15) = v;
      }
  }
  // positional work-list = what MSCKF::update() leaves in feature_tracks_to_residualize_ (msckf.h:249-262):
  // cam_state_indices are positions in cam_states_, cam_states are copies of those entries
  void set_tracks(int F, const int* M, const int* slots, const double* obs) override {
    f.feature_tracks_to_residualize_.clear();
    int o = 0;
    for (int t = 0; t < F; ++t) {
      featureTrackToResidualize<S> tr;
      tr.feature_id = (size_t)t;
      for (int k = 0; k < M[t]; ++k) {
        tr.observations.push_back(Vector2<S>(S(obs[2 * (o + k)]), S(obs[2 * (o + k) + 1])));
        tr.cam_state_indices.push_back((size_t)slots[o + k]);
        tr.cam_states.push_back(f.cam_states_[(size_t)slots[o + k]]);
      }
      o += M[t];
      f.feature_tracks_to_residualize_.push_back(tr);
    }
  }
  int get_tracks(int* M, int* slots, double* obs, int cap_f, int cap_m) override {
    const auto& tr = f.feature_tracks_to_residualize_;
    int F = (int)tr.size(); if (F > cap_f) return -F;
    for (int t = 0; t < F; ++t) {
      M[t] = (int)tr[t].observations.size();
      for (int k = 0; k < M[t] && k < cap_m; ++k) {
        slots[t * cap_m + k] = (int)tr[t].cam_state_indices[k];
        obs[2 * (t * cap_m + k)] = tr[t].observations[k](0); obs[2 * (t * cap_m + k) + 1] = tr[t].observations[k](1);
      }
    }
    return F;
  }
  // additive (the steady-state window of the synthetic configs): drop the n oldest camera states with the
  // reference's own slicing helpers, as pruneEmptyStates does (msckf.h:712-757)
  void drop_oldest(int n) override {
    const int num = (int)f.cam_states_.size();
    if (n <= 0) return;
    if (n > num) n = num;
    for (int i = 0; i < n; ++i) f.pruned_states_.push_back(f.cam_states_[(size_t)i]);
    f.cam_states_.erase(f.cam_states_.begin(), f.cam_states_.begin() + n);
    Eigen::VectorXi keep(6 * (num - n));
    for (int i = 0; i < 6 * (num - n); ++i) keep(i) = 6 * n + i;
    MatrixX<S> pc; square_slice(f.cam_covar_, keep, pc); f.cam_covar_ = pc;
    Eigen::Matrix<S, 15, Eigen::Dynamic> pic; column_slice(f.imu_cam_covar_, keep, pic); f.imu_cam_covar_ = pic;
  }
  int map_points(double* out, int cap) override { auto m = f.getMap(); int n = (int)m.size(); for (int i = 0; i < n && i < cap; ++i) pack3(out + 3 * i, m[(size_t)i]); return n; }
  int pruned_ids(int* out, int cap) override { auto p = f.getPrunedStates(); int n = (int)p.size(); for (int i = 0; i < n && i < cap; ++i) out[i] = p[(size_t)i].state_id; return n; }
  int pruned_states(double* out9, int cap) override {
    auto p = f.getPrunedStates(); int n = (int)p.size();
    for (int i = 0; i < n && i < cap; ++i) { double* o = out9 + 9 * i; pack4(o, p[(size_t)i].q_CG); pack3(o + 4, p[(size_t)i].p_C_G); o[7] = p[(size_t)i].time; o[8] = p[(size_t)i].state_id; }
    return n;
  }
  long num_residualized() override { return (long)f.num_feature_tracks_residualized_; }
  void set_num_residualized(long n) override { f.num_feature_tracks_residualized_ = (size_t)n; }
  Base* clone() override { return new Impl<S>(*this); }
};
}  // namespace

extern "C" {
int oracle_is_reference(void) { return 1; }
void* oracle_create(int dtype, int /*mode*/) { return dtype == 0 ? (Base*)new Impl<float>() : (Base*)new Impl<double>(); }
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
int oracle_get_cam_meta(void* h, double* time, int* ntracked, int* last_corr, int cap) { return ((Base*)h)->get_cam_meta(time, ntracked, last_corr, cap); }
void oracle_set_cam_pose(void* h, int i, const double* qp) { ((Base*)h)->set_cam_pose(i, qp); }
void oracle_get_covariance(void* h, double* P) { ((Base*)h)->get_covariance(P); }
void oracle_set_covariance(void* h, const double* P, int D) { ((Base*)h)->set_covariance(P, D); }
void oracle_set_tracks(void* h, int F, const int* M, const int* slots, const double* obs) { ((Base*)h)->set_tracks(F, M, slots, obs); }
int oracle_get_tracks(void* h, int* M, int* slots, double* obs, int cap_f, int cap_m) { return ((Base*)h)->get_tracks(M, slots, obs, cap_f, cap_m); }
void oracle_drop_oldest(void* h, int n) { ((Base*)h)->drop_oldest(n); }
// the reference computes its per-update counters and discards them (msckf.h:338-346): nothing to report
void oracle_last_stats(void*, int* out) { for (int i = 0; i < 7; ++i) out[i] = -1; }
int oracle_last_tracks(void*, double*, int) { return 0; }
int oracle_last_deltax(void*, double*, int) { return 0; }
int oracle_map_points(void* h, double* out, int cap) { return ((Base*)h)->map_points(out, cap); }
int oracle_pruned_ids(void* h, int* out, int cap) { return ((Base*)h)->pruned_ids(out, cap); }
int oracle_pruned_states(void* h, double* out9, int cap) { return ((Base*)h)->pruned_states(out9, cap); }
long oracle_num_residualized(void* h) { return ((Base*)h)->num_residualized(); }
void oracle_set_num_residualized(void* h, long n) { ((Base*)h)->set_num_residualized(n); }
void oracle_set_mode(void*, int) {}
void* oracle_clone(void* h) { return ((Base*)h)->clone(); }
void oracle_set_whiten(void*, int) {}
void oracle_set_colpiv_null(void*, int) {}
void oracle_set_tiny_row_tol(void*, double) {}
void oracle_set_capture(void*, int) {}
// chi_squared_test_table as the reference built it (msckf.h:91-95), for the table test
int oracle_chi2_table(void* h, double* out, int cap) {
  Impl<double>* d = dynamic_cast<Impl<double>*>((Base*)h);
  if (!d) return -1;
  const int n = (int)d->f.chi_squared_test_table.size();
  for (int i = 0; i < n && i < cap; ++i) out[i] = d->f.chi_squared_test_table[(size_t)i];
  return n;
}

// same contract as oracle_time_updates in oracle_capi.cpp: the reference's per-image call sequence
// (asl_msckf.cpp:227-294 minus pruneRedundantStates) on n_threads host threads, one filter per thread at a time
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

// ---- the shim's own linear algebra, exposed so that tests can hold it against numpy/scipy (all column-major) ----
namespace {
template <class S> Eigen::Matrix<S, Eigen::Dynamic, Eigen::Dynamic> load(const double* a, int m, int n) {
  Eigen::Matrix<S, Eigen::Dynamic, Eigen::Dynamic> M(m, n);
  for (int j = 0; j < n; ++j) for (int i = 0; i < m; ++i) M(i, j) = S(a[(size_t)j * m + i]);
  return M;
}
template <class M> void store(const M& A, double* out) {
  for (int j = 0; j < (int)A.cols(); ++j) for (int i = 0; i < (int)A.rows(); ++i) out[(size_t)j * A.rows() + i] = (double)A(i, j);
}
template <class S> void t_expm(int n, const double* A, double* out) { store(load<S>(A, n, n).exp(), out); }
template <class S> void t_qr(int m, int n, const double* A, double* Q, double* R) {
  Eigen::HouseholderQR<Eigen::Matrix<S, Eigen::Dynamic, Eigen::Dynamic>> qr(load<S>(A, m, n));
  Eigen::Matrix<S, Eigen::Dynamic, Eigen::Dynamic> q = qr.householderQ();
  Eigen::Matrix<S, Eigen::Dynamic, Eigen::Dynamic> r = qr.matrixQR().template triangularView<Eigen::Upper>();
  store(q, Q); store(r, R);
}
template <class S> void t_svd(int m, int n, const double* A, double* U, double* V, double* sv) {
  Eigen::JacobiSVD<Eigen::Matrix<S, Eigen::Dynamic, Eigen::Dynamic>> svd(load<S>(A, m, n), Eigen::ComputeFullU | Eigen::ComputeThinV);
  store(svd.matrixU(), U); store(svd.matrixV(), V); store(svd.singularValues(), sv);
}
template <class S> void t_solve(int n, int nrhs, const double* A, const double* b, double* x_ldlt, double* inv, double* det) {
  auto M = load<S>(A, n, n); auto B = load<S>(b, n, nrhs);
  Eigen::Matrix<S, Eigen::Dynamic, Eigen::Dynamic> x = M.ldlt().solve(B);
  store(x, x_ldlt); store(M.inverse(), inv); *det = (double)M.determinant();
}
}  // namespace
extern "C" {
void shim_expm(int dtype, int n, const double* A, double* out) { dtype == 0 ? t_expm<float>(n, A, out) : t_expm<double>(n, A, out); }
void shim_qr(int dtype, int m, int n, const double* A, double* Q, double* R) { dtype == 0 ? t_qr<float>(m, n, A, Q, R) : t_qr<double>(m, n, A, Q, R); }
void shim_svd(int dtype, int m, int n, const double* A, double* U, double* V, double* sv) { dtype == 0 ? t_svd<float>(m, n, A, U, V, sv) : t_svd<double>(m, n, A, U, V, sv); }
void shim_solve(int dtype, int n, int nrhs, const double* A, const double* b, double* x_ldlt, double* inv, double* det) {
  dtype == 0 ? t_solve<float>(n, nrhs, A, b, x_ldlt, inv, det) : t_solve<double>(n, nrhs, A, b, x_ldlt, inv, det);
}
}
