// oracle/msckf_oracle.hpp -- TEST INFRASTRUCTURE ONLY.
//
// CPU restatement (plain C++17, no Eigen/Boost) of the EKF hot path of the reference
//   /root/reference/include/msckf_mono/msckf.h  (template<typename _S> class MSCKF)
//   /root/reference/include/msckf_mono/types.h, matrix_utils.h
// used as the checker for the HIP path and as the `cpu_baseline` leg of bench.py.  Nothing in the
// product (msckf_mono_amd/csrc, include/) may include or link this file.
//
// PARITY PINNED TO REFERENCE SOURCE: the reference has no tests / golden vectors, and Eigen / Boost.Math are absent
// from this image, but its three headers compile unmodified against the minimal Eigen/Boost shim of oracle/ref_shim
// (oracle/Makefile target _ref -> oracle/_ref/lib_ref.so, wrapper oracle/ref_capi.cpp with this file's C-ABI).
// tests/test_ref_vs_oracle.py holds this restatement against that library (1e-8 double / 1e-3 float, free-running,
// public API path, golden fixtures, cfg3 window); an independent numpy/scipy implementation (oracle/np_oracle.py,
// tests/test_oracle_vs_numpy.py) and the invariants of SURVEY.md section 4 pin it a second time.  Limit: the shim's
// decompositions are this repository's code (checked against scipy), not Eigen's.
//
// Two modes, identical results up to rounding:
//   FAITHFUL  same algorithmic steps and asymptotic costs as msckf.h (dense P re-assembly per use,
//             P.determinant() in augmentState, full-U null space, dense per-feature gate, full m x m
//             Householder Q and dense m x m R_o, explicit S^-1) -- this is the CPU baseline that is timed.
//   LEAN      thin QR, block-diagonal R_o, no copies, no determinant.
//   GRAM      as LEAN, but [T_H | r_n] is the Cholesky factor of [H_o | r_o]^T [H_o | r_o] accumulated in double with
//             semi-definite pivot skipping -- the compression route of the HIP library (kernels_gram.hip) restated on
//             the CPU, so that its long-run behaviour in float can be checked without a GPU.  Isotropic or whitened
//             noise only (R_n = sigma^2 I); otherwise it behaves as LEAN.
// Documented deviations from the reference:
//   Q1b  A_j := last 2M-3 columns of the *Householder* Q of H_f_j (the reference takes them from
//        JacobiSVD's full U, msckf.h:954-955; only the span is defined by the maths, and for
//        u_var' == v_var' every quantity downstream is basis-invariant).
//   D1   the second loop of marginalize uses track.p_f_G (msckf.h:370) instead of p_f_G_vec[iter]
//        (msckf.h:419), which is mis-indexed after a motion-rejected track.
//   D3   observation erased at the index computed *before* erasing the cam_state index (msckf.h:601-604).
//   D6   finish() (msckf.h:765-807) drops the stale feature_tracks_to_residualize_ left by the previous update()
//        instead of re-residualizing it with out-of-date positional indices (the list is only cleared at :218).
#ifndef ORACLE_MSCKF_ORACLE_HPP
#define ORACLE_MSCKF_ORACLE_HPP

#include <cstdio>
#include <set>

#include "chi2_table.h"
#include "la.hpp"

namespace oracle {

// ---- types.h:49-126 restated as PODs
template <class S> struct Camera { S c_u = 0, c_v = 0, f_u = 0, f_v = 0, b = 0; Quat<S> q_CI; V3<S> p_C_I; };
template <class S> struct CamState {
  V3<S> p_C_G; Quat<S> q_CG; S time = 0; int state_id = 0; int last_correlated_id = -1;
  std::vector<size_t> tracked_feature_ids;
};
template <class S> struct ImuState {
  V3<S> p_I_G, p_I_G_null, v_I_G, b_g, b_a, g, v_I_G_null; Quat<S> q_IG, q_IG_null;
};
template <class S> struct ImuReading { V3<S> omega, a; S dT = 0; };
template <class S> struct NoiseParams {
  S u_var_prime = 0, v_var_prime = 0;
  Mat<S> Q_imu{12, 12};
  Mat<S> initial_imu_covar{15, 15};
};
template <class S> struct MSCKFParams {
  S max_gn_cost_norm = 0, min_rcond = 0, translation_threshold = 0;
  S redundancy_angle_thresh = 0, redundancy_distance_thresh = 0;
  int min_track_length = 0, max_track_length = 0, max_cam_states = 0;
};
template <class S> struct V2 { S x = 0, y = 0; };
template <class S> struct FeatureTrackToResidualize {
  size_t feature_id = 0;
  std::vector<V2<S>> observations;
  std::vector<CamState<S>> cam_states;
  std::vector<size_t> cam_state_indices;
  bool initialized = false;
  V3<S> p_f_G;
};
template <class S> struct FeatureTrack {
  size_t feature_id = 0;
  std::vector<V2<S>> observations;
  std::vector<size_t> cam_state_indices;  // state_ids
  bool initialized = false;
  V3<S> p_f_G;
};

enum Mode { FAITHFUL = 0, LEAN = 1, GRAM = 2 };

struct TrackDebug {  // per residualized track, in input order (for kernel-level parity tests)
  int motion_ok = 1, tri_valid = 0, gate_pass = 0, rows = 0;
  double gamma = 0, p_f_G[3] = {0, 0, 0};
};
struct UpdateStats {
  int n_tracks = 0, n_motion_rejected = 0, n_tri_rejected = 0, n_gate_rejected = 0, n_passed = 0;
  int m_rows = 0, r_rows = 0;
};

template <class S>
class MSCKF {
 public:
  Mode mode = LEAN;
  // Anisotropic pixel noise (u_var' != v_var'): when set, every observation row is pre-whitened by 1/sigma_u resp.
  // 1/sigma_v and the filter then runs with unit isotropic noise -- the construction the HIP path uses (DESIGN.md
  // section 3).  When clear the reference's A_j^T R_j A_j / Q_1^T R_o Q_1 construction is restated literally.
  bool whiten = false;
  // null-space basis A_j: the reference takes the trailing columns of JacobiSVD's full U, i.e. of the Q of a
  // column-pivoted Householder QR of H_f_j (true); false = unpivoted reflectors in column order.  With isotropic
  // pixel noise the update does not depend on the choice.
  bool colpiv_null = true;
  // experiment: > 0 drops the rows of R whose entries are all below tiny_row_tol * max|R| (rounding-level rows of the
  // gauge directions); 0 = the reference's rule (a row is kept if any entry is non-zero, msckf.h:1347)
  double tiny_row_tol = 0;
  UpdateStats last_stats;
  std::vector<TrackDebug> last_tracks;
  Mat<S> last_deltaX;
  // debug capture of the last measurementUpdate (tests only): stacked [H_o, r_o], compressed T_H, r_n, R_n and S
  bool capture = false;
  Mat<S> last_Ho, last_ro, last_TH, last_rn, last_Rn, last_S;
  // per track whose Jacobian was formed in the last marginalize(), in list order: camera slots, H_x_j as [M][2][6], r_j [2M],
  // gated in (stacked) or not
  mutable std::vector<std::vector<int>> cap_slots; mutable std::vector<std::vector<double>> cap_hx, cap_r; mutable std::vector<int> cap_pass;

  // ---------------------------------------------------------------- msckf.h:72-97
  void initialize(const Camera<S>& camera, const NoiseParams<S>& noise_params,
                  const MSCKFParams<S>& msckf_params, const ImuState<S>& imu_state) {
    camera_ = camera; noise_params_ = noise_params; msckf_params_ = msckf_params;
    num_feature_tracks_residualized_ = 0;
    imu_state_ = imu_state;
    imu_state_.p_I_G_null = imu_state_.p_I_G;
    imu_state_.v_I_G_null = imu_state_.v_I_G;
    imu_state_.q_IG_null = imu_state_.q_IG;
    imu_covar_ = noise_params.initial_imu_covar;
    cam_covar_.resize(0, 0); imu_cam_covar_.resize(15, 0);
    cam_states_.clear(); feature_tracks_.clear(); tracked_feature_ids_.clear();
    feature_tracks_to_residualize_.clear(); tracks_to_remove_.clear(); pruned_states_.clear(); map_.clear();
    chi_squared_test_table.resize(99);
    for (int i = 1; i < 100; ++i) chi_squared_test_table[i - 1] = S(kOracleChi2Q05[i - 1]);
  }

  // ---------------------------------------------------------------- msckf.h:101-145
  void propagate(const ImuReading<S>& m) {
    calcF(imu_state_, m);
    calcG(imu_state_);
    ImuState<S> prop = propogateImuStateRK(imu_state_, m);
    for (auto& v : F_.a) v *= m.dT;              // :108
    Phi_ = expm(F_);                             // :111
    // observability constraints :116-132
    M3<S> R_kk_1 = imu_state_.q_IG_null.toRot();
    set3(Phi_, 0, 0, prop.q_IG.toRot() * R_kk_1.t());
    V3<S> u = R_kk_1 * imu_state_.g;
    const S uu = dot(u, u);
    V3<S> s = (S(1) / uu) * u;  // row vector (u^T u)^-1 u^T
    M3<S> A1 = get3(Phi_, 6, 0);
    V3<S> tmp = imu_state_.v_I_G_null - prop.v_I_G;
    V3<S> w1 = skew(tmp) * imu_state_.g;
    set3(Phi_, 6, 0, A1 - outer((A1 * u) - w1, s));
    M3<S> A2 = get3(Phi_, 12, 0);
    tmp = (m.dT * imu_state_.v_I_G_null) + imu_state_.p_I_G_null - prop.p_I_G;
    V3<S> w2 = skew(tmp) * imu_state_.g;
    set3(Phi_, 12, 0, A2 - outer((A2 * u) - w2, s));
    // :134  Phi (P_II + G Q G^T dT) Phi^T
    Mat<S> GQ = mul(G_, noise_params_.Q_imu);
    Mat<S> GQG = mul_abt(GQ, G_);
    for (auto& v : GQG.a) v *= m.dT;
    Mat<S> inner = add(imu_covar_, GQG);
    Mat<S> prop_cov = mul_abt(mul(Phi_, inner), Phi_);
    imu_state_ = prop;                           // :138-141
    imu_state_.q_IG_null = imu_state_.q_IG;
    imu_state_.v_I_G_null = imu_state_.v_I_G;
    imu_state_.p_I_G_null = imu_state_.p_I_G;
    symmetrize(prop_cov);                        // :143
    imu_covar_ = prop_cov;
    if (imu_cam_covar_.c) imu_cam_covar_ = mul(Phi_, imu_cam_covar_);  // :144
  }

  // ---------------------------------------------------------------- msckf.h:148-212
  void augmentState(int state_id, S time) {
    map_.clear();
    Quat<S> q_CG = camera_.q_CI * imu_state_.q_IG;
    q_CG.normalize();
    CamState<S> cs;
    cs.last_correlated_id = -1;
    cs.q_CG = q_CG;
    cs.p_C_G = imu_state_.p_I_G + imu_state_.q_IG.inverse().rotate(camera_.p_C_I);
    cs.time = time; cs.state_id = state_id;
    const int n = (int)cam_states_.size(), D = 15 + 6 * n;
    Mat<S> P = fullP();
    if (mode == FAITHFUL) { volatile S det = determinant(P); (void)det; }   // :176 (result unused)
    M3<S> Jtt = camera_.q_CI.toRot();
    M3<S> Jpt = skew(imu_state_.q_IG.inverse().rotate(camera_.p_C_I));
    Mat<S> P_aug(D + 6, D + 6);
    if (mode == FAITHFUL) {                      // :189-195 two dense (D+6) x D products
      Mat<S> T(D + 6, D);
      for (int i = 0; i < D; ++i) T(i, i) = 1;
      for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { T(D + i, j) = Jtt.m[i][j]; T(D + 3 + i, j) = Jpt.m[i][j]; }
      for (int i = 0; i < 3; ++i) T(D + 3 + i, 12 + i) = 1;
      P_aug = mul_abt(mul(T, P), T);
    } else {                                     // same product using J's sparsity
      Mat<S> JP(6, D);
      for (int c = 0; c < D; ++c) for (int i = 0; i < 3; ++i) {
        S a = 0, b = 0;
        for (int k = 0; k < 3; ++k) { a += Jtt.m[i][k] * P(k, c); b += Jpt.m[i][k] * P(k, c); }
        JP(i, c) = a; JP(3 + i, c) = b + P(12 + i, c);
      }
      P_aug.set_block(0, 0, P);
      for (int c = 0; c < D; ++c) for (int i = 0; i < 6; ++i) { P_aug(D + i, c) = JP(i, c); P_aug(c, D + i) = JP(i, c); }
      for (int i = 0; i < 6; ++i) for (int j = 0; j < 3; ++j) {
        S a = 0, b = 0;
        for (int k = 0; k < 3; ++k) { a += JP(i, k) * Jtt.m[j][k]; b += JP(i, k) * Jpt.m[j][k]; }
        P_aug(D + i, D + j) = a; P_aug(D + i, D + 3 + j) = b + JP(i, 12 + j);
      }
    }
    symmetrize(P_aug);                           // :197
    cam_states_.push_back(cs);
    splitP(P_aug);                               // :203-209
  }

  // ---------------------------------------------------------------- msckf.h:215-299
  void update(const std::vector<V2<S>>& measurements, const std::vector<size_t>& feature_ids) {
    feature_tracks_to_residualize_.clear();
    tracks_to_remove_.clear();
    int id_iter = 0;
    for (size_t feature_id : tracked_feature_ids_) {
      auto it = std::find(feature_ids.begin(), feature_ids.end(), feature_id);
      const bool is_valid = (it != feature_ids.end());
      FeatureTrack<S>& track = feature_tracks_[id_iter];
      if (is_valid) {
        size_t d = (size_t)std::distance(feature_ids.begin(), it);
        track.observations.push_back(measurements[d]);
        CamState<S>& last = cam_states_.back();
        last.tracked_feature_ids.push_back(feature_id);
        track.cam_state_indices.push_back((size_t)last.state_id);
      }
      if (!is_valid || track.observations.size() >= (size_t)msckf_params_.max_track_length) {
        FeatureTrackToResidualize<S> ttr;
        removeTrackedFeature(feature_id, ttr.cam_states, ttr.cam_state_indices);
        if (ttr.cam_states.size() >= (size_t)msckf_params_.min_track_length) {
          ttr.feature_id = track.feature_id;
          ttr.observations = track.observations;
          ttr.initialized = track.initialized;
          if (track.initialized) ttr.p_f_G = track.p_f_G;
          feature_tracks_to_residualize_.push_back(ttr);
        }
        tracks_to_remove_.push_back(feature_id);
      }
      id_iter++;
    }
    for (size_t feature_id : tracks_to_remove_) {
      for (auto ti = feature_tracks_.begin(); ti != feature_tracks_.end(); ++ti) {
        if (ti->feature_id == feature_id) {
          size_t last_id = ti->cam_state_indices.back();
          for (size_t index : ti->cam_state_indices)
            for (auto& cs : cam_states_)
              if (!cs.tracked_feature_ids.size() && (size_t)cs.state_id == index) cs.last_correlated_id = (int)last_id;
          feature_tracks_.erase(ti);
          break;
        }
      }
      auto cid = std::find(tracked_feature_ids_.begin(), tracked_feature_ids_.end(), feature_id);
      if (cid != tracked_feature_ids_.end()) tracked_feature_ids_.erase(cid);
    }
  }

  // ---------------------------------------------------------------- msckf.h:302-332
  void addFeatures(const std::vector<V2<S>>& features, const std::vector<size_t>& feature_ids) {
    for (size_t i = 0; i < features.size(); i++) {
      size_t id = feature_ids[i];
      if (std::find(tracked_feature_ids_.begin(), tracked_feature_ids_.end(), id) == tracked_feature_ids_.end()) {
        FeatureTrack<S> track;
        track.feature_id = id;
        track.observations.push_back(features[i]);
        CamState<S>& last = cam_states_.back();
        last.tracked_feature_ids.push_back(id);
        track.cam_state_indices.push_back((size_t)last.state_id);
        feature_tracks_.push_back(track);
        tracked_feature_ids_.push_back(id);
      } else {
        return;  // :328-329 (reference prints and returns early)
      }
    }
  }

  // ---------------------------------------------------------------- msckf.h:336-449
  void marginalize() {
    last_stats = UpdateStats(); last_tracks.clear(); last_deltaX.resize(0, 0);
    cap_slots.clear(); cap_hx.clear(); cap_r.clear(); cap_pass.clear();
    if (feature_tracks_to_residualize_.empty()) return;
    last_stats.n_tracks = (int)feature_tracks_to_residualize_.size();
    last_tracks.resize(feature_tracks_to_residualize_.size());
    std::vector<bool> valid_tracks;
    int total_nObs = 0, num_passed = 0, ti = 0;
    for (auto& track : feature_tracks_to_residualize_) {
      TrackDebug& dbg = last_tracks[ti++];
      if (num_feature_tracks_residualized_ > 3 && !checkMotion(track.observations.front(), track.cam_states)) {
        dbg.motion_ok = 0; last_stats.n_motion_rejected++;
        valid_tracks.push_back(false);
        continue;
      }
      V3<S> p_f_G;
      bool isvalid = initializePosition(track.cam_states, track.observations, p_f_G);
      dbg.p_f_G[0] = p_f_G.x; dbg.p_f_G[1] = p_f_G.y; dbg.p_f_G[2] = p_f_G.z;
      dbg.tri_valid = isvalid;
      if (isvalid) { track.initialized = true; track.p_f_G = p_f_G; map_.push_back(p_f_G); }
      if (!isvalid) { last_stats.n_tri_rejected++; valid_tracks.push_back(false); }
      else {
        num_passed++; valid_tracks.push_back(true);
        total_nObs += (int)track.observations.size();
        num_feature_tracks_residualized_ += 1;
      }
    }
    if (!num_passed) return;
    const int D = 15 + 6 * (int)cam_states_.size();
    const int mmax = 2 * total_nObs - 3 * num_passed;
    Mat<S> H_o(mmax, D), r_o(mmax, 1);
    Mat<S> R_o;                                   // dense m x m only in FAITHFUL (:406)
    std::vector<Mat<S>> R_blocks; std::vector<int> R_off;
    if (mode == FAITHFUL) R_o.resize(mmax, mmax);
    int stack = 0;
    for (size_t iter = 0; iter < feature_tracks_to_residualize_.size(); iter++) {
      if (!valid_tracks[iter]) continue;
      FeatureTrackToResidualize<S>& track = feature_tracks_to_residualize_[iter];
      const V3<S> p_f_G = track.p_f_G;            // D1
      Mat<S> r_j = calcResidual(p_f_G, track.cam_states, track.observations);
      whitenRows(r_j);
      const int nObs = (int)track.observations.size();
      Mat<S> H_o_j, A_j;
      calcMeasJacobian(p_f_G, track.cam_state_indices, H_o_j, A_j);
      Mat<S> r_o_j = mul_atb(A_j, r_j);           // :430
      Mat<S> R_o_j = projectedNoise(A_j, nObs);   // :431
      double gamma = 0;
      const bool pass = gatingTest(H_o_j, r_o_j, (int)track.cam_states.size() - 1, &gamma);
      last_tracks[iter].gamma = gamma; last_tracks[iter].gate_pass = pass; last_tracks[iter].rows = H_o_j.r;
      if (capture) { std::vector<double> rr(r_j.r); for (int i = 0; i < r_j.r; ++i) rr[i] = (double)r_j(i, 0); cap_r.push_back(rr); cap_pass.push_back(pass ? 1 : 0); }
      if (pass) {
        for (int i = 0; i < H_o_j.r; ++i) r_o(stack + i, 0) = r_o_j(i, 0);
        H_o.set_block(stack, 0, H_o_j);
        if (mode == FAITHFUL) R_o.set_block(stack, stack, R_o_j);
        else { R_blocks.push_back(R_o_j); R_off.push_back(stack); }
        stack += H_o_j.r;
        last_stats.n_passed++;
      } else last_stats.n_gate_rejected++;
    }
    // conservativeResize :443-445
    Mat<S> H2 = H_o.block(0, 0, stack, D), r2 = r_o.block(0, 0, stack, 1);
    if (mode == FAITHFUL) { Mat<S> R2 = R_o.block(0, 0, stack, stack); measurementUpdate(H2, r2, &R2, nullptr, nullptr); }
    else measurementUpdate(H2, r2, nullptr, &R_blocks, &R_off);
  }

  // ---------------------------------------------------------------- msckf.h:453-682
  void pruneRedundantStates() {
    if (cam_states_.size() < 20) return;
    std::vector<size_t> rm_ids;
    findRedundantCamStates(rm_ids);
    for (auto& feature : feature_tracks_) {                        // :466-534
      std::vector<size_t> involved; size_t obs_id = 0;
      for (size_t cam_id : rm_ids) {
        auto it = std::find(feature.cam_state_indices.begin(), feature.cam_state_indices.end(), cam_id);
        if (it != feature.cam_state_indices.end()) { involved.push_back(cam_id); obs_id = (size_t)std::distance(feature.cam_state_indices.begin(), it); }
      }
      if (involved.empty()) continue;
      if (involved.size() == 1) {
        feature.observations.erase(feature.observations.begin() + obs_id);
        feature.cam_state_indices.erase(feature.cam_state_indices.begin() + obs_id);
        continue;
      }
      if (!feature.initialized) {
        std::vector<CamState<S>> assoc;
        for (const auto& cs : cam_states_)
          if (std::find(feature.cam_state_indices.begin(), feature.cam_state_indices.end(), (size_t)cs.state_id) != feature.cam_state_indices.end())
            assoc.push_back(cs);
        V3<S> p_f_G;
        if (!checkMotion(feature.observations.front(), assoc) || !initializePosition(assoc, feature.observations, p_f_G)) {
          eraseInvolved(feature, involved);
          continue;
        }
        feature.initialized = true; feature.p_f_G = p_f_G; map_.push_back(p_f_G);
      }
    }
    const int D = 15 + 6 * (int)cam_states_.size();
    std::vector<Mat<S>> Hs, rs, Rs;
    for (auto& feature : feature_tracks_) {                        // :545-607
      std::vector<size_t> involved; std::vector<V2<S>> involved_obs;
      for (size_t cam_id : rm_ids) {
        auto it = std::find(feature.cam_state_indices.begin(), feature.cam_state_indices.end(), cam_id);
        if (it != feature.cam_state_indices.end()) {
          involved.push_back(cam_id);
          involved_obs.push_back(feature.observations[std::distance(feature.cam_state_indices.begin(), it)]);
        }
      }
      const size_t nObs = involved.size();
      if (nObs == 0) continue;
      std::vector<CamState<S>> involved_cs; std::vector<size_t> pos_idx;
      int pos = 0;
      for (const auto& cs : cam_states_) {
        if (std::find(involved.begin(), involved.end(), (size_t)cs.state_id) != involved.end()) { involved_cs.push_back(cs); pos_idx.push_back(pos); }
        pos++;
      }
      Mat<S> r_j = calcResidual(feature.p_f_G, involved_cs, involved_obs);
      whitenRows(r_j);
      Mat<S> H_x_j, A_j;
      calcMeasJacobian(feature.p_f_G, pos_idx, H_x_j, A_j);
      Mat<S> r_x_j = mul_atb(A_j, r_j);
      Mat<S> R_x_j = projectedNoise(A_j, (int)nObs);
      if (gatingTest(H_x_j, r_x_j, (int)nObs - 1, nullptr)) { Hs.push_back(H_x_j); rs.push_back(r_x_j); Rs.push_back(R_x_j); }
      eraseInvolved(feature, involved);
    }
    int m = 0; for (auto& h : Hs) m += h.r;
    Mat<S> H(m, D), r(m, 1); std::vector<int> off; int stack = 0;
    for (size_t i = 0; i < Hs.size(); ++i) { H.set_block(stack, 0, Hs[i]); for (int k = 0; k < rs[i].r; ++k) r(stack + k, 0) = rs[i](k, 0); off.push_back(stack); stack += Hs[i].r; }
    if (mode == FAITHFUL) {
      Mat<S> R(m, m); for (size_t i = 0; i < Rs.size(); ++i) R.set_block(off[i], off[i], Rs[i]);
      measurementUpdate(H, r, &R, nullptr, nullptr);
    } else measurementUpdate(H, r, nullptr, &Rs, &off);
    // prune :616-681
    std::vector<int> keep;
    std::vector<CamState<S>> kept;
    for (size_t i = 0; i < cam_states_.size(); ++i) {
      if (std::find(rm_ids.begin(), rm_ids.end(), (size_t)cam_states_[i].state_id) != rm_ids.end()) pruned_states_.push_back(cam_states_[i]);
      else { keep.push_back((int)i); kept.push_back(cam_states_[i]); }
    }
    if (kept.size() != cam_states_.size()) { sliceCovariance(keep); cam_states_ = kept; }
  }

  // ---------------------------------------------------------------- msckf.h:685-761
  void pruneEmptyStates() {
    const int max_states = msckf_params_.max_cam_states;
    if ((int)cam_states_.size() < max_states) return;
    const int num = (int)cam_states_.size();
    int last_to_remove = num - max_states - 1;
    if (cam_states_.front().tracked_feature_ids.size()) return;
    for (int i = 1; i < num - max_states; i++)
      if (cam_states_[i].tracked_feature_ids.size()) { last_to_remove = i - 1; break; }
    if (last_to_remove < 0) return;
    std::vector<int> keep;
    for (int i = 0; i <= last_to_remove; ++i) pruned_states_.push_back(cam_states_[i]);
    for (int i = last_to_remove + 1; i < num; ++i) keep.push_back(i);
    sliceCovariance(keep);
    cam_states_.erase(cam_states_.begin(), cam_states_.begin() + last_to_remove + 1);
  }

  // ---------------------------------------------------------------- msckf.h:765-807
  void finish() {
    feature_tracks_to_residualize_.clear();   // D6
    for (size_t i = 0; i < tracked_feature_ids_.size(); i++) {
      std::vector<size_t> idx; std::vector<CamState<S>> cs;
      removeTrackedFeature(tracked_feature_ids_[i], cs, idx);
      if (cs.size() >= (size_t)msckf_params_.min_track_length) {
        FeatureTrackToResidualize<S> t;
        for (auto& ft : feature_tracks_) if (ft.feature_id == tracked_feature_ids_[i]) {
          t.feature_id = ft.feature_id; t.observations = ft.observations; t.initialized = ft.initialized;
          if (ft.initialized) t.p_f_G = ft.p_f_G;
          break;
        }
        t.cam_states = cs; t.cam_state_indices = idx;
        feature_tracks_to_residualize_.push_back(t);
      }
      tracks_to_remove_.push_back(tracked_feature_ids_[i]);
    }
    marginalize();
  }

  // ---------------------------------------------------------------- getters :810-848 (+ additive ones)
  size_t getNumCamStates() const { return cam_states_.size(); }
  ImuState<S> getImuState() const { return imu_state_; }
  std::vector<V3<S>> getMap() const { return map_; }
  Camera<S> getCamera() const { return camera_; }
  CamState<S> getCamState(size_t i) const { return cam_states_[i]; }
  std::vector<CamState<S>> getCamStates() const { return cam_states_; }
  std::vector<CamState<S>> getPrunedStates() {
    std::stable_sort(pruned_states_.begin(), pruned_states_.end(), [](const CamState<S>& a, const CamState<S>& b) { return a.state_id < b.state_id; });
    return pruned_states_;
  }
  // additive (the reference keeps the covariance private, msckf.h:52-54)
  Mat<S> getCovariance() const { return fullP(); }
  void setCovariance(const Mat<S>& P) { splitP(P); }
  void setImuState(const ImuState<S>& s) { imu_state_ = s; }
  void setCamPose(size_t i, const Quat<S>& q, const V3<S>& p) { cam_states_[i].q_CG = q; cam_states_[i].p_C_G = p; }
  size_t numResidualized() const { return num_feature_tracks_residualized_; }
  void setNumResidualized(size_t n) { num_feature_tracks_residualized_ = n; }
  const std::vector<FeatureTrackToResidualize<S>>& tracksToResidualize() const { return feature_tracks_to_residualize_; }
  // inject a ready-made work-list (the batched path's "track dump" input): positional slots + obs
  void setTracksToResidualize(const std::vector<std::vector<int>>& slots, const std::vector<std::vector<V2<S>>>& obs) {
    feature_tracks_to_residualize_.clear();
    for (size_t t = 0; t < slots.size(); ++t) {
      FeatureTrackToResidualize<S> tr;
      tr.feature_id = t; tr.observations = obs[t];
      for (int s : slots[t]) { tr.cam_state_indices.push_back((size_t)s); tr.cam_states.push_back(cam_states_[s]); }
      feature_tracks_to_residualize_.push_back(tr);
    }
  }
  // drop cam slots [0, n) unconditionally (steady-state window of SURVEY 8d: oldest state leaves each frame)
  void dropOldest(int n) {
    std::vector<int> keep;
    for (int i = n; i < (int)cam_states_.size(); ++i) keep.push_back(i);
    for (int i = 0; i < n; ++i) pruned_states_.push_back(cam_states_[i]);
    sliceCovariance(keep);
    cam_states_.erase(cam_states_.begin(), cam_states_.begin() + n);
  }

 private:
  Camera<S> camera_; NoiseParams<S> noise_params_; MSCKFParams<S> msckf_params_;
  std::vector<FeatureTrack<S>> feature_tracks_;
  std::vector<size_t> tracked_feature_ids_;
  std::vector<FeatureTrackToResidualize<S>> feature_tracks_to_residualize_;
  size_t num_feature_tracks_residualized_ = 0;
  std::vector<size_t> tracks_to_remove_;
  ImuState<S> imu_state_;
  std::vector<CamState<S>> cam_states_, pruned_states_;
  std::vector<V3<S>> map_;
  Mat<S> imu_covar_{15, 15}, cam_covar_, imu_cam_covar_{15, 0};
  std::vector<S> chi_squared_test_table;
  Mat<S> F_{15, 15}, Phi_{15, 15}, G_{15, 12};

  bool whitening() const { return whiten && noise_params_.u_var_prime != noise_params_.v_var_prime; }
  S uvar() const { return whitening() ? S(1) : noise_params_.u_var_prime; }
  S vvar() const { return whitening() ? S(1) : noise_params_.v_var_prime; }
  S roww(int i) const { return whitening() ? S(1) / std::sqrt((i % 2 == 0) ? noise_params_.u_var_prime : noise_params_.v_var_prime) : S(1); }
  void whitenRows(Mat<S>& r) const { if (whitening()) for (int i = 0; i < r.r; ++i) for (int j = 0; j < r.c; ++j) r(i, j) *= roww(i); }

  static void set3(Mat<S>& M, int i0, int j0, const M3<S>& b) { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) M(i0 + i, j0 + j) = b.m[i][j]; }
  static M3<S> get3(const Mat<S>& M, int i0, int j0) { M3<S> b; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) b.m[i][j] = M(i0 + i, j0 + j); return b; }
  static M3<S> outer(V3<S> a, V3<S> b) {
    M3<S> r; S av[3] = {a.x, a.y, a.z}, bv[3] = {b.x, b.y, b.z};
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = av[i] * bv[j];
    return r;
  }

  Mat<S> fullP() const {  // :166-174, :1104-1110, :1330-1336
    const int n = cam_covar_.r;
    Mat<S> P(15 + n, 15 + n);
    P.set_block(0, 0, imu_covar_);
    if (n) { P.set_block(0, 15, imu_cam_covar_); P.set_block(15, 0, imu_cam_covar_.t()); P.set_block(15, 15, cam_covar_); }
    return P;
  }
  void splitP(const Mat<S>& P) {
    const int n = P.r - 15;
    imu_covar_ = P.block(0, 0, 15, 15);
    cam_covar_ = P.block(15, 15, n, n);
    imu_cam_covar_ = P.block(0, 15, 15, n);
  }
  void sliceCovariance(const std::vector<int>& keep_states) {  // square_slice / column_slice, matrix_utils.h:58-87
    std::vector<int> idx;
    for (int s : keep_states) for (int k = 0; k < 6; ++k) idx.push_back(6 * s + k);
    const int n = (int)idx.size();
    Mat<S> cc(n, n), ic(15, n);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) cc(i, j) = cam_covar_(idx[i], idx[j]);
    for (int i = 0; i < 15; ++i) for (int j = 0; j < n; ++j) ic(i, j) = imu_cam_covar_(i, idx[j]);
    cam_covar_ = cc; imu_cam_covar_ = ic;
  }
  void eraseInvolved(FeatureTrack<S>& feature, const std::vector<size_t>& involved) {
    for (size_t cam_id : involved) {
      auto it = std::find(feature.cam_state_indices.begin(), feature.cam_state_indices.end(), cam_id);
      if (it != feature.cam_state_indices.end()) {
        size_t idx = (size_t)std::distance(feature.cam_state_indices.begin(), it);
        feature.cam_state_indices.erase(it);
        feature.observations.erase(feature.observations.begin() + idx);
      }
    }
  }

  // ---- :851-872
  static Quat<S> buildUpdateQuat(V3<S> dtheta) {
    V3<S> dq = S(0.5) * dtheta;
    Quat<S> q;
    S cs = dot(dq, dq);
    q.w = (cs > S(1)) ? S(1) : std::sqrt(S(1) - cs);
    q.x = -dq.x; q.y = -dq.y; q.z = -dq.z;
    q.normalize();
    return q;
  }
  // ---- :874-890
  void calcF(const ImuState<S>& st, const ImuReading<S>& m) {
    F_.resize(15, 15);
    V3<S> omegaHat = m.omega - st.b_g, aHat = m.a - st.b_a;
    M3<S> C_IG = st.q_IG.toRot();
    set3(F_, 0, 0, neg(skew(omegaHat)));
    set3(F_, 0, 3, neg(M3<S>::identity()));
    set3(F_, 6, 0, neg(C_IG.t() * skew(aHat)));
    set3(F_, 6, 9, neg(C_IG.t()));
    set3(F_, 12, 6, M3<S>::identity());
  }
  // ---- :892-903
  void calcG(const ImuState<S>& st) {
    G_.resize(15, 12);
    M3<S> C_IG = st.q_IG.toRot();
    set3(G_, 0, 0, neg(M3<S>::identity()));
    set3(G_, 3, 3, M3<S>::identity());
    set3(G_, 6, 6, neg(C_IG.t()));
    set3(G_, 9, 9, M3<S>::identity());
  }
  // ---- :1425-1467
  ImuState<S> propogateImuStateRK(const ImuState<S>& st, const ImuReading<S>& m) {
    ImuState<S> out = st;
    const S dT = m.dT;
    V3<S> w = m.omega - st.b_g;
    // omega_psi = 0.5 * omegaMat(w), matrix_utils.h:20-30
    S O[4][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    M3<S> ns = neg(skew(w));
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) O[i][j] = S(0.5) * ns.m[i][j];
    O[0][3] = S(0.5) * w.x; O[1][3] = S(0.5) * w.y; O[2][3] = S(0.5) * w.z;
    O[3][0] = -S(0.5) * w.x; O[3][1] = -S(0.5) * w.y; O[3][2] = -S(0.5) * w.z;
    struct V4 { S v[4]; };
    auto mulO = [&](const V4& y) { V4 r; for (int i = 0; i < 4; ++i) { S s = 0; for (int j = 0; j < 4; ++j) s += O[i][j] * y.v[j]; r.v[i] = s; } return r; };
    auto comb = [&](const V4& y0, std::initializer_list<std::pair<S, const V4*>> terms) {
      V4 acc{{0, 0, 0, 0}};
      for (auto& t : terms) for (int i = 0; i < 4; ++i) acc.v[i] += t.first * t.second->v[i];
      V4 r; for (int i = 0; i < 4; ++i) r.v[i] = y0.v[i] + acc.v[i] * dT; return r;
    };
    V4 y0{{-st.q_IG.x, -st.q_IG.y, -st.q_IG.z, st.q_IG.w}};
    V4 k0 = mulO(y0);
    V4 k1 = mulO(comb(y0, {{S(1) / S(4), &k0}}));
    V4 k2 = mulO(comb(y0, {{S(1) / S(8), &k0}, {S(1) / S(8), &k1}}));
    V4 k3 = mulO(comb(y0, {{-S(1) / S(2), &k1}, {S(1), &k2}}));
    V4 k4 = mulO(comb(y0, {{S(3) / S(16), &k0}, {S(9) / S(16), &k3}}));
    V4 k5 = mulO(comb(y0, {{-S(3) / S(7), &k0}, {S(2) / S(7), &k1}, {S(12) / S(7), &k2}, {-S(12) / S(7), &k3}, {S(8) / S(7), &k4}}));
    V4 yt;
    for (int i = 0; i < 4; ++i)
      yt.v[i] = y0.v[i] + (S(7) * k0.v[i] + S(32) * k2.v[i] + S(12) * k3.v[i] + S(32) * k4.v[i] + S(7) * k5.v[i]) * dT / S(90);
    Quat<S> q{yt.v[3], -yt.v[0], -yt.v[1], -yt.v[2]};
    q.normalize();
    out.q_IG = q;
    V3<S> dv = dT * ((st.q_IG.toRot().t() * (m.a - st.b_a)) + st.g);
    out.v_I_G = st.v_I_G + dv;
    out.p_I_G = st.p_I_G + (dT * st.v_I_G);
    return out;
  }
  // ---- :1469-1485
  void removeTrackedFeature(size_t featureID, std::vector<CamState<S>>& featCamStates, std::vector<size_t>& camStateIndices) {
    featCamStates.clear(); camStateIndices.clear();
    for (size_t c_i = 0; c_i < cam_states_.size(); c_i++) {
      auto& ids = cam_states_[c_i].tracked_feature_ids;
      auto it = std::find(ids.begin(), ids.end(), featureID);
      if (it != ids.end()) { ids.erase(it); camStateIndices.push_back(c_i); featCamStates.push_back(cam_states_[c_i]); }
    }
  }
  // ---- :980-1025
  bool checkMotion(const V2<S>& first_obs, const std::vector<CamState<S>>& cs) const {
    if (cs.size() < 2) return false;
    const CamState<S>& first = cs.front();
    M3<S> R0 = first.q_CG.toRot().t();
    V3<S> dir{first_obs.x, first_obs.y, S(1)};
    dir = (S(1) / norm(dir)) * dir;
    dir = R0 * dir;
    S max_ortho = 0;
    for (size_t i = 1; i < cs.size(); ++i) {
      V3<S> t = cs[i].p_C_G - first.p_C_G;
      S par = dot(t, dir);
      V3<S> orth = t - (par * dir);
      if (norm(orth) > max_ortho) max_ortho = norm(orth);
    }
    return max_ortho > msckf_params_.translation_threshold;
  }
  // ---- :1027-1047, :1287-1323 ; pose = (R, t) taking c0-frame vectors to ci
  struct Pose { M3<S> R; V3<S> t; };
  static S cost(const Pose& T, const V3<S>& x, const V2<S>& z) {
    V3<S> h = (T.R * V3<S>{x.x, x.y, S(1)}) + (x.z * T.t);
    S dx = h.x / h.z - z.x, dy = h.y / h.z - z.y;
    return dx * dx + dy * dy;
  }
  static void jacobian(const Pose& T, const V3<S>& x, const V2<S>& z, S J[2][3], S r[2], S& w) {
    V3<S> h = (T.R * V3<S>{x.x, x.y, S(1)}) + (x.z * T.t);
    S W[3][3];
    for (int i = 0; i < 3; ++i) { W[i][0] = T.R.m[i][0]; W[i][1] = T.R.m[i][1]; }
    W[0][2] = T.t.x; W[1][2] = T.t.y; W[2][2] = T.t.z;
    for (int j = 0; j < 3; ++j) {
      J[0][j] = S(1) / h.z * W[0][j] - h.x / (h.z * h.z) * W[2][j];
      J[1][j] = S(1) / h.z * W[1][j] - h.y / (h.z * h.z) * W[2][j];
    }
    r[0] = h.x / h.z - z.x; r[1] = h.y / h.z - z.y;
    S e = std::sqrt(r[0] * r[0] + r[1] * r[1]);
    const S huber = S(0.01);
    w = (e <= huber) ? S(1) : huber / (S(2) * e);
  }
  // ---- :1126-1145
  static V3<S> generateInitialGuess(const Pose& T, const V2<S>& z1, const V2<S>& z2) {
    V3<S> m = T.R * V3<S>{z1.x, z1.y, S(1)};
    S A0 = m.x - z2.x * m.z, A1 = m.y - z2.y * m.z;
    S b0 = z2.x * T.t.z - T.t.x, b1 = z2.y * T.t.z - T.t.y;
    S depth = (S(1) / (A0 * A0 + A1 * A1)) * (A0 * b0 + A1 * b1);
    return {z1.x * depth, z1.y * depth, depth};
  }
  // ---- :1147-1285
  bool initializePosition(const std::vector<CamState<S>>& camStates, const std::vector<V2<S>>& meas, V3<S>& p_f_G) const {
    const int n = (int)camStates.size();
    std::vector<Pose> poses(n);
    // cam0_pose = (C^T, p); pose.inverse() * T_c0_w  ->  R_i = C_i C_0^T, t_i = C_i (p_0 - p_i)
    M3<S> C0 = camStates[0].q_CG.toRot(); V3<S> p0 = camStates[0].p_C_G;
    for (int i = 0; i < n; ++i) {
      M3<S> Ci = camStates[i].q_CG.toRot();
      poses[i].R = Ci * C0.t();
      poses[i].t = Ci * (p0 - camStates[i].p_C_G);
    }
    V3<S> init = generateInitialGuess(poses[n - 1], meas[0], meas[meas.size() - 1]);
    V3<S> sol{init.x / init.z, init.y / init.z, S(1) / init.z};
    S lambda = S(1e-3);
    const int inner_max = 10, outer_max = 10;
    const S precision = S(5e-7);
    int inner = 0, outer = 0;
    bool reduced = false;
    S delta_norm = 0, total_cost = 0;
    for (int i = 0; i < n; ++i) total_cost += cost(poses[i], sol, meas[i]);
    do {
      Mat<S> A(3, 3); S b[3] = {0, 0, 0};
      for (int i = 0; i < n; ++i) {
        S J[2][3], r[2], w;
        jacobian(poses[i], sol, meas[i], J, r, w);
        const S w2 = (w == S(1)) ? S(1) : w * w;
        for (int a = 0; a < 3; ++a) {
          for (int c = 0; c < 3; ++c) A(a, c) += w2 * (J[0][a] * J[0][c] + J[1][a] * J[1][c]);
          b[a] += w2 * (J[0][a] * r[0] + J[1][a] * r[1]);
        }
      }
      do {
        Mat<S> Ad = A; for (int a = 0; a < 3; ++a) Ad(a, a) += lambda;
        std::vector<S> delta = ldlt_solve(Ad, std::vector<S>{b[0], b[1], b[2]});
        V3<S> ns{sol.x - delta[0], sol.y - delta[1], sol.z - delta[2]};
        delta_norm = std::sqrt(delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2]);
        S new_cost = 0;
        for (int i = 0; i < n; ++i) new_cost += cost(poses[i], ns, meas[i]);
        if (new_cost < total_cost) {
          reduced = true; sol = ns; total_cost = new_cost;
          lambda = lambda / 10 > S(1e-10) ? lambda / 10 : S(1e-10);
        } else {
          reduced = false;
          lambda = lambda * 10 < S(1e12) ? lambda * 10 : S(1e12);
        }
      } while (inner++ < inner_max && !reduced);
      inner = 0;
    } while (outer++ < outer_max && delta_norm > precision);
    V3<S> fin{sol.x / sol.z, sol.y / sol.z, S(1) / sol.z};
    bool valid = true;
    for (const auto& T : poses) { V3<S> p = (T.R * fin) + T.t; if (p.z <= 0) { valid = false; break; } }
    S normalized_cost = total_cost / (S(2) * S(n) * S(n));
    if (normalized_cost > msckf_params_.max_gn_cost_norm) valid = false;
    p_f_G = (C0.t() * fin) + p0;
    return valid;
  }
  // ---- :960-978
  Mat<S> calcResidual(const V3<S>& p_f_G, const std::vector<CamState<S>>& cs, const std::vector<V2<S>>& obs) const {
    Mat<S> r(2 * (int)cs.size(), 1);
    for (size_t i = 0; i < cs.size(); ++i) {
      V3<S> p = cs[i].q_CG.toRot() * (p_f_G - cs[i].p_C_G);
      r(2 * (int)i, 0) = obs[i].x - p.x / p.z;
      r(2 * (int)i + 1, 0) = obs[i].y - p.y / p.z;
    }
    return r;
  }
  // ---- :905-958.  H_x (2x6 per observation) with the OC projection; A_j from the Householder Q of H_f_j.
  void calcMeasJacobian(const V3<S>& p_f_G, const std::vector<size_t>& idx, Mat<S>& H_o_j, Mat<S>& A_j) const {
    const int M = (int)idx.size(), D = 15 + 6 * (int)cam_states_.size();
    Mat<S> H_f(2 * M, 3), H_x(2 * M, D);
    for (int c = 0; c < M; ++c) {
      const CamState<S>& cs = cam_states_[idx[c]];
      M3<S> C = cs.q_CG.toRot();
      V3<S> pc = C * (p_f_G - cs.p_C_G);
      const S X = pc.x, Y = pc.y, Z = pc.z;
      S Ji[2][3];  // reference: J_i << 1,0,-X/Z, 0,1,-Y/Z; J_i *= 1/Z   (:929-931)
      Ji[0][0] = S(1) * (S(1) / Z); Ji[0][1] = 0; Ji[0][2] = (-X / Z) * (S(1) / Z);
      Ji[1][0] = 0; Ji[1][1] = S(1) * (S(1) / Z); Ji[1][2] = (-Y / Z) * (S(1) / Z);
      M3<S> sk = skew(pc);
      S A[2][6];
      for (int i = 0; i < 2; ++i) for (int j = 0; j < 3; ++j) {
        S a = 0, b = 0;
        for (int k = 0; k < 3; ++k) { a += Ji[i][k] * sk.m[k][j]; b += Ji[i][k] * C.m[k][j]; }
        A[i][j] = a; A[i][3 + j] = -b;
      }
      V3<S> uh = C * imu_state_.g;
      V3<S> ut = skew(p_f_G - cs.p_C_G) * imu_state_.g;
      S u[6] = {uh.x, uh.y, uh.z, ut.x, ut.y, ut.z};
      S uu = 0; for (int k = 0; k < 6; ++k) uu += u[k] * u[k];
      for (int i = 0; i < 2; ++i) {
        S Au = 0; for (int k = 0; k < 6; ++k) Au += A[i][k] * u[k];
        for (int k = 0; k < 6; ++k) {
          S hx = A[i][k] - Au * (S(1) / uu) * u[k];
          H_x(2 * c + i, 15 + 6 * (int)idx[c] + k) = hx;
          if (k >= 3) H_f(2 * c + i, k - 3) = -hx;
        }
      }
    }
    whitenRows(H_f); whitenRows(H_x);
    if (capture) {
      std::vector<int> sl(M); std::vector<double> hx((size_t)M * 12);
      for (int c = 0; c < M; ++c) {
        sl[c] = (int)idx[c];
        for (int i = 0; i < 2; ++i) for (int k = 0; k < 6; ++k) hx[(size_t)c * 12 + i * 6 + k] = (double)H_x(2 * c + i, 15 + 6 * (int)idx[c] + k);
      }
      cap_slots.push_back(sl); cap_hx.push_back(hx);
    }
    const int rows = 2 * M;
    Mat<S> QR = H_f; std::vector<S> tau;
    // JacobiSVD(ComputeFullU) of the tall H_f_j = column-pivoted Householder QR preconditioner; its trailing
    // 2M-3 columns of Q are matrixU().rightCols(2M-3) (msckf.h:954-955).  colpiv_null = false: unpivoted reflectors.
    if (colpiv_null) householder_qr_colpiv_inplace(QR, tau); else householder_qr_inplace(QR, tau);
    if (mode == FAITHFUL) {                       // full U, rightCols, dense A^T H_x  (:954-957)
      Mat<S> Q = form_q_cols(QR, tau, 0, rows);
      A_j = Q.block(0, 3, rows, rows - 3);
      H_o_j = mul_atb(A_j, H_x);
    } else {
      A_j = form_q_cols(QR, tau, 3, rows - 3);
      Mat<S> QtH = H_x; apply_qt(QR, tau, QtH);
      H_o_j = QtH.block(3, 0, rows - 3, D);
    }
  }
  // R_o_j = A_j^T diag(u',v',u',v',...) A_j   (:423,:431)
  Mat<S> projectedNoise(const Mat<S>& A_j, int nObs) const {
    Mat<S> RA = A_j;
    for (int j = 0; j < RA.c; ++j) for (int i = 0; i < 2 * nObs; ++i) RA(i, j) *= (i % 2 == 0) ? uvar() : vvar();
    return mul_atb(A_j, RA);
  }
  // ---- :1103-1124
  bool gatingTest(const Mat<S>& H, const Mat<S>& r, int dof, double* gamma_out) const {
    Mat<S> P1;
    if (mode == FAITHFUL) { Mat<S> P = fullP(); P1 = mul_abt(mul(H, P), H); }
    else {  // only the cam columns of H are non-zero (:949)
      const int n = cam_covar_.r;
      Mat<S> Hc = H.block(0, 15, H.r, n);
      P1 = mul_abt(mul(Hc, cam_covar_), Hc);
    }
    for (int i = 0; i < P1.r; ++i) P1(i, i) += uvar();
    std::vector<S> rv(r.r); for (int i = 0; i < r.r; ++i) rv[i] = r(i, 0);
    std::vector<S> x = ldlt_solve(P1, rv);
    S gamma = 0; for (int i = 0; i < r.r; ++i) gamma += rv[i] * x[i];
    if (gamma_out) *gamma_out = (double)gamma;
    return gamma < chi_squared_test_table[dof + 1];
  }
  // ---- :1049-1098
  void findRedundantCamStates(std::vector<size_t>& rm) const {
    if (cam_states_.size() < 5) return;
    const S dist_thresh = msckf_params_.redundancy_distance_thresh, angle_thresh = msckf_params_.redundancy_angle_thresh;
    size_t kf = 0;
    V3<S> kf_pos = cam_states_[0].p_C_G; Quat<S> kf_q = cam_states_[0].q_CG;
    const size_t prot = cam_states_.size() - 3;
    size_t next = 1;
    while (next != prot) {
      S distance = norm(cam_states_[next].p_C_G - kf_pos);
      S angle = kf_q.angularDistance(cam_states_[next].q_CG);
      if (distance < dist_thresh && angle < angle_thresh) rm.push_back((size_t)cam_states_[next].state_id);
      else { kf = next; kf_pos = cam_states_[kf].p_C_G; kf_q = cam_states_[kf].q_CG; }
      ++next;
      int remaining = (int)cam_states_.size() - (int)rm.size();
      if (remaining <= msckf_params_.max_cam_states) break;
    }
    int over = ((int)cam_states_.size() - (int)rm.size()) - msckf_params_.max_cam_states;
    for (int i = 0; i < over; i++)
      if (rm.end() == std::find(rm.begin(), rm.end(), (size_t)cam_states_[i].state_id)) rm.push_back((size_t)cam_states_[i].state_id);
    if (rm.size() < 2) rm.clear();
    std::sort(rm.begin(), rm.end());
  }

  // ---- :1325-1423
  // K = P T_H^T S^-1, state injection and the Joseph-form covariance update (msckf.h:1368-1418) for a compressed
  // measurement (T_H, r_n, R_n)
  void applyUpdate(const Mat<S>& T_H, const Mat<S>& r_n, const Mat<S>& R_n, const Mat<S>& P) {
    const int D = T_H.c;
    Mat<S> PHt = mul_abt(P, T_H);                 // D x nr
    Mat<S> Smat = add(mul(T_H, PHt), R_n);        // :1369
    if (capture) { last_TH = T_H; last_rn = r_n; last_Rn = R_n; last_S = Smat; }
    Mat<S> K = mul(PHt, inverse(Smat));           // :1370
    Mat<S> dX = mul(K, r_n);                      // :1373
    last_deltaX = dX;
    auto seg = [&](int o) { return V3<S>{dX(o, 0), dX(o + 1, 0), dX(o + 2, 0)}; };
    imu_state_.q_IG = buildUpdateQuat(seg(0)) * imu_state_.q_IG;   // :1376-1383
    imu_state_.b_g = imu_state_.b_g + seg(3);
    imu_state_.b_a = imu_state_.b_a + seg(9);
    imu_state_.v_I_G = imu_state_.v_I_G + seg(6);
    imu_state_.p_I_G = imu_state_.p_I_G + seg(12);
    for (size_t c = 0; c < cam_states_.size(); ++c) {              // :1386-1391
      Quat<S> q = buildUpdateQuat(seg(15 + 6 * (int)c)) * cam_states_[c].q_CG;
      cam_states_[c].q_CG = q.normalized();
      cam_states_[c].p_C_G = cam_states_[c].p_C_G + seg(18 + 6 * (int)c);
    }
    Mat<S> tempMat = add(Mat<S>::identity(D), mul(K, T_H), S(-1));  // :1394-1396
    Mat<S> Pc = add(mul_abt(mul(tempMat, P), tempMat), mul_abt(mul(K, R_n), K));  // :1399
    symmetrize(Pc);                               // :1401-1403
    splitP(Pc);
  }

  // GRAM mode: [T_H | r_n] = chol([H_o | r_o]^T [H_o | r_o]) over the camera columns, accumulated and factored in
  // double whatever S is; a pivot below 64 eps of its original diagonal is an unobservable direction of the stack
  // and gives a zero row.
  void measurementUpdateGram(const Mat<S>& H_o, const Mat<S>& r_o) {
    const int m = H_o.r, D = H_o.c, n = D - 15;
    std::vector<double> A((size_t)(n + 1) * (n + 1), 0.0);
    auto at = [&](int i, int j) -> double& { return A[(size_t)i * (n + 1) + j]; };
    for (int row = 0; row < m; ++row) {
      std::vector<double> x(n + 1);
      for (int c = 0; c < n; ++c) x[c] = (double)H_o(row, 15 + c);
      x[n] = (double)r_o(row, 0);
      for (int i = 0; i <= n; ++i) { if (x[i] == 0.0) continue; for (int j = 0; j <= i; ++j) at(i, j) += x[i] * x[j]; }
    }
    std::vector<double> d0(n);
    for (int k = 0; k < n; ++k) d0[k] = at(k, k);
    Mat<S> T_H(n, D), r_n(n, 1), R_n(n, n);
    int rank = 0;
    const double tol = 64.0 * 2.220446049250313e-16;
    for (int k = 0; k < n; ++k) {
      const double p = at(k, k);
      if (!(p > tol * d0[k])) continue;                       // zero row of T_H
      ++rank;
      const double dinv = 1.0 / std::sqrt(p);
      std::vector<double> l(n + 1, 0.0);
      for (int i = k; i <= n; ++i) l[i] = at(i, k) * dinv;
      for (int i = k; i < n; ++i) T_H(k, 15 + i) = (S)l[i];
      r_n(k, 0) = (S)l[n];
      for (int i = k + 1; i <= n; ++i) for (int j = k + 1; j <= i; ++j) at(i, j) -= l[i] * l[j];
    }
    for (int k = 0; k < n; ++k) R_n(k, k) = uvar();
    last_stats.r_rows = rank;
    applyUpdate(T_H, r_n, R_n, fullP());
  }

  void measurementUpdate(const Mat<S>& H_o, const Mat<S>& r_o, const Mat<S>* R_o_dense,
                         const std::vector<Mat<S>>* R_blocks, const std::vector<int>* R_off) {
    const int m = H_o.r;
    last_stats.m_rows = m;
    if (m == 0) return;
    const int D = H_o.c;
    if (capture) { last_Ho = H_o; last_ro = r_o; }
    if (mode == GRAM && (uvar() == vvar())) { measurementUpdateGram(H_o, r_o); return; }
    Mat<S> P = fullP();
    Mat<S> QR = H_o; std::vector<S> tau;
    householder_qr_inplace(QR, tau, tiny_row_tol);              // :1343 (tiny_row_tol > 0: thresholded zero-tail rule)
    const int steps = std::min(m, D);
    std::vector<int> kept;                        // nonZeroRows of the upper-triangular view :1345-1348
    S rmax = 0;
    if (tiny_row_tol > 0) for (int r = 0; r < steps; ++r) for (int c = r; c < D; ++c) rmax = std::max(rmax, (S)std::fabs(QR(r, c)));
    for (int r = 0; r < steps; ++r) {
      bool any = false;
      for (int c = r; c < D && !any; ++c) any = tiny_row_tol > 0 ? (std::fabs(QR(r, c)) > S(tiny_row_tol) * rmax) : (QR(r, c) != S(0));
      if (any) kept.push_back(r);
    }
    const int nr = (int)kept.size();
    last_stats.r_rows = nr;
    Mat<S> T_H(nr, D), r_n(nr, 1), R_n(nr, nr);
    for (int k = 0; k < nr; ++k) for (int c = kept[k]; c < D; ++c) T_H(k, c) = QR(kept[k], c);
    const bool iso = (uvar() == vvar());
    if (mode == FAITHFUL) {
      Mat<S> Q = form_q_cols(QR, tau, 0, m);      // full m x m Q :1344
      Mat<S> Q1(m, nr);
      for (int k = 0; k < nr; ++k) for (int i = 0; i < m; ++i) Q1(i, k) = Q(i, kept[k]);
      r_n = mul_atb(Q1, r_o);                     // :1365
      R_n = mul_atb(Q1, mul(*R_o_dense, Q1));     // :1366
    } else {
      Mat<S> qtr = r_o; apply_qt(QR, tau, qtr);
      for (int k = 0; k < nr; ++k) r_n(k, 0) = qtr(kept[k], 0);
      if (iso) { for (int k = 0; k < nr; ++k) R_n(k, k) = uvar(); }
      else {
        Mat<S> Qs = form_q_cols(QR, tau, 0, steps);
        Mat<S> Q1(m, nr);
        for (int k = 0; k < nr; ++k) for (int i = 0; i < m; ++i) Q1(i, k) = Qs(i, kept[k]);
        Mat<S> RQ(m, nr);
        for (size_t b = 0; b < R_blocks->size(); ++b) {
          const Mat<S>& Rb = (*R_blocks)[b]; const int o = (*R_off)[b];
          RQ.set_block(o, 0, mul(Rb, Q1.block(o, 0, Rb.r, nr)));
        }
        R_n = mul_atb(Q1, RQ);
      }
    }
    applyUpdate(T_H, r_n, R_n, P);
  }
};

}  // namespace oracle
#endif
