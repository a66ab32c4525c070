/* msckf_hip.h -- C-ABI of the MI355X-native batched MSCKF filter core (libmsckf_hip.so).
 *
 * Drop-in boundary for the EKF hot path of daniilidis-group/msckf_mono.  The reference has no FFI layer:
 * its operator boundary is the public interface of the header-only class template
 *     msckf_mono::MSCKF<_S>            (include/msckf_mono/msckf.h:66-848)
 * instantiated as MSCKF<float> by src/ros_interface.cpp:80 and datasets/asl_msckf.cpp:57.  Each entry point
 * below replaces one member function of that class for ONE trajectory `b` of a batch handle (the reference
 * object is the B = 1 case); include/msckf_mono/msckf.h in this repository is the C++ shim that keeps the
 * reference's class/member names on top of these calls.
 *
 * Conventions: plain pointers and sizes only; every scalar crosses the boundary as double (narrowed to
 * the handle's dtype inside); matrices are column-major; quaternions are (w,x,y,z) in Eigen's Hamilton
 * convention; return value 0 = success, negative = -errno style failure (never throws; the reference
 * reports nothing at all, msckf.h:328,1405-1409).  A handle is bound to one HIP device and one stream; calls
 * on one handle must not be concurrent (the reference object is not re-entrant either).
 *
 * Packed argument layouts
 *   cam12    c_u c_v f_u f_v b | q_CI(w,x,y,z) | p_C_I(3)                               types.h:48-55
 *   noise29  u_var_prime v_var_prime | diag(Q_imu)(12) | diag(initial_imu_covar)(15)   types.h:86-92
 *            (every caller of the reference passes diagonal matrices: asl_msckf.cpp:86-108)
 *   params8  max_gn_cost_norm min_rcond translation_threshold redundancy_angle_thresh
 *            redundancy_distance_thresh min_track_length max_track_length max_cam_states types.h:94-99
 *   imu29    q_IG(4) b_g(3) v_I_G(3) b_a(3) p_I_G(3) g(3) q_IG_null(4) v_I_G_null(3) p_I_G_null(3)
 *                                                                                       types.h:69-76
 *   reading7 omega(3) a(3) dT                                                           types.h:78-84
 *   cam7     q_CG(4) p_C_G(3)                                                           types.h:57-67
 *
 * Pixel noise: with u_var_prime == v_var_prime (the configuration the throughput metric is quoted on) the update is
 * independent of the null-space basis and of the compression order and matches the reference to rounding.  With
 * u_var_prime != v_var_prime (EuRoC intrinsics, asl_msckf.cpp:77-78) the library builds the reference's own construction on
 * the device (msckf_hip_set_anisotropic_noise, mode 0, default): R_o_j = A_j^T R_j A_j with A_j the trailing columns of the
 * column-pivoted Householder Q of H_f_j (= JacobiSVD's trailing U columns, msckf.h:954-955), HouseholderQR of the stack in
 * the reference's column and row order with the zero-tail rule (rows 0..14 of H_o pass verbatim), R_n = Q_1^T R_o Q_1
 * (msckf.h:423-431, 1343-1366).  The reference's own result there is not reproducible beyond ~1e-4 in the biases per
 * update: a column that depends on the previous ones (the window's gauge directions) has a rounding-level tail, the
 * reflector built from it is rounding noise, and its row enters R_n with O(1) weight (measured with the reference's own
 * source under two roundings, tests/test_ref_vs_oracle.py).  tail_tol > 0 treats such a tail (below tail_tol * |column|) as
 * the zero it is in exact arithmetic -- the reference's algorithm in its exact-arithmetic limit, reproducible to rounding
 * and as close to either rounding of the reference as they are to each other; tail_tol = 0 is msckf.h's rule to the
 * letter (where rows exist below the first 15 + 6N the tail comes out of an f64 Gram matrix, which resolves it to ~1e-4 of
 * the column at best: tail_tol is taken no finer than 3e-4 there).  Mode 1 pre-whitens every observation row by 1/sigma_u resp. 1/sigma_v and runs the filter with unit noise (the
 * full generalized-least-squares update: statistically the better estimator, but not the reference's).
 */
#ifndef MSCKF_HIP_H
#define MSCKF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct msckf_hip_batch* msckf_hip_handle;

/* MSCKF_HIP_F16H_F32P (BASELINE.json configs[4]): state, covariance and every accumulation in f32 (Gram matrix in f64),
 * the 2 x 6 measurement Jacobian blocks H_x rounded to and stored as fp16. */
enum { MSCKF_HIP_F32 = 0, MSCKF_HIP_F64 = 1, MSCKF_HIP_F16H_F32P = 2 };

/* ---- lifecycle --------------------------------------------------------------------------------------- */
/* B trajectories; capacity n_cap camera states (>= max(max_cam_states, max_track_length)+1),
 * f_cap tracks per update, m_cap observations per track (<= 64).  replaces: MSCKF() ctor, msckf.h:69. */
int msckf_hip_create(int B, int n_cap, int f_cap, int m_cap, int dtype, int device, msckf_hip_handle* out);
int msckf_hip_destroy(msckf_hip_handle h);
/* Value semantics of the reference object (MSCKF<_S> is copyable, msckf.h:31-67): dst <- the filter state of every trajectory
 * of src (states, parameters, covariance, counters, flags, host-side track bookkeeping, mode switches).  Both handles must
 * have been created with the same B, capacities and dtype; work buffers and a resident scenario are not copied. */
int msckf_hip_copy_state(msckf_hip_handle dst, msckf_hip_handle src);
const char* msckf_hip_last_error(void);

/* ---- the reference's public member functions, per trajectory b ---------------------------------------- */
/* MSCKF::initialize(camera, noise_params, msckf_params, imu_state)                   msckf.h:72-97  */
int msckf_hip_initialize(msckf_hip_handle h, int b, const double* cam12, const double* noise29,
                         const double* params8, const double* imu29);
/* MSCKF::propagate(imuReading&), K consecutive readings fused into one launch        msckf.h:101-145 */
int msckf_hip_propagate(msckf_hip_handle h, int b, const double* readings7, int K);
/* MSCKF::augmentState(state_id, time)                                                msckf.h:148-212 */
int msckf_hip_augment_state(msckf_hip_handle h, int b, int state_id, double time);
/* MSCKF::update(measurements, feature_ids)   (host-side track bookkeeping)           msckf.h:215-299 */
int msckf_hip_update(msckf_hip_handle h, int b, const double* meas2, const uint64_t* ids, int n);
/* MSCKF::addFeatures(features, feature_ids)  (host-side track bookkeeping)           msckf.h:302-332 */
int msckf_hip_add_features(msckf_hip_handle h, int b, const double* meas2, const uint64_t* ids, int n);
/* MSCKF::marginalize()                                                               msckf.h:336-449 */
int msckf_hip_marginalize(msckf_hip_handle h, int b);
/* MSCKF::pruneEmptyStates()                                                          msckf.h:685-761 */
int msckf_hip_prune_empty_states(msckf_hip_handle h, int b);
/* MSCKF::pruneRedundantStates(): keyframe selection + observation surgery on the host, triangulation of
 * not-yet-initialized features and the second update on the device                   msckf.h:453-682 */
int msckf_hip_prune_redundant_states(msckf_hip_handle h, int b);
/* MSCKF::finish()                                                                    msckf.h:765-807 */
int msckf_hip_finish(msckf_hip_handle h, int b);
/* getters, msckf.h:810-848 */
int msckf_hip_get_num_cam_states(msckf_hip_handle h, int b);                 /* getNumCamStates :810 */
int msckf_hip_get_imu_state(msckf_hip_handle h, int b, double* imu29);       /* getImuState     :815 */
int msckf_hip_get_cam_states(msckf_hip_handle h, int b, double* cam7, int* state_ids, int cap); /* getCamStates :835 */
int msckf_hip_get_map(msckf_hip_handle h, int b, double* xyz, int cap);      /* getMap :820; returns count */
int msckf_hip_get_pruned_state_ids(msckf_hip_handle h, int b, int* ids, int cap);  /* getPrunedStates :840 */
/* getPrunedStates() :840-848 in full: the camState of every pruned state as it was when pruned (pose after the
 * last update that touched it, time, ids; types.h:57-67), sorted by state_id as the reference sorts them.  Any output
 * pointer may be NULL.  Read by asl_msckf.cpp:409-424 (pruned-camera path).  Returns the count. */
int msckf_hip_get_pruned_states(msckf_hip_handle h, int b, double* cam7, double* time, int* state_ids,
                                int* last_correlated_ids, int cap);
/* the non-pose members of getCamStates()[i] (types.h:57-67): time, tracked_feature_ids.size(), last_correlated_id
 * (asl_msckf.cpp:384-388 publishes them); host-side bookkeeping, no device access.  Returns the count. */
int msckf_hip_get_cam_meta(msckf_hip_handle h, int b, double* time, int* n_tracked, int* last_correlated_ids, int cap);
/* getCamState(i).tracked_feature_ids :830 */
int msckf_hip_get_tracked_feature_ids(msckf_hip_handle h, int b, int cam_index, uint64_t* ids, int cap);

/* ---- additive accessors (the reference keeps these private, msckf.h:52-54; needed for parity tests) ---- */
int msckf_hip_get_covariance(msckf_hip_handle h, int b, double* P, int ld);  /* D x D, D = 15 + 6 N */
int msckf_hip_set_covariance(msckf_hip_handle h, int b, const double* P, int D);
int msckf_hip_set_imu_state(msckf_hip_handle h, int b, const double* imu29);
int msckf_hip_set_cam_pose(msckf_hip_handle h, int b, int slot, const double* cam7);
int msckf_hip_get_num_residualized(msckf_hip_handle h, int b, long long* n);
int msckf_hip_set_num_residualized(msckf_hip_handle h, int b, long long n);
/* last marginalize: out[0..6] = n_tracks, motion-rejected, triangulation-rejected, gate-rejected, passed,
 * stacked rows m, kept rows r.  Also the place where the trajectory's sticky device-side error flags surface (out7 is
 * filled regardless): -EOVERFLOW camera-state capacity exceeded in augmentState; -EDOM a factorization of
 * S = T_H P T_H^T + R_n (msckf.h:1369) met a non-positive pivot since the flags were last cleared -- the covariance lost
 * positive definiteness (the pivot is clamped and the run continues; the reference's explicit S.inverse() would return
 * garbage silently).  (No kernel waits for another workgroup without a fall-back: the in-place prune is one workgroup per
 * trajectory, the gain solve's shared product is re-formed locally when a sibling does not show up.) */
int msckf_hip_last_stats(msckf_hip_handle h, int b, int* out7);
int msckf_hip_clear_error_flags(msckf_hip_handle h, int b);
/* the sticky flags themselves, without turning them into a return code: bit 0 capacity overflow (-EOVERFLOW above), bit 1
 * non-positive pivot (-EDOM above).  For callers that poll many trajectories and want the statistics either way. */
int msckf_hip_get_error_flags(msckf_hip_handle h, int b, int* flags);
/* per track of the last marginalize: out[t*8 + ..] = motion_ok tri_valid gate_pass included gamma p_f_G(3) */
int msckf_hip_last_tracks(msckf_hip_handle h, int b, double* out8, int cap);
int msckf_hip_last_deltax(msckf_hip_handle h, int b, double* dx, int cap);

/* ---- batched path: work-lists ("front-end track dump": positional camera slots + normalized coords) ---- */
/* Replace trajectory b's work-list by F tracks; slots/obs are flattened over tracks (sum(M) entries).
 * This is what MSCKF::update() hands to marginalize() as feature_tracks_to_residualize_ (msckf.h:249-262). */
int msckf_hip_set_tracks(msckf_hip_handle h, int b, int F, const int* M, const int* slots, const double* obs2);
/* propagate / augment / marginalize / prune for the trajectory range [b0, b0+nb) in single launches */
int msckf_hip_propagate_range(msckf_hip_handle h, int b0, int nb, const double* readings7, int K); /* [nb][K][7] */
int msckf_hip_augment_range(msckf_hip_handle h, int b0, int nb);
int msckf_hip_marginalize_range(msckf_hip_handle h, int b0, int nb);
int msckf_hip_drop_oldest_range(msckf_hip_handle h, int b0, int nb, int n_drop);
/* The ASL runner's per-image cycle (datasets/asl_msckf.cpp:269-294: augmentState, update, addFeatures, marginalize,
 * pruneRedundantStates :289, pruneEmptyStates) for trajectories b0 .. b0 + nb - 1 IN LOCKSTEP: the feature bookkeeping of every
 * trajectory (msckf.h:215-332, 453-534, 685-717, 1049-1098) runs on the host as in the per-filter entries, every device stage is one
 * launch sequence over the range and every read-back (poses for findRedundantCamStates, triangulated points, pruned states' poses)
 * one copy + one wait for the whole range.  Same results per trajectory as msckf_hip_augment_state / _update / _add_features /
 * _marginalize / _prune_redundant_states / _prune_empty_states called filter by filter, bit for bit.  IMU samples go in beforehand
 * through msckf_hip_propagate_range.  state_ids[nb], times[nb] (may be null): augmentState's arguments; upd_* / new_*: update()'s
 * and addFeatures()'s arguments of the trajectories, concatenated (upd_n[i] / new_n[i] entries each; normalized coordinates, 2 per id).
 * flags: 1 = pruneRedundantStates, 2 = pruneEmptyStates. */
int msckf_hip_image_cycle_range(msckf_hip_handle h, int b0, int nb, const int* state_ids, const double* times,
                                const double* upd_meas2, const uint64_t* upd_ids, const int* upd_n,
                                const double* new_meas2, const uint64_t* new_ids, const int* new_n, int flags);

/* ---- batched path: HBM-resident scenario (inputs uploaded once, then frames run without host syncs) ---- */
int msckf_hip_scenario_alloc(msckf_hip_handle h, int n_frames, int K);
/* stage one (frame, trajectory) cell on the host side of the handle */
int msckf_hip_scenario_set(msckf_hip_handle h, int frame, int b, const double* readings7 /*[K][7]*/, int F,
                           const int* M, const int* slots, const double* obs2, int n_drop);
int msckf_hip_scenario_commit(msckf_hip_handle h);   /* H2D of everything staged */
/* one filter update per trajectory per frame: K x propagate + augmentState + marginalize + prune, for
 * frames [f0, f1), asynchronously on the handle's stream.  Results do not depend on how a range is cut into calls.
 * (In the square-root gain form the prune of every frame but a call's last rides on the covariance downdate, which
 * writes the pruned covariance into the handle's second buffer -- one launch less per frame; environment
 * MSCKF_HIP_FUSE_PRUNE=0 at create time keeps the separate prune launch, for A/B runs.) */
int msckf_hip_run_frames(msckf_hip_handle h, int f0, int f1);
/* The same frames with the inputs handed over per frame, as the reference's callers do (IMU samples and the image's
 * tracks arrive with the image, asl_msckf.cpp:227-284): frame f's IMU samples and its COMPACT work-list (sum M_j slot /
 * observation entries + per-track offsets, not padded [f_cap][m_cap] rows) are copied from page-locked host memory into one
 * of `depth` device staging sets on a copy stream, up to depth - 1 frames ahead of the kernels that read them.  Results are
 * bit-identical to msckf_hip_run_frames; the difference is the PCIe leg inside the timed region. */
int msckf_hip_run_frames_streamed(msckf_hip_handle h, int f0, int f1);
/* Build the page-locked per-frame blocks of frames [f0, f1) (and size the device staging ring) ahead of time;
 * msckf_hip_run_frames_streamed does it on first use otherwise.  Only frames that are streamed are ever page-locked;
 * -ENOMEM leaves msckf_hip_run_frames on the resident scenario unaffected. */
int msckf_hip_scenario_pin(msckf_hip_handle h, int f0, int f1);
/* Staging ring of msckf_hip_run_frames_streamed: depth 2..8 (default 6); mode 0 (default) hands frames over between the
 * uploading thread and the slices' enqueue threads on the HOST (no stream waits for another stream's event on the device),
 * mode 1 uses hipStreamWaitEvent hand-overs.  Same results. */
int msckf_hip_set_upload_ring(msckf_hip_handle h, int depth, int mode);
int msckf_hip_sync(msckf_hip_handle h);
/* HIP-event stage timing: enable, run, sync, then read accumulated milliseconds and launch counts for
 * stages 0 propagate, 1 augment, 2 k_feature, 3 compression A (k_gram | TSQR stage 1), 4 compression B
 * (k_chol_mfma | TSQR merge), 5 kalman, 6 prune, 7 k_select (information form: k_select_diag, which also reduces the
 * block-diagonal part of the Gram matrix).  (While profiling, run_frames launches propagate and
 * augmentState separately; otherwise they share one launch.) */
int msckf_hip_profile_enable(msckf_hip_handle h, int on);
int msckf_hip_profile_read(msckf_hip_handle h, double* ms8, int* count8);
/* The same with the stages beyond the first eight (cap >= 8 entries are written, unknown ones as zero): 8 k_lit_pre, 9 k_lit_gamma,
 * 10 k_literal -- the three launches of the literal anisotropic compression (u_var' != v_var', msckf.h:423-431, 1343-1366;
 * kernels_literal.hip), which a profiled run brackets one by one (stage 3 is then k_gram alone). */
int msckf_hip_profile_read_ex(msckf_hip_handle h, double* ms, int* count, int cap);
/* Milliseconds an event pair with nothing between its two records reads on the handle's stream (mean of 64 pairs): every
 * stage timer above brackets its launches with such a pair, so a single-kernel stage reads the kernel's duration plus
 * this.  (The reference's StageTiming message, asl_msckf.cpp:229-296, is host wall-clock and has no such term.) */
int msckf_hip_profile_event_overhead(msckf_hip_handle h, double* ms);
/* run_frames on n = 1..8 HIP streams: the batch is cut into n slices of independent trajectories that run the
 * same kernel sequence concurrently (latency-bound stages of one slice overlap chip-filling stages of another). */
int msckf_hip_set_streams(msckf_hip_handle h, int n);
/* Host cores for the threads of msckf_hip_run_frames / _streamed: cpus[0] for the calling thread while it uploads frames
 * (restored on return), cpus[1 + i] for the enqueue thread of slice i.  The hand-overs between them are spin waits; without
 * this the threads run wherever the scheduler puts them (n = 0 clears the list).  One core each, ideally on the GPU's NUMA
 * node. */
int msckf_hip_set_host_affinity(msckf_hip_handle h, const int* cpus, int n);
/* Exact early accept of the chi-square gate (gatingTest, msckf.h:1103-1124), OFF by default: S = H_o P H_o^T + sigma^2 I
 * >= sigma^2 I, so gamma <= |r_o|^2 / sigma^2; when that bound is below half the threshold the track passes without
 * forming S.  Same decisions as the reference; the reported gamma of such a track is the bound (status bit 32). */
int msckf_hip_set_gate_early_accept(msckf_hip_handle h, int on);
/* Compression of the stacked Jacobian (HouseholderQR + Q_1^T r_o of measurementUpdate, msckf.h:1338-1366):
 * -1 default for the window size, 0 Householder TSQR (kernels_qr.hip), 3 information form: [T | r_n] = chol(H_o^T H_o),
 * accumulated in f64 on the matrix cores (kernels_gram.hip) and factored by the blocked matrix-core Cholesky k_chol_mfma
 * (kernels_chol.hip; the default; two levels for windows of more than 31 cameras, 6 n_cap + 1 > 192).  1 and 2 selected the
 * register-resident factorizations of rounds 1-2, which are gone: they are accepted and mean 3.  The information form needs
 * 6 n_cap + 1 <= 384 and f_cap <= 1024 (-ENOTSUP otherwise).  Both routes give the reference's update (tests keep them
 * together). */
int msckf_hip_set_compression(msckf_hip_handle h, int route);
/* Covariance update of measurementUpdate (msckf.h:1368-1418): 0 (default) the square-root gain form -- S = L L^T,
 * W = P T_H^T L^-T, dx = W L^-1 r_n, P <- P - W W^T (= (I - K T_H) P, written symmetrically); 1 the reference's literal
 * Joseph sequence K, (I - K T_H) P (I - K T_H)^T + K R_n K^T, symmetrise; 2 the square-root gain form with the
 * register-resident solve of round 1 instead of the blocked matrix-core one.  Identical in exact arithmetic, equal to
 * rounding in the tests. */
int msckf_hip_set_covariance_update(msckf_hip_handle h, int form);
/* run_frames: launch a frame's per-track kernel on a side stream concurrently with the same frame's propagate + augmentState
 * whenever no track of the frame observes the camera that augmentState adds (it then depends only on what the previous
 * frame left behind).  Bit-identical results; OFF by default -- on MI355X at the benchmark configuration the per-track
 * kernel starves the latency-bound propagate/augment workgroups of CUs (100 k -> 82 k updates/s). */
int msckf_hip_set_feature_overlap(msckf_hip_handle h, int on);
/* Anisotropic pixel noise, u_var_prime != v_var_prime (see the header comment): mode 0 (default) the reference's
 * R_o_j = A_j^T R_j A_j / HouseholderQR in column order / R_n = Q_1^T R_o Q_1 on the device (msckf.h:423-431, 1343-1366;
 * f64: the Householder sweep for its decisions, the result as a projection onto range(Q_1), kernels_literal.hip), handed to the
 * update as the information matrix [T_H | r_n]^T R_n^-1 [T_H | r_n]; mode 1 rows pre-whitened by 1/sigma (generalized least squares).  tail_tol: zero-tail tolerance of mode 0, < 0 = default (1e-10 double, 8e-4 float), 0 = the reference's
 * rule to the letter.  Applies to every trajectory of the handle, initialized or not.  -ENOMEM when the work space does not fit
 * (per trajectory about sixteen (6 n_cap)^2 matrices of doubles, the (2 * 6 n_cap + 16)^2 elimination matrix, and per TRACK six rows of
 * ldR doubles -- f_cap * 6 * ldR * 8 bytes, 1.8 MB at 200 tracks and a 30-camera window); -ENOTSUP on a device that does not grant
 * the route's kernels their 94 KB of LDS per workgroup (gfx950 does). */
int msckf_hip_set_anisotropic_noise(msckf_hip_handle h, int mode, double tail_tol);
/* last marginalize of trajectory b on the literal route: out[0..5] = stacked rows m, kept rows r of R (msckf.h:1347),
 * Householder steps that reflected, steps whose non-zero tail fell under tail_tol, route taken (3: the sequence of steps on
 * the compressed representation -- first 15 + 6N rows explicit, the rest through their Gram matrix; 2: the sweep over the
 * dense stack, environment MSCKF_HIP_LITERAL_ROUTE=1 at create time), leading steps that meet the zero IMU columns (15),
 * kept rows that a dependent column handed through in the middle of the sweep (route 3); out[7] reserved. */
int msckf_hip_literal_info(msckf_hip_handle h, int b, int* out8);

#ifdef __cplusplus
}
#endif
#endif
