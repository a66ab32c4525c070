// msckf_hip.hip -- host side of libmsckf_hip.so: batch handle, track bookkeeping, C-ABI (include/msckf_hip.h).
//
// The host keeps exactly what the reference keeps on the host side of its hot loop -- the integer /
// std::find bookkeeping of MSCKF::update / addFeatures / removeTrackedFeature / pruneEmptyStates
// (msckf.h:215-332, 685-717, 1469-1485) -- and turns it into positional work-lists for the device.
// Everything numerical (msckf.h:101-212, 336-449, 905-1423) runs in the HIP kernels of kernels_*.hip; there
// is no CPU fallback: if no HIP device is usable msckf_hip_create fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cerrno>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <dlfcn.h>
#include <cstring>
#include <string>
#include <vector>
#include <chrono>
#include <unordered_map>
#include <unordered_set>

#include "../../include/msckf_hip.h"
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <pthread.h>
#include <sched.h>
#include "dev_common.h"


namespace {
using namespace msckf;

thread_local std::string g_err;
int fail(int code, const std::string& msg) { g_err = msg; return code; }

// rocTX ranges under the reference's stage names (asl_msckf.cpp:229-296: imu_prop, msckf_augment_state, msckf_update,
// msckf_add_features, msckf_marginalize, msckf_prune_redundant, msckf_prune_empty_states) around the host side of every stage,
// so that a `rocprofv3 --marker-trace` timeline of a caller reads like the reference's StageTiming message.  Off unless
// MSCKF_HIP_ROCTX=1; the marker library is opened at run time (no link dependency).
struct Roctx {
  int (*push)(const char*) = nullptr; int (*pop)() = nullptr;
  Roctx() {
    const char* e = getenv("MSCKF_HIP_ROCTX");
    if (!e || !atoi(e)) return;
    void* h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return;
    push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
    pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
    if (!push || !pop) { push = nullptr; pop = nullptr; }
  }
};
const Roctx& roctx() { static Roctx r; return r; }
struct StageRange {
  bool on;
  explicit StageRange(const char* name) : on(roctx().push != nullptr) { if (on) roctx().push(name); }
  ~StageRange() { if (on) roctx().pop(); }
};

#define HIPCHK(expr)                                                                         \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess) return fail(-EIO, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

struct CamMeta { int state_id; double time; int last_correlated_id; std::vector<uint64_t> tracked; };
struct PrunedState { int state_id; double time; int last_correlated_id; double pose[7]; };   // camState at the moment it was pruned (msckf.h:631,714)
struct Track { uint64_t id; std::vector<double> obs; std::vector<int> cam_ids; bool initialized = false; double p_f_G[3] = {0, 0, 0}; };
struct TrackToResid { uint64_t id; std::vector<double> obs; std::vector<int> slots; };
struct HostTraj {
  bool initialized = false;
  int max_cam_states = 0, min_track_length = 0, max_track_length = 0;
  double redundancy_angle_thresh = 0, redundancy_distance_thresh = 0;
  std::vector<CamMeta> cams;
  std::vector<Track> tracks;
  std::vector<uint64_t> tracked_ids;
  std::vector<TrackToResid> to_resid;
  std::vector<PrunedState> pruned;
  std::vector<double> map;   // xyz triples of the last marginalize
  int map_pending = 0;       // > 0: the last marginalize()'s triangulated points of this many tracks are still on the device
                             // (read back when somebody asks -- getMap(), pruneRedundantStates(), a copy -- or dropped by the next marginalize())
  int wl_F = 0;              // tracks in the device work-list of this trajectory
};

struct BatchBase {
  virtual ~BatchBase() {}
  int B = 0, n_cap = 0, f_cap = 0, m_cap = 0, dtype = 0, device = 0;
  bool h16 = false;   // dtype MSCKF_HIP_F16H_F32P: fp16 measurement Jacobian, f32 state and covariance
  std::vector<HostTraj> traj;
  virtual int init(int b, const double* cam, const double* noise, const double* params, const double* imu) = 0;
  virtual int propagate(int b0, int nb, const double* rd, int K, bool mirror = false) = 0;
  virtual int augment(int b0, int nb) = 0;
  virtual int set_tracks(int b, int F, const int* M, const int* slots, const double* obs) = 0;
  virtual int marginalize(int b0, int nb) = 0;
  virtual int set_given_positions(int b, int F, const double* pf3) = 0;   // mode-1 work-list: stored p_f_G per track
  virtual int feature_only(int b, int* status, double* pf3, int cap) = 0;  // checkMotion + triangulation of the work-list
  virtual int marginalize_given(int b) = 0;                               // second update of pruneRedundantStates
  virtual int prune_keep(int b, const std::vector<int>& keep) = 0;
  // range forms for the batched image cycle (host_image_cycle): one copy / one launch for trajectories b0 .. b0 + nb - 1
  struct WorkList { std::vector<int> M, slots; std::vector<double> obs; };                     // one trajectory's tracks to residualize (set_tracks' arguments)
  virtual int set_tracks_range(int b0, int nb, const std::vector<WorkList>& wl) = 0;           // every trajectory's list in one pinned block, two copies
  virtual int cams_range(int b0, int nb, double* poses7) = 0;                                  // [nb][n_cap][7], one read + one wait
  virtual int feature_only_range(int b0, int nb, int* status, double* pf3, bool launch) = 0;   // [nb][f_cap], [nb][f_cap][3]; launch = false: only read what the last launch left
  virtual int set_given_range(int b0, int nb, const double* pf3) = 0;                          // [nb][f_cap][3]
  virtual int marginalize_given_range(int b0, int nb) = 0;
  virtual int prune_keep_range(int b0, int nb, const std::vector<std::vector<int>>& keep) = 0; // keep[i]: ascending slots of trajectory b0 + i
  virtual int drop_oldest(int b0, int nb, int n) = 0;
  virtual int get_ncam(int b, int* n) = 0;
  virtual int ncam_host(int b) const = 0;   // the window size from the host's own count (kept through every entry that changes it): no device read
  virtual int get_imu(int b, double* o) = 0;
  virtual int set_imu(int b, const double* in) = 0;
  virtual int get_cams(int b, double* o, int cap, int* n) = 0;
  virtual int get_cams_known(int b, double* o, int n) = 0;   // n known to the caller: cameras + (into the host copy) the IMU state, one wait
  virtual int set_cam(int b, int slot, const double* in) = 0;
  virtual int get_cov(int b, double* P, int ldo) = 0;
  virtual int set_cov(int b, const double* P, int D) = 0;
  virtual int get_nres(int b, long long* n) = 0;
  virtual int set_nres(int b, long long n) = 0;
  virtual int stats(int b, int* out) = 0;
  virtual int track_info(int b, double* out, int cap) = 0;
  virtual int deltax(int b, double* out, int cap) = 0;
  virtual int scen_alloc(int n_frames, int K) = 0;
  virtual int scen_set(int f, int b, const double* rd, int F, const int* M, const int* slots, const double* obs, int n_drop) = 0;
  virtual int scen_commit() = 0;
  virtual int run_frames(int f0, int f1) = 0;
  virtual int run_frames_streamed(int f0, int f1) = 0;
  virtual int scen_pin(int f0, int f1) = 0;
  virtual int set_upload_ring(int depth, int mode) = 0;
  virtual int sync() = 0;
  virtual int prof_enable(int on) = 0;
  virtual int prof_read(double* ms, int* cnt, int cap) = 0;
  virtual int prof_event_overhead(double* ms) = 0;
  virtual int set_streams(int n) = 0;
  virtual int set_host_affinity(const int* cpus, int n) = 0;
  virtual int set_gate_early(int on) = 0;
  virtual int set_compression(int route) = 0;
  virtual int set_cov_update(int form) = 0;
  virtual int set_feature_overlap(int on) = 0;
  virtual int clear_stats(int b) = 0;
  virtual int clear_errors(int b) = 0;
  virtual int set_aniso(int mode, double tol) = 0;
  virtual int error_flags(int b, int* flags) = 0;
  virtual int copy_from(BatchBase* src) = 0;
  virtual int lit_info(int b, int* out8) = 0;
};

int resolve_map(BatchBase* B, int b);   // (defined with the host-side bookkeeping below)
constexpr int NSTAGE = 11;   // 0..7: msckf_hip_profile_read; 8 k_lit_pre, 9 k_lit_gamma, 10 k_literal (msckf_hip_profile_read_ex)

// Persistent enqueue threads of a batch (one per slice of run_frames / run_frames_streamed): a K-frame window is a few
// milliseconds, creating and joining three std::threads per call was 1-2 % of it.
struct Workers {
  std::vector<std::thread> th;
  std::mutex m;
  std::condition_variable cv, cv_done;
  std::function<void(int)> job;
  unsigned long gen = 0;
  int active = 0, pending = 0;
  bool stop = false;
  // worker idx runs on cpus[idx + 1] (cpus[0] is the calling thread's: the uploader of run_frames_streamed) when a list was
  // given (msckf_hip_set_host_affinity): the hand-overs between the uploading thread and the slices' enqueue threads are
  // spin waits, and a waiter that the scheduler moves or parks costs a frame's worth of time
  std::vector<int> cpus;
  // cpu >= 0: pin the calling worker to it (its original mask is saved on the first pin); cpu < 0: back to the original mask
  static void pin_self(int cpu, cpu_set_t* orig, bool* have_orig) {
    if (cpu < 0) {
      if (*have_orig) (void)pthread_setaffinity_np(pthread_self(), sizeof(*orig), orig);
      return;
    }
    if (!*have_orig) { if (pthread_getaffinity_np(pthread_self(), sizeof(*orig), orig) == 0) *have_orig = true; }
    cpu_set_t set; CPU_ZERO(&set); CPU_SET(cpu, &set);
    (void)pthread_setaffinity_np(pthread_self(), sizeof(set), &set);
  }
  void loop(int idx) {
    unsigned long seen = 0;
    int pinned = -2;
    cpu_set_t orig; bool have_orig = false;
    for (;;) {
      std::function<void(int)> f;
      {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return stop || gen != seen; });
        if (stop) return;
        seen = gen;
        if (idx >= active) continue;
        f = job;
        const int want = idx + 1 < (int)cpus.size() ? cpus[idx + 1] : -1;
        if (want != pinned) { pin_self(want, &orig, &have_orig); pinned = want; }
      }
      f(idx);
      { std::lock_guard<std::mutex> lk(m); if (--pending == 0) cv_done.notify_all(); }
    }
  }
  // run f(0) .. f(n - 1) on the worker threads (not on the caller); wait() returns when all have finished
  void start(int n, std::function<void(int)> f) {
    while ((int)th.size() < n) { const int idx = (int)th.size(); th.emplace_back([this, idx] { loop(idx); }); }
    { std::lock_guard<std::mutex> lk(m); job = std::move(f); active = n; pending = n; ++gen; }
    cv.notify_all();
  }
  void wait() { std::unique_lock<std::mutex> lk(m); cv_done.wait(lk, [&] { return pending == 0; }); }
  ~Workers() {
    { std::lock_guard<std::mutex> lk(m); stop = true; }
    cv.notify_all();
    for (auto& t : th) t.join();
  }
};

template <class S>
struct Batch : BatchBase {
  Dev<S> d{};
  hipStream_t st = nullptr;
  static constexpr int MAXS = 8;     // run_frames can run up to MAXS slices of the batch concurrently
  hipStream_t stx[MAXS] = {nullptr}; // stx[0] == st
  hipEvent_t ev_fork = nullptr, ev_join[MAXS] = {nullptr};
  int nstreams = 1;
  std::vector<void*> allocs;
  // pinned host staging of the per-call inputs (single-filter API): filled, copied asynchronously, reused only after
  // ev_stage says the previous copy has left it -- the calls themselves do not wait for the device
  // (a ring of NSTG areas: a call takes the next one and only waits if the copy made out of it NSTG calls ago is still in
  // flight -- with a single area every propagate() of the per-sample API waited for the previous sample's copy)
  static constexpr int NSTG = 8;
  unsigned char* h_stage[NSTG] = {nullptr}; size_t h_stage_bytes[NSTG] = {0}; hipEvent_t ev_stage[NSTG] = {nullptr}; bool stage_busy[NSTG] = {false};
  int stage_cur = 0;
  // host mirror of ncam[] (every call that changes the window size updates it) and, per scenario cell, the largest camera
  // slot its work-list touches: run_frames overlaps k_feature with propagate + augment when no track sees the newest camera
  std::vector<int> h_ncam, h_maxslot;
  hipStream_t sty[MAXS] = {nullptr}; hipEvent_t ev_fa[MAXS] = {nullptr}, ev_fb[MAXS] = {nullptr};
  S* P_spare = nullptr;   // second covariance buffer: target of a downdate that carries the frame's prune (Dev::Pout)
  int fuse_prune = 1;     // run_frames: prune rides on the downdate (MSCKF_HIP_FUSE_PRUNE=0: separate k_prune_inplace launch)
  int small_update = 84;     // windows of at most this many camera columns (6 x cameras) take the one-launch update k_update_small; MSCKF_HIP_SMALL_UPDATE=0 switches it off
  int overlap_feature = 0;   // measured on MI355X at cfg3: 100 k -> 82 k updates/s with the overlap on (k_feature floods the CUs the
                             // latency-bound propagate/augment workgroups need); kept selectable, off by default
  int compress_route = -1;   // -1 default, 0 Householder TSQR, != 0 information form + blocked matrix-core Cholesky
  int test_fail_upload = -1;  // test hook (MSCKF_HIP_TEST_FAIL_UPLOAD): run_frames_streamed pretends that this frame's copy failed
  // anisotropic pixel noise (u_var' != v_var'): 0 = the reference's construction R_o_j = A_j^T R_j A_j, R_n = Q_1^T R_o Q_1 on
  // the device (kernels_literal.hip; default), 1 = rows pre-whitened by 1/sigma (generalized least squares, unit noise)
  int aniso_mode = 0;
  int lit_route = 0;         // 0 the compact route; 1 the sweep over the dense stack (tests, A/B)
  double lit_tol = -1;       // zero-tail tolerance of the literal route; < 0: 1e-10 (double) / 8e-4 (float: H_x is float-rounded)
  std::vector<double> h_uv;  // [B][2] u_var', v_var' as initialize() got them
  // Host mirror of the IMU state for the single-filter API: getImuState() is called once per IMU sample by the reference's
  // runner (asl_msckf.cpp:231) and must be synchronously available on the host; between two images only
  // propagate() changes it, and propogateImuStateRK (msckf.h:1425-1467) is a few hundred FLOP -- so msckf_hip_propagate
  // advances this copy with the reference's own RK sequence while the device advances the state the filter uses, and
  // msckf_hip_get_imu_state answers from it without a device round trip.  Any other device-side change of the state
  // (marginalize, the batched calls) invalidates it; the next getter reads the device and re-validates.
  std::vector<S> h_imu; std::vector<char> h_imu_ok;
  std::vector<char> h_lit;   // [B] trajectory runs the literal route
  int n_lit = 0;
  // single-call staging on device
  S* d_rd = nullptr; int rd_cap = 0;               // [B][rd_cap][7]
  S* d_pfin = nullptr;                              // [B][f_cap][4] stored feature positions (mode 1)
  // single-call work-lists: per trajectory ONE block of ints [n, 0, 0, 0 | M[f4] | slots[f_cap][m_cap]] (wl_ib ints; one copy per
  // set_tracks instead of three) and the observations [2 wl_ib] (the kernels index both with the same stride)
  int* wl_i = nullptr; long wl_ib = 0; int wl_f4 = 0; S* wl_obs = nullptr;
  S* h_rb = nullptr;           // page-locked landing area of the single-filter state read (get_cams_known): [n_cap][CAM_STRIDE] + [IMU_STRIDE]
  // scenario.  Work-lists are COMPACT: a cell (frame, trajectory) holds sum M_j (slot, observation) entries, track t of the
  // cell starts at off[cell][t] counted from the frame's first entry (Dev::trk_off) -- not [f_cap][m_cap] padded rows (1.9x
  // the payload at cfg3's track lengths, on the host, in HBM and in every per-frame upload).
  int sc_frames = 0, sc_K = 0;
  bool committed = false;
  S* sc_rd = nullptr; int* sc_n = nullptr; int* sc_M = nullptr; int* sc_off = nullptr; int* sc_drop = nullptr;
  int* sc_slots = nullptr; S* sc_obs = nullptr; size_t sc_total = 0;      // sum over all cells
  std::vector<S> h_rd; std::vector<int> h_n, h_M, h_off, h_drop;
  std::vector<std::vector<int>> c_slots; std::vector<std::vector<S>> c_obs;   // per cell
  std::vector<size_t> fr_base;                                             // [frames + 1] first entry of a frame in sc_slots / sc_obs
  std::vector<void*> sc_allocs;
  // streamed inputs (run_frames_streamed): frame f's block [rd | n | drop | M | off | slots | obs] is copied from page-locked
  // host memory into one of `ring` device staging sets on a copy stream, `ring` - 1 frames ahead of the kernels that read it.
  // Page-locked blocks are built on demand (scen_pin, or the first streamed run over a frame), only for frames that are
  // streamed: a run_frames-only user never pays for them.
  static constexpr int RING_MAX = 8;
  int ring = 6, up_mode = 0;   // up_mode 0: the host threads hand frames over (no device-side cross-stream wait); 1: hipStreamWaitEvent
  hipStream_t stc = nullptr; hipEvent_t ev_up[RING_MAX] = {nullptr}; hipEvent_t ev_use[RING_MAX][MAXS] = {{nullptr}};
  unsigned char* sg_blk[RING_MAX] = {nullptr}; size_t sg_bytes = 0;
  struct PinFrame { unsigned char* p = nullptr; size_t bytes = 0, off_obs = 0; int chunk = -1; };
  struct PinChunk { void* p = nullptr; int live = 0; };   // page-locked block of several frames; freed when its last frame is invalidated
  std::vector<PinFrame> pinf; std::vector<PinChunk> pin_chunks;
  void unpin_frame(int f) {
    const int c = pinf[f].chunk;
    if (c >= 0 && c < (int)pin_chunks.size() && pin_chunks[c].p && --pin_chunks[c].live == 0) {
      if (stc) (void)hipStreamSynchronize(stc);     // the last copy out of the block has completed
      (void)hipHostFree(pin_chunks[c].p);
      pin_chunks[c].p = nullptr;
    }
    pinf[f] = PinFrame();
  }
  size_t pk_rd = 0, pk_n = 0, pk_drop = 0, pk_M = 0, pk_off = 0, pk_slots = 0;   // section offsets (256-byte aligned); obs follows the frame's slots
  Workers workers;   // enqueue threads of the slices
  // profiling
  bool prof = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool[NSTAGE];
  size_t ev_used[NSTAGE] = {0};
  double prof_ms[NSTAGE] = {0}; int prof_cnt[NSTAGE] = {0};

  template <class T> int dalloc(T** p, size_t count) {
    void* q = nullptr;
    HIPCHK(hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(T)));
    HIPCHK(hipMemsetAsync(q, 0, std::max<size_t>(count, 1) * sizeof(T), st));
    allocs.push_back(q);
    *p = (T*)q;
    return 0;
  }
  // (flush_pending: the IMU samples of propagate() calls that have not reached the device yet, see propagate())
#define DEVICE_ENTER() do { HIPCHK(hipSetDevice(device)); if (pend_b >= 0) { const int rc_p_ = flush_pending(); if (rc_p_) return rc_p_; } } while (0)
  int create() {
    HIPCHK(hipSetDevice(device));
    feature_device_setup(); qr_device_setup(); kalman_device_setup(); gram_device_setup(); literal_device_setup();   // per device: constant tables, dynamic-LDS limits
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (int i = 0; i < NSTG; ++i) HIPCHK(hipEventCreateWithFlags(&ev_stage[i], hipEventDisableTiming));
    stx[0] = st;
    for (int i = 1; i < MAXS; ++i) HIPCHK(hipStreamCreateWithFlags(&stx[i], hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
    for (int i = 1; i < MAXS; ++i) HIPCHK(hipEventCreateWithFlags(&ev_join[i], hipEventDisableTiming));
    for (int i = 0; i < MAXS; ++i) {
      HIPCHK(hipStreamCreateWithFlags(&sty[i], hipStreamNonBlocking));
      HIPCHK(hipEventCreateWithFlags(&ev_fa[i], hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&ev_fb[i], hipEventDisableTiming));
    }
    h_ncam.assign(B, 0); h_uv.assign((size_t)2 * B, 0.0); h_lit.assign(B, 0); h_imu.assign((size_t)B * IMU_STRIDE, S(0)); h_imu_ok.assign(B, 0);
    d.B = B; d.n_cap = n_cap; d.f_cap = f_cap; d.m_cap = m_cap;
    d.n6cap = 6 * n_cap;
    d.ld = ((15 + 6 * n_cap + 15) / 16) * 16;
    d.ldR = ((6 * n_cap + 1 + 63) / 64) * 64;
    // 6 n_cap + 1 <= 384 (n_cap <= 63): the compression kernels' column capacity; it also bounds everything indexed by a state
    // column or a camera slot further down (k_prune_inplace keeps ceil(ld / 16) x ceil(ld / 256) <= 25 x 2 elements per thread: ld <= 400; 6-bit slot fields of trk_first)
    if (d.ldR / 64 > 6) return fail(-ENOTSUP, "n_cap too large: 6*n_cap+1 must be <= 384 (at most 63 camera states)");
    int nch = 1;
    while (nch < 8 && (long)B * nch * 2 <= 256) nch *= 2;    // TSQR route: chunks x trajectories ~ one workgroup per CU
    d.nchunk = nch;
    const size_t Bz = B, pl = (size_t)d.ld * d.ld, nl = (size_t)d.n6cap * d.n6cap, dn = (size_t)d.ld * d.n6cap;
    const size_t TF = Bz * f_cap;
    int rc = 0;
    rc |= dalloc(&d.imu, Bz * IMU_STRIDE); rc |= dalloc(&d.cam, Bz * n_cap * CAM_STRIDE); rc |= dalloc(&d.prm, Bz * PRM_STRIDE);
    rc |= dalloc(&d.P, Bz * pl); rc |= dalloc(&P_spare, Bz * pl); d.Pout = nullptr; d.fuse_drop = nullptr; d.ncam_defer = 0;
    if (const char* e = getenv("MSCKF_HIP_FUSE_PRUNE")) fuse_prune = atoi(e) != 0;
    if (const char* e = getenv("MSCKF_HIP_TEST_FAIL_UPLOAD")) test_fail_upload = atoi(e);   // test hook: the upload of this frame fails
    if (const char* e = getenv("MSCKF_HIP_LITERAL_ROUTE")) lit_route = atoi(e);   // A/B runs and tests: 1 = the sweep over the dense stack
    rc |= dalloc(&d.ncam, Bz); rc |= dalloc(&d.n_resid, Bz);
    rc |= dalloc(&d.trk_status, TF); rc |= dalloc(&d.trk_pf, TF * 4); rc |= dalloc(&d.trk_gamma, TF);
    d.h16 = h16 ? 1 : 0; d.trk_Hx = nullptr; d.trk_Hx16 = nullptr;
    if (h16) rc |= dalloc(&d.trk_Hx16, TF * m_cap * 12); else rc |= dalloc(&d.trk_Hx, TF * m_cap * 12);
    rc |= dalloc(&d.trk_V, TF * 2 * m_cap * 4); rc |= dalloc(&d.trk_Zf, TF * 3 * (size_t)d.ldR);
    rc |= dalloc(&d.trk_ro, TF * 2 * m_cap); rc |= dalloc(&d.trk_first, TF);
    rc |= dalloc(&d.row_start, Bz * (f_cap + 1)); rc |= dalloc(&d.trk_order, TF); rc |= dalloc(&d.stats, Bz * STAT_STRIDE);
    rc |= dalloc(&d.Rbuf, Bz * d.nchunk * (size_t)d.n6cap * d.ldR);
    // information-form compression (kernels_gram.hip + kernels_chol.hip)
    d.compress = (d.ldR <= 384 && f_cap <= 1024) ? 3 : 0;   // blocked matrix-core Cholesky (kernels_chol.hip), two levels beyond 192 columns
    d.Mp = nullptr; d.Mp2 = nullptr;
    if (d.n6cap > 192) rc |= dalloc(&d.Mp2, Bz * 24 * 256);
    if (d.compress) {
      if (d.ldR > 192) rc |= dalloc(&d.Mp, Bz * 12 * 256);
      rc |= dalloc(&d.trk_B, TF * 3 * (size_t)d.ldR); rc |= dalloc(&d.trk_rw, TF * 2 * m_cap); rc |= dalloc(&d.trk_inv, TF * n_cap);
      rc |= dalloc(&d.Dg, Bz * n_cap * DG_STRIDE);
      d.lam_part = d.ldR <= 192 ? (long)(Bz * (size_t)d.ldR * d.ldR) : 0;     // up to four copies of Lam^ for the split-K SYRK (windows up to 31 cameras)
      d.gram_parts = 3;   // P = 4 (sixteen workgroups per trajectory) measured: k_gram 56 -> 54 us, the Cholesky's extra load round 64 -> 66 us
      rc |= dalloc(&d.Lam, Bz * (size_t)d.ldR * d.ldR * (d.lam_part ? 4 : 1));
    }
    rc |= dalloc(&d.PHt, Bz * dn); rc |= dalloc(&d.Smat, Bz * nl); rc |= dalloc(&d.Linv, Bz * nl); rc |= dalloc(&d.W, Bz * dn);
    rc |= dalloc(&d.K, Bz * dn); rc |= dalloc(&d.A, Bz * pl); rc |= dalloc(&d.AP, Bz * pl); rc |= dalloc(&d.X, Bz * pl); rc |= dalloc(&d.dx, Bz * d.ld);
    rc |= dalloc(&d.keep, Bz * n_cap); rc |= dalloc(&d.nkeep, Bz); rc |= dalloc(&d.ncam_upd, Bz); rc |= dalloc(&d.nres_upd, Bz);
    rc |= dalloc(&d_pfin, TF * 4); d.trk_pfin = d_pfin; d.mode = 0; d.joseph = 0; d.ncam_bias = 0;
    { const char* e = getenv("MSCKF_HIP_FUSED_S"); d.gain_fused_s = e ? atoi(e) : 2; }
    { const char* e = getenv("MSCKF_HIP_FEATURE_PAIR"); d.feat_pair = e ? atoi(e) : 1; }
    { const char* e = getenv("MSCKF_HIP_SMALL_UPDATE"); if (e) small_update = atoi(e); }
    rc |= dalloc(&d.gain_bar, Bz * 32);   // 0: the S GEMM as a launch of its own (A/B runs)
    rd_cap = 64;
    rc |= dalloc(&d_rd, Bz * rd_cap * RD_STRIDE);
    HIPCHK(hipHostMalloc((void**)&h_rb, ((size_t)n_cap * CAM_STRIDE + IMU_STRIDE) * sizeof(S), hipHostMallocDefault));
    wl_f4 = (f_cap + 3) & ~3; wl_ib = (4 + wl_f4 + (long)f_cap * m_cap + 3) & ~3L;
    rc |= dalloc(&wl_i, Bz * wl_ib); rc |= dalloc(&wl_obs, Bz * wl_ib * 2);
    if (rc) return rc;
    use_single_worklists();
    if (feature_lds_bytes(m_cap, sizeof(S)) > 160 * 1024) return fail(-EINVAL, "m_cap too large for the feature kernel's LDS budget");
    traj.assign(B, HostTraj());
    HIPCHK(hipStreamSynchronize(st));
    return 0;
  }
  ~Batch() override {
    hipSetDevice(device);
    if (st) hipStreamSynchronize(st);
    for (void* p : allocs) hipFree(p);
    if (h_rb) hipHostFree(h_rb);
    for (int s = 0; s < NSTAGE; ++s) for (auto& e : ev_pool[s]) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
    for (int i = 1; i < MAXS; ++i) { if (stx[i]) hipStreamDestroy(stx[i]); if (ev_join[i]) hipEventDestroy(ev_join[i]); }
    for (int i = 0; i < MAXS; ++i) { if (sty[i]) hipStreamDestroy(sty[i]); if (ev_fa[i]) hipEventDestroy(ev_fa[i]); if (ev_fb[i]) hipEventDestroy(ev_fb[i]); }
    if (ev_fork) hipEventDestroy(ev_fork);
    for (int i = 0; i < NSTG; ++i) if (ev_stage[i]) hipEventDestroy(ev_stage[i]);
    unpin_host();
    for (int k = 0; k < RING_MAX; ++k) { if (ev_up[k]) hipEventDestroy(ev_up[k]); for (int i = 0; i < MAXS; ++i) if (ev_use[k][i]) hipEventDestroy(ev_use[k][i]); if (sg_blk[k]) hipFree(sg_blk[k]); }
    if (stc) hipStreamDestroy(stc);

    for (int i = 0; i < NSTG; ++i) if (h_stage[i]) hipHostFree(h_stage[i]);
    if (st) hipStreamDestroy(st);
  }
  // pinned staging area of at least `bytes`, safe to overwrite (the previous asynchronous copy out of it has finished)
  int stage_acquire(size_t bytes, unsigned char** out) {
    stage_cur = (stage_cur + 1) % NSTG;
    const int k = stage_cur;
    if (stage_busy[k]) { HIPCHK(hipEventSynchronize(ev_stage[k])); stage_busy[k] = false; }
    if (bytes > h_stage_bytes[k]) {
      if (h_stage[k]) HIPCHK(hipHostFree(h_stage[k]));
      h_stage[k] = nullptr; h_stage_bytes[k] = 0;
      const size_t nb = std::max<size_t>(bytes, 1 << 12);
      HIPCHK(hipHostMalloc((void**)&h_stage[k], nb, hipHostMallocDefault));
      h_stage_bytes[k] = nb;
    }
    *out = h_stage[k];
    return 0;
  }
  int stage_release() { HIPCHK(hipEventRecord(ev_stage[stage_cur], st)); stage_busy[stage_cur] = true; return 0; }
  // every slot of the ring at least `bytes` (a commit walks the ring once per frame: without this each slot would be freed and
  // re-allocated as the frames grow)
  int stage_reserve(size_t bytes) {
    for (int k = 0; k < NSTG; ++k) {
      if (bytes <= h_stage_bytes[k]) continue;
      if (stage_busy[k]) { HIPCHK(hipEventSynchronize(ev_stage[k])); stage_busy[k] = false; }
      if (h_stage[k]) HIPCHK(hipHostFree(h_stage[k]));
      h_stage[k] = nullptr; h_stage_bytes[k] = 0;
      const size_t nb = std::max<size_t>(bytes, 1 << 12);
      HIPCHK(hipHostMalloc((void**)&h_stage[k], nb, hipHostMallocDefault));
      h_stage_bytes[k] = nb;
    }
    return 0;
  }
  void use_single_worklists() {
    d.trk_n = wl_i; d.trk_M = wl_i + 4; d.trk_slots = wl_i + 4 + wl_f4; d.trk_obs = wl_obs; d.trk_off = nullptr;
    d.wl_stride_n = wl_ib; d.wl_stride_f = wl_ib; d.wl_stride_o = wl_ib;
  }
  // Dev view whose work-list pointers start at trajectory b0 (kernels index work-lists by b - b0)
  Dev<S> view(int b0) const {
    Dev<S> v = d;
    v.trk_n += (long)b0 * d.wl_stride_n; v.trk_M += (long)b0 * d.wl_stride_f;
    v.trk_slots += (long)b0 * d.wl_stride_o; v.trk_obs += 2 * (long)b0 * d.wl_stride_o;
    return v;
  }
  // ---- profiling helpers
  void stage_begin(int s, hipStream_t q) {
    if (!prof) return;
    if (ev_used[s] == ev_pool[s].size()) {
      hipEvent_t a, b2; hipEventCreate(&a); hipEventCreate(&b2);
      ev_pool[s].push_back({a, b2});
    }
    hipEventRecord(ev_pool[s][ev_used[s]].first, q);
  }
  void stage_end(int s, hipStream_t q) {
    if (!prof) return;
    hipEventRecord(ev_pool[s][ev_used[s]].second, q);
    ev_used[s]++;
  }

  // a track observes a camera at most once (the kernels rely on it: slot -> observation map, contiguous-range test)
  static bool repeated_slot(const int* s, int M) {
    unsigned long long seen = 0;
    for (int k = 0; k < M; ++k) { const unsigned long long bit = 1ull << (s[k] & 63); if (s[k] >= 0 && s[k] < 64 && (seen & bit)) return true; seen |= bit; }
    return false;
  }
  int chk(int b) const { return (b < 0 || b >= B) ? -EINVAL : 0; }
  // A run_frames / run_frames_streamed call that failed after some of its frames were enqueued leaves the slices at different
  // frames: which covariance buffer is current (the fused prune flips them per frame) and whether a window size is still
  // deferred differ per slice, and nothing can put that right.  The handle refuses further work instead of answering from a
  // stale buffer; the caller destroys it.
  bool poisoned = false;
  int poison(int rc, const std::string& msg) {
    (void)hipDeviceSynchronize();
    poisoned = true;
    return fail(rc, msg + " -- frames of this call were already enqueued: the filter states of this handle are undefined, destroy it");
  }
#define POISON_GUARD() do { if (poisoned) return fail(-EIO, "handle unusable after a failed run_frames call (destroy it)"); } while (0)
  int chk_range(int b0, int nb) const { return (b0 < 0 || nb < 0 || b0 + nb > B) ? -EINVAL : 0; }

  // The five derived noise parameters PRM_WU .. PRM_LIT of trajectory b for the batch's anisotropic-noise mode
  // (dev_common.h); allocates the literal route's work space when the first trajectory needs it.
  int derive_noise(int b, S* out5) {
    const double u = h_uv[2 * (size_t)b], v = h_uv[2 * (size_t)b + 1];
    const bool was = h_lit[b] != 0;
    bool lit = false;
    if (u == v) { out5[0] = 1; out5[1] = 1; out5[2] = (S)u; out5[3] = (S)u; out5[4] = 0; }
    else if (aniso_mode == 0 && !h16 && d.trk_B) { out5[0] = 1; out5[1] = 1; out5[2] = 1; out5[3] = (S)u; out5[4] = 1; lit = true; }
    else { out5[0] = (S)(1.0 / std::sqrt(u)); out5[1] = (S)(1.0 / std::sqrt(v)); out5[2] = 1; out5[3] = 1; out5[4] = 0; }
    if (lit && !d.lit.W2) { const int rc = lit_alloc(); if (rc) return rc; }
    if (lit != was) { n_lit += lit ? 1 : -1; h_lit[b] = lit ? 1 : 0; }
    return 0;
  }
  // work space of kernels_literal.hip (a few (6 n_cap)^2 matrices per trajectory); only allocated when a trajectory has
  // u_var' != v_var' on the literal route
  int lit_alloc() {
    if (!literal_lds_available()) return fail(-ENOTSUP, "this device does not grant the 94 KB of LDS per workgroup the literal anisotropic route needs (msckf_hip_set_anisotropic_noise(h, 1, 0) selects pre-whitening)");
    LitBufs& L = d.lit;
    const size_t Bz = B, n1 = (size_t)d.n6cap + 1;
    L.ldx = ((f_cap * std::max(2 * m_cap - 3, 1) + 7) / 8) * 8;
    L.r_cap = d.n6cap + 15; L.ldg = f_cap * m_cap + 8; L.ldz = L.r_cap + (int)n1; L.kept_stride = 6 * (d.n6cap + 16) + 64;
    L.tol = lit_tol >= 0 ? lit_tol : (sizeof(S) == 4 ? 8e-4 : 1e-10);
    L.route = lit_route;
    int rc = 0;
    rc |= dalloc(&L.tau, Bz * (2 * n1 + 2));
    rc |= dalloc(&L.Vf, Bz * f_cap * 2 * m_cap * 3); rc |= dalloc(&L.Tf, Bz * f_cap * 9);
    rc |= dalloc(&L.row0, Bz * (f_cap + 1)); rc |= dalloc(&L.obs0, Bz * (f_cap + 1)); rc |= dalloc(&L.otrk, Bz * L.ldg); rc |= dalloc(&L.kept, Bz * L.kept_stride);
    rc |= dalloc(&L.TH, Bz * L.r_cap * n1); rc |= dalloc(&L.Z, Bz * (size_t)L.ldz * L.ldz);
    L.w2_stride = lit_ws_doubles(d.n6cap, m_cap, L.r_cap);
    rc |= dalloc(&L.W2, Bz * (size_t)L.w2_stride);
    // the sweep over the dense stack (MSCKF_HIP_LITERAL_ROUTE=1: tests, A/B runs) needs the stack itself and the u-rows of A Q_1
    if (lit_route == 1) { rc |= dalloc(&L.X, Bz * L.ldx * n1); rc |= dalloc(&L.G, Bz * (size_t)L.ldg * L.r_cap); }
    rc |= dalloc(&L.info, Bz * 8);
    rc |= dalloc(&L.BD, Bz * f_cap * 6 * (size_t)d.ldR); rc |= dalloc(&L.Gam, Bz * (size_t)d.ldR * d.ldR); rc |= dalloc(&L.Du, Bz * n_cap * 24);
    if (const char* e = getenv("MSCKF_HIP_LITERAL_SERIAL")) L.serial = atoi(e);
    if (const char* e = getenv("MSCKF_HIP_LITERAL_TIMERS")) if (atoi(e)) rc |= dalloc(&L.tim, Bz * LIT_TIM_SLOTS);
    if (rc) { L.W2 = nullptr; return fail(-ENOMEM, "work space of the literal anisotropic route (msckf_hip_set_anisotropic_noise(h, 1, 0) selects pre-whitening)"); }
    return 0;
  }
  int set_aniso(int mode, double tol) override {
    if (mode < 0 || mode > 1) return fail(-EINVAL, "mode: 0 the reference's R_n = Q_1^T R_o Q_1 on the device, 1 pre-whitened rows");
    DEVICE_ENTER();
    aniso_mode = mode; lit_tol = tol;
    d.lit.tol = tol >= 0 ? tol : (sizeof(S) == 4 ? 8e-4 : 1e-10);
    d.lit.route = lit_route;
    for (int b = 0; b < B; ++b) {
      if (!traj[b].initialized) continue;
      S out5[5];
      const int rc = derive_noise(b, out5);
      if (rc) return rc;
      HIPCHK(hipMemcpyAsync(d.prm + (size_t)b * PRM_STRIDE + PRM_WU, out5, sizeof(out5), hipMemcpyHostToDevice, st));
      HIPCHK(hipStreamSynchronize(st));
    }
    return 0;
  }
  int lit_info(int b, int* out8) override {
    if (chk(b)) return fail(-EINVAL, "trajectory index out of range");
    if (!d.lit.info) { for (int i = 0; i < 8; ++i) out8[i] = 0; return 0; }
    DEVICE_ENTER();
    HIPCHK(hipMemcpyAsync(out8, d.lit.info + (size_t)b * 8, 8 * sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (d.lit.tim) {     // MSCKF_HIP_LITERAL_TIMERS=1 (profiling runs): phase durations of the last launch in microseconds on stderr
      long long t[LIT_TIM_SLOTS];
      HIPCHK(hipMemcpyAsync(t, d.lit.tim + (size_t)b * LIT_TIM_SLOTS, sizeof(t), hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
      std::fprintf(stderr, "[k_literal b=%d] us: explicit rows %.0f Gram %.0f sweep %.0f kept %.0f handed-through rows %.0f basis products %.0f Z fill %.0f eliminate %.0f store %.0f total %.0f\n", b,
                   (t[2] - t[1]) * 0.01, (t[3] - t[2]) * 0.01, (t[4] - t[3]) * 0.01, (t[5] - t[4]) * 0.01, (t[6] - t[5]) * 0.01,
                   (t[8] - t[6]) * 0.01, (t[10] - t[8]) * 0.01, (t[11] - t[10]) * 0.01, (t[9] - t[11]) * 0.01, (t[9] - t[0]) * 0.01);
      std::fprintf(stderr, "[k_literal b=%d] us: the sweep's panels = stage %.0f + core %.0f + rows / columns %.0f + results and trailing pass %.0f; basis products = staging %.0f + Z %.0f + rest %.0f\n", b,
                   t[12] * 0.01, t[13] * 0.01, t[14] * 0.01, t[15] * 0.01, (t[7] - t[6]) * 0.01, (t[23] - t[7]) * 0.01, (t[8] - t[23]) * 0.01);
      if (out8[6] > 0)
        std::fprintf(stderr, "[k_literal b=%d] us: the %d kept handed-through rows = column operations %.0f + Gram of the start %.0f + its products %.0f + reflectors %.0f + t~ %.0f + products with the explicit rows, Gam y %.0f + pair products %.0f\n", b, out8[6],
                     (t[16] - t[5]) * 0.01, (t[17] - t[16]) * 0.01, (t[18] - t[17]) * 0.01, (t[19] - t[18]) * 0.01, (t[20] - t[19]) * 0.01, (t[21] - t[20]) * 0.01, (t[22] - t[21]) * 0.01);
    }
    return 0;
  }
  int init(int b, const double* cam, const double* noise, const double* params, const double* imu) override {
    POISON_GUARD();
    if (chk(b)) return fail(-EINVAL, "trajectory index out of range");
    if (!(noise[0] > 0) || !(noise[1] > 0)) return fail(-EINVAL, "u_var_prime / v_var_prime must be positive");
    DEVICE_ENTER();
    S prm[PRM_STRIDE] = {0}, st_imu[IMU_STRIDE] = {0};
    for (int i = 0; i < 12; ++i) prm[i] = (S)cam[i];
    prm[PRM_UVAR] = (S)noise[0]; prm[PRM_VVAR] = (S)noise[1];
    for (int i = 0; i < 12; ++i) prm[PRM_Q + i] = (S)noise[2 + i];
    for (int i = 0; i < 8; ++i) prm[PRM_GN + i] = (S)params[i];
    h_uv[2 * (size_t)b] = noise[0]; h_uv[2 * (size_t)b + 1] = noise[1];
    { const int rc0 = derive_noise(b, prm + PRM_WU); if (rc0) return rc0; }
    for (int i = 0; i < 29; ++i) st_imu[i] = (S)imu[i];
    for (int i = 0; i < 4; ++i) st_imu[IQN + i] = st_imu[IQ + i];        // msckf.h:83-85
    for (int i = 0; i < 3; ++i) { st_imu[IVN + i] = st_imu[IV + i]; st_imu[IPN + i] = st_imu[IP + i]; }
    std::vector<S> P((size_t)d.ld * d.ld, S(0));
    for (int i = 0; i < 15; ++i) P[(size_t)i * d.ld + i] = (S)noise[14 + i];
    std::copy(st_imu, st_imu + IMU_STRIDE, h_imu.begin() + (size_t)b * IMU_STRIDE); h_imu_ok[b] = 1;
    HIPCHK(hipMemcpyAsync(d.prm + (size_t)b * PRM_STRIDE, prm, sizeof(prm), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d.imu + (size_t)b * IMU_STRIDE, st_imu, sizeof(st_imu), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d.P + (size_t)b * d.ld * d.ld, P.data(), P.size() * sizeof(S), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemsetAsync(d.ncam + b, 0, sizeof(int), st));
    h_ncam[b] = 0;
    HIPCHK(hipMemsetAsync(d.n_resid + b, 0, sizeof(long long), st));
    HIPCHK(hipMemsetAsync(d.stats + (size_t)b * STAT_STRIDE, 0, sizeof(int) * STAT_STRIDE, st));
    HIPCHK(hipMemsetAsync(wl_i + (size_t)b * wl_ib, 0, sizeof(int), st));
    HIPCHK(hipStreamSynchronize(st));
    HostTraj& t = traj[b];
    t = HostTraj();
    t.initialized = true;
    t.min_track_length = (int)params[5]; t.max_track_length = (int)params[6]; t.max_cam_states = (int)params[7];
    t.redundancy_angle_thresh = params[3]; t.redundancy_distance_thresh = params[4];
    return 0;
  }
  // propogateImuStateRK (msckf.h:1425-1467) + the anchors of msckf.h:138-141 on the host copy of trajectory b, in S arithmetic
  static void host_rk(S* x, const double* rd7) {
    const S dT = (S)rd7[6];
    const S w[3] = {(S)rd7[0] - x[IBG], (S)rd7[1] - x[IBG + 1], (S)rd7[2] - x[IBG + 2]};
    // 0.5 * omegaMat(w) applied to y = (-x, -y, -z, w) of q_IG (matrix_utils.h:19-30)
    auto op = [&](const S y[4], S o[4]) {
      o[0] = S(0.5) * (w[2] * y[1] - w[1] * y[2] + w[0] * y[3]);
      o[1] = S(0.5) * (-w[2] * y[0] + w[0] * y[2] + w[1] * y[3]);
      o[2] = S(0.5) * (w[1] * y[0] - w[0] * y[1] + w[2] * y[3]);
      o[3] = S(0.5) * (-w[0] * y[0] - w[1] * y[1] - w[2] * y[2]);
    };
    const S y0[4] = {-x[IQ + 1], -x[IQ + 2], -x[IQ + 3], x[IQ]};
    S k0[4], k1[4], k2[4], k3[4], k4[4], k5[4], t[4];
    op(y0, k0);
    for (int i = 0; i < 4; ++i) t[i] = y0[i] + (k0[i] / S(4)) * dT;
    op(t, k1);
    for (int i = 0; i < 4; ++i) t[i] = y0[i] + (k0[i] / S(8) + k1[i] / S(8)) * dT;
    op(t, k2);
    for (int i = 0; i < 4; ++i) t[i] = y0[i] + (-k1[i] / S(2) + k2[i]) * dT;
    op(t, k3);
    for (int i = 0; i < 4; ++i) t[i] = y0[i] + (k0[i] * S(3) / S(16) + k3[i] * S(9) / S(16)) * dT;
    op(t, k4);
    for (int i = 0; i < 4; ++i) t[i] = y0[i] + (-k0[i] * S(3) / S(7) + k1[i] * S(2) / S(7) + k2[i] * S(12) / S(7) - k3[i] * S(12) / S(7) + k4[i] * S(8) / S(7)) * dT;
    op(t, k5);
    S yt[4];
    for (int i = 0; i < 4; ++i) yt[i] = y0[i] + (S(7) * k0[i] + S(32) * k2[i] + S(12) * k3[i] + S(32) * k4[i] + S(7) * k5[i]) * dT / S(90);
    S q[4] = {yt[3], -yt[0], -yt[1], -yt[2]};
    const S nq = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    // v += (C_IG^T (a - b_a) + g) dT with the OLD attitude, p += v_old dT
    const S qw = x[IQ], qx = x[IQ + 1], qy = x[IQ + 2], qz = x[IQ + 3];
    const S R[3][3] = {{1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw)},
                       {2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw)},
                       {2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)}};
    const S am[3] = {(S)rd7[3] - x[IBA], (S)rd7[4] - x[IBA + 1], (S)rd7[5] - x[IBA + 2]};
    for (int i = 0; i < 3; ++i) x[IP + i] += x[IV + i] * dT;
    for (int i = 0; i < 3; ++i) x[IV + i] += (R[0][i] * am[0] + R[1][i] * am[1] + R[2][i] * am[2] + x[IG + i]) * dT;
    for (int i = 0; i < 4; ++i) x[IQ + i] = q[i] / nq;
    for (int i = 0; i < 4; ++i) x[IQN + i] = x[IQ + i];
    for (int i = 0; i < 3; ++i) { x[IVN + i] = x[IV + i]; x[IPN + i] = x[IP + i]; }
  }
  void invalidate_imu(int b0, int nb) { for (int b = b0; b < b0 + nb && b < B; ++b) h_imu_ok[b] = 0; }
  // Single-filter API (the shim's propagate(), one call per IMU sample, msckf.h:101): while the host copy of the IMU state is
  // valid it answers getImuState(), so the samples need not reach the device one by one -- they wait here and go as ONE copy +
  // ONE k_propagate launch when anything else touches the device (DEVICE_ENTER at the head of every other entry).  Ten calls per
  // image were ten pinned-memory copies and ten launches (~6 us of host time each) for the same device-side result.
  std::vector<double> pend_rd; int pend_b = -1;
  int flush_pending(bool then_augment = false) {
    if (pend_b < 0) return 0;
    const int b = pend_b; pend_b = -1;
    std::vector<double> rd; rd.swap(pend_rd);
    if (hipSetDevice(device) != hipSuccess) { h_imu_ok[b] = 0; return fail(-EIO, "hipSetDevice failed"); }
    const int rc = propagate_device(b, 1, rd.data(), (int)(rd.size() / RD_STRIDE), then_augment);
    if (rc) h_imu_ok[b] = 0;   // the samples are gone and the device never saw them: the host copy is ahead of the filter, drop it (getImuState() re-reads the device)
    return rc;
  }
  int propagate(int b0, int nb, const double* rd, int K, bool mirror) override {
    POISON_GUARD();
    if (chk_range(b0, nb)) return fail(-EINVAL, "trajectory range out of bounds");
    if (K < 0) return fail(-EINVAL, "negative sample count");
    if (mirror && nb == 1 && h_imu_ok[b0]) {
      if (pend_b >= 0 && pend_b != b0) { const int rc = flush_pending(); if (rc) return rc; }   // first: a failure here must not leave b0's host copy advanced with nothing queued
      for (int k = 0; k < K; ++k) host_rk(h_imu.data() + (size_t)b0 * IMU_STRIDE, rd + (size_t)k * RD_STRIDE);
      pend_b = b0; pend_rd.insert(pend_rd.end(), rd, rd + (size_t)K * RD_STRIDE);
      return 0;
    }
    if (!(mirror && nb == 1)) invalidate_imu(b0, nb);
    DEVICE_ENTER();
    return propagate_device(b0, nb, rd, K);
  }
  // then_augment: augmentState follows for the same trajectories -- k_propagate's fused variant on the last chunk (as run_frames)
  int propagate_device(int b0, int nb, const double* rd, int K, bool then_augment = false) {
    for (int k0 = 0; k0 < K; k0 += rd_cap) {
      const int kk = std::min(rd_cap, K - k0);
      const size_t cnt = (size_t)nb * kk * RD_STRIDE;
      unsigned char* raw = nullptr;
      int rc = stage_acquire(cnt * sizeof(S), &raw);
      if (rc) return rc;
      S* tmp = reinterpret_cast<S*>(raw);
      for (int i = 0; i < nb; ++i)
        for (int k = 0; k < kk; ++k)
          for (int c = 0; c < RD_STRIDE; ++c) tmp[((size_t)i * kk + k) * RD_STRIDE + c] = (S)rd[((size_t)i * K + k0 + k) * RD_STRIDE + c];
      HIPCHK(hipMemcpyAsync(d_rd, tmp, cnt * sizeof(S), hipMemcpyHostToDevice, st));
      rc = stage_release();
      if (rc) return rc;
      launch_propagate<S>(d, b0, nb, d_rd, (long)kk * RD_STRIDE, kk, st, then_augment && k0 + kk >= K);
      HIPCHK(hipGetLastError());
    }
    return 0;
  }
  int augment(int b0, int nb) override {
    POISON_GUARD();
    if (chk_range(b0, nb)) return fail(-EINVAL, "trajectory range out of bounds");
    HIPCHK(hipSetDevice(device));
    if (pend_b >= 0 && pend_b == b0 && nb == 1 && !pend_rd.empty()) {     // the image's IMU samples are still here: one launch for both
      const int rc = flush_pending(true);
      if (rc) return rc;
    } else {
      DEVICE_ENTER();
      launch_augment<S>(d, b0, nb, st);
    }
    for (int b = b0; b < b0 + nb; ++b) if (h_ncam[b] < n_cap) h_ncam[b]++;
    HIPCHK(hipGetLastError());
    return 0;
  }
  int set_tracks(int b, int F, const int* M, const int* slots, const double* obs) override {
    if (chk(b)) return fail(-EINVAL, "trajectory index out of range");
    if (F < 0 || F > f_cap) return fail(-E2BIG, "more tracks than f_cap");
    DEVICE_ENTER();
    for (int t = 0; t < F; ++t) if (M[t] > m_cap || M[t] < 0) return fail(-E2BIG, "track longer than m_cap");
    { size_t o = 0; for (int t = 0; t < F; ++t) { if (repeated_slot(slots + o, M[t])) return fail(-EINVAL, "camera slot repeated within a track"); o += M[t]; } }
    // only the F rows in use travel, as the device holds them: [n, 0, 0, 0 | M[f4] | F rows of slots] in one copy, F rows of
    // coordinates in a second one, both out of one pinned block
    const size_t nI = 4 + (size_t)wl_f4 + (size_t)F * m_cap, nO = (size_t)F * m_cap * 2;
    const size_t offO = ((nI * sizeof(int) + 15) / 16) * 16;
    unsigned char* raw = nullptr;
    int rc = stage_acquire(offO + nO * sizeof(S) + 16, &raw);
    if (rc) return rc;
    int* hI = reinterpret_cast<int*>(raw); int* hM = hI + 4; int* hS = hI + 4 + wl_f4; S* hO = reinterpret_cast<S*>(raw + offO);
    std::memset(raw, 0, offO + nO * sizeof(S));
    hI[0] = F;
    size_t o = 0;
    for (int t = 0; t < F; ++t) {
      hM[t] = M[t];
      for (int k = 0; k < M[t]; ++k) {
        if (slots[o + k] < 0 || slots[o + k] >= n_cap) return fail(-EINVAL, "camera slot out of range");
        hS[(size_t)t * m_cap + k] = slots[o + k];
        hO[((size_t)t * m_cap + k) * 2] = (S)obs[2 * (o + k)];
        hO[((size_t)t * m_cap + k) * 2 + 1] = (S)obs[2 * (o + k) + 1];
      }
      o += M[t];
    }
    HIPCHK(hipMemcpyAsync(wl_i + (size_t)b * wl_ib, hI, (F ? nI : 4) * sizeof(int), hipMemcpyHostToDevice, st));
    if (F) HIPCHK(hipMemcpyAsync(wl_obs + (size_t)b * wl_ib * 2, hO, nO * sizeof(S), hipMemcpyHostToDevice, st));
    rc = stage_release();
    if (rc) return rc;
    traj[b].wl_F = F;
    return 0;
  }
  int set_tracks_range(int b0, int nb, const std::vector<BatchBase::WorkList>& wl) override {
    if (chk_range(b0, nb) || (int)wl.size() != nb) return fail(-EINVAL, "trajectory range out of bounds");
    DEVICE_ENTER();
    for (int i = 0; i < nb; ++i) {
      const int F = (int)wl[i].M.size();
      if (F > f_cap) return fail(-E2BIG, "more tracks than f_cap");
      size_t o = 0;
      for (int t = 0; t < F; ++t) {
        const int m = wl[i].M[t];
        if (m > m_cap || m < 0) return fail(-E2BIG, "track longer than m_cap");
        if (repeated_slot(wl[i].slots.data() + o, m)) return fail(-EINVAL, "camera slot repeated within a track");
        for (int k = 0; k < m; ++k) if (wl[i].slots[o + k] < 0 || wl[i].slots[o + k] >= n_cap) return fail(-EINVAL, "camera slot out of range");
        o += m;
      }
    }
    // the device's padded single-call layout for the whole range: [nb][wl_ib] ints ([n, 0, 0, 0 | M[f4] | f_cap rows of slots]) and
    // [nb][wl_ib * 2] coordinates, out of one pinned block, two copies
    const size_t nI = (size_t)nb * wl_ib, nO = (size_t)nb * wl_ib * 2;
    const size_t offO = ((nI * sizeof(int) + 15) / 16) * 16;
    unsigned char* raw = nullptr;
    int rc = stage_acquire(offO + nO * sizeof(S) + 16, &raw);
    if (rc) return rc;
    int* hI = reinterpret_cast<int*>(raw); S* hO = reinterpret_cast<S*>(raw + offO);
    const long ib = wl_ib; const int f4 = wl_f4, mc = m_cap;
    parallel_for(nb, [&](int i) {
      int* I = hI + (size_t)i * ib; S* O = hO + (size_t)i * ib * 2;
      const int F = (int)wl[i].M.size();
      std::memset(I, 0, (size_t)(4 + f4 + (size_t)F * mc) * sizeof(int));
      std::memset(O, 0, (size_t)F * mc * 2 * sizeof(S));
      I[0] = F;
      int* hM = I + 4; int* hS = I + 4 + f4;
      size_t o = 0;
      for (int t = 0; t < F; ++t) {
        const int m = wl[i].M[t];
        hM[t] = m;
        for (int k = 0; k < m; ++k) {
          hS[(size_t)t * mc + k] = wl[i].slots[o + k];
          O[((size_t)t * mc + k) * 2] = (S)wl[i].obs[2 * (o + k)];
          O[((size_t)t * mc + k) * 2 + 1] = (S)wl[i].obs[2 * (o + k) + 1];
        }
        o += m;
      }
      return 0;
    });
    HIPCHK(hipMemcpyAsync(wl_i + (size_t)b0 * wl_ib, hI, nI * sizeof(int), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(wl_obs + (size_t)b0 * wl_ib * 2, hO, nO * sizeof(S), hipMemcpyHostToDevice, st));
    rc = stage_release();
    if (rc) return rc;
    for (int i = 0; i < nb; ++i) traj[b0 + i].wl_F = (int)wl[i].M.size();
    return 0;
  }
  // last_stats of a marginalize() that had nothing to residualize (the reference returns early, msckf.h:337)
  int clear_stats(int b) override {
    if (chk(b)) return fail(-EINVAL, "trajectory index out of range");
    DEVICE_ENTER();
    HIPCHK(hipMemsetAsync(d.stats + (size_t)b * STAT_STRIDE, 0, sizeof(int) * STAT_ERR, st));
    return 0;
  }
  int clear_errors(int b) override {
    if (chk(b)) return fail(-EINVAL, "trajectory index out of range");
    DEVICE_ENTER();
    HIPCHK(hipMemsetAsync(d.stats + (size_t)b * STAT_STRIDE + STAT_ERR, 0, sizeof(int), st));
    return 0;
  }
  // every slice ran the same frames: all but the last of a call carry their prune on the downdate and flip the buffers
  void commit_buffer_parity(int f0, int f1) {
    const bool fuse = fuse_prune && !prof && !overlap_feature && d.joseph == 0;
    if (fuse && f1 - f0 >= 2 && ((f1 - f0 - 1) & 1)) std::swap(d.P, P_spare);
  }
  // compression route of an update's launches (k_feature publishes B^ only for the information form): the handle's choice,
  // except that a batch with a trajectory on the literal anisotropic route always takes the information form (that route hands
  // over an information matrix: Cholesky tail).  Every launch of an update -- the early k_feature of the overlap path too --
  // asks here.
  int update_compress(int base) const {
    int cmp = base;
    if (compress_route >= 0) cmp = (compress_route && d.trk_B) ? 3 : 0;
    if (n_lit > 0 && !cmp) cmp = d.compress;
    return cmp;
  }
  // ncam_ahead: run_frames advances its host mirror of the window sizes after the frame's launches (the frame's augmentState is
  // not in h_ncam yet when its update is enqueued).  Which route a frame takes depends on the window sizes only -- never on how a
  // frame range is cut into calls or on the API used (bit-identical results either way)
  void launch_update(const Dev<S>& vin, int b0, int nb, hipStream_t q, bool feature_done = false, int ncam_ahead = 0) {
    invalidate_imu(b0, nb);            // the update corrects the IMU state on the device (msckf.h:1376-1383)
    Dev<S> v = vin;
    v.compress = update_compress(v.compress);
    if (!feature_done) { stage_begin(2, q); launch_feature<S>(v, b0, nb, q); stage_end(2, q); }
    // information form: k_select and the block-diagonal reduction share a launch (both only read k_feature's outputs)
    stage_begin(7, q); if (v.compress) launch_select_diag<S>(v, b0, nb, q); else launch_select<S>(v, b0, nb, q); stage_end(7, q);
    // short windows (single filters, BASELINE configs[1]): everything after the selection in ONE launch (k_update_small)
    if (small_update && v.compress && n_lit == 0 && d.joseph == 0) {
      int nmax = 0;
      for (int b = b0; b < b0 + nb; ++b) nmax = std::max(nmax, 6 * std::min(h_ncam[b] + ncam_ahead, n_cap));
      if (nmax > 0 && nmax <= small_update) {
        stage_begin(5, q);
        const bool ok = launch_update_small<S>(v, b0, nb, q, nmax);
        stage_end(5, q);
        if (ok) return;
      }
    }
    if (v.compress) {
      // anisotropic pixel noise, literal route: the information matrix of the reference's (T_H, r_n, R_n) replaces H_o^T H_o
      // for those trajectories (kernels_literal.hip)
      stage_begin(3, q);
      launch_gram<S>(v, b0, nb, q, 3);                    // (the literal route starts from the same f64 Gram matrix)
      stage_end(3, q);
      if (n_lit > 0) {
        if (prof) for (int part = 1; part <= 3; ++part) { stage_begin(7 + part, q); launch_literal<S>(v, b0, nb, q, part); stage_end(7 + part, q); }
        else launch_literal<S>(v, b0, nb, q);
      }
      stage_begin(4, q); launch_gram<S>(v, b0, nb, q, 2); stage_end(4, q);
    } else {
      stage_begin(3, q); launch_compress<S>(v, b0, nb, q, 1); stage_end(3, q);
      stage_begin(4, q); launch_compress<S>(v, b0, nb, q, 2); stage_end(4, q);
    }
    stage_begin(5, q); launch_kalman<S>(v, b0, nb, q); stage_end(5, q);
  }
  int marginalize(int b0, int nb) override {
    POISON_GUARD();
    if (chk_range(b0, nb)) return fail(-EINVAL, "trajectory range out of bounds");
    DEVICE_ENTER();
    use_single_worklists();
    launch_update(view(b0), b0, nb, st);
    HIPCHK(hipGetLastError());
    return 0;
  }
  int set_given_positions(int b, int F, const double* pf3) override {
    if (chk(b) || F > f_cap) return fail(-EINVAL, "bad arguments");
    DEVICE_ENTER();
    std::vector<S> tmp((size_t)std::max(F, 1) * 4, S(0));
    for (int t = 0; t < F; ++t) for (int k = 0; k < 3; ++k) tmp[4 * t + k] = (S)pf3[3 * t + k];
    HIPCHK(hipMemcpyAsync(d_pfin + (size_t)b * f_cap * 4, tmp.data(), tmp.size() * sizeof(S), hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
    return 0;
  }
  int feature_only(int b, int* status, double* pf3, int cap) override {
    if (chk(b)) return fail(-EINVAL, "trajectory index out of range");
    DEVICE_ENTER();
    use_single_worklists();
    const int F = traj[b].wl_F;
    if (F > cap) return fail(-E2BIG, "output buffer too small");
    if (F == 0) return 0;
    launch_feature<S>(view(b), b, 1, st);
    HIPCHK(hipGetLastError());
    std::vector<int> stt(F); std::vector<S> pf((size_t)F * 4);
    HIPCHK(hipMemcpyAsync(stt.data(), d.trk_status + (size_t)b * f_cap, F * sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(pf.data(), d.trk_pf + (size_t)b * f_cap * 4, (size_t)F * 4 * sizeof(S), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (int t = 0; t < F; ++t) { status[t] = stt[t]; for (int k = 0; k < 3; ++k) pf3[3 * t + k] = (double)pf[4 * t + k]; }
    return F;
  }
  int marginalize_given(int b) override {
    POISON_GUARD();
    if (chk(b)) return fail(-EINVAL, "trajectory index out of range");
    DEVICE_ENTER();
    use_single_worklists();
    Dev<S> v = view(b);
    v.mode = 1;
    launch_update(v, b, 1, st);
    HIPCHK(hipGetLastError());
    return 0;
  }
  int prune_keep(int b, const std::vector<int>& keep) override {
    POISON_GUARD();
    DEVICE_ENTER();
    const int nk = (int)keep.size();
    // the list goes through the pinned ring like every other input: no wait for the stream in the middle of the image's chain
    // (the update before it, k_prune and the state read after it are one uninterrupted queue)
    unsigned char* raw = nullptr;
    int rc = stage_acquire(((size_t)nk + 1) * sizeof(int), &raw);
    if (rc) return rc;
    int* hk = reinterpret_cast<int*>(raw);
    hk[0] = nk; for (int i = 0; i < nk; ++i) hk[1 + i] = keep[i];
    if (nk) HIPCHK(hipMemcpyAsync(d.keep + (size_t)b * n_cap, hk + 1, nk * sizeof(int), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d.nkeep + b, hk, sizeof(int), hipMemcpyHostToDevice, st));
    rc = stage_release();
    if (rc) return rc;
    launch_prune<S>(d, b, 1, st);
    h_ncam[b] = nk;
    HIPCHK(hipGetLastError());
    return 0;
  }
  // ---- range forms (host_image_cycle)
  int cams_range(int b0, int nb, double* poses7) override {
    POISON_GUARD();
    if (chk_range(b0, nb)) return fail(-EINVAL, "trajectory range out of bounds");
    DEVICE_ENTER();
    const size_t per = (size_t)n_cap * CAM_STRIDE;
    std::vector<S> tmp(per * nb);
    HIPCHK(hipMemcpyAsync(tmp.data(), d.cam + (size_t)b0 * per, tmp.size() * sizeof(S), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (int i = 0; i < nb; ++i)
      for (int c = 0; c < n_cap; ++c)
        for (int k = 0; k < 7; ++k) poses7[((size_t)i * n_cap + c) * 7 + k] = (double)tmp[(size_t)i * per + (size_t)c * CAM_STRIDE + k];
    return 0;
  }
  int feature_only_range(int b0, int nb, int* status, double* pf3, bool launch) override {
    POISON_GUARD();
    if (chk_range(b0, nb)) return fail(-EINVAL, "trajectory range out of bounds");
    DEVICE_ENTER();
    if (launch) { use_single_worklists(); launch_feature<S>(view(b0), b0, nb, st); HIPCHK(hipGetLastError()); }
    std::vector<int> stt((size_t)nb * f_cap); std::vector<S> pf((size_t)nb * f_cap * 4);
    HIPCHK(hipMemcpyAsync(stt.data(), d.trk_status + (size_t)b0 * f_cap, stt.size() * sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(pf.data(), d.trk_pf + (size_t)b0 * f_cap * 4, pf.size() * sizeof(S), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (size_t t = 0; t < stt.size(); ++t) { status[t] = stt[t]; for (int k = 0; k < 3; ++k) pf3[3 * t + k] = (double)pf[4 * t + k]; }
    return 0;
  }
  int set_given_range(int b0, int nb, const double* pf3) override {
    if (chk_range(b0, nb)) return fail(-EINVAL, "trajectory range out of bounds");
    DEVICE_ENTER();
    const size_t cnt = (size_t)nb * f_cap * 4;
    unsigned char* raw = nullptr;
    int rc = stage_acquire(cnt * sizeof(S), &raw);
    if (rc) return rc;
    S* tmp = reinterpret_cast<S*>(raw);
    for (size_t t = 0; t < (size_t)nb * f_cap; ++t) { for (int k = 0; k < 3; ++k) tmp[4 * t + k] = (S)pf3[3 * t + k]; tmp[4 * t + 3] = S(0); }
    HIPCHK(hipMemcpyAsync(d_pfin + (size_t)b0 * f_cap * 4, tmp, cnt * sizeof(S), hipMemcpyHostToDevice, st));
    return stage_release();
  }
  int marginalize_given_range(int b0, int nb) override {
    POISON_GUARD();
    if (chk_range(b0, nb)) return fail(-EINVAL, "trajectory range out of bounds");
    DEVICE_ENTER();
    use_single_worklists();
    Dev<S> v = view(b0);
    v.mode = 1;
    launch_update(v, b0, nb, st);
    HIPCHK(hipGetLastError());
    return 0;
  }
  int prune_keep_range(int b0, int nb, const std::vector<std::vector<int>>& keep) override {
    POISON_GUARD();
    if (chk_range(b0, nb) || (int)keep.size() != nb) return fail(-EINVAL, "trajectory range out of bounds");
    DEVICE_ENTER();
    // [nb][n_cap] slots + [nb] counts through the pinned ring, two copies, one launch (a trajectory that keeps everything is a no-op there)
    const size_t nI = (size_t)nb * n_cap + nb;
    unsigned char* raw = nullptr;
    int rc = stage_acquire(nI * sizeof(int), &raw);
    if (rc) return rc;
    int* hk = reinterpret_cast<int*>(raw); int* hn = hk + (size_t)nb * n_cap;
    std::memset(hk, 0, nI * sizeof(int));
    for (int i = 0; i < nb; ++i) {
      const int nk = (int)keep[i].size();
      if (nk > n_cap) return fail(-EINVAL, "keep list longer than n_cap");
      hn[i] = nk;
      for (int k = 0; k < nk; ++k) hk[(size_t)i * n_cap + k] = keep[i][k];
    }
    HIPCHK(hipMemcpyAsync(d.keep + (size_t)b0 * n_cap, hk, (size_t)nb * n_cap * sizeof(int), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d.nkeep + b0, hn, nb * sizeof(int), hipMemcpyHostToDevice, st));
    rc = stage_release();
    if (rc) return rc;
    launch_prune<S>(d, b0, nb, st);
    for (int i = 0; i < nb; ++i) h_ncam[b0 + i] = (int)keep[i].size();
    HIPCHK(hipGetLastError());
    return 0;
  }
  int drop_oldest(int b0, int nb, int n) override;
  int ncam_host(int b) const override { return (b < 0 || b >= B) ? -EINVAL : (poisoned ? -EIO : h_ncam[b]); }   // a poisoned handle's count is not to be trusted (its slices may have stopped at different frames)
  int get_ncam(int b, int* n) override {
    POISON_GUARD();
    if (chk(b)) return fail(-EINVAL, "trajectory index out of range");
    DEVICE_ENTER();
    HIPCHK(hipMemcpyAsync(n, d.ncam + b, sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return 0;
  }
  int get_imu(int b, double* o) override {
    POISON_GUARD();
    if (chk(b)) return fail(-EINVAL, "trajectory index out of range");
    S* tmp = h_imu.data() + (size_t)b * IMU_STRIDE;
    if (!h_imu_ok[b]) {
      DEVICE_ENTER();
      HIPCHK(hipMemcpyAsync(tmp, d.imu + (size_t)b * IMU_STRIDE, IMU_STRIDE * sizeof(S), hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
      h_imu_ok[b] = 1;
    }
    for (int i = 0; i < 29; ++i) o[i] = (double)tmp[i];
    return 0;
  }
  int set_imu(int b, const double* in) override {
    if (chk(b)) return fail(-EINVAL, "trajectory index out of range");
    DEVICE_ENTER();
    S tmp[IMU_STRIDE] = {0};
    for (int i = 0; i < 29; ++i) tmp[i] = (S)in[i];
    HIPCHK(hipMemcpyAsync(d.imu + (size_t)b * IMU_STRIDE, tmp, sizeof(tmp), hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
    std::copy(tmp, tmp + IMU_STRIDE, h_imu.begin() + (size_t)b * IMU_STRIDE); h_imu_ok[b] = 1;
    return 0;
  }
  int get_cams(int b, double* o, int cap, int* nout) override {
    POISON_GUARD();
    int n = 0;
    int rc = get_ncam(b, &n);
    if (rc) return rc;
    *nout = n;
    if (n > cap) return fail(-E2BIG, "output buffer too small");
    std::vector<S> tmp((size_t)std::max(n, 1) * CAM_STRIDE);
    if (n) HIPCHK(hipMemcpyAsync(tmp.data(), d.cam + (size_t)b * n_cap * CAM_STRIDE, (size_t)n * CAM_STRIDE * sizeof(S), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (int i = 0; i < n; ++i) for (int k = 0; k < 7; ++k) o[7 * i + k] = (double)tmp[(size_t)i * CAM_STRIDE + k];
    return 0;
  }
  // the single-filter API knows its window size on the host: no count read first, and the IMU state rides along into the host
  // copy (nothing changes it between an update and the next propagate), so that the getImuState() that follows costs no wait
  int get_cams_known(int b, double* o, int n) override {
    POISON_GUARD();
    if (chk(b) || n < 0 || n > n_cap) return fail(-EINVAL, "index out of range");
    DEVICE_ENTER();
    S* tmp = h_rb; S* tim = h_rb + (size_t)n_cap * CAM_STRIDE;        // (page-locked: a pageable destination goes through the runtime's own staging copy)
    if (n) HIPCHK(hipMemcpyAsync(tmp, d.cam + (size_t)b * n_cap * CAM_STRIDE, (size_t)n * CAM_STRIDE * sizeof(S), hipMemcpyDeviceToHost, st));
    const bool want_imu = !h_imu_ok[b];
    if (want_imu) HIPCHK(hipMemcpyAsync(tim, d.imu + (size_t)b * IMU_STRIDE, IMU_STRIDE * sizeof(S), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (want_imu) { std::copy(tim, tim + IMU_STRIDE, h_imu.begin() + (size_t)b * IMU_STRIDE); h_imu_ok[b] = 1; }
    for (int i = 0; i < n; ++i) for (int k = 0; k < 7; ++k) o[7 * i + k] = (double)tmp[(size_t)i * CAM_STRIDE + k];
    return 0;
  }
  int set_cam(int b, int slot, const double* in) override {
    if (chk(b) || slot < 0 || slot >= n_cap) return fail(-EINVAL, "index out of range");
    DEVICE_ENTER();
    S tmp[CAM_STRIDE] = {0};
    for (int k = 0; k < 7; ++k) tmp[k] = (S)in[k];
    HIPCHK(hipMemcpyAsync(d.cam + ((size_t)b * n_cap + slot) * CAM_STRIDE, tmp, sizeof(tmp), hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
    return 0;
  }
  int get_cov(int b, double* P, int ldo) override {
    POISON_GUARD();
    int n = 0;
    int rc = get_ncam(b, &n);
    if (rc) return rc;
    const int D = 15 + 6 * n;
    if (ldo < D) return fail(-EINVAL, "ld smaller than D");
    std::vector<S> tmp((size_t)d.ld * d.ld);
    HIPCHK(hipMemcpyAsync(tmp.data(), d.P + (size_t)b * d.ld * d.ld, tmp.size() * sizeof(S), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (int j = 0; j < D; ++j) for (int i = 0; i < D; ++i) P[(size_t)j * ldo + i] = (double)tmp[(size_t)j * d.ld + i];
    return 0;
  }
  int set_cov(int b, const double* P, int D) override {
    if (chk(b)) return fail(-EINVAL, "trajectory index out of range");
    if (D < 15 || (D - 15) % 6 || (D - 15) / 6 > n_cap) return fail(-EINVAL, "bad covariance dimension");
    DEVICE_ENTER();
    std::vector<S> tmp((size_t)d.ld * d.ld, S(0));
    for (int j = 0; j < D; ++j) for (int i = 0; i < D; ++i) tmp[(size_t)j * d.ld + i] = (S)P[(size_t)j * D + i];
    const int n = (D - 15) / 6;
    HIPCHK(hipMemcpyAsync(d.P + (size_t)b * d.ld * d.ld, tmp.data(), tmp.size() * sizeof(S), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d.ncam + b, &n, sizeof(int), hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
    h_ncam[b] = n;
    return 0;
  }
  int get_nres(int b, long long* n) override {
    if (chk(b)) return fail(-EINVAL, "trajectory index out of range");
    DEVICE_ENTER();
    HIPCHK(hipMemcpyAsync(n, d.n_resid + b, sizeof(long long), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return 0;
  }
  int set_nres(int b, long long n) override {
    if (chk(b)) return fail(-EINVAL, "trajectory index out of range");
    DEVICE_ENTER();
    HIPCHK(hipMemcpyAsync(d.n_resid + b, &n, sizeof(long long), hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
    return 0;
  }
  int stats(int b, int* out) override {
    POISON_GUARD();
    if (chk(b)) return fail(-EINVAL, "trajectory index out of range");
    DEVICE_ENTER();
    int tmp[STAT_STRIDE];
    HIPCHK(hipMemcpyAsync(tmp, d.stats + (size_t)b * STAT_STRIDE, sizeof(tmp), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (int i = 0; i < 7; ++i) out[i] = tmp[i];
    if (tmp[STAT_ERR] & STAT_ERR_NCAP) return fail(-EOVERFLOW, "camera-state capacity n_cap exceeded in augmentState");
    if (tmp[STAT_ERR] & STAT_ERR_PIVOT)
      return fail(-EDOM, "non-positive pivot in the factorization of S = T_H P T_H^T + R_n: the covariance lost positive definiteness "
                         "(msckf_hip_set_covariance_update(h, 1) selects the reference's Joseph form)");
    return 0;
  }
  // value semantics of the reference object (MSCKF<_S> is copyable, msckf.h:31-67): the filter state of every trajectory
  // -- IMU / camera states, parameters, covariance, window size, counters, flags -- and the host-side track bookkeeping;
  // work buffers and a resident scenario are not state and are not copied
  int copy_from(BatchBase* src) override {
    POISON_GUARD();
    Batch<S>* o = dynamic_cast<Batch<S>*>(src);
    if (!o || o->B != B || o->n_cap != n_cap || o->f_cap != f_cap || o->m_cap != m_cap || o->h16 != h16)
      return fail(-EINVAL, "copy_state: handles differ in shape or dtype");
    if (o->poisoned) return fail(-EIO, "copy_state: the source handle is unusable after a failed run_frames call (its filter states are undefined)");
    { const int rcf = o->flush_pending(); if (rcf) return rcf; }
    for (int b = 0; b < o->B; ++b) { const int rcm = resolve_map(o, b); if (rcm) return rcm; }   // (work buffers are not copied: points still on the device first)
    DEVICE_ENTER();
    HIPCHK(hipStreamSynchronize(o->st));
    const size_t Bz = B, pl = (size_t)d.ld * d.ld;
    auto cp = [&](void* dst, const void* sp, size_t bytes) { return hipMemcpyAsync(dst, sp, bytes, hipMemcpyDeviceToDevice, st); };
    HIPCHK(cp(d.imu, o->d.imu, Bz * IMU_STRIDE * sizeof(S))); HIPCHK(cp(d.cam, o->d.cam, Bz * n_cap * CAM_STRIDE * sizeof(S)));
    HIPCHK(cp(d.prm, o->d.prm, Bz * PRM_STRIDE * sizeof(S))); HIPCHK(cp(d.P, o->d.P, Bz * pl * sizeof(S)));
    HIPCHK(cp(d.ncam, o->d.ncam, Bz * sizeof(int))); HIPCHK(cp(d.n_resid, o->d.n_resid, Bz * sizeof(long long)));
    HIPCHK(cp(d.stats, o->d.stats, Bz * STAT_STRIDE * sizeof(int))); HIPCHK(cp(d.ncam_upd, o->d.ncam_upd, Bz * sizeof(int)));
    traj = o->traj; h_ncam = o->h_ncam; h_uv = o->h_uv; h_imu = o->h_imu; h_imu_ok = o->h_imu_ok;
    compress_route = o->compress_route; d.joseph = o->d.joseph; d.gate_early = o->d.gate_early; nstreams = o->nstreams;
    overlap_feature = o->overlap_feature; d.gain_fused_s = o->d.gain_fused_s; fuse_prune = o->fuse_prune;
    HIPCHK(hipStreamSynchronize(st));
    std::fill(h_lit.begin(), h_lit.end(), 0); n_lit = 0;   // which trajectories run the literal route is re-derived from the copied parameters
    return set_aniso(o->aniso_mode, o->lit_tol);   // re-derives the per-trajectory noise parameters, allocates the literal route's work space if needed
  }
  int error_flags(int b, int* flags) override {
    if (chk(b)) return fail(-EINVAL, "trajectory index out of range");
    DEVICE_ENTER();
    HIPCHK(hipMemcpyAsync(flags, d.stats + (size_t)b * STAT_STRIDE + STAT_ERR, sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return 0;
  }
  int track_info(int b, double* out, int cap) override {
    if (chk(b)) return fail(-EINVAL, "trajectory index out of range");
    DEVICE_ENTER();
    int tmp[STAT_STRIDE];
    HIPCHK(hipMemcpyAsync(tmp, d.stats + (size_t)b * STAT_STRIDE, sizeof(tmp), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    const int F = tmp[STAT_NTRACKS];
    if (F > cap) return fail(-E2BIG, "output buffer too small");
    std::vector<int> stt(std::max(F, 1)); std::vector<S> gm(std::max(F, 1)), pf((size_t)std::max(F, 1) * 4);
    if (F) {
      HIPCHK(hipMemcpyAsync(stt.data(), d.trk_status + (size_t)b * f_cap, F * sizeof(int), hipMemcpyDeviceToHost, st));
      HIPCHK(hipMemcpyAsync(gm.data(), d.trk_gamma + (size_t)b * f_cap, F * sizeof(S), hipMemcpyDeviceToHost, st));
      HIPCHK(hipMemcpyAsync(pf.data(), d.trk_pf + (size_t)b * f_cap * 4, (size_t)F * 4 * sizeof(S), hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
    }
    for (int t = 0; t < F; ++t) {
      double* o = out + 8 * t;
      const bool skipped = stt[t] & ST_MOTION_SKIPPED;
      o[0] = (skipped || (stt[t] & ST_MOTION_OK)) ? 1 : 0;
      o[1] = (stt[t] & ST_TRI_VALID) ? 1 : 0; o[2] = (stt[t] & ST_GATE_PASS) ? 1 : 0; o[3] = (stt[t] & ST_INCLUDED) ? 1 : 0;
      o[4] = (double)gm[t]; o[5] = (double)pf[4 * t]; o[6] = (double)pf[4 * t + 1]; o[7] = (double)pf[4 * t + 2];
    }
    return F;
  }
  int deltax(int b, double* out, int cap) override {
    int n = 0;
    int rc = get_ncam(b, &n);
    if (rc) return rc;
    const int D = 15 + 6 * n;
    if (D > cap) return fail(-E2BIG, "output buffer too small");
    std::vector<S> tmp(D);
    HIPCHK(hipMemcpyAsync(tmp.data(), d.dx + (size_t)b * d.ld, D * sizeof(S), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (int i = 0; i < D; ++i) out[i] = (double)tmp[i];
    return D;
  }
  // ---- scenario
  void free_scenario_device() {
    for (void* q : sc_allocs) {                                  // a previous scenario is replaced, not leaked
      hipFree(q);
      allocs.erase(std::remove(allocs.begin(), allocs.end(), q), allocs.end());
    }
    sc_allocs.clear();
    sc_rd = nullptr; sc_n = sc_M = sc_off = sc_drop = sc_slots = nullptr; sc_obs = nullptr; sc_total = 0;
  }
  template <class T> int sc_dalloc(T** p, size_t count) {
    const size_t mark = allocs.size();
    const int rc = dalloc(p, count);
    sc_allocs.insert(sc_allocs.end(), allocs.begin() + mark, allocs.end());
    return rc;
  }
  int scen_alloc(int n_frames, int K) override {
    if (n_frames <= 0 || K <= 0) return fail(-EINVAL, "bad scenario size");
    DEVICE_ENTER();
    HIPCHK(hipStreamSynchronize(st));
    free_scenario_device();
    sc_frames = 0; committed = false;
    unpin_host();
    const size_t Bz = B, FB = (size_t)n_frames * Bz;
    h_rd.assign(FB * K * RD_STRIDE, S(0)); h_n.assign(FB, 0); h_M.assign(FB * f_cap, 0); h_off.assign(FB * f_cap, 0);
    h_drop.assign(FB, 0); h_maxslot.assign(FB, -1);
    c_slots.assign(FB, std::vector<int>()); c_obs.assign(FB, std::vector<S>());
    fr_base.assign((size_t)n_frames + 1, 0);
    pinf.assign((size_t)n_frames, PinFrame());
    int rc = 0;
    rc |= sc_dalloc(&sc_rd, h_rd.size()); rc |= sc_dalloc(&sc_n, h_n.size()); rc |= sc_dalloc(&sc_M, h_M.size());
    rc |= sc_dalloc(&sc_off, h_off.size()); rc |= sc_dalloc(&sc_drop, h_drop.size());
    if (rc) return rc;
    {   // fixed sections of a streamed frame block; the frame's slots start at pk_slots, its observations follow them
      auto al = [](size_t x) { return (x + 255) / 256 * 256; };
      pk_rd = 0; pk_n = al(pk_rd + Bz * K * RD_STRIDE * sizeof(S)); pk_drop = al(pk_n + Bz * sizeof(int));
      pk_M = al(pk_drop + Bz * sizeof(int)); pk_off = al(pk_M + Bz * f_cap * sizeof(int)); pk_slots = al(pk_off + Bz * f_cap * sizeof(int));
    }
    sc_frames = n_frames; sc_K = K;
    return 0;
  }
  int scen_set(int f, int b, const double* rd, int F, const int* M, const int* slots, const double* obs, int n_drop) override {
    if (f < 0 || f >= sc_frames || chk(b)) return fail(-EINVAL, "scenario cell out of range");
    if (F < 0 || F > f_cap) return fail(-E2BIG, "more tracks than f_cap");
    if (n_drop < 0) return fail(-EINVAL, "negative n_drop");
    const size_t cell = (size_t)f * B + b;
    size_t tot = 0;
    {   // validate before touching the staged cell (same rules as set_tracks)
      for (int t = 0; t < F; ++t) {
        if (M[t] > m_cap || M[t] < 0) return fail(-E2BIG, "track longer than m_cap");
        for (int k = 0; k < M[t]; ++k) if (slots[tot + k] < 0 || slots[tot + k] >= n_cap) return fail(-EINVAL, "camera slot out of range");
        if (repeated_slot(slots + tot, M[t])) return fail(-EINVAL, "camera slot repeated within a track");
        tot += M[t];
      }
    }
    for (int k = 0; k < sc_K; ++k) for (int c = 0; c < RD_STRIDE; ++c) h_rd[(cell * sc_K + k) * RD_STRIDE + c] = (S)rd[k * RD_STRIDE + c];
    h_n[cell] = F; h_drop[cell] = n_drop;
    for (int t = 0; t < f_cap; ++t) h_M[cell * f_cap + t] = t < F ? M[t] : 0;
    c_slots[cell].assign(slots, slots + tot);
    c_obs[cell].resize(2 * tot);
    int mx = -1;
    for (size_t e = 0; e < tot; ++e) { mx = std::max(mx, slots[e]); c_obs[cell][2 * e] = (S)obs[2 * e]; c_obs[cell][2 * e + 1] = (S)obs[2 * e + 1]; }
    h_maxslot[cell] = mx;
    committed = false;               // offsets move: the resident copy and the frame's page-locked block are stale until the next commit
    unpin_frame(f);                  // (its chunk is released with the last of its frames: patch -> commit -> stream cycles do not grow)
    return 0;
  }
  // H2D of everything staged.  The host copy is kept, so cells may be patched with scenario_set and committed again.
  int scen_commit() override {
    if (sc_frames <= 0) return fail(-EINVAL, "no scenario allocated");
    DEVICE_ENTER();
    HIPCHK(hipStreamSynchronize(st));
    const size_t Bz = B;
    size_t total = 0;
    for (int f = 0; f < sc_frames; ++f) {
      fr_base[f] = total;
      size_t in_frame = 0;
      for (size_t b = 0; b < Bz; ++b) {
        const size_t cell = (size_t)f * Bz + b;
        size_t o = in_frame;
        for (int t = 0; t < f_cap; ++t) { h_off[cell * f_cap + t] = (int)o; o += h_M[cell * f_cap + t]; }
        in_frame += c_slots[cell].size();
      }
      if (in_frame > 0x7fffffffu) return fail(-E2BIG, "a frame's work-lists exceed 2^31 observations");
      total += in_frame;
    }
    fr_base[sc_frames] = total;
    if (total != sc_total || !sc_slots) {          // patched cells may have changed the compact size
      for (void* q : {(void*)sc_slots, (void*)sc_obs})
        if (q) { hipFree(q); allocs.erase(std::remove(allocs.begin(), allocs.end(), q), allocs.end()); sc_allocs.erase(std::remove(sc_allocs.begin(), sc_allocs.end(), q), sc_allocs.end()); }
      sc_slots = nullptr; sc_obs = nullptr;
      int rc = sc_dalloc(&sc_slots, total); rc |= sc_dalloc(&sc_obs, 2 * total);
      if (rc) return rc;
      sc_total = total;
    }
    HIPCHK(hipMemcpyAsync(sc_rd, h_rd.data(), h_rd.size() * sizeof(S), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(sc_n, h_n.data(), h_n.size() * sizeof(int), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(sc_M, h_M.data(), h_M.size() * sizeof(int), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(sc_off, h_off.data(), h_off.size() * sizeof(int), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(sc_drop, h_drop.data(), h_drop.size() * sizeof(int), hipMemcpyHostToDevice, st));
    {   // size the pinned staging ring once, for the largest frame
      size_t mx = 0;
      for (int f = 0; f < sc_frames; ++f) mx = std::max(mx, fr_base[f + 1] - fr_base[f]);
      const int rc = stage_reserve(mx * (sizeof(int) + 2 * sizeof(S)));
      if (rc) return rc;
    }
    for (int f = 0; f < sc_frames; ++f) {            // one frame at a time through the pinned staging area (bounded host memory)
      const size_t nf = fr_base[f + 1] - fr_base[f];
      if (!nf) continue;
      unsigned char* raw = nullptr;
      int rc = stage_acquire(nf * (sizeof(int) + 2 * sizeof(S)), &raw);
      if (rc) return rc;
      int* hs = reinterpret_cast<int*>(raw); S* ho = reinterpret_cast<S*>(raw + nf * sizeof(int));
      size_t o = 0;
      for (size_t b = 0; b < Bz; ++b) {
        const size_t cell = (size_t)f * Bz + b, n = c_slots[cell].size();
        if (n) { std::memcpy(hs + o, c_slots[cell].data(), n * sizeof(int)); std::memcpy(ho + 2 * o, c_obs[cell].data(), 2 * n * sizeof(S)); }
        o += n;
      }
      HIPCHK(hipMemcpyAsync(sc_slots + fr_base[f], hs, nf * sizeof(int), hipMemcpyHostToDevice, st));
      HIPCHK(hipMemcpyAsync(sc_obs + 2 * fr_base[f], ho, 2 * nf * sizeof(S), hipMemcpyHostToDevice, st));
      rc = stage_release();
      if (rc) return rc;
    }
    HIPCHK(hipStreamSynchronize(st));
    committed = true;
    return 0;
  }
  // page-locked per-frame blocks for run_frames_streamed, frames [f0, f1) that do not have one yet; also sizes the device
  // staging ring.  Called by run_frames_streamed itself; call it beforehand to keep the pinning out of a timed region.
  int scen_pin(int f0, int f1) override {
    if (f0 < 0 || f1 > sc_frames || f0 > f1) return fail(-EINVAL, "frame range out of bounds");
    if (!committed) return fail(-EINVAL, "scenario not committed");
    DEVICE_ENTER();
    auto al = [](size_t x) { return (x + 255) / 256 * 256; };
    const size_t Bz = B;
    size_t need = 0, maxb = sg_bytes;
    std::vector<int> todo;
    for (int f = f0; f < f1; ++f) {
      const size_t nf = fr_base[f + 1] - fr_base[f];
      const size_t off_obs = al(pk_slots + nf * sizeof(int)), bytes = al(off_obs + 2 * nf * sizeof(S));
      maxb = std::max(maxb, bytes);
      if (pinf[f].p) continue;
      pinf[f].bytes = bytes; pinf[f].off_obs = off_obs;
      need += bytes; todo.push_back(f);
    }
    if (!todo.empty()) {
      unsigned char* chunk = nullptr;
      if (hipHostMalloc((void**)&chunk, need, hipHostMallocDefault) != hipSuccess) {
        for (int f : todo) pinf[f] = PinFrame();
        return fail(-ENOMEM, "could not page-lock the frames to stream (run_frames on the resident scenario is unaffected)");
      }
      int ci = -1;
      for (size_t q = 0; q < pin_chunks.size(); ++q) if (!pin_chunks[q].p) { ci = (int)q; break; }
      if (ci < 0) { pin_chunks.push_back(PinChunk()); ci = (int)pin_chunks.size() - 1; }
      pin_chunks[ci].p = chunk; pin_chunks[ci].live = (int)todo.size();
      size_t o = 0;
      for (int f : todo) {
        unsigned char* blk = chunk + o;
        const size_t c0 = (size_t)f * Bz;
        std::memcpy(blk + pk_rd, h_rd.data() + c0 * sc_K * RD_STRIDE, Bz * sc_K * RD_STRIDE * sizeof(S));
        std::memcpy(blk + pk_n, h_n.data() + c0, Bz * sizeof(int));
        std::memcpy(blk + pk_drop, h_drop.data() + c0, Bz * sizeof(int));
        std::memcpy(blk + pk_M, h_M.data() + c0 * f_cap, Bz * f_cap * sizeof(int));
        std::memcpy(blk + pk_off, h_off.data() + c0 * f_cap, Bz * f_cap * sizeof(int));
        int* hs = reinterpret_cast<int*>(blk + pk_slots); S* ho = reinterpret_cast<S*>(blk + pinf[f].off_obs);
        size_t e = 0;
        for (size_t b = 0; b < Bz; ++b) {
          const size_t cell = c0 + b, n = c_slots[cell].size();
          if (n) { std::memcpy(hs + e, c_slots[cell].data(), n * sizeof(int)); std::memcpy(ho + 2 * e, c_obs[cell].data(), 2 * n * sizeof(S)); }
          e += n;
        }
        pinf[f].p = blk; pinf[f].chunk = ci;
        o += pinf[f].bytes;
      }
    }
    if (maxb > sg_bytes || !sg_blk[0]) {
      HIPCHK(hipStreamSynchronize(st));
      if (stc) HIPCHK(hipStreamSynchronize(stc));
      for (int k = 0; k < RING_MAX; ++k) { if (sg_blk[k]) hipFree(sg_blk[k]); sg_blk[k] = nullptr; }
      for (int k = 0; k < RING_MAX; ++k) HIPCHK(hipMalloc((void**)&sg_blk[k], maxb));
      sg_bytes = maxb;
    }
    return 0;
  }
  int run_frames(int f0, int f1) override;
  int run_frames_streamed(int f0, int f1) override;
  void unpin_host() {
    for (auto& q : pin_chunks) if (q.p) hipHostFree(q.p);
    pin_chunks.clear();
    for (auto& pf : pinf) pf = PinFrame();
  }
  int set_upload_ring(int depth, int mode) override {
    if (depth < 2 || depth > RING_MAX || mode < 0 || mode > 1) return fail(-EINVAL, "ring depth 2..8; mode 0 host hand-over, 1 device-side event waits");
    ring = depth; up_mode = mode;
    return 0;
  }
  // streams of the nh slices of a run
  int slice_streams(int nh, hipStream_t* qs) {
    for (int i = 0; i < nh; ++i) qs[i] = stx[i];
    return 0;
  }
  int fork_slices(int nh, hipStream_t* qs, hipStream_t extra = nullptr) {
    if (nh <= 1 && !extra) return 0;
    HIPCHK(hipEventRecord(ev_fork, st));
    for (int i = 0; i < nh; ++i) if (qs[i] != st) HIPCHK(hipStreamWaitEvent(qs[i], ev_fork, 0));
    if (extra) HIPCHK(hipStreamWaitEvent(extra, ev_fork, 0));
    return 0;
  }
  int join_slices(int nh, hipStream_t* qs) {
    for (int i = 0; i < nh; ++i)
      if (qs[i] != st) { HIPCHK(hipEventRecord(ev_join[i], qs[i])); HIPCHK(hipStreamWaitEvent(st, ev_join[i], 0)); }
    return 0;
  }
  int sync() override {
    DEVICE_ENTER();
    HIPCHK(hipStreamSynchronize(st));
    return 0;
  }
  int prof_enable(int on) override {
    prof = on != 0;
    for (int s = 0; s < NSTAGE; ++s) { ev_used[s] = 0; prof_ms[s] = 0; prof_cnt[s] = 0; }
    return 0;
  }
  int set_gate_early(int on) override { d.gate_early = on ? 1 : 0; return 0; }
  int set_feature_overlap(int on) override { overlap_feature = on ? 1 : 0; return 0; }
  int set_cov_update(int form) override {
    if (form < 0 || form > 2) return fail(-EINVAL, "form: 0 square-root gain (P - W W^T), 1 Joseph, 2 square-root gain with the register-resident solve");
    d.joseph = form;
    return 0;
  }
  int set_compression(int route) override {
    if (route < -1 || route > 3) return fail(-EINVAL, "route: -1 default, 0 Householder TSQR, 1..3 information form (blocked matrix-core Cholesky; 1 and 2 named retired factorizations)");
    if (route >= 1 && !d.trk_B) return fail(-ENOTSUP, "information form not available for this window size (6 n_cap + 1 > 384 or f_cap > 1024)");
    compress_route = route;
    return 0;
  }
  int set_host_affinity(const int* cpus, int n) override {
    std::lock_guard<std::mutex> lk(workers.m);
    workers.cpus.assign(cpus, cpus + std::max(n, 0));
    return 0;
  }
  int set_streams(int n) override {
    if (n < 1 || n > MAXS) return fail(-EINVAL, "1 to 8 streams");
    nstreams = n;
    return 0;
  }
  // what an event pair with NOTHING between its records measures on this stream (the marker packets themselves): the stage
  // timers of prof_read hold one such pair per launch, so a single-kernel stage reads kernel time + this
  int prof_event_overhead(double* ms) override {
    DEVICE_ENTER();
    hipEvent_t a, b2;
    HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b2));
    HIPCHK(hipStreamSynchronize(st));
    double tot = 0; const int reps = 64;
    for (int i = 0; i < reps; ++i) {
      HIPCHK(hipEventRecord(a, st)); HIPCHK(hipEventRecord(b2, st));
      HIPCHK(hipEventSynchronize(b2));
      float t = 0; HIPCHK(hipEventElapsedTime(&t, a, b2));
      tot += t;
    }
    hipEventDestroy(a); hipEventDestroy(b2);
    *ms = tot / reps;
    return 0;
  }
  int prof_read(double* ms, int* cnt, int cap) override {
    DEVICE_ENTER();
    HIPCHK(hipStreamSynchronize(st));
    for (int s = 0; s < NSTAGE; ++s) {
      for (size_t i = 0; i < ev_used[s]; ++i) {
        float t = 0;
        HIPCHK(hipEventElapsedTime(&t, ev_pool[s][i].first, ev_pool[s][i].second));
        prof_ms[s] += t; prof_cnt[s]++;
      }
      ev_used[s] = 0;
      if (s < cap) { ms[s] = prof_ms[s]; cnt[s] = prof_cnt[s]; }
    }
    return 0;
  }
};

template <class S>
int Batch<S>::drop_oldest(int b0, int nb, int n) {
  POISON_GUARD();
  if (chk_range(b0, nb)) return fail(-EINVAL, "trajectory range out of bounds");
  DEVICE_ENTER();
  launch_prune<S>(d, b0, nb, st, nullptr, std::max(n, 0));
  for (int b = b0; b < b0 + nb; ++b) h_ncam[b] -= std::max(0, std::min(n, h_ncam[b]));
  HIPCHK(hipGetLastError());
  return 0;
}

template <class S>
int Batch<S>::run_frames(int f0, int f1) {
  POISON_GUARD();
  if (f0 < 0 || f1 > sc_frames || f0 > f1) return fail(-EINVAL, "frame range out of bounds");
  if (!committed) return fail(-EINVAL, "scenario not committed");
  DEVICE_ENTER();
  // Trajectories are independent, so the batch may be cut into slices that run the same kernel sequence on
  // separate streams: the latency-bound stages of one slice (gain solve, Cholesky, propagate: one workgroup per
  // trajectory) overlap with the chip-filling stages of the others.  Stage profiling forces a single stream.
  const int nh = prof ? 1 : std::max(1, std::min(nstreams, B));
  hipStream_t qs[MAXS];
  int rc = slice_streams(nh, qs);
  if (rc) return rc;
  rc = fork_slices(nh, qs);
  if (rc) return rc;
  // one host thread per slice enqueues that slice's kernels for all frames: ~13 launches per frame and slice would
  // otherwise serialise on one thread and make more than two slices launch-bound
  int slice_rc[MAXS] = {0};
  auto enqueue = [&](int hh) {
    (void)hipSetDevice(device);
    (void)hipGetLastError();
    hipStream_t q = qs[hh];
    const int b0 = (int)((long)B * hh / nh);
    const int nb = (int)((long)B * (hh + 1) / nh) - b0;
    S* curP = d.P; S* spare = P_spare;   // a frame whose prune rides on the downdate leaves the covariance in the other buffer
    bool pending = false;                // ... and the new window size for the next k_propagate to commit
    for (int f = f0; f < f1; ++f) {
      const size_t cell0 = (size_t)f * B;
      Dev<S> v = d;
      v.P = curP; v.ncam_defer = pending ? 1 : 0;
      // the call's last frame prunes with its own launch: ncam must be final when run_frames returns
      const bool fuse = fuse_prune && !prof && !overlap_feature && d.joseph == 0 && f + 1 < f1;
      v.trk_n = sc_n + cell0 + b0; v.trk_M = sc_M + (cell0 + b0) * f_cap; v.trk_off = sc_off + (cell0 + b0) * f_cap;
      v.trk_slots = sc_slots + fr_base[f]; v.trk_obs = sc_obs + 2 * fr_base[f];
      v.wl_stride_n = 1; v.wl_stride_f = f_cap; v.wl_stride_o = 0;
      // k_feature reads only what the previous frame's prune left behind -- camera states and P blocks of slots below the
      // newest one, the constant gravity vector -- unless a track observes the camera this frame's augmentState adds.
      // When none does (host mirror of the window sizes, slots known since scenario_set) it runs on a side stream
      // concurrently with the latency-bound propagate + augment of the same frame.
      bool early = overlap_feature && !prof && f > f0;   // f0: the previous call need not have ended with a prune (ncam_upd)
      for (int b = b0; b < b0 + nb && early; ++b) {
        const int n_after = std::min(h_ncam[b] + 1, n_cap);
        early = h_ncam[b] < n_cap && h_maxslot[cell0 + b] <= n_after - 2;
      }
      if (early) {
        (void)hipEventRecord(ev_fa[hh], q);
        (void)hipStreamWaitEvent(sty[hh], ev_fa[hh], 0);
        Dev<S> v2 = v; v2.ncam_bias = 1;
        v2.compress = update_compress(v2.compress);
        launch_feature<S>(v2, b0, nb, sty[hh]);
        (void)hipEventRecord(ev_fb[hh], sty[hh]);
      }
      // propagate and augmentState are always back to back here: one launch (the per-stage profile keeps them apart)
      {
        StageRange r("imu_prop+msckf_augment_state");
        stage_begin(0, q); launch_propagate<S>(v, b0, nb, sc_rd + (cell0 + b0) * sc_K * RD_STRIDE, (long)sc_K * RD_STRIDE, sc_K, q, !prof); stage_end(0, q);
        if (prof) { stage_begin(1, q); launch_augment<S>(v, b0, nb, q); stage_end(1, q); }
      }
      if (early) (void)hipStreamWaitEvent(q, ev_fb[hh], 0);
      v.ncam_defer = 0;
      if (fuse) { v.Pout = spare; v.fuse_drop = (const int*)(sc_drop + cell0 + b0); }
      { StageRange r(fuse ? "msckf_marginalize+msckf_prune_empty_states" : "msckf_marginalize"); launch_update(v, b0, nb, q, early, 1); }
      if (fuse) { std::swap(curP, spare); pending = true; }
      else {
        StageRange r("msckf_prune_empty_states");
        stage_begin(6, q);
        launch_prune<S>(v, b0, nb, q, (const int*)(sc_drop + cell0 + b0), 0);
        stage_end(6, q);
        pending = false;
      }
      for (int b = b0; b < b0 + nb; ++b) {   // host mirror of the window size: augment, then drop n_drop (clamped as k_make_keep does)
        if (h_ncam[b] < n_cap) h_ncam[b]++;
        h_ncam[b] -= std::max(0, std::min(h_drop[cell0 + b], h_ncam[b]));
      }
    }
    // hipGetLastError is per host thread: a failed launch of this slice must not vanish with the thread
    const hipError_t e = hipGetLastError();
    slice_rc[hh] = (int)e;
  };
  if (nh == 1) enqueue(0);
  else {
    workers.start(nh - 1, [&](int idx) { enqueue(idx + 1); });
    enqueue(0);
    workers.wait();
  }
  rc = join_slices(nh, qs);
  if (rc) return poison(rc, "joining the slices' streams failed");
  for (int i = 0; i < nh; ++i)
    if (slice_rc[i]) return poison(-EIO, std::string("kernel launch failed on slice ") + std::to_string(i) + ": " + hipGetErrorString((hipError_t)slice_rc[i]));
  commit_buffer_parity(f0, f1);
  return 0;
}

// run_frames with the inputs handed over per frame, as the reference's callers do (asl_msckf.cpp:227-284: IMU samples and
// the image's tracks arrive with the image): frame f's block -- IMU samples + compact work-list, what the frame really
// holds, not a padded maximum -- goes from page-locked host memory into staging set f % ring on a copy stream, up to
// ring - 1 frames ahead of the kernels that consume it.  Hand-over (up_mode 0): the uploading thread waits for its copy on
// the HOST and publishes the frame number; a slice's enqueue thread launches frame f only after that, and the uploader
// reuses a set only after every slice's "consumed" event of frame f - ring has completed -- no stream ever waits for
// another stream's event on the device (those waits cost 0.15-0.25 ms per frame with two staging sets).  up_mode 1 keeps
// the device-side hipStreamWaitEvent protocol, for comparison.
template <class S>
int Batch<S>::run_frames_streamed(int f0, int f1) {
  POISON_GUARD();
  if (f0 < 0 || f1 > sc_frames || f0 > f1) return fail(-EINVAL, "frame range out of bounds");
  if (!committed) return fail(-EINVAL, "scenario not committed");
  DEVICE_ENTER();
  {
    bool all = sg_blk[0] != nullptr;
    for (int f = f0; f < f1 && all; ++f) all = pinf[f].p != nullptr;
    if (!all) { int rc = scen_pin(f0, f1); if (rc) return rc; }
  }
  if (!stc) {
    HIPCHK(hipStreamCreateWithFlags(&stc, hipStreamNonBlocking));
    for (int k = 0; k < RING_MAX; ++k) {
      HIPCHK(hipEventCreateWithFlags(&ev_up[k], hipEventDisableTiming));
      for (int i = 0; i < MAXS; ++i) HIPCHK(hipEventCreateWithFlags(&ev_use[k][i], hipEventDisableTiming));
    }
  }
  const int nh = prof ? 1 : std::max(1, std::min(nstreams, B));
  const int R = ring, mode = up_mode;
  hipStream_t qs[MAXS];
  int rc = slice_streams(nh, qs);
  if (rc) return rc;
  rc = fork_slices(nh, qs, stc);
  if (rc) return rc;
  // up_rdy = frames whose block may be read (mode 0: the copy has completed; mode 1: copy + event record are enqueued --
  // an event must be recorded before a wait on it is enqueued); use_enq[s] = frames whose "consumed" record is enqueued.
  std::atomic<int> up_rdy{f0};
  std::atomic<int> use_enq[MAXS];
  for (int i = 0; i < MAXS; ++i) use_enq[i].store(f0);
  std::atomic<int> failed{0};
  int slice_rc[MAXS] = {0};
  auto slice = [&](int hh) {
    (void)hipSetDevice(device);
    (void)hipGetLastError();
    hipStream_t q = qs[hh];
    const int b0 = (int)((long)B * hh / nh);
    const int nb = (int)((long)B * (hh + 1) / nh) - b0;
    S* curP = d.P; S* spare = P_spare;
    bool pending = false;
    for (int f = f0; f < f1; ++f) {
      while (up_rdy.load(std::memory_order_acquire) <= f && !failed.load()) std::this_thread::yield();
      if (failed.load()) break;
      const int k = (f - f0) % R;
      unsigned char* blk = sg_blk[k];
      if (mode == 1) (void)hipStreamWaitEvent(q, ev_up[k], 0);
      Dev<S> v = d;
      v.trk_n = reinterpret_cast<int*>(blk + pk_n) + b0; v.trk_M = reinterpret_cast<int*>(blk + pk_M) + (size_t)b0 * f_cap;
      v.trk_off = reinterpret_cast<int*>(blk + pk_off) + (size_t)b0 * f_cap;
      v.trk_slots = reinterpret_cast<int*>(blk + pk_slots); v.trk_obs = reinterpret_cast<S*>(blk + pinf[f].off_obs);
      v.wl_stride_n = 1; v.wl_stride_f = f_cap; v.wl_stride_o = 0;
      v.P = curP; v.ncam_defer = pending ? 1 : 0;
      const bool fuse = fuse_prune && !prof && !overlap_feature && d.joseph == 0 && f + 1 < f1;   // as in run_frames
      stage_begin(0, q); launch_propagate<S>(v, b0, nb, reinterpret_cast<S*>(blk + pk_rd) + (size_t)b0 * sc_K * RD_STRIDE, (long)sc_K * RD_STRIDE, sc_K, q, !prof); stage_end(0, q);
      if (prof) { stage_begin(1, q); launch_augment<S>(v, b0, nb, q); stage_end(1, q); }
      v.ncam_defer = 0;
      if (fuse) { v.Pout = spare; v.fuse_drop = (const int*)(reinterpret_cast<int*>(blk + pk_drop) + b0); }
      launch_update(v, b0, nb, q, false, 1);
      if (fuse) { std::swap(curP, spare); pending = true; }
      else {
        stage_begin(6, q);
        launch_prune<S>(v, b0, nb, q, (const int*)(reinterpret_cast<int*>(blk + pk_drop) + b0), 0);
        stage_end(6, q);
        pending = false;
      }
      (void)hipEventRecord(ev_use[k][hh], q);
      for (int b = b0; b < b0 + nb; ++b) {   // host mirror of the window size
        if (h_ncam[b] < n_cap) h_ncam[b]++;
        h_ncam[b] -= std::max(0, std::min(h_drop[(size_t)f * B + b], h_ncam[b]));
      }
      use_enq[hh].store(f + 1, std::memory_order_release);
    }
    slice_rc[hh] = (int)hipGetLastError();
  };
  // the uploading (calling) thread on its own core for the duration of the call, when a list of cores was given
  cpu_set_t old_mask; bool repin = false;
  if (!workers.cpus.empty() && workers.cpus[0] >= 0 && pthread_getaffinity_np(pthread_self(), sizeof(old_mask), &old_mask) == 0) {
    cpu_set_t one; CPU_ZERO(&one); CPU_SET(workers.cpus[0], &one);
    (void)pthread_setaffinity_np(pthread_self(), sizeof(one), &one); repin = true;
  }
  workers.start(nh, slice);
  int rc_up = 0;
  for (int f = f0; f < f1 && !rc_up; ++f) {
    const int k = (f - f0) % R;
    if (f - f0 >= R)
      for (int i = 0; i < nh && !rc_up; ++i) {   // the set is free again once every slice has consumed frame f - R
        while (use_enq[i].load(std::memory_order_acquire) <= f - R) std::this_thread::yield();
        const hipError_t e = mode == 0 ? hipEventSynchronize(ev_use[k][i]) : hipStreamWaitEvent(stc, ev_use[k][i], 0);
        if (e != hipSuccess) rc_up = -EIO;
      }
    if (!rc_up && (f == test_fail_upload || hipMemcpyAsync(sg_blk[k], pinf[f].p, pinf[f].bytes, hipMemcpyHostToDevice, stc) != hipSuccess)) rc_up = -EIO;
    if (!rc_up && (mode == 0 ? hipStreamSynchronize(stc) : hipEventRecord(ev_up[k], stc)) != hipSuccess) rc_up = -EIO;
    if (rc_up) failed.store(1);
    up_rdy.store(f + 1, std::memory_order_release);
  }
  if (rc_up) failed.store(1);
  workers.wait();
  if (repin) (void)pthread_setaffinity_np(pthread_self(), sizeof(old_mask), &old_mask);
  if (rc_up) return poison(rc_up, "input upload failed");
  rc = join_slices(nh, qs);
  if (rc) return poison(rc, "joining the slices' streams failed");
  if (hipEventRecord(ev_join[1], stc) != hipSuccess || hipStreamWaitEvent(st, ev_join[1], 0) != hipSuccess) return poison(-EIO, "joining the copy stream failed");
  for (int i = 0; i < nh; ++i)
    if (slice_rc[i]) return poison(-EIO, std::string("kernel launch failed on slice ") + std::to_string(i) + ": " + hipGetErrorString((hipError_t)slice_rc[i]));
  commit_buffer_parity(f0, f1);
  return 0;
}

// -------------------------------------------------------------------------------------------------
// host bookkeeping shared by both dtypes (restates msckf.h:215-332, 685-717, 765-807, 1469-1485)
// -------------------------------------------------------------------------------------------------
void remove_tracked_feature(HostTraj& t, uint64_t fid, std::vector<int>& slots) {
  slots.clear();
  for (size_t c = 0; c < t.cams.size(); ++c) {
    auto& ids = t.cams[c].tracked;
    auto it = std::find(ids.begin(), ids.end(), fid);
    if (it != ids.end()) { ids.erase(it); slots.push_back((int)c); }
  }
}

// key -> int table without a heap node per key (open addressing, power-of-two capacity, one instance per host thread reused
// from call to call): the bookkeeping below makes a few hundred to a few thousand look-ups per image and trajectory, and a
// std::unordered_map's allocations were most of their cost
struct FlatIndex {
  std::vector<uint64_t> key; std::vector<int> val; unsigned shift = 64; size_t mask = 0;
  void reset(size_t n) {
    size_t cap = 16; unsigned lg = 4;
    while (cap < 2 * n + 2) { cap <<= 1; ++lg; }
    if (key.size() < cap) key.resize(cap);
    val.assign(cap, -1);
    mask = cap - 1; shift = 64 - lg;
  }
  size_t slot(uint64_t k) const { return (size_t)((k * 0x9E3779B97F4A7C15ull) >> shift) & mask; }
  // keeps the FIRST value given for a key (std::find returns the first occurrence); returns the value held
  int insert_first(uint64_t k, int v) {
    for (size_t s = slot(k);; s = (s + 1) & mask) {
      if (val[s] < 0) { key[s] = k; val[s] = v; return v; }
      if (key[s] == k) return val[s];
    }
  }
  int find(uint64_t k) const {
    for (size_t s = slot(k);; s = (s + 1) & mask) {
      if (val[s] < 0) return -1;
      if (key[s] == k) return val[s];
    }
  }
};

// update(), msckf.h:215-300, with the reference's results and none of its quadratic searches.  The reference looks every tracked
// feature up in the incoming ids by linear search (:226), removes an ended feature from every camera state's list by linear
// search + erase (removeTrackedFeature :1469-1485) and erases the ended tracks one by one (:283-298): O(tracked x incoming) +
// O(ended x cameras x tracked) per image -- 43 us of the single filter's 286 us at 50 features per image (round-5 verdict), tens of
// milliseconds per filter at the benchmark's 200.  Here: one table of the incoming ids (first occurrence, as std::find
// returns), one table "ended feature -> camera slots that list it" built from the lists as they stand after this image's
// registrations, ONE stable filter pass per camera list and per track list.  Every list ends in the order the reference leaves it.
int host_update(BatchBase* B, int b, const double* meas, const uint64_t* ids, int n) {
  HostTraj& t = B->traj[b];
  if (!t.initialized) return fail(-EINVAL, "trajectory not initialized");
  if (t.cams.empty()) return fail(-EINVAL, "update() before augmentState() (msckf.h:238 dereferences cam_states_.end()-1)");
  t.to_resid.clear();
  static thread_local FlatIndex first, where;
  first.reset((size_t)n);
  for (int k = 0; k < n; ++k) first.insert_first(ids[k], k);                  // keeps the first occurrence (std::find, :226)
  // pass 1 (:224-247): register this image's observation; which tracks end here
  const size_t nt = t.tracked_ids.size();
  static thread_local std::vector<char> ended;
  ended.assign(nt, 0);
  size_t n_ended = 0;
  for (size_t i = 0; i < nt; ++i) {
    const uint64_t fid = t.tracked_ids[i];
    Track& tr = t.tracks[i];
    const int k = first.find(fid);
    const bool valid = k >= 0;
    if (valid) {
      tr.obs.push_back(meas[2 * (size_t)k]); tr.obs.push_back(meas[2 * (size_t)k + 1]);
      t.cams.back().tracked.push_back(fid);
      tr.cam_ids.push_back(t.cams.back().state_id);
    }
    if (!valid || tr.obs.size() / 2 >= (size_t)t.max_track_length) { ended[i] = 1; ++n_ended; }
  }
  if (!n_ended) return 0;
  // pass 2 (:249-265 + removeTrackedFeature): the camera slots that list an ended feature, in camera order
  where.reset(n_ended);
  std::vector<std::vector<int>> slots_of(n_ended);
  {
    int e = 0;
    for (size_t i = 0; i < nt; ++i) if (ended[i]) where.insert_first(t.tracked_ids[i], e++);
  }
  for (size_t c = 0; c < t.cams.size(); ++c) {
    auto& lst = t.cams[c].tracked;
    size_t w = 0;
    for (size_t r = 0; r < lst.size(); ++r) {
      const int e = where.find(lst[r]);
      if (e >= 0 && (slots_of[e].empty() || slots_of[e].back() != (int)c)) { slots_of[e].push_back((int)c); continue; }   // first occurrence in this list leaves it
      lst[w++] = lst[r];
    }
    lst.resize(w);
  }
  {
    int e = 0;
    for (size_t i = 0; i < nt; ++i) {
      if (!ended[i]) continue;
      Track& tr = t.tracks[i];
      std::vector<int>& slots = slots_of[e++];
      if (slots.size() >= (size_t)t.min_track_length) {
        TrackToResid r;
        r.id = tr.id; r.obs = std::move(tr.obs); r.slots = std::move(slots);   // (the track is erased below: its observations move, they are not copied)
        t.to_resid.push_back(std::move(r));
      }
    }
  }
  // pass 3 (:283-298): last_correlated_id of the camera states an ended track leaves empty, then the tracks themselves
  for (size_t i = 0; i < nt; ++i) {
    if (!ended[i] || t.tracks[i].cam_ids.empty()) continue;
    const int last_id = t.tracks[i].cam_ids.back();
    for (int idx : t.tracks[i].cam_ids)
      for (auto& cs : t.cams)
        if (cs.state_id == idx) { if (cs.tracked.empty()) cs.last_correlated_id = last_id; break; }
  }
  {
    size_t w = 0;
    for (size_t i = 0; i < nt; ++i) {
      if (ended[i]) continue;
      if (w != i) { t.tracks[w] = std::move(t.tracks[i]); t.tracked_ids[w] = t.tracked_ids[i]; }
      ++w;
    }
    t.tracks.resize(w); t.tracked_ids.resize(w);
  }
  return 0;
}

int host_add_features(BatchBase* B, int b, const double* meas, const uint64_t* ids, int n) {
  HostTraj& t = B->traj[b];
  if (!t.initialized) return fail(-EINVAL, "trajectory not initialized");
  if (t.cams.empty()) return fail(-EINVAL, "addFeatures() before augmentState() (msckf.h:320)");
  static thread_local FlatIndex known;
  known.reset(t.tracked_ids.size() + (size_t)n);
  for (size_t i = 0; i < t.tracked_ids.size(); ++i) known.insert_first(t.tracked_ids[i], (int)i);
  for (int i = 0; i < n; ++i) {
    if (known.find(ids[i]) >= 0)
      return fail(-EEXIST, "added new feature that was already being tracked");   // msckf.h:328-329 prints and returns
    known.insert_first(ids[i], (int)t.tracked_ids.size());
    Track tr; tr.id = ids[i];
    tr.obs.push_back(meas[2 * i]); tr.obs.push_back(meas[2 * i + 1]);
    t.cams.back().tracked.push_back(ids[i]);
    tr.cam_ids.push_back(t.cams.back().state_id);
    t.tracks.push_back(std::move(tr));
    t.tracked_ids.push_back(ids[i]);
  }
  return 0;
}

int resolve_map(BatchBase* B, int b);
int host_marginalize(BatchBase* B, int b) {
  HostTraj& t = B->traj[b];
  t.map.clear(); t.map_pending = 0;
  if (t.to_resid.empty()) { int z = 0; int rc = B->set_tracks(b, 0, &z, &z, nullptr); return rc ? rc : B->clear_stats(b); }
  const int F = (int)t.to_resid.size();
  std::vector<int> M(F), slots; std::vector<double> obs;
  for (int i = 0; i < F; ++i) {
    M[i] = (int)t.to_resid[i].slots.size();
    slots.insert(slots.end(), t.to_resid[i].slots.begin(), t.to_resid[i].slots.end());
    obs.insert(obs.end(), t.to_resid[i].obs.begin(), t.to_resid[i].obs.end());
  }
  int rc = B->set_tracks(b, F, M.data(), slots.data(), obs.data());
  if (rc) return rc;
  rc = B->marginalize(b, 1);
  if (rc) return rc;
  t.map_pending = F;      // map_ (msckf.h:371) is read back when it is asked for: no wait for the device inside marginalize()
  return 0;
}

// the triangulated points of the last marginalize() (msckf.h:371: map_.push_back(p_f_G)), fetched on demand
int resolve_map(BatchBase* B, int b) {
  HostTraj& t = B->traj[b];
  const int F = t.map_pending;
  if (F <= 0) return 0;
  t.map_pending = 0;
  std::vector<double> info((size_t)F * 8);
  const int rc = B->track_info(b, info.data(), F);
  if (rc < 0) return rc;
  for (int i = 0; i < F; ++i)
    if (info[8 * i] != 0 && info[8 * i + 1] != 0) { t.map.push_back(info[8 * i + 5]); t.map.push_back(info[8 * i + 6]); t.map.push_back(info[8 * i + 7]); }
  return 0;
}

int host_prune_empty(BatchBase* B, int b) {
  HostTraj& t = B->traj[b];
  const int max_states = t.max_cam_states, num = (int)t.cams.size();
  if (num < max_states) return 0;
  if (!t.cams.front().tracked.empty()) return 0;
  int last_to_remove = num - max_states - 1;
  for (int i = 1; i < num - max_states; i++)
    if (!t.cams[i].tracked.empty()) { last_to_remove = i - 1; break; }
  if (last_to_remove < 0) return 0;
  std::vector<int> keep;
  // pruned_states_ keeps the whole camState (msckf.h:714; read by asl_msckf.cpp:409-424): poses come back once, here
  std::vector<double> poses((size_t)num * 7);
  int rc = B->get_cams_known(b, poses.data(), num);
  if (rc) return rc;
  for (int i = 0; i <= last_to_remove; ++i) {
    PrunedState ps{t.cams[i].state_id, t.cams[i].time, t.cams[i].last_correlated_id, {0}};
    std::copy(&poses[7 * (size_t)i], &poses[7 * (size_t)i] + 7, ps.pose);
    t.pruned.push_back(ps);
  }
  for (int i = last_to_remove + 1; i < num; ++i) keep.push_back(i);
  rc = B->prune_keep(b, keep);
  if (rc) return rc;
  t.cams.erase(t.cams.begin(), t.cams.begin() + last_to_remove + 1);
  return 0;
}

// findRedundantCamStates, msckf.h:1049-1098 (poses: n x 7 = q_CG(w,x,y,z) p_C_G)
static void find_redundant(const HostTraj& t, const std::vector<double>& poses, std::vector<int>& rm) {
  const int n = (int)t.cams.size();
  if (n < 5) return;
  auto qp = [&](int i) { return &poses[7 * i]; };
  int kf = 0;
  const int prot = n - 3;
  int next = 1;
  while (next != prot) {
    const double* a = qp(kf); const double* c = qp(next);
    const double dx = c[4] - a[4], dy = c[5] - a[5], dz = c[6] - a[6];
    const double distance = std::sqrt(dx * dx + dy * dy + dz * dz);
    // Eigen angularDistance: d = kf_q * conj(cam_q); 2*atan2(|vec(d)|, |d.w|)
    const double aw = a[0], ax = a[1], ay = a[2], az = a[3], bw = c[0], bx = -c[1], by = -c[2], bz = -c[3];
    const double dw = aw * bw - ax * bx - ay * by - az * bz;
    const double vx = aw * bx + ax * bw + ay * bz - az * by, vy = aw * by + ay * bw + az * bx - ax * bz, vz = aw * bz + az * bw + ax * by - ay * bx;
    const double angle = 2 * std::atan2(std::sqrt(vx * vx + vy * vy + vz * vz), std::fabs(dw));
    if (distance < t.redundancy_distance_thresh && angle < t.redundancy_angle_thresh) rm.push_back(t.cams[next].state_id);
    else kf = next;
    ++next;
    if (n - (int)rm.size() <= t.max_cam_states) break;
  }
  const int over = (n - (int)rm.size()) - t.max_cam_states;
  for (int i = 0; i < over; i++)
    if (std::find(rm.begin(), rm.end(), t.cams[i].state_id) == rm.end()) rm.push_back(t.cams[i].state_id);
  if (rm.size() < 2) rm.clear();
  std::sort(rm.begin(), rm.end());
}

static void erase_involved(Track& tr, const std::vector<int>& involved) {
  for (int cam_id : involved) {
    auto it = std::find(tr.cam_ids.begin(), tr.cam_ids.end(), cam_id);
    if (it != tr.cam_ids.end()) {
      const size_t idx = (size_t)(it - tr.cam_ids.begin());
      tr.cam_ids.erase(it);
      tr.obs.erase(tr.obs.begin() + 2 * idx, tr.obs.begin() + 2 * idx + 2);
    }
  }
}

// MSCKF::pruneRedundantStates, msckf.h:453-682: keyframe selection and observation surgery on the host, the
// triangulation of not-yet-initialized features and the second measurement update on the device.
int host_prune_redundant(BatchBase* B, int b) {
  HostTraj& t = B->traj[b];
  if (!t.initialized) return fail(-EINVAL, "trajectory not initialized");
  if (t.cams.size() < 20) return 0;                                           // :455
  const int n = (int)t.cams.size();
  std::vector<double> poses((size_t)n * 7);
  int rc = resolve_map(B, b);          // this call appends to map_ (:528) and reuses the device work-list
  if (rc) return rc;
  rc = B->get_cams_known(b, poses.data(), n);
  if (rc) return rc;
  std::vector<int> rm;
  find_redundant(t, poses, rm);
  auto involved_of = [&](const Track& tr) {
    std::vector<int> inv;
    for (int cam_id : rm) if (std::find(tr.cam_ids.begin(), tr.cam_ids.end(), cam_id) != tr.cam_ids.end()) inv.push_back(cam_id);
    return inv;
  };
  auto slot_of = [&](int cam_id) { for (int i = 0; i < n; ++i) if (t.cams[i].state_id == cam_id) return i; return -1; };
  // ---- first loop :466-534
  std::vector<size_t> cand;   // not-yet-initialized features with >= 2 involved states: need motion check + triangulation
  for (size_t i = 0; i < t.tracks.size(); ++i) {
    Track& tr = t.tracks[i];
    std::vector<int> inv = involved_of(tr);
    if (inv.empty()) continue;
    if (inv.size() == 1) { erase_involved(tr, inv); continue; }
    if (!tr.initialized) cand.push_back(i);
  }
  if (!cand.empty()) {
    if ((int)cand.size() > B->f_cap) return fail(-E2BIG, "more candidate features than f_cap");
    std::vector<int> M, slots; std::vector<double> obs;
    for (size_t ci : cand) {
      const Track& tr = t.tracks[ci];
      int m = 0;
      for (int p = 0; p < n; ++p) {                                           // feature_associated_cam_states in cam order (:490-495)
        auto it = std::find(tr.cam_ids.begin(), tr.cam_ids.end(), t.cams[p].state_id);
        if (it == tr.cam_ids.end()) continue;
        const size_t k = (size_t)(it - tr.cam_ids.begin());
        slots.push_back(p); obs.push_back(tr.obs[2 * k]); obs.push_back(tr.obs[2 * k + 1]); ++m;
      }
      M.push_back(m);
    }
    rc = B->set_tracks(b, (int)cand.size(), M.data(), slots.data(), obs.data());
    if (rc) return rc;
    std::vector<int> status(cand.size()); std::vector<double> pf(3 * cand.size());
    rc = B->feature_only(b, status.data(), pf.data(), (int)cand.size());
    if (rc < 0) return rc;
    for (size_t c = 0; c < cand.size(); ++c) {
      Track& tr = t.tracks[cand[c]];
      const bool ok = (status[c] & ST_MOTION_OK) && (status[c] & ST_TRI_VALID);
      if (!ok) erase_involved(tr, involved_of(tr));                           // :496-524
      else { tr.initialized = true; for (int k = 0; k < 3; ++k) tr.p_f_G[k] = pf[3 * c + k]; t.map.insert(t.map.end(), &pf[3 * c], &pf[3 * c] + 3); }
    }
  }
  // ---- second loop :545-607
  {
    std::vector<int> M, slots; std::vector<double> obs, pf;
    std::vector<size_t> used;
    for (size_t i = 0; i < t.tracks.size(); ++i) {
      Track& tr = t.tracks[i];
      std::vector<int> inv = involved_of(tr);
      if (inv.empty()) continue;
      for (int cam_id : inv) {
        const size_t k = (size_t)(std::find(tr.cam_ids.begin(), tr.cam_ids.end(), cam_id) - tr.cam_ids.begin());
        slots.push_back(slot_of(cam_id)); obs.push_back(tr.obs[2 * k]); obs.push_back(tr.obs[2 * k + 1]);
      }
      M.push_back((int)inv.size());
      for (int k = 0; k < 3; ++k) pf.push_back(tr.p_f_G[k]);
      used.push_back(i);
    }
    const int F = (int)M.size();
    if (F > B->f_cap) return fail(-E2BIG, "more features than f_cap");
    if (F > 0) {
      rc = B->set_tracks(b, F, M.data(), slots.data(), obs.data());
      if (rc) return rc;
      rc = B->set_given_positions(b, F, pf.data());
      if (rc) return rc;
      rc = B->marginalize_given(b);
      if (rc) return rc;
    }
    for (size_t i : used) erase_involved(t.tracks[i], involved_of(t.tracks[i]));
  }
  // ---- prune the removed camera states :616-681
  std::vector<int> keep;
  std::vector<CamMeta> kept;
  if (!rm.empty()) {                                  // poses as corrected by the second update (msckf.h:614 precedes :631)
    rc = B->get_cams_known(b, poses.data(), n);
    if (rc) return rc;
  }
  for (int i = 0; i < n; ++i) {
    if (std::find(rm.begin(), rm.end(), t.cams[i].state_id) != rm.end()) {
      PrunedState ps{t.cams[i].state_id, t.cams[i].time, t.cams[i].last_correlated_id, {0}};
      std::copy(&poses[7 * (size_t)i], &poses[7 * (size_t)i] + 7, ps.pose);
      t.pruned.push_back(ps);
    }
    else { keep.push_back(i); kept.push_back(t.cams[i]); }
  }
  if ((int)keep.size() != n) {
    rc = B->prune_keep(b, keep);
    if (rc) return rc;
    t.cams = kept;
  }
  return 0;
}

int host_finish(BatchBase* B, int b) {
  HostTraj& t = B->traj[b];
  // D6: the reference appends to the stale feature_tracks_to_residualize_ of the previous update() (cleared only
  // at msckf.h:218), whose positional indices and pose copies are invalid once states were corrected/pruned;
  // the stale list is dropped here (oracle/msckf_oracle.hpp does the same).
  t.to_resid.clear();
  for (size_t i = 0; i < t.tracked_ids.size(); i++) {
    TrackToResid r;
    remove_tracked_feature(t, t.tracked_ids[i], r.slots);
    if (r.slots.size() >= (size_t)t.min_track_length) {
      for (auto& tr : t.tracks) if (tr.id == t.tracked_ids[i]) { r.id = tr.id; r.obs = tr.obs; break; }
      t.to_resid.push_back(r);
    }
  }
  return host_marginalize(B, b);
}


// -------------------------------------------------------------------------------------------------
// One image of the ASL runner's loop (asl_msckf.cpp:269-294) for trajectories b0 .. b0 + nb - 1 of a batch IN LOCKSTEP:
//   augmentState -> update -> addFeatures -> marginalize -> [pruneRedundantStates] -> [pruneEmptyStates]
// The bookkeeping of every trajectory (msckf.h:215-332, 453-534, 685-717, 1049-1098) runs on the host exactly as in the
// per-filter entries above -- same functions --, the device work of a stage goes out as ONE launch sequence over the range
// (the kernels index trajectories; a run of trajectories with nothing to do is skipped) and what a stage needs back -- poses for
// findRedundantCamStates, triangulated points of not-yet-initialized features, poses of the states about to be pruned --
// comes back in one read and one wait per stage for the whole range, instead of one per filter.
// Same arithmetic per trajectory as the per-filter calls (tests/test_gpu_parity.py: bit for bit).
// -------------------------------------------------------------------------------------------------
// the bookkeeping of the trajectories of a range is independent: spread over host threads (created per call: tens of microseconds
// against milliseconds of list surgery at the benchmark's 200 tracks per image); fn(i) returns 0 or an error code
template <class Fn> static int parallel_for(int n, Fn fn) {
  unsigned hw = std::thread::hardware_concurrency();
  int nt = (int)std::min<unsigned>(hw ? hw : 1u, 32u);
  if (const char* e = getenv("MSCKF_HIP_HOST_THREADS")) nt = std::max(1, atoi(e));
  nt = std::min(nt, n);
  if (nt <= 1) { for (int i = 0; i < n; ++i) { const int rc = fn(i); if (rc) return rc; } return 0; }
  std::atomic<int> next(0), err(0);
  auto work = [&]() { for (;;) { const int i = next.fetch_add(1); if (i >= n || err.load()) return; const int rc = fn(i); if (rc) err.store(rc); } };
  std::vector<std::thread> th;
  for (int k = 1; k < nt; ++k) th.emplace_back(work);
  work();
  for (auto& x : th) x.join();
  return err.load();
}
template <class Fn> static int for_runs(const std::vector<char>& on, int b0, Fn fn) {
  const int nb = (int)on.size();
  for (int i = 0; i < nb;) {
    if (!on[i]) { ++i; continue; }
    int j = i;
    while (j < nb && on[j]) ++j;
    const int rc = fn(b0 + i, j - i);
    if (rc) return rc;
    i = j;
  }
  return 0;
}

int host_image_cycle(BatchBase* B, int b0, int nb, const int* state_ids, const double* times,
                     const double* upd_meas, const uint64_t* upd_ids, const int* upd_n,
                     const double* new_meas, const uint64_t* new_ids, const int* new_n, int flags) {
  if (b0 < 0 || nb <= 0 || b0 + nb > B->B) return fail(-EINVAL, "trajectory range out of bounds");
  const int n_cap = B->n_cap, f_cap = B->f_cap;
  for (int i = 0; i < nb; ++i) {
    const HostTraj& t = B->traj[b0 + i];
    if (!t.initialized) return fail(-EINVAL, "trajectory not initialized");
    if ((int)t.cams.size() >= n_cap) return fail(-EOVERFLOW, "camera-state capacity n_cap exceeded");
  }
  static const bool tim = getenv("MSCKF_HIP_CYCLE_TIMERS") != nullptr;
  double tph[8] = {0}; auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double tl = now();
  auto tick = [&](int k) { if (tim) { const double t2 = now(); tph[k] += t2 - tl; tl = t2; } };
  // ---- augmentState :148-212
  int rc = B->augment(b0, nb);
  if (rc) return rc;
  for (int i = 0; i < nb; ++i) {
    HostTraj& t = B->traj[b0 + i];
    t.cams.push_back(CamMeta{state_ids[i], times ? times[i] : 0.0, -1, {}});
    t.map.clear(); t.map_pending = 0;
  }
  tick(0);
  // ---- update :215-300, addFeatures :302-332 (host)
  {
    std::vector<size_t> ou(nb + 1, 0), on(nb + 1, 0);
    for (int i = 0; i < nb; ++i) { ou[i + 1] = ou[i] + (size_t)upd_n[i]; on[i + 1] = on[i] + (size_t)new_n[i]; }
    rc = parallel_for(nb, [&](int i) {
      int r2 = host_update(B, b0 + i, upd_meas + 2 * ou[i], upd_ids + ou[i], upd_n[i]);
      if (!r2) r2 = host_add_features(B, b0 + i, new_meas + 2 * on[i], new_ids + on[i], new_n[i]);
      return r2;
    });
    if (rc) return rc;
  }
  tick(1);
  // ---- marginalize :336-449
  std::vector<char> has(nb, 0);
  {
    std::vector<BatchBase::WorkList> wl(nb);
    parallel_for(nb, [&](int i) {
      const HostTraj& t = B->traj[b0 + i];
      const int F = (int)t.to_resid.size();
      wl[i].M.assign(F, 0);
      for (int k = 0; k < F; ++k) {
        wl[i].M[k] = (int)t.to_resid[k].slots.size();
        wl[i].slots.insert(wl[i].slots.end(), t.to_resid[k].slots.begin(), t.to_resid[k].slots.end());
        wl[i].obs.insert(wl[i].obs.end(), t.to_resid[k].obs.begin(), t.to_resid[k].obs.end());
      }
      return 0;
    });
    tick(2);
    rc = B->set_tracks_range(b0, nb, wl);
    if (rc) return rc;
    for (int i = 0; i < nb; ++i) {
      const int F = (int)wl[i].M.size();
      if (!F) { rc = B->clear_stats(b0 + i); if (rc) return rc; }
      has[i] = F > 0;
      B->traj[b0 + i].map_pending = F;
    }
    tick(3);
    rc = for_runs(has, b0, [&](int s0, int n) { return B->marginalize(s0, n); });
    if (rc) return rc;
  }
  tick(4);
  // ---- pruneRedundantStates :453-682
  if (flags & 1) {
    std::vector<char> act(nb, 0);
    bool any = false;
    for (int i = 0; i < nb; ++i) { act[i] = B->traj[b0 + i].cams.size() >= 20; any |= act[i] != 0; }   // :455
    if (any) {
      std::vector<int> status((size_t)nb * f_cap);
      std::vector<double> pf((size_t)nb * f_cap * 3), poses((size_t)nb * n_cap * 7);
      // what resolve_map() fetches per filter: the points of the marginalize just launched (the work-lists are reused below)
      rc = B->feature_only_range(b0, nb, status.data(), pf.data(), false);
      if (rc) return rc;
      rc = B->cams_range(b0, nb, poses.data());
      if (rc) return rc;
      std::vector<std::vector<int>> rm(nb);
      std::vector<std::vector<size_t>> cand(nb);
      std::vector<char> has_cand(nb, 0), has_upd(nb, 0);
      auto involved_of = [&](int i, const Track& tr) {
        std::vector<int> inv;
        for (int cam_id : rm[i]) if (std::find(tr.cam_ids.begin(), tr.cam_ids.end(), cam_id) != tr.cam_ids.end()) inv.push_back(cam_id);
        return inv;
      };
      for (int i = 0; i < nb; ++i) {
        if (!act[i]) continue;
        HostTraj& t = B->traj[b0 + i];
        const int F = t.map_pending; t.map_pending = 0;
        for (int k = 0; k < F; ++k) {
          const int sx = status[(size_t)i * f_cap + k];
          if (((sx & ST_MOTION_SKIPPED) || (sx & ST_MOTION_OK)) && (sx & ST_TRI_VALID)) t.map.insert(t.map.end(), &pf[((size_t)i * f_cap + k) * 3], &pf[((size_t)i * f_cap + k) * 3] + 3);
        }
        const int n = (int)t.cams.size();
        std::vector<double> pz(&poses[(size_t)i * n_cap * 7], &poses[(size_t)i * n_cap * 7] + (size_t)n * 7);
        find_redundant(t, pz, rm[i]);
        if (rm[i].empty()) continue;                   // no camera state to remove: both loops over the tracks find nothing involved
        // first loop :466-534
        for (size_t k = 0; k < t.tracks.size(); ++k) {
          Track& tr = t.tracks[k];
          std::vector<int> inv = involved_of(i, tr);
          if (inv.empty()) continue;
          if (inv.size() == 1) { erase_involved(tr, inv); continue; }
          if (!tr.initialized) cand[i].push_back(k);
        }
        if ((int)cand[i].size() > f_cap) return fail(-E2BIG, "more candidate features than f_cap");
        if (!cand[i].empty()) {
          std::vector<int> M, slots; std::vector<double> obs;
          for (size_t ci : cand[i]) {
            const Track& tr = t.tracks[ci];
            int m = 0;
            for (int p2 = 0; p2 < n; ++p2) {
              auto it = std::find(tr.cam_ids.begin(), tr.cam_ids.end(), t.cams[p2].state_id);
              if (it == tr.cam_ids.end()) continue;
              const size_t k = (size_t)(it - tr.cam_ids.begin());
              slots.push_back(p2); obs.push_back(tr.obs[2 * k]); obs.push_back(tr.obs[2 * k + 1]); ++m;
            }
            M.push_back(m);
          }
          rc = B->set_tracks(b0 + i, (int)cand[i].size(), M.data(), slots.data(), obs.data());
          if (rc) return rc;
          has_cand[i] = 1;
        }
      }
      // checkMotion + initializePosition of the not-yet-initialized features, every trajectory's in one launch per run
      {
        bool anyc = false;
        for (int i = 0; i < nb; ++i) anyc |= has_cand[i] != 0;
        if (anyc) {
          // (feature_only_range launches over the runs; the read-back covers the whole range once)
          rc = for_runs(has_cand, b0, [&](int s0, int n) { return B->feature_only_range(s0, n, status.data() + (size_t)(s0 - b0) * f_cap, pf.data() + (size_t)(s0 - b0) * f_cap * 3, true); });
          if (rc) return rc;
          for (int i = 0; i < nb; ++i) {
            if (!has_cand[i]) continue;
            HostTraj& t = B->traj[b0 + i];
            for (size_t c = 0; c < cand[i].size(); ++c) {
              Track& tr = t.tracks[cand[i][c]];
              const int sx = status[(size_t)i * f_cap + c];
              const double* pc = &pf[((size_t)i * f_cap + c) * 3];
              const bool ok = (sx & ST_MOTION_OK) && (sx & ST_TRI_VALID);
              if (!ok) erase_involved(tr, involved_of(i, tr));                      // :496-524
              else { tr.initialized = true; for (int k = 0; k < 3; ++k) tr.p_f_G[k] = pc[k]; t.map.insert(t.map.end(), pc, pc + 3); }
            }
          }
        }
      }
      // second loop :545-607: the work-lists of the second update
      std::vector<std::vector<size_t>> used(nb);
      std::vector<double> pfin((size_t)nb * f_cap * 3, 0.0);
      for (int i = 0; i < nb; ++i) {
        if (!act[i] || rm[i].empty()) continue;
        HostTraj& t = B->traj[b0 + i];
        const int n = (int)t.cams.size();
        auto slot_of = [&](int cam_id) { for (int q = 0; q < n; ++q) if (t.cams[q].state_id == cam_id) return q; return -1; };
        std::vector<int> M, slots; std::vector<double> obs;
        for (size_t k = 0; k < t.tracks.size(); ++k) {
          Track& tr = t.tracks[k];
          std::vector<int> inv = involved_of(i, tr);
          if (inv.empty()) continue;
          for (int cam_id : inv) {
            const size_t q = (size_t)(std::find(tr.cam_ids.begin(), tr.cam_ids.end(), cam_id) - tr.cam_ids.begin());
            slots.push_back(slot_of(cam_id)); obs.push_back(tr.obs[2 * q]); obs.push_back(tr.obs[2 * q + 1]);
          }
          for (int q = 0; q < 3; ++q) pfin[((size_t)i * f_cap + M.size()) * 3 + q] = tr.p_f_G[q];
          M.push_back((int)inv.size());
          used[i].push_back(k);
          if ((int)M.size() > f_cap) return fail(-E2BIG, "more features than f_cap");
        }
        if (!M.empty()) {
          rc = B->set_tracks(b0 + i, (int)M.size(), M.data(), slots.data(), obs.data());
          if (rc) return rc;
          has_upd[i] = 1;
        }
      }
      {
        bool anyu = false;
        for (int i = 0; i < nb; ++i) anyu |= has_upd[i] != 0;
        if (anyu) {
          rc = B->set_given_range(b0, nb, pfin.data());
          if (rc) return rc;
          rc = for_runs(has_upd, b0, [&](int s0, int n) { return B->marginalize_given_range(s0, n); });
          if (rc) return rc;
        }
      }
      for (int i = 0; i < nb; ++i) {
        HostTraj& t = B->traj[b0 + i];
        for (size_t k : used[i]) erase_involved(t.tracks[k], involved_of(i, t.tracks[k]));
      }
      // prune the removed camera states :616-681 (poses as corrected by the second update: :614 precedes :631)
      bool anyrm = false;
      for (int i = 0; i < nb; ++i) anyrm |= !rm[i].empty();
      if (anyrm) {
        rc = B->cams_range(b0, nb, poses.data());
        if (rc) return rc;
        std::vector<std::vector<int>> keep(nb);
        for (int i = 0; i < nb; ++i) {
          HostTraj& t = B->traj[b0 + i];
          const int n = (int)t.cams.size();
          std::vector<CamMeta> kept;
          for (int q = 0; q < n; ++q) {
            if (!rm[i].empty() && std::find(rm[i].begin(), rm[i].end(), t.cams[q].state_id) != rm[i].end()) {
              PrunedState ps{t.cams[q].state_id, t.cams[q].time, t.cams[q].last_correlated_id, {0}};
              std::copy(&poses[((size_t)i * n_cap + q) * 7], &poses[((size_t)i * n_cap + q) * 7] + 7, ps.pose);
              t.pruned.push_back(ps);
            } else { keep[i].push_back(q); kept.push_back(t.cams[q]); }
          }
          if ((int)keep[i].size() != n) t.cams = kept;
        }
        rc = B->prune_keep_range(b0, nb, keep);
        if (rc) return rc;
      }
    }
  }
  tick(5);
  // ---- pruneEmptyStates :685-761
  if (flags & 2) {
    std::vector<int> last(nb, -1);
    bool any = false;
    for (int i = 0; i < nb; ++i) {
      const HostTraj& t = B->traj[b0 + i];
      const int max_states = t.max_cam_states, num = (int)t.cams.size();
      if (num < max_states || !t.cams.front().tracked.empty()) continue;
      int last_to_remove = num - max_states - 1;
      for (int q = 1; q < num - max_states; q++)
        if (!t.cams[q].tracked.empty()) { last_to_remove = q - 1; break; }
      last[i] = last_to_remove;
      any |= last_to_remove >= 0;
    }
    if (any) {
      std::vector<double> poses((size_t)nb * n_cap * 7);
      rc = B->cams_range(b0, nb, poses.data());    // pruned_states_ keeps the whole camState (msckf.h:714)
      if (rc) return rc;
      std::vector<std::vector<int>> keep(nb);
      for (int i = 0; i < nb; ++i) {
        HostTraj& t = B->traj[b0 + i];
        const int num = (int)t.cams.size();
        for (int q = 0; q <= last[i]; ++q) {
          PrunedState ps{t.cams[q].state_id, t.cams[q].time, t.cams[q].last_correlated_id, {0}};
          std::copy(&poses[((size_t)i * n_cap + q) * 7], &poses[((size_t)i * n_cap + q) * 7] + 7, ps.pose);
          t.pruned.push_back(ps);
        }
        for (int q = last[i] + 1; q < num; ++q) keep[i].push_back(q);
        if (last[i] >= 0) t.cams.erase(t.cams.begin(), t.cams.begin() + last[i] + 1);
      }
      rc = B->prune_keep_range(b0, nb, keep);
      if (rc) return rc;
    }
  }
  tick(6);
  if (tim) fprintf(stderr, "cycle nb=%d ms: augment %.2f update+add %.2f lists %.2f upload %.2f launch %.2f redundant %.2f empty %.2f\n", nb, tph[0], tph[1], tph[2], tph[3], tph[4], tph[5], tph[6]);
  return 0;
}

BatchBase* H(msckf_hip_handle h) { return reinterpret_cast<BatchBase*>(h); }
}  // namespace

#ifdef MSCKF_ABLATE
namespace msckf { void qr_debug_set(int idx, int val); void feat_debug_set(int val); void featp_cycles_read(unsigned long long* out8, int reset); extern int g_gram_dbg; void chol_cycles_read(unsigned long long* out16, int reset); void chol_sub_read(unsigned long long* out32, int reset); void chol_debug_set(int v); void prop_cycles_read(unsigned long long* out8, int reset); void gram_cycles_read(unsigned long long* out40, int reset); void gemm_cycles_read(unsigned long long* out16, int reset); void gemm_trace_read(unsigned long long* out); void us_cycles_read(unsigned long long* out16, int reset); }
#endif

extern "C" {

#ifdef MSCKF_ABLATE
// ablation knobs of the -DMSCKF_ABLATE build (scripts/*_ablate.py); the product library does not export this symbol
void msckf_hip_debug_set(int idx, int val) {
  if (idx == 200) { msckf::feat_debug_set(val); return; }
  if (idx == 300) { msckf::g_gram_dbg = val; return; }
  if (idx == 400) { msckf::chol_debug_set(val); return; }
  msckf::qr_debug_set(idx, val);
}
void msckf_hip_debug_chol_cycles(unsigned long long* out16, int reset) { msckf::chol_cycles_read(out16, reset); }
void msckf_hip_debug_chol_sub(unsigned long long* out32, int reset) { msckf::chol_sub_read(out32, reset); }
void msckf_hip_debug_prop_cycles(unsigned long long* out8, int reset) { msckf::prop_cycles_read(out8, reset); }
void msckf_hip_debug_featp_cycles(unsigned long long* out8, int reset) { msckf::featp_cycles_read(out8, reset); }
void msckf_hip_debug_gram_cycles(unsigned long long* out40, int reset) { msckf::gram_cycles_read(out40, reset); }
void msckf_hip_debug_gemm_cycles(unsigned long long* out16, int reset) { msckf::gemm_cycles_read(out16, reset); }
void msckf_hip_debug_gemm_trace(unsigned long long* out) { msckf::gemm_trace_read(out); }
void msckf_hip_debug_us_cycles(unsigned long long* out16, int reset) { msckf::us_cycles_read(out16, reset); }
#endif

const char* msckf_hip_last_error(void) { return g_err.c_str(); }

int msckf_hip_create(int B, int n_cap, int f_cap, int m_cap, int dtype, int device, msckf_hip_handle* out) {
  if (!out) return fail(-EINVAL, "null out pointer");
  *out = nullptr;
  if (B <= 0 || n_cap <= 0 || f_cap <= 0 || m_cap < 2 || m_cap > 64) return fail(-EINVAL, "bad capacities (need B,n_cap,f_cap > 0 and 2 <= m_cap <= 64)");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(-ENODEV, "no HIP device available (this library has no CPU fallback)");
  if (device < 0 || device >= ndev) return fail(-ENODEV, "HIP device index out of range");
  BatchBase* b = nullptr;
  if (dtype == MSCKF_HIP_F32 || dtype == MSCKF_HIP_F16H_F32P) b = new Batch<float>();
  else if (dtype == MSCKF_HIP_F64) b = new Batch<double>();
  else return fail(-EINVAL, "dtype must be MSCKF_HIP_F32, MSCKF_HIP_F64 or MSCKF_HIP_F16H_F32P");
  b->B = B; b->n_cap = n_cap; b->f_cap = f_cap; b->m_cap = m_cap; b->dtype = dtype; b->device = device;
  b->h16 = dtype == MSCKF_HIP_F16H_F32P;
  int rc = dtype != MSCKF_HIP_F64 ? static_cast<Batch<float>*>(b)->create() : static_cast<Batch<double>*>(b)->create();
  if (rc) { delete b; return rc; }
  *out = reinterpret_cast<msckf_hip_handle>(b);
  return 0;
}
int msckf_hip_destroy(msckf_hip_handle h) { delete H(h); return 0; }

int msckf_hip_initialize(msckf_hip_handle h, int b, const double* cam12, const double* noise29, const double* params8, const double* imu29) {
  return H(h)->init(b, cam12, noise29, params8, imu29);
}
int msckf_hip_propagate(msckf_hip_handle h, int b, const double* readings7, int K) { StageRange r("imu_prop"); return H(h)->propagate(b, 1, readings7, K, true); }
int msckf_hip_augment_state(msckf_hip_handle h, int b, int state_id, double time) {
  BatchBase* B = H(h);
  if (b < 0 || b >= B->B) return fail(-EINVAL, "trajectory index out of range");
  if ((int)B->traj[b].cams.size() >= B->n_cap) return fail(-EOVERFLOW, "camera-state capacity n_cap exceeded");
  StageRange r("msckf_augment_state");
  int rc = B->augment(b, 1);
  if (rc) return rc;
  B->traj[b].cams.push_back(CamMeta{state_id, time, -1, {}});
  B->traj[b].map.clear();   // msckf.h:149
  B->traj[b].map_pending = 0;   // (the points of the previous marginalize() still on the device belong to the map just cleared)
  return 0;
}
int msckf_hip_update(msckf_hip_handle h, int b, const double* meas2, const uint64_t* ids, int n) {
  if (b < 0 || b >= H(h)->B) return fail(-EINVAL, "trajectory index out of range");
  StageRange r("msckf_update");
  return host_update(H(h), b, meas2, ids, n);
}
int msckf_hip_add_features(msckf_hip_handle h, int b, const double* meas2, const uint64_t* ids, int n) {
  if (b < 0 || b >= H(h)->B) return fail(-EINVAL, "trajectory index out of range");
  StageRange r("msckf_add_features");
  return host_add_features(H(h), b, meas2, ids, n);
}
int msckf_hip_marginalize(msckf_hip_handle h, int b) {
  if (b < 0 || b >= H(h)->B) return fail(-EINVAL, "trajectory index out of range");
  StageRange r("msckf_marginalize");
  return host_marginalize(H(h), b);
}
int msckf_hip_prune_empty_states(msckf_hip_handle h, int b) {
  if (b < 0 || b >= H(h)->B) return fail(-EINVAL, "trajectory index out of range");
  StageRange r("msckf_prune_empty_states");
  return host_prune_empty(H(h), b);
}
int msckf_hip_image_cycle_range(msckf_hip_handle h, int b0, int nb, const int* state_ids, const double* times,
                                const double* upd_meas2, const uint64_t* upd_ids, const int* upd_n,
                                const double* new_meas2, const uint64_t* new_ids, const int* new_n, int flags) {
  if (!h) return fail(-EINVAL, "null handle");
  if (!state_ids || !upd_n || !new_n) return fail(-EINVAL, "null argument");
  StageRange r("msckf_image_cycle_range");
  return host_image_cycle(H(h), b0, nb, state_ids, times, upd_meas2, upd_ids, upd_n, new_meas2, new_ids, new_n, flags);
}
int msckf_hip_prune_redundant_states(msckf_hip_handle h, int b) {
  if (b < 0 || b >= H(h)->B) return fail(-EINVAL, "trajectory index out of range");
  StageRange r("msckf_prune_redundant");
  return host_prune_redundant(H(h), b);
}
int msckf_hip_finish(msckf_hip_handle h, int b) {
  if (b < 0 || b >= H(h)->B) return fail(-EINVAL, "trajectory index out of range");
  return host_finish(H(h), b);
}
int msckf_hip_get_num_cam_states(msckf_hip_handle h, int b) {   // cam_states_.size(): the host keeps the count (augment, prune, drop and run_frames all update it)
  const int n = H(h)->ncam_host(b);
  if (n == -EIO) return fail(-EIO, "handle unusable after a failed run_frames call (destroy it)");
  return n < 0 ? fail(-EINVAL, "trajectory index out of range") : n;
}
int msckf_hip_get_imu_state(msckf_hip_handle h, int b, double* imu29) { return H(h)->get_imu(b, imu29); }
int msckf_hip_get_cam_states(msckf_hip_handle h, int b, double* cam7, int* state_ids, int cap) {
  int n = 0;
  int rc = H(h)->get_cams(b, cam7, cap, &n);
  if (rc) return rc;
  if (state_ids) {
    const auto& cams = H(h)->traj[b].cams;
    for (int i = 0; i < n; ++i) state_ids[i] = i < (int)cams.size() ? cams[i].state_id : -1;
  }
  return n;
}
int msckf_hip_get_map(msckf_hip_handle h, int b, double* xyz, int cap) {
  if (b < 0 || b >= H(h)->B) return fail(-EINVAL, "trajectory index out of range");
  { const int rc = resolve_map(H(h), b); if (rc) return rc; }
  const auto& m = H(h)->traj[b].map;
  const int n = (int)m.size() / 3;
  if (n > cap) return fail(-E2BIG, "output buffer too small");
  std::copy(m.begin(), m.end(), xyz);
  return n;
}
static std::vector<PrunedState> sorted_pruned(const HostTraj& t) {
  std::vector<PrunedState> p = t.pruned;
  std::stable_sort(p.begin(), p.end(), [](const PrunedState& a, const PrunedState& c) { return a.state_id < c.state_id; });   // msckf.h:842-846
  return p;
}
int msckf_hip_get_pruned_state_ids(msckf_hip_handle h, int b, int* ids, int cap) {
  if (b < 0 || b >= H(h)->B) return fail(-EINVAL, "trajectory index out of range");
  const std::vector<PrunedState> p = sorted_pruned(H(h)->traj[b]);
  if ((int)p.size() > cap) return fail(-E2BIG, "output buffer too small");
  for (size_t i = 0; i < p.size(); ++i) ids[i] = p[i].state_id;
  return (int)p.size();
}
int msckf_hip_get_pruned_states(msckf_hip_handle h, int b, double* cam7, double* time, int* state_ids, int* last_correlated_ids, int cap) {
  if (b < 0 || b >= H(h)->B) return fail(-EINVAL, "trajectory index out of range");
  const std::vector<PrunedState> p = sorted_pruned(H(h)->traj[b]);
  if ((int)p.size() > cap) return fail(-E2BIG, "output buffer too small");
  for (size_t i = 0; i < p.size(); ++i) {
    if (cam7) std::copy(p[i].pose, p[i].pose + 7, cam7 + 7 * i);
    if (time) time[i] = p[i].time;
    if (state_ids) state_ids[i] = p[i].state_id;
    if (last_correlated_ids) last_correlated_ids[i] = p[i].last_correlated_id;
  }
  return (int)p.size();
}
int msckf_hip_get_cam_meta(msckf_hip_handle h, int b, double* time, int* n_tracked, int* last_correlated_ids, int cap) {
  if (b < 0 || b >= H(h)->B) return fail(-EINVAL, "trajectory index out of range");
  const auto& cams = H(h)->traj[b].cams;
  if ((int)cams.size() > cap) return fail(-E2BIG, "output buffer too small");
  for (size_t i = 0; i < cams.size(); ++i) {
    if (time) time[i] = cams[i].time;
    if (n_tracked) n_tracked[i] = (int)cams[i].tracked.size();
    if (last_correlated_ids) last_correlated_ids[i] = cams[i].last_correlated_id;
  }
  return (int)cams.size();
}
int msckf_hip_get_tracked_feature_ids(msckf_hip_handle h, int b, int cam_index, uint64_t* ids, int cap) {
  if (b < 0 || b >= H(h)->B) return fail(-EINVAL, "trajectory index out of range");
  const auto& cams = H(h)->traj[b].cams;
  if (cam_index < 0 || cam_index >= (int)cams.size()) return fail(-EINVAL, "camera index out of range");
  const auto& tr = cams[(size_t)cam_index].tracked;
  if ((int)tr.size() > cap) return fail(-E2BIG, "output buffer too small");
  std::copy(tr.begin(), tr.end(), ids);
  return (int)tr.size();
}
int msckf_hip_get_covariance(msckf_hip_handle h, int b, double* P, int ld) { return H(h)->get_cov(b, P, ld); }
int msckf_hip_set_covariance(msckf_hip_handle h, int b, const double* P, int D) { return H(h)->set_cov(b, P, D); }
int msckf_hip_set_imu_state(msckf_hip_handle h, int b, const double* imu29) { return H(h)->set_imu(b, imu29); }
int msckf_hip_set_cam_pose(msckf_hip_handle h, int b, int slot, const double* cam7) { return H(h)->set_cam(b, slot, cam7); }
int msckf_hip_get_num_residualized(msckf_hip_handle h, int b, long long* n) { return H(h)->get_nres(b, n); }
int msckf_hip_set_num_residualized(msckf_hip_handle h, int b, long long n) { return H(h)->set_nres(b, n); }
int msckf_hip_last_stats(msckf_hip_handle h, int b, int* out7) { return H(h)->stats(b, out7); }
int msckf_hip_clear_error_flags(msckf_hip_handle h, int b) { if (!h) return fail(-EINVAL, "null handle"); return H(h)->clear_errors(b); }
int msckf_hip_last_tracks(msckf_hip_handle h, int b, double* out8, int cap) { return H(h)->track_info(b, out8, cap); }
int msckf_hip_last_deltax(msckf_hip_handle h, int b, double* dx, int cap) { return H(h)->deltax(b, dx, cap); }

int msckf_hip_set_tracks(msckf_hip_handle h, int b, int F, const int* M, const int* slots, const double* obs2) { return H(h)->set_tracks(b, F, M, slots, obs2); }
int msckf_hip_propagate_range(msckf_hip_handle h, int b0, int nb, const double* readings7, int K) { return H(h)->propagate(b0, nb, readings7, K); }
int msckf_hip_augment_range(msckf_hip_handle h, int b0, int nb) { return H(h)->augment(b0, nb); }
int msckf_hip_marginalize_range(msckf_hip_handle h, int b0, int nb) { return H(h)->marginalize(b0, nb); }
int msckf_hip_drop_oldest_range(msckf_hip_handle h, int b0, int nb, int n_drop) { return H(h)->drop_oldest(b0, nb, n_drop); }

int msckf_hip_scenario_alloc(msckf_hip_handle h, int n_frames, int K) { return H(h)->scen_alloc(n_frames, K); }
int msckf_hip_scenario_set(msckf_hip_handle h, int frame, int b, const double* readings7, int F, const int* M, const int* slots, const double* obs2, int n_drop) {
  return H(h)->scen_set(frame, b, readings7, F, M, slots, obs2, n_drop);
}
int msckf_hip_scenario_commit(msckf_hip_handle h) { return H(h)->scen_commit(); }
int msckf_hip_run_frames(msckf_hip_handle h, int f0, int f1) { return H(h)->run_frames(f0, f1); }
int msckf_hip_run_frames_streamed(msckf_hip_handle h, int f0, int f1) { return H(h)->run_frames_streamed(f0, f1); }
int msckf_hip_sync(msckf_hip_handle h) { return H(h)->sync(); }
int msckf_hip_profile_enable(msckf_hip_handle h, int on) { return H(h)->prof_enable(on); }
int msckf_hip_profile_read(msckf_hip_handle h, double* ms7, int* count7) { return H(h)->prof_read(ms7, count7, 8); }
int msckf_hip_profile_read_ex(msckf_hip_handle h, double* ms, int* count, int cap) { if (!h || !ms || !count || cap < 8) return fail(-EINVAL, "bad arguments"); for (int s = NSTAGE; s < cap; ++s) { ms[s] = 0; count[s] = 0; } return H(h)->prof_read(ms, count, cap); }
int msckf_hip_profile_event_overhead(msckf_hip_handle h, double* ms) { if (!h || !ms) return fail(-EINVAL, "null argument"); return H(h)->prof_event_overhead(ms); }
int msckf_hip_set_host_affinity(msckf_hip_handle h, const int* cpus, int n) { if (!h || (n > 0 && !cpus)) return fail(-EINVAL, "null argument"); return H(h)->set_host_affinity(cpus, n); }
int msckf_hip_set_streams(msckf_hip_handle h, int n) { return H(h)->set_streams(n); }
int msckf_hip_scenario_pin(msckf_hip_handle h, int f0, int f1) { if (!h) return fail(-EINVAL, "null handle"); return H(h)->scen_pin(f0, f1); }
int msckf_hip_set_upload_ring(msckf_hip_handle h, int depth, int mode) { if (!h) return fail(-EINVAL, "null handle"); return H(h)->set_upload_ring(depth, mode); }
int msckf_hip_set_feature_overlap(msckf_hip_handle h, int on) { if (!h) return fail(-EINVAL, "null handle"); return H(h)->set_feature_overlap(on); }
int msckf_hip_set_covariance_update(msckf_hip_handle h, int form) { if (!h) return fail(-EINVAL, "null handle"); return H(h)->set_cov_update(form); }
int msckf_hip_set_compression(msckf_hip_handle h, int route) { if (!h) return fail(-EINVAL, "null handle"); return H(h)->set_compression(route); }
int msckf_hip_set_gate_early_accept(msckf_hip_handle h, int on) { if (!h) return fail(-EINVAL, "null handle"); return H(h)->set_gate_early(on); }
int msckf_hip_set_anisotropic_noise(msckf_hip_handle h, int mode, double tail_tol) { if (!h) return fail(-EINVAL, "null handle"); return H(h)->set_aniso(mode, tail_tol); }
int msckf_hip_copy_state(msckf_hip_handle dst, msckf_hip_handle src) { if (!dst || !src) return fail(-EINVAL, "null handle"); return H(dst)->copy_from(H(src)); }
int msckf_hip_get_error_flags(msckf_hip_handle h, int b, int* flags) { if (!h || !flags) return fail(-EINVAL, "null argument"); return H(h)->error_flags(b, flags); }
int msckf_hip_literal_info(msckf_hip_handle h, int b, int* out8) { if (!h || !out8) return fail(-EINVAL, "null argument"); return H(h)->lit_info(b, out8); }

}  // extern "C"
