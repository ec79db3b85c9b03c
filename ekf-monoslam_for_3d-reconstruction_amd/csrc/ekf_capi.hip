// Host side of libekfslam_hip.so: the filter object (device buffers, feature table, launch
// sequences) and the C ABI declared in include/ekf_monoslam.h.  gfx950 only.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/ekf_monoslam.h"
#include "ekf_dense.hpp"
#include "ekf_image.hpp"
#include "ekf_syrk6.hpp"
#include "ekf_chain.hpp"
#include "ekf_small.hpp"
#include "ekf_step.hpp"
#include "ekf_kernels.hpp"
#include "ekf_shard.hpp"

namespace ekf {

static thread_local std::string g_create_error;

enum KernelId : int {
  KID_PREDICT_CAMERA = 0,
  KID_PROPAGATE_STRIPS,
  KID_PROPAGATE_STREAMING,
  KID_MEASURE,
  KID_INNOVATION,
  KID_SIGMA_HT,
  KID_INNOVATION_COV,
  KID_CHOL_DIAG,
  KID_CHOL_PANEL,
  KID_CHOL_TRAILING,
  KID_STATE_UPDATE,
  KID_DOWNDATE,
  KID_SOLVE,
  KID_NORMALIZE,
  KID_ADD_FEATURE,
  KID_COMPACT,
  KID_MISC,
  KID_WUPDATE,
  KID_GATHER_H,
  KID_GATHER_S,
  KID_GATHER_V,
  KID_GATHER_SIGMA,
  KID_COUNT
};

static const char* kKernelNames[KID_COUNT] = {
    "predict_camera",  "propagate_strips", "propagate_streaming", "measure",
    "innovation",      "sigma_ht",         "innovation_cov",      "chol_diag",
    "chol_panel",      "chol_trailing",    "state_update",        "downdate_syrk",
    "solve_trmm",      "normalize_quat",  "add_feature",      "compact_transform",   "misc",
    "w_update", "allgather_h", "allgather_s", "allgather_v", "allgather_sigma"};

// Which launch structure an update actually took (ekf_launch_count): host-side counters, always on, one increment per
// launch.  The order is the ABI's `enum ekf_launch_kind`.
static const char* kLaunchNames[EKF_LAUNCH_KINDS] = {
    "downdate_bf16x6", "downdate_f32", "downdate_f32_fused_wu", "downdate_f32_half_tail", "downdate_f32_t64",
    "row_rider", "row_gemv", "row_tile_gemm", "w_update_gemm", "w_recompute",
    "chain_step_launches", "chain_persistent", "solve", "solve_two_groups", "update_oneblock", "update_allinone",
    "chain_trail_diag", "split_image", "state_update_tail", "update_onelaunch", "chain_dist_gather", "chain_step_fused"};

static inline int round_up(int v, int a) { return (v + a - 1) / a * a; }

// CU-masked streams are never destroyed: hipStreamDestroy of a stream made by hipExtStreamCreateWithCUMask stalls for good about
// once in a few hundred calls on this runtime (ROCm 7.2; tools/lifecycle_soak.py, DESIGN.md 8: the cause of the two test hangs of
// round 6).  A filter that goes away hands its stream back -- idle: its work was synchronised -- and the next filter of the same
// device and mask takes it over.  What is left in the pool at process exit is destroyed by an atexit handler registered on first
// use -- i.e. behind the runtime's own initialisation, so it runs in front of the runtime's (and a profiler's) tear-down: a masked
// stream still alive at that point crashes rocprofv3's exit (SIGSEGV in __cxa_finalize); one destroy per process and mask, on an
// idle device, instead of one per filter.
struct MaskedStreamPool {
  struct Entry { int device, num_cus, reserved; hipStream_t st; };
  std::mutex mu;
  std::vector<Entry> idle;
  bool hooked = false;
  static void at_exit();
  hipStream_t take(int device, int num_cus, int reserved) {
    std::lock_guard<std::mutex> lk(mu);
    for (size_t i = 0; i < idle.size(); ++i)
      if (idle[i].device == device && idle[i].num_cus == num_cus && idle[i].reserved == reserved) {
        hipStream_t st = idle[i].st;
        idle.erase(idle.begin() + i);
        return st;
      }
    return nullptr;
  }
  void give(int device, int num_cus, int reserved, hipStream_t st) {
    std::lock_guard<std::mutex> lk(mu);
    idle.push_back({device, num_cus, reserved, st});
    if (!hooked) { hooked = true; atexit(&MaskedStreamPool::at_exit); }
  }
};
static MaskedStreamPool g_masked_streams;
void MaskedStreamPool::at_exit() {
  std::lock_guard<std::mutex> lk(g_masked_streams.mu);
  for (const Entry& e : g_masked_streams.idle) {
    if (hipSetDevice(e.device) != hipSuccess) continue;
    (void)hipStreamSynchronize(e.st);
    (void)hipStreamDestroy(e.st);
  }
  g_masked_streams.idle.clear();
}

struct FilterBase {
  std::string err;
  bool diag_synced = false;   // sharded filter: the owners' diagonal blocks were gathered and nothing changed Sigma or the layout since (shard_sync_diag_blocks)
  virtual ~FilterBase() {}
  virtual int set_dt(double) = 0;
  virtual double get_dt() const = 0;
  virtual int set_stream(void*) = 0;
  virtual int set_option(int, int) = 0;
  virtual int synchronize() = 0;
  virtual int add_feature(double, double) = 0;
  virtual int remove_features(const int*, int) = 0;
  virtual int predict(const void*, const void*, int) = 0;
  virtual int measure() = 0;
  virtual int motion_jacobian(void*, void*) = 0;
  virtual int get_predictions(void*, unsigned char*, unsigned char*, void*, void*, void*) = 0;
  virtual int update(const void*, const int*, int, int, bool) = 0;
  virtual int innovation_covariance(const int*, int, int, void*) = 0;
  virtual int get_gain(void*) = 0;
  virtual int last_rows() const = 0;
  virtual int convert(int index, bool all) = 0;
  virtual int num_features() const = 0;
  virtual int state_dim() const = 0;
  virtual int get_layout(int*, int*) const = 0;
  virtual int get_state(void*, int, int) = 0;
  virtual int set_state(const void*, int, int) = 0;
  virtual int get_sigma(void*, int, int, int, int) = 0;
  virtual int peek_work(int, void*, int, int, int, int) = 0;
  virtual int set_sigma(const void*, int, int, int, int) = 0;
  virtual int covariance_parameter(double*) = 0;
  virtual int check_invariants(double*, double*, double*) = 0;
  virtual int feature_xyz(int, void*, void*) = 0;
  virtual int profile_read(int, double*, long long*) = 0;
  virtual int profile_reset() = 0;
  virtual int profile_work(int, double*) = 0;
  virtual int chunk_plan(int*, int, int*, int*) = 0;
  long long launch_cnt[EKF_LAUNCH_KINDS] = {};           // ekf_launch_count: since ekf_create / ekf_profile_reset
  virtual void* dev_mu() = 0;
  virtual void* dev_sigma(int*) = 0;
  virtual int export_points(void*, int) = 0;
  virtual int export_points_table(void*, int, int*) = 0;
  virtual int feature_ids(int*, int*) const = 0;
  virtual int set_feature_meta(int, int, int) = 0;
  virtual int num_archived() const = 0;
  virtual int rescue(const void*, const void*, const int*, int, double, unsigned char*) = 0;
  virtual int search_ellipses(int, int*) = 0;
  virtual int ransac(const void*, const int*, int, double, int*, unsigned char*, int*) = 0;
  virtual int update_two_stage(const void*, const int*, int, int, unsigned int, double, double, unsigned char*,
                               unsigned char*, int*) = 0;
  virtual int set_frame(const unsigned char*, int, int, int) = 0;
  virtual int set_patch(int, const unsigned char*) = 0;
  virtual int get_patch(int, int, unsigned char*) = 0;
  virtual int blur_predictions(void*) = 0;
  virtual int find_matches(double, void*, unsigned char*, float*) = 0;
  virtual int shard_configure(int, int, ekf_allgather_fn, void*) = 0;
  virtual int shard_info(ekf_shard_info*) = 0;
  virtual int shard_update(const void*, const int*, int, int) = 0;
  virtual int shard_rebalance() = 0;
};

#define HIPCHK(expr)                                                                         \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess) {                                                                  \
      char _b[512];                                                                          \
      snprintf(_b, sizeof(_b), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
               __LINE__);                                                                    \
      err = _b;                                                                              \
      return EKF_ERR_DEVICE;                                                                 \
    }                                                                                        \
  } while (0)

#define FAIL(code, msg) \
  do {                  \
    err = (msg);        \
    return (code);      \
  } while (0)

template <typename T>
struct Filter : FilterBase {
  ekf_config cfg;
  CamParams cam;
  int camera_dim = 14;
  int capN = 0, cap_n = 0, n_pad = 0, ld = 0;
  int N = 0, n = 0;
  int device = 0;
  double dT = 1.0;                                     // vR.cpp:155
  double vmax[6];
  int sigma_pixel_2 = 4;
  std::vector<int> pos, coding;
  // Patch::real_index (= patchnumbre at creation, vR.cpp:148, 318-319) and Patch::n_find (Patch.cpp:86, 145) per feature;
  // the patches archived at removal (vR.cpp:394-404): real_index on the host, XYZ + 3x3 covariance on the device
  std::vector<int> real_index, n_find;
  int patchnumbre = 1;
  std::vector<int> arch_real;
  T* d_archive = nullptr;
  size_t arch_cap = 0;
  int* d_arch_idx = nullptr;
  size_t arch_idx_cap = 0;
  bool layout_dirty = true;
  int *d_pos = nullptr, *d_coding = nullptr;
  T* d_mu[2] = {nullptr, nullptr};
  int cur_mu = 0;
  T* d_S[2] = {nullptr, nullptr};
  int cur = 0;
  int extent[2] = {0, 0};                              // largest n ever written per Sigma buffer
  T* d_scr = nullptr;
  T *d_h = nullptr, *d_Hc = nullptr, *d_Hf = nullptr, *d_Sd = nullptr;
  unsigned char *d_flags = nullptr, *d_cflag = nullptr;
  T *d_Jy = nullptr, *d_Yxyz = nullptr;
  int *d_map_src = nullptr, *d_map_conv = nullptr;
  // update workspace
  int m_cap = 0, ldy = 0, w_rows = 0;
  T* d_Y = nullptr;                                     // [S; Z], 2 ldy rows
  T* d_W = nullptr;                                     // [W; nu block], n_pad + 128 rows
  T* d_V = nullptr;                                     // [V; y block]
  T* d_Dinv = nullptr;
  T* d_z = nullptr;
  int* d_midx = nullptr;
  int* d_status = nullptr;
  T* d_tmp = nullptr;                                   // small D2H staging (>= 16 T)
  T* d_K = nullptr;                                     // lazily allocated gain buffer
  size_t K_elems = 0;
  int last_m = 0, last_m_pad = 0, last_n = 0;
  bool have_meas = false;
  bool have_sd = false;                                  // 2x2 St blocks of the current h/H evaluated?
  bool have_update = false;
  // options
  int opt_streaming = 0, opt_mfma = 1, opt_profile = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  hipStream_t stream_b = nullptr;                       // solve pieces / downdate pieces, overlapped with the chain
  hipEvent_t ev_chain[8] = {}, ev_solve[8] = {}, ev_b = nullptr, ev_wu = nullptr;
  hipStream_t stream_g = nullptr;                       // sharded step: the all-gathers of V_g, beside the rank's solves
  bool stream_b_masked = false;                         // stream_b carries a CU mask: handed back to the pool, never destroyed
  int stream_b_reserved = 0;                            // the mask it was made with (reserved_cus may be changed later)
  hipEvent_t ev_gath[8] = {}, ev_g = nullptr;
  int solve64_off = 0, solve6464_off = 0, tri64_off = 0, tri64_count = 0;
  int tri6_off = 0;                                      // the lower-triangular list, diagonal tiles first (k_syrk_bf16x6)
  int trih_off = 0, trih_count = 0;                      // the 128 x 128 list with its last opt_split_tail tiles as 64 x 128 halves
  // EKF_SPLIT_TAIL (-1: 1.5 tiles per CU, at most a third of the list): tiles at the end of the LAST downdate's queued list
  // that go out as 64 x 128 halves.  Measured at N = 1000: 384 halves 1.348 -> 1.329 ms (on the CU-masked side stream: no gain).
  int opt_split_tail = -1;
  double opt_feature_noise = 0.0;                       // EKF_OPT_FEATURE_NOISE: variance added to every feature state per predict
  int opt_split_bf16 = 1;                               // EKF_OPT_SPLIT_BF16 (default on, round 5): downdate of large maps on the bf16 matrix pipe, 3 x bf16 per operand, six products (ekf_syrk6.hpp)
  s6_u32x4* d_Vimg = nullptr;                           // plane image of V: (n_pad + 128) / 128 row blocks x ldy / 16 records of 12 KB
  int last_nchunks = 1, last_cend[8] = {};
  // image side (8f4): current frame, templates (original / matching), blur-pose predictions, match results
  unsigned char* d_frame = nullptr;
  size_t frame_cap = 0;
  int frame_w = 0, frame_h = 0;
  bool have_frame = false, have_blur = false;
  unsigned char *d_patch[2] = {nullptr, nullptr}, *d_mpatch[2] = {nullptr, nullptr};
  int cur_patch = 0;
  T* d_hb = nullptr;
  T* d_zm = nullptr;
  unsigned char* d_found = nullptr;
  float* d_score = nullptr;
  int* d_keep = nullptr;
  int opt_panel_direct = 1;                             // EKF_PANEL_DIRECT=0: panel through the general tile GEMM
  // the trailing update of step j and the diagonal factor of step j + 1 as ONE launch (k_trail_diag, ekf_chain.hpp);
  // EKF_CHAIN_FUSED_DIAG=0: diag -> panel -> trailing, three launches per block step (rounds 1-5; A/B and bit-identity check)
  int opt_chain_fused_diag = 1;
  // the whole update of a small map (n_pad <= 256, one diagonal block) as ONE launch (k_update_small_onelaunch, ekf_small.hpp);
  // EKF_SMALL_ONELAUNCH=0: W, S, the factor and the rest as four launches (rounds 3-5; A/B and bit-identity check)
  int opt_small_onelaunch = 1;
  // a block step of the chain (factor, panel, trailing update) as ONE launch where the step has fewer workgroups than the chip has
  // CUs and the chain runs alone (one column chunk: N up to ~230; k_chain_step_fused, ekf_step.hpp); EKF_STEP_FUSED=0: three launches
  int opt_step_fused = 1;
  struct StepPlan { int nblk = 0, s1 = 0; std::vector<int> off, cnt; } sfp;
  int* d_sf_lists = nullptr;
  bool sf_now = false;                                  // this update's chain may take the fused step (set by update())
  unsigned long long* d_small_stamps = nullptr;         // EKF_SMALL_STAMPS=1: phase stamps of its workgroup 0 (ekf_peek_workspace, which = 3)
  unsigned small_gate_total = 0;                        // arrivals the gate word (d_status[9]) has seen when every launch so far is over
  int opt_su_tail = 1;                                  // EKF_SU_TAIL=0: k_state_update as its own launch on the second stream beside the last downdate (round 5)
  int opt_fuse_split = 1;                               // EKF_FUSE_SPLIT=0: the plane image of V_g by its own launch behind the solve (rounds 5)
  bool vimg_done = false;                               // this chunk's solve has written the plane image of V_g
  int opt_chain_defer = 1;                              // EKF_CHAIN_DEFER=0: a chunk's event behind the trailing update of its last step (rounds 1-5)
  int td_min_blocks = 24;                               // EKF_TD_MIN_BLOCKS: steps with fewer blocks in their update keep the three launches
  int td_max_blocks = 1 << 30;                          // EKF_TD_MAX_BLOCKS: ... and so do steps with more (many rounds of blocks: the 64 x 64 tile kernel's occupancy wins)
  int* d_td_blocks = nullptr;
  std::vector<int> td_off, td_cnt;                      // per block step: its list of (I, K) blocks inside d_td_blocks
  int td_nblk = 0, td_nchunks = 0, td_cend[8] = {};
  int chain_diag_ahead = -1;                            // block step whose diagonal factor the last k_trail_diag launch has already done
  // EKF_SYRK_STAGGER="h,m": de-phasing of the bf16x6 downdate's workgroups (Syrk6Args).  Round 6, N = 1000, knob A/B: every
  // (0, m) with m = 1 .. 6 measures 0.897-0.903 ms per step against 0.917-0.922 without; a late second half (h > 0) gains nothing
  int opt_syrk_stag_half = 0, opt_syrk_stag_mod4 = 2;
  // EKF_CHAIN_PERSISTENT=1: the chain as one look-ahead launch per column chunk (ekf_chain.hpp).  Bit-identical to the
  // per-step launches and NOT faster (round 6, measured: profiles/r6_chain_persistent_trace.txt): 45-50 us per block
  // step against 35-39 -- the critical workgroup moves ~360 KB per step through ONE CU, whose write-through stores run at
  // 10-50 GB/s.  Off by default.
  int opt_chain_persistent = 0;
  ChainPlan chain_plan;
  ChainTask* d_chain_tasks = nullptr;
  unsigned* d_chain_flags = nullptr;
  int chain_flags_cap = 0;
  unsigned chain_epoch = 0;
  unsigned* d_chain_trace = nullptr;                    // EKF_CHAIN_TRACE=1 (diagnostics, tools/chain_trace.py): per-task time stamps of the last update
  static constexpr int kChainTraceCap = 1 << 16;
  int opt_solve_s2 = 1;                                 // EKF_SOLVE_S2: latency-bound solve launches on two wave groups (halves of K)
  bool solve_s2_now = false;
  // A solve launch of under ~one round of tiles is bounded by the K steps of its heaviest tile: two wave groups per
  // workgroup then take half of K each (k_gemm_mfma<.., S2>).  The rule only looks at the chunk width and the size of the
  // WHOLE state (never at the rows one rank holds), so the plain and the sharded path sum every element in the same order.
  bool want_solve_s2(int width, int npad_live) const {
    return kIsF32 && opt_mfma && opt_solve_s2 && (opt_solve_s2 > 1 || (width / 128) * (npad_live / 64) <= 2 * num_cus);   // (2: always, for A/B runs)
  }
  int opt_fused = 1;                                    // EKF_OPT_FUSED_LAUNCHES: k_predict_fused, k_solve_state_oneblock, k_update_oneblock_small
  int opt_solve_one_per_cu = 1;                         // EKF_SOLVE_ONE_PER_CU: the last solve on one workgroup per CU when it has 1 .. 2 tiles per CU
  bool solve_one_per_cu_now = false;
  int opt_wrecompute = 1;                               // EKF_OPT_W_RECOMPUTE / EKF_W_RECOMPUTE: next chunk's W re-evaluated from the downdated Sigma
  int opt_row_gemv = 1;                                 // EKF_ROW_GEMV=0: the innovation-row update through the tile GEMM (A/B, bit-identity check)
  int opt_fuse_wu = 1;                                  // EKF_FUSE_WU: 0 never, 1 every overlapped chunk but the one before the last, 2 every overlapped chunk
  int env_chunks[8] = {}, env_nchunks = 0;               // EKF_CHUNKS="5,10,14,16": tuning knob (block steps)
  int opt_pipeline = -1;                                 // -1 auto: on when the chain has >= 8 block steps
  int* d_tilemap = nullptr;                             // work lists: [lower-tri super-tiles | solve heavy-first]
  int tilemap_nt = 0, tilemap_ntc = 0, tri_count = 0, solve_off = 0;
  const T* cur_z = nullptr;                              // measured pixels / list of the update in flight
  const int* cur_midx = nullptr;
  int w_zeroed_n = -1;                                   // n for which the pad rows of W were last cleared
  int* d_counters = nullptr;                            // one work-queue head per queued launch of an update
  int counter_next = 0;
  int num_cus = 256, reserved_cus = 32;
  // profiling
  struct Pending { int kid; hipEvent_t a, b; };
  std::vector<Pending> pending;
  std::vector<hipEvent_t> pool;
  static constexpr int kProfileSamplePeriod = 8;
  long long frame_seq = 0;                              // updates since the last profile reset (EKF_OPT_PROFILE = 3 samples on it)
  double prof_ms[KID_COUNT];
  long long prof_cnt[KID_COUNT];
  double prof_work[KID_COUNT];                          // algorithmic flop of the timed launches (downdate only)

  static constexpr bool kIsF32 = sizeof(T) == 4;
  int NB() const { return (kIsF32 && opt_mfma) ? 128 : 64; }

  ~Filter() override {
    const bool dbg = getenv("EKF_DEBUG_DTOR") != nullptr;       // (diagnostics: which call of the tear-down a stall sits in)
    auto mark = [&](const char* what) { if (dbg) { fprintf(stderr, "[ekf dtor %p] %s\n", (void*)this, what); fflush(stderr); } };
    hipSetDevice(device);
    mark("sync main stream");
    if (stream) hipStreamSynchronize(stream);
    mark("sync side streams");
    if (stream_b) hipStreamSynchronize(stream_b);
    if (stream_g) hipStreamSynchronize(stream_g);
    mark("events");
    for (auto& p : pending) { hipEventDestroy(p.a); hipEventDestroy(p.b); }
    for (auto e : pool) hipEventDestroy(e);
    mark("device memory");
    void* ptrs[] = {d_pos, d_coding, d_mu[0], d_mu[1], d_S[0], d_S[1], d_scr, d_h, d_Hc, d_Hf, d_Sd,
                    d_flags, d_cflag, d_Jy, d_Yxyz, d_map_src, d_map_conv, d_Y, d_W, d_V, d_Dinv, d_z, d_midx,
                    d_status, d_tmp, d_K, d_tilemap, d_counters, d_ibuf, d_rmask, d_pts, d_tab,
                    d_frame, d_patch[0], d_patch[1], d_mpatch[0], d_mpatch[1], d_hb, d_zm, d_found, d_score, d_keep,
                    d_Vimg, d_stage_send, d_stage_recv, d_archive, d_arch_idx, d_panel_tiles, d_shard_solve, d_shard_syrk,
                    d_chain_tasks, d_chain_flags, d_chain_trace, d_td_blocks, d_small_stamps,
                    d_dist_lists, d_dist_counters, d_dist_send, d_dist_recv, d_sf_lists};
    for (void* p : ptrs) if (p) hipFree(p);
    mark("host memory");
    for (int s = 0; s < kInSlots; ++s) { if (h_in[s]) hipHostFree(h_in[s]); if (ev_in[s]) hipEventDestroy(ev_in[s]); }
    if (h_pred) hipHostFree(h_pred);
    if (h_ransac) hipHostFree(h_ransac);
    if (h_gate) hipHostFree(h_gate);
    if (h_rb) hipHostFree(h_rb);
    mark("streams");
    if (own_stream && stream) hipStreamDestroy(stream);
    mark("stream b");
    const char* pool_env = getenv("EKF_MASKED_STREAM_POOL");        // =0: destroy it, as rounds 1-5 did (A/B; can stall, see the pool)
    if (stream_b && stream_b_masked && !(pool_env && atoi(pool_env) == 0)) g_masked_streams.give(device, num_cus, stream_b_reserved, stream_b);
    else if (stream_b) hipStreamDestroy(stream_b);
    mark("stream g");
    if (stream_g) hipStreamDestroy(stream_g);
    mark("last events");
    for (auto e : ev_gath) if (e) hipEventDestroy(e);
    if (ev_g) hipEventDestroy(ev_g);
    for (auto e : ev_chain) if (e) hipEventDestroy(e);
    for (auto e : ev_solve) if (e) hipEventDestroy(e);
    if (ev_b) hipEventDestroy(ev_b);
    if (ev_wu) hipEventDestroy(ev_wu);
    mark("done");
  }

  // ---- profiling helpers ---------------------------------------------------------------
  bool prof_on(int kid) const {
    if (opt_profile == 2) return true;
    const bool dominant = (kid == KID_DOWNDATE || kid == KID_PROPAGATE_STREAMING);
    if (opt_profile == 1) return dominant;
    if (opt_profile == 3) return dominant && (frame_seq % kProfileSamplePeriod == 0);   // every 8th frame: an event pair costs ~6 us of queue time
    return false;
  }
  hipEvent_t get_event() {
    if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
    hipEvent_t e;
    hipEventCreate(&e);
    return e;
  }
  struct Scope {
    Filter* f; int kid; hipEvent_t a = nullptr, b = nullptr; bool on; hipStream_t st;
    Scope(Filter* f_, int kid_, hipStream_t st_ = nullptr)
        : f(f_), kid(kid_), on(f_->prof_on(kid_)), st(st_ ? st_ : f_->stream) {
      if (on) { a = f->get_event(); b = f->get_event(); hipEventRecord(a, st); }
    }
    ~Scope() {
      if (on) { hipEventRecord(b, st); f->pending.push_back({kid, a, b}); }
    }
  };
  void resolve_profile() {
    if (pending.empty()) return;
    hipStreamSynchronize(stream);
    hipStreamSynchronize(stream_b);
    if (stream_g) hipStreamSynchronize(stream_g);
    for (auto& p : pending) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) { prof_ms[p.kid] += ms; prof_cnt[p.kid] += 1; }
      pool.push_back(p.a); pool.push_back(p.b);
    }
    pending.clear();
  }

  // ---- construction ---------------------------------------------------------------------
  int init(const ekf_config* c, int cdim, int capacity, int dev) {
    cfg = *c;
    camera_dim = cdim;
    device = dev;
    capN = capacity;
    cap_n = camera_dim + 6 * capN;
    n_pad = round_up(cap_n, 128);
    ld = n_pad;
    m_cap = 2 * capN + 3;
    ldy = round_up(m_cap, 128);
    w_rows = n_pad + 128;
    memset(prof_ms, 0, sizeof(prof_ms));
    memset(prof_cnt, 0, sizeof(prof_cnt));
    memset(prof_work, 0, sizeof(prof_work));
    cam.fx = c->fx; cam.fy = c->fy; cam.u0 = c->u0; cam.v0 = c->v0;
    cam.k1 = c->k1; cam.k2 = c->k2; cam.k3 = c->k3; cam.p1 = c->p1; cam.p2 = c->p2;
    cam.width = c->image_width; cam.height = c->image_height; cam.half_window = c->window_size / 2;
    sigma_pixel_2 = c->sigma_pixel * c->sigma_pixel;               // ints, vR.cpp:150-151
    const float sv[6] = {c->sigma_vx, c->sigma_vy, c->sigma_vz, c->sigma_wx, c->sigma_wy, c->sigma_wz};
    for (int i = 0; i < 6; ++i) vmax[i] = double(T(sv[i]) * T(sv[i]));  // vR.cpp:194-200
    HIPCHK(hipSetDevice(device));
    {
      int lo = 0, hi = 0;
      HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
      HIPCHK(hipStreamCreateWithPriority(&stream, hipStreamNonBlocking, hi));   // the serial chain goes first
    }
    own_stream = true;
    for (auto& e : ev_chain) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto& e : ev_solve) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&ev_b, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&ev_wu, hipEventDisableTiming));
    const size_t sig = (size_t)n_pad * ld;
    HIPCHK(hipMalloc(&d_S[0], sig * sizeof(T)));
    HIPCHK(hipMalloc(&d_S[1], sig * sizeof(T)));
    HIPCHK(hipMemsetAsync(d_S[0], 0, sig * sizeof(T), stream));
    HIPCHK(hipMemsetAsync(d_S[1], 0, sig * sizeof(T), stream));
    HIPCHK(hipMalloc(&d_mu[0], (size_t)n_pad * sizeof(T)));
    HIPCHK(hipMalloc(&d_mu[1], (size_t)n_pad * sizeof(T)));
    HIPCHK(hipMemsetAsync(d_mu[0], 0, (size_t)n_pad * sizeof(T), stream));
    HIPCHK(hipMemsetAsync(d_mu[1], 0, (size_t)n_pad * sizeof(T), stream));
    HIPCHK(hipMalloc(&d_scr, SCR_SIZE * sizeof(T)));
    const size_t cn = (size_t)std::max(capN, 1);
    HIPCHK(hipMalloc(&d_pos, cn * sizeof(int)));
    HIPCHK(hipMalloc(&d_coding, cn * sizeof(int)));
    HIPCHK(hipMalloc(&d_h, cn * 2 * sizeof(T)));
    HIPCHK(hipMalloc(&d_Hc, cn * 14 * sizeof(T)));
    HIPCHK(hipMalloc(&d_Hf, cn * 12 * sizeof(T)));
    HIPCHK(hipMalloc(&d_Sd, cn * 4 * sizeof(T)));
    HIPCHK(hipMalloc(&d_flags, cn));
    HIPCHK(hipMalloc(&d_cflag, cn));
    HIPCHK(hipMalloc(&d_Jy, cn * 18 * sizeof(T)));
    HIPCHK(hipMalloc(&d_Yxyz, cn * 3 * sizeof(T)));
    HIPCHK(hipMalloc(&d_map_src, (size_t)n_pad * sizeof(int)));
    HIPCHK(hipMalloc(&d_map_conv, (size_t)n_pad * sizeof(int)));
    HIPCHK(hipMalloc(&d_Y, (size_t)2 * ldy * ldy * sizeof(T)));
    HIPCHK(hipMemsetAsync(d_Y, 0, (size_t)2 * ldy * ldy * sizeof(T), stream));
    HIPCHK(hipMalloc(&d_W, (size_t)w_rows * ldy * sizeof(T)));
    HIPCHK(hipMemsetAsync(d_W, 0, (size_t)w_rows * ldy * sizeof(T), stream));
    HIPCHK(hipMalloc(&d_V, (size_t)w_rows * ldy * sizeof(T)));
    HIPCHK(hipMemsetAsync(d_V, 0, (size_t)w_rows * ldy * sizeof(T), stream));
    HIPCHK(hipMalloc(&d_Dinv, (size_t)(ldy / 64) * 128 * 128 * sizeof(T)));
    HIPCHK(hipMalloc(&d_z, (size_t)ldy * sizeof(T)));
    HIPCHK(hipMalloc(&d_midx, cn * sizeof(int)));
    HIPCHK(hipMalloc(&d_status, 16 * sizeof(int)));         // [0] pivot <= 0, [1] bad device index list, [3] a bounded device-side wait gave up; [4..7] scratch of ekf_check_invariants; [8] arrival gate of k_predict_fused, [9] of k_update_small_onelaunch
    HIPCHK(hipMemsetAsync(d_status, 0, 16 * sizeof(int), stream));
    HIPCHK(hipMalloc(&d_tmp, 64 * sizeof(T)));
    HIPCHK(hipMalloc(&d_counters, kQueueCounters * sizeof(int)));
    HIPCHK(hipMemset(d_counters, 0, kQueueCounters * sizeof(int)));
    {
      hipDeviceProp_t prop;
      HIPCHK(hipGetDeviceProperties(&prop, device));
      num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    {
      // second stream for the pipelined solve / downdate pieces: kept off `reserved_cus` CUs so that
      // the serial chain on the main stream always finds a free CU (a chain workgroup sharing its SIMDs
      // with MFMA-saturating tile-GEMM waves runs ~4x slower: tools/coresidency.hip)
      reserved_cus = std::min(32, num_cus / 4);
      if (const char* e = getenv("EKF_RESERVED_CUS")) reserved_cus = std::max(1, std::min(num_cus / 2, atoi(e)));   // tuning knob
      std::vector<uint32_t> mask((num_cus + 31) / 32, 0xffffffffu);
      for (int i = 0; i < reserved_cus; ++i) mask[i / 32] &= ~(1u << (i % 32));
      // (a runtime that refuses CU masks still gets a second stream: the overlap works, only less well)
      stream_b = g_masked_streams.take(device, num_cus, reserved_cus);      // (a masked stream is reused, never destroyed: see the pool)
      stream_b_masked = stream_b != nullptr;
      if (!stream_b) {
        if (hipExtStreamCreateWithCUMask(&stream_b, (uint32_t)mask.size(), mask.data()) == hipSuccess) {
          stream_b_masked = true;
        } else {
          (void)hipGetLastError();
          stream_b = nullptr;
          reserved_cus = 0;
          HIPCHK(hipStreamCreateWithFlags(&stream_b, hipStreamNonBlocking));
        }
      }
      stream_b_reserved = reserved_cus;

      if (const char* e = getenv("EKF_FUSE_WU")) opt_fuse_wu = atoi(e);
      if (const char* e = getenv("EKF_ROW_GEMV")) opt_row_gemv = atoi(e) ? 1 : 0;
      if (const char* e = getenv("EKF_SOLVE_ONE_PER_CU")) opt_solve_one_per_cu = atoi(e) ? 1 : 0;
      if (const char* e = getenv("EKF_W_RECOMPUTE")) opt_wrecompute = atoi(e) ? 1 : 0;   // = EKF_OPT_W_RECOMPUTE, for A/B runs
      if (const char* e = getenv("EKF_SOLVE_S2")) opt_solve_s2 = atoi(e);
      if (const char* e = getenv("EKF_FUSED_LAUNCHES")) opt_fused = atoi(e) ? 1 : 0;   // = EKF_OPT_FUSED_LAUNCHES, for A/B runs
      if (const char* e = getenv("EKF_PANEL_DIRECT")) opt_panel_direct = atoi(e);
      if (const char* e = getenv("EKF_SYRK_STAGGER")) {
        opt_syrk_stag_half = atoi(e);
        if (const char* c = strchr(e, ',')) opt_syrk_stag_mod4 = atoi(c + 1);
      }
      if (const char* e = getenv("EKF_CHAIN_PERSISTENT")) opt_chain_persistent = atoi(e) ? 1 : 0;
      if (const char* e = getenv("EKF_CHAIN_FUSED_DIAG")) opt_chain_fused_diag = atoi(e) ? 1 : 0;
      if (const char* e = getenv("EKF_FUSE_SPLIT")) opt_fuse_split = atoi(e) ? 1 : 0;
      if (const char* e = getenv("EKF_SU_TAIL")) opt_su_tail = atoi(e) ? 1 : 0;
      if (const char* e = getenv("EKF_SMALL_ONELAUNCH")) opt_small_onelaunch = atoi(e) ? 1 : 0;
      if (const char* e = getenv("EKF_STEP_FUSED")) opt_step_fused = atoi(e) ? 1 : 0;
      if (const char* e = getenv("EKF_SHARD_DIST_CHAIN")) opt_shard_dist_chain = atoi(e) ? 1 : 0;
      if (const char* e = getenv("EKF_SHARD_DIST_MIN_BLOCKS")) shard_dist_min_blocks = std::max(2, atoi(e));
      if (const char* e = getenv("EKF_SMALL_STAMPS")) {
        if (atoi(e)) { HIPCHK(hipMalloc(&d_small_stamps, 16 * sizeof(unsigned long long))); HIPCHK(hipMemset(d_small_stamps, 0, 16 * sizeof(unsigned long long))); }
      }
      if (const char* e = getenv("EKF_CHAIN_DEFER")) opt_chain_defer = atoi(e) ? 1 : 0;
      if (const char* e = getenv("EKF_TD_MIN_BLOCKS")) td_min_blocks = std::max(1, atoi(e));
      if (const char* e = getenv("EKF_TD_MAX_BLOCKS")) td_max_blocks = std::max(1, atoi(e));
      if (const char* e = getenv("EKF_CHAIN_TRACE")) {
        if (atoi(e)) HIPCHK(hipMalloc(&d_chain_trace, (size_t)(8 + 8 * kChainTraceCap) * sizeof(unsigned)));
      }
      if (const char* e = getenv("EKF_SPLIT_TAIL")) opt_split_tail = atoi(e);
      if (const char* e = getenv("EKF_SPLIT_BF16")) opt_split_bf16 = atoi(e) ? 1 : 0;   // = EKF_OPT_SPLIT_BF16, for A/B runs
      if (const char* e = getenv("EKF_CHUNKS")) {           // tuning knob: chunk ends in block steps
        for (const char* q = e; *q && env_nchunks < 8;) {
          env_chunks[env_nchunks++] = atoi(q);
          while (*q && *q != ',') ++q;
          if (*q == ',') ++q;
        }
      }
    }
    // mu0 / Sigma0 (vR.cpp:163-180, 211-216)
    std::vector<T> mu0(camera_dim, T(0));
    mu0[3] = T(0.0); mu0[4] = T(0.0); mu0[5] = T(-0.707106781); mu0[6] = T(0.707106781);
    if (camera_dim == 14) mu0[13] = T(1);
    n = camera_dim;
    N = 0;
    HIPCHK(hipMemcpyAsync(d_mu[0], mu0.data(), camera_dim * sizeof(T), hipMemcpyHostToDevice, stream));
    std::vector<T> S0((size_t)camera_dim * camera_dim, T(0));
    for (int i = 0; i < camera_dim; ++i) S0[(size_t)i * camera_dim + i] = T(0.0000000004);
    if (camera_dim == 14) S0[13 * 14 + 13] = T(0.09);
    const T sv2 = T(0.0004) * T(0.0004);
    for (int i = 7; i < 13; ++i) S0[(size_t)i * camera_dim + i] = sv2;
    HIPCHK(hipMemcpy2DAsync(d_S[0], (size_t)ld * sizeof(T), S0.data(), camera_dim * sizeof(T),
                            camera_dim * sizeof(T), camera_dim, hipMemcpyHostToDevice, stream));
    HIPCHK(hipStreamSynchronize(stream));
    extent[0] = n; extent[1] = 0;
    // the diagonal-block kernel needs > 64 KiB of LDS
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_chol_diag<T, 64>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, diag_lds(64)));
    // (f32, NB = 128 uses k_chol_diag_packed: 66 KiB of static LDS)
    if constexpr (kIsF32) {
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_chain_persistent),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)kChainLds));
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_trail_diag),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)kChainLds));
    }
    return EKF_OK;
  }
  static int diag_lds(int nb) { return 2 * nb * (nb + 1) * (int)sizeof(T); }

  T* S() { return d_S[cur]; }
  T* mu() { return d_mu[cur_mu]; }

  int sync_layout() {
    if (!layout_dirty) return EKF_OK;
    if (N > 0) {
      HIPCHK(hipMemcpyAsync(d_pos, pos.data(), N * sizeof(int), hipMemcpyHostToDevice, stream));
      HIPCHK(hipMemcpyAsync(d_coding, coding.data(), N * sizeof(int), hipMemcpyHostToDevice, stream));
      HIPCHK(hipStreamSynchronize(stream));   // host vectors may change right after
    }
    layout_dirty = false;
    return EKF_OK;
  }

  // z (2 M scalars) and the index list (M ints) of a host caller -> d_z / d_midx through pinned slot `in_slot`
  static constexpr int kInSlots = 4;
  void* h_in[kInSlots] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_in[kInSlots] = {nullptr, nullptr, nullptr, nullptr};
  bool in_used[kInSlots] = {false, false, false, false};
  int in_slot = 0;
  void* h_pred = nullptr;                               // host-mapped read-back buffer of ekf_get_predictions
  void* h_ransac = nullptr;                             // ... of ekf_ransac_1point: counts, camera pose, small inlier masks
  static constexpr size_t kRansacMaskBytes = 65536;
  const unsigned char* ransac_mask_host = nullptr;      // the mask of the last ransac() call, if it came back with the counts
  int ransac_mask_M = 0;
  double ransac_cam[7] = {0, 0, 0, 0, 0, 0, 0};
  bool ransac_cam_valid = false;
  void* h_gate = nullptr;                               // host-mapped gate flags of ekf_rescue_high_innovation
  int stage_inputs(const void* z, const int* idx, int M, const void* cam7 = nullptr) {
    const size_t zb = (size_t)2 * M * sizeof(T), ib = (size_t)M * sizeof(int);
    const int s = in_slot;
    in_slot = (in_slot + 1) % kInSlots;
    if (!h_in[s]) {
      HIPCHK(hipHostMalloc(&h_in[s], (size_t)capN * (2 * sizeof(T) + sizeof(int)) + 8 * sizeof(T) + 64, hipHostMallocDefault));
      HIPCHK(hipEventCreateWithFlags(&ev_in[s], hipEventDisableTiming));
    }
    if (in_used[s]) HIPCHK(hipEventSynchronize(ev_in[s]));      // the copies that last used this slot are long done
    char* base = static_cast<char*>(h_in[s]);
    memcpy(base, z, zb);
    memcpy(base + (size_t)2 * capN * sizeof(T), idx, ib);
    HIPCHK(hipMemcpyAsync(d_z, base, zb, hipMemcpyHostToDevice, stream));
    HIPCHK(hipMemcpyAsync(d_midx, base + (size_t)2 * capN * sizeof(T), ib, hipMemcpyHostToDevice, stream));
    if (cam7) {                                         // (the rescue's camera pose of the state before the first update)
      char* pc7 = base + (size_t)capN * (2 * sizeof(T) + sizeof(int));
      memcpy(pc7, cam7, 7 * sizeof(T));
      HIPCHK(hipMemcpyAsync(d_tmp, pc7, 7 * sizeof(T), hipMemcpyHostToDevice, stream));
    }
    HIPCHK(hipEventRecord(ev_in[s], stream));
    in_used[s] = true;
    return EKF_OK;
  }

  // Small device -> host reads of the getters: through ONE pinned bounce buffer, several pieces and the status words per
  // synchronisation (a copy into pageable memory is staged by the runtime and costs ~25 us; every getter used to pay that
  // twice, once for its data and once for the status words).
  void* h_rb = nullptr;
  size_t rb_cap = 0, rb_off = 0;
  struct RbPiece { void* dst; size_t off, bytes; };
  std::vector<RbPiece> rb_pend;
  int rb_reserve(size_t bytes) {
    if (rb_off + bytes + 64 <= rb_cap) return EKF_OK;
    if (!rb_pend.empty()) {                            // pieces already queued: deliver them before the buffer moves
      hipError_t e = hipStreamSynchronize(stream);
      if (e == hipSuccess)
        for (const RbPiece& p : rb_pend) memcpy(p.dst, static_cast<const char*>(h_rb) + p.off, p.bytes);
      rb_pend.clear();
      rb_off = 0;
      HIPCHK(e);
      if (bytes + 64 <= rb_cap) return EKF_OK;
    }
    const size_t want = std::max<size_t>(1 << 16, 2 * (bytes + 64));
    if (h_rb) HIPCHK(hipHostFree(h_rb));
    h_rb = nullptr;
    HIPCHK(hipHostMalloc(&h_rb, want, hipHostMallocDefault));
    rb_cap = want;
    rb_off = 0;
    return EKF_OK;
  }
  // queue `bytes` from device `src` for host `dst` (copied out by rb_finish)
  int rb_add(void* dst, const void* src, size_t bytes) {
    int rc = rb_reserve(bytes);
    if (rc) { rb_pend.clear(); rb_off = 0; return rc; }    // (a failed batch leaves nothing queued behind)
    hipError_t e = hipMemcpyAsync(static_cast<char*>(h_rb) + rb_off, src, bytes, hipMemcpyDeviceToHost, stream);
    if (e != hipSuccess) { rb_pend.clear(); rb_off = 0; }
    HIPCHK(e);
    rb_pend.push_back({dst, rb_off, bytes});
    rb_off += (bytes + 15) & ~size_t(15);
    return EKF_OK;
  }
  // queue a rows x cols block (device pitch in bytes), packed row-major at dst
  int rb_add_2d(void* dst, const void* src, size_t pitch, size_t row_bytes, int rows) {
    int rc = rb_reserve(row_bytes * rows);
    if (rc) { rb_pend.clear(); rb_off = 0; return rc; }
    hipError_t e = hipMemcpy2DAsync(static_cast<char*>(h_rb) + rb_off, row_bytes, src, pitch, row_bytes, rows,
                                    hipMemcpyDeviceToHost, stream);
    if (e != hipSuccess) { rb_pend.clear(); rb_off = 0; }
    HIPCHK(e);
    rb_pend.push_back({dst, rb_off, row_bytes * rows});
    rb_off += (row_bytes * rows + 15) & ~size_t(15);
    return EKF_OK;
  }
  // one synchronisation for everything queued (+ the status words when asked for), then the pieces go to their places
  int rb_finish(bool with_status) {
    int st[4] = {0, 0, 0, 0};
    if (with_status) { int rc = rb_add(st, d_status, sizeof(st)); if (rc) { rb_pend.clear(); rb_off = 0; return rc; } }
    hipError_t e = hipStreamSynchronize(stream);
    if (e == hipSuccess)
      for (const RbPiece& p : rb_pend) memcpy(p.dst, static_cast<const char*>(h_rb) + p.off, p.bytes);
    rb_pend.clear();
    rb_off = 0;
    HIPCHK(e);
    return with_status ? eval_status(st) : EKF_OK;
  }
  int eval_status(const int* st) {
    if (st[0] || st[1] || st[2] || st[3]) {
      HIPCHK(hipMemsetAsync(d_status, 0, 4 * sizeof(int), stream));
      if (st[3]) {
        // the arrival gate of k_predict_fused was left mid-count: every launch that could still add to it has to be
        // over before it is cleared, and the fused predict launch stays off for this filter from here on
        HIPCHK(hipStreamSynchronize(stream));
        HIPCHK(hipMemsetAsync(d_status + 8, 0, 2 * sizeof(int), stream));      // ... and the gate of k_update_small_onelaunch
        small_gate_total = 0;
        opt_fused = 0;
        FAIL(EKF_ERR_DEVICE, "a bounded device-side wait gave up (fused launch); EKF_OPT_FUSED_LAUNCHES is now off for this filter");
      }
      if (st[2]) {
        // a hand-over of the persistent chain kernel was never published within its bound: the launch gave up (every
        // workgroup left); the per-step launches take over for this filter
        opt_chain_persistent = 0;
        chain_epoch = 0;
        FAIL(EKF_ERR_DEVICE, "a bounded device-side wait gave up (persistent chain); EKF_CHAIN_PERSISTENT is now off for this filter");
      }
      if (st[1])
        FAIL(EKF_ERR_ARG, "ekf_update_device: a device-resident index is outside [0, N) or the list is not strictly "
                          "ascending (indices were clamped; the state is not meaningful)");
      FAIL(EKF_ERR_NUMERIC, "innovation covariance is not positive definite (Cholesky pivot <= 0)");
    }
    return EKF_OK;
  }

  int check_status() {
    return rb_finish(true);
  }
  // ---- simple accessors -----------------------------------------------------------------
  int set_dt(double v) override { if (!(v > 0)) FAIL(EKF_ERR_ARG, "dT must be positive"); dT = v; return EKF_OK; }
  double get_dt() const override { return dT; }
  int set_stream(void* s) override {
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipStreamSynchronize(stream));
    resolve_profile();
    if (own_stream) { hipStreamDestroy(stream); own_stream = false; }
    stream = reinterpret_cast<hipStream_t>(s);
    return EKF_OK;
  }
  int set_option(int o, int v) override {
    switch (o) {
      case EKF_OPT_PROPAGATE_STREAMING: opt_streaming = v ? 1 : 0; return EKF_OK;
      case EKF_OPT_USE_MFMA: opt_mfma = v ? 1 : 0; w_zeroed_n = -1; return EKF_OK;   // tile size changes the pads
      case EKF_OPT_PROFILE: resolve_profile(); opt_profile = v; return EKF_OK;
      case EKF_OPT_PIPELINE: opt_pipeline = (v < 0) ? -1 : v; return EKF_OK;
      case EKF_OPT_SPLIT_BF16: opt_split_bf16 = v ? 1 : 0; return EKF_OK;
      case EKF_OPT_FEATURE_NOISE: opt_feature_noise = (v > 0) ? 1e-12 * v : 0.0; return EKF_OK;
      case EKF_OPT_FUSED_LAUNCHES: opt_fused = v ? 1 : 0; return EKF_OK;
      case EKF_OPT_W_RECOMPUTE: opt_wrecompute = v ? 1 : 0; return EKF_OK;
      default: FAIL(EKF_ERR_ARG, "unknown option");
    }
  }
  int synchronize() override {
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipStreamSynchronize(stream));
    return check_status();
  }
  int num_features() const override { return N; }
  int state_dim() const override { return n; }
  int last_rows() const override { return last_m; }
  int get_layout(int* p, int* c) const override {
    for (int i = 0; i < N; ++i) { if (p) p[i] = pos[i]; if (c) c[i] = coding[i]; }
    return EKF_OK;
  }
  // (the caller may write through these pointers: the gathered diagonal blocks of a sharded filter are stale from here on)
  void* dev_mu() override { diag_synced = false; return mu(); }
  void* dev_sigma(int* l) override { diag_synced = false; if (l) *l = ld; return S(); }

  // ---- a12 add feature --------------------------------------------------------------------
  int add_feature(double u, double v) override {
    HIPCHK(hipSetDevice(device));
    const T uT = T(u), vT = T(v);
    const T hw = T(cam.half_window);
    if (!((uT > hw) && (vT > hw) && (uT < T(cam.width) - hw) && (vT < T(cam.height) - hw))) return 0;
    if (N >= capN) { err = "capacity_features exceeded"; return -EKF_ERR_CAPACITY; }
    {
      Scope sc(this, KID_ADD_FEATURE);
      k_add_prepare<T><<<1, 64, 0, stream>>>(mu(), S(), ld, n, cam, uT, vT, T(cfg.rho_0), T(sigma_pixel_2),
                                            T(cfg.sigma_rho_0), d_scr);
      k_add_border<T><<<(n + 255) / 256, 256, 0, stream>>>(S(), ld, n, d_scr);
    }
    if (hipGetLastError() != hipSuccess) { err = "add_feature launch failed"; return -EKF_ERR_DEVICE; }
    if (have_frame) {
      // Patch::Patch(cv::Mat(frame, cv::Rect(pf.x - w/2, pf.y - w/2, w, w)), ...) (vR.cpp:318): float -> int truncation
      const int w = cfg.window_size;
      const int x0 = (int)(float(u) - float(w / 2)), y0 = (int)(float(v) - float(w / 2));
      k_capture_patch<<<1, 256, 0, stream>>>(d_frame, frame_w, frame_h, x0, y0, w,
                                            d_patch[cur_patch] + (size_t)N * w * w, d_mpatch[cur_patch] + (size_t)N * w * w);
    }
    pos.push_back(n);
    coding.push_back(0);
    real_index.push_back(patchnumbre++);
    n_find.push_back(1);
    N += 1;
    n += 6;
    shard_after_add();
    extent[cur] = std::max(extent[cur], n);
    layout_dirty = true;
    have_meas = false;
    return 1;
  }

  // ---- image side: frame, templates, predicted blur, NCC search ---------------
  int ensure_image_buffers() {
    if (d_patch[0]) return EKF_OK;
    const int w = cfg.window_size;
    if (w < 1 || w > kMaxWindow) FAIL(EKF_ERR_UNSUPPORTED, "window_size outside [1, 32] for the device matcher");
    const size_t bytes = (size_t)std::max(capN, 1) * w * w;
    for (int b = 0; b < 2; ++b) {
      HIPCHK(hipMalloc(&d_patch[b], bytes));
      HIPCHK(hipMalloc(&d_mpatch[b], bytes));
      HIPCHK(hipMemsetAsync(d_patch[b], 0, bytes, stream));
      HIPCHK(hipMemsetAsync(d_mpatch[b], 0, bytes, stream));
    }
    HIPCHK(hipMalloc(&d_hb, (size_t)std::max(capN, 1) * 2 * sizeof(T)));
    HIPCHK(hipMalloc(&d_zm, (size_t)std::max(capN, 1) * 2 * sizeof(T)));
    HIPCHK(hipMalloc(&d_found, (size_t)std::max(capN, 1)));
    HIPCHK(hipMalloc(&d_score, (size_t)std::max(capN, 1) * sizeof(float)));
    HIPCHK(hipMalloc(&d_keep, (size_t)std::max(capN, 1) * sizeof(int)));
    return EKF_OK;
  }
  int set_frame(const unsigned char* gray, int width, int height, int stride) override {
    HIPCHK(hipSetDevice(device));
    if (!gray || width <= 0 || height <= 0 || stride < width) FAIL(EKF_ERR_ARG, "bad frame");
    if (width != cam.width || height != cam.height)
      FAIL(EKF_ERR_ARG, "frame size differs from ekf_config image_width / image_height");
    int rc = ensure_image_buffers();
    if (rc) return rc;
    const size_t need = (size_t)width * height;
    if (need > frame_cap) {
      if (d_frame) HIPCHK(hipFree(d_frame));
      d_frame = nullptr;
      HIPCHK(hipMalloc(&d_frame, need));
      frame_cap = need;
    }
    HIPCHK(hipMemcpy2DAsync(d_frame, (size_t)width, gray, (size_t)stride, (size_t)width, height, hipMemcpyHostToDevice,
                            stream));
    HIPCHK(hipStreamSynchronize(stream));            // the caller's buffer may be reused at once
    frame_w = width; frame_h = height;
    have_frame = true;
    return EKF_OK;
  }
  int set_patch(int index, const unsigned char* data) override {
    HIPCHK(hipSetDevice(device));
    if (index < 0 || index >= N || !data) FAIL(EKF_ERR_ARG, "feature index out of range");
    int rc = ensure_image_buffers();
    if (rc) return rc;
    const size_t w2 = (size_t)cfg.window_size * cfg.window_size;
    HIPCHK(hipMemcpyAsync(d_patch[cur_patch] + index * w2, data, w2, hipMemcpyHostToDevice, stream));
    HIPCHK(hipMemcpyAsync(d_mpatch[cur_patch] + index * w2, data, w2, hipMemcpyHostToDevice, stream));
    HIPCHK(hipStreamSynchronize(stream));
    return EKF_OK;
  }
  int get_patch(int index, int matching, unsigned char* out) override {
    HIPCHK(hipSetDevice(device));
    if (index < 0 || index >= N || !out) FAIL(EKF_ERR_ARG, "feature index out of range");
    if (!d_patch[0]) FAIL(EKF_ERR_STATE, "no templates: call ekf_set_frame / ekf_set_patch first");
    const size_t w2 = (size_t)cfg.window_size * cfg.window_size;
    const unsigned char* src = (matching ? d_mpatch[cur_patch] : d_patch[cur_patch]) + index * w2;
    HIPCHK(hipMemcpyAsync(out, src, w2, hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    return EKF_OK;
  }
  // predicted blur of every visible feature's template (inside predict(), vR.cpp:496-500, 546-548, 575-576)
  int launch_blur() {
    have_blur = false;
    if (!d_patch[0] || N == 0) return EKF_OK;
    Scope sc(this, KID_MISC);
    k_blur_points<T><<<(N + 63) / 64, 64, 0, stream>>>(mu(), d_pos, d_coding, N, cam, T(cfg.T_camera), T(dT), d_hb);
    k_blur_templates<T><<<N, 256, 0, stream>>>(d_h, d_hb, d_flags, cfg.window_size, cfg.kernel_size, d_patch[cur_patch],
                                               d_mpatch[cur_patch]);
    HIPCHK(hipGetLastError());
    have_blur = true;
    return EKF_OK;
  }
  int blur_predictions(void* out) override {
    HIPCHK(hipSetDevice(device));
    if (!have_blur) FAIL(EKF_ERR_STATE, "no blur predictions: ekf_predict with templates present computes them");
    if (N) HIPCHK(hipMemcpyAsync(out, d_hb, (size_t)N * 2 * sizeof(T), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    return EKF_OK;
  }
  int find_matches(double threshold, void* z, unsigned char* found, float* score) override {
    HIPCHK(hipSetDevice(device));
    if (!have_frame) FAIL(EKF_ERR_STATE, "ekf_find_matches needs ekf_set_frame");
    if (!have_meas) FAIL(EKF_ERR_STATE, "ekf_find_matches needs the predictions of ekf_predict / ekf_measure");
    if (N == 0) return EKF_OK;
    // (sharded filter: the frame, the templates and -- after ekf_predict's "reassemble H" -- h are replicated, the 2x2 St
    // blocks are owner-computed and all-gathered by ensure_sd(), a COLLECTIVE; every rank then searches every feature:
    // the matching templates the search rewrites stay identical on every rank, and nothing else is exchanged)
    int rc = ensure_sd();
    if (rc) return rc;
    {
      Scope sc(this, KID_MISC);
      k_ncc_search<T><<<N, 256, 0, stream>>>(d_frame, frame_w, frame_h, d_h, d_Sd, d_flags, cfg.window_size,
                                             float(cfg.sigma_size), float(threshold), d_mpatch[cur_patch], d_zm, d_found,
                                             d_score);
    }
    HIPCHK(hipGetLastError());
    // the three results through the pinned bounce buffer: one synchronisation
    if (z) { rc = rb_add(z, d_zm, (size_t)N * 2 * sizeof(T)); if (rc) return rc; }
    if (found) { rc = rb_add(found, d_found, (size_t)N); if (rc) return rc; }
    if (score) { rc = rb_add(score, d_score, (size_t)N * sizeof(float)); if (rc) return rc; }
    return rb_finish(false);
  }

  // zero everything of buffer `b` outside the live n x n (up to what was ever written there)
  int zero_border(int b, int n_live) {
    const int ext = extent[b];
    if (ext > n_live) {
      // rows [n_live, ext): full width; rows [0, n_live): columns [n_live, ext)
      HIPCHK(hipMemset2DAsync(d_S[b] + (size_t)n_live * ld, (size_t)ld * sizeof(T), 0, (size_t)ext * sizeof(T),
                              ext - n_live, stream));
      HIPCHK(hipMemset2DAsync(d_S[b] + n_live, (size_t)ld * sizeof(T), 0, (size_t)(ext - n_live) * sizeof(T),
                              n_live, stream));
    }
    extent[b] = n_live;
    return EKF_OK;
  }

  // ---- a13 / a14: one out-of-place pass for a set of removals and conversions -------------
  int ensure_arch_idx(size_t count) {
    if (count <= arch_idx_cap) return EKF_OK;
    if (d_arch_idx) HIPCHK(hipFree(d_arch_idx));
    d_arch_idx = nullptr;
    arch_idx_cap = std::max<size_t>(2 * count, 256);
    HIPCHK(hipMalloc(&d_arch_idx, arch_idx_cap * sizeof(int)));
    return EKF_OK;
  }

  // vR.cpp:394-404: a removed XYZ feature that was found more than 5 times is kept (XYZ_pos, cov_4_delete, real_index)
  int archive_removed(const std::vector<char>& rm) {
    std::vector<int> ap, ar;
    for (int i = 0; i < N; ++i)
      if (rm[i] && coding[i] != 0 && n_find[i] > 5) { ap.push_back(pos[i]); ar.push_back(real_index[i]); }
    if (ap.empty()) return EKF_OK;
    // sharded: the 3 x 3 block of a removed feature is valid on its owner only -- gathered first, so that every rank
    // archives the same 12 scalars (the list above is the same on every rank: host metadata is replicated)
    int rc = shard_sync_diag_blocks();
    diag_synced = false;                                     // the removal that follows compacts Sigma
    if (rc) return rc;
    const size_t have = arch_real.size(), need = have + ap.size();
    if (need > arch_cap) {
      const size_t cap = std::max<size_t>(2 * need, 256);
      T* nbuf = nullptr;
      HIPCHK(hipMalloc(&nbuf, cap * 12 * sizeof(T)));
      hipError_t e1 = have ? hipMemcpyAsync(nbuf, d_archive, have * 12 * sizeof(T), hipMemcpyDeviceToDevice, stream) : hipSuccess;
      if (e1 == hipSuccess) e1 = hipStreamSynchronize(stream);
      if (e1 != hipSuccess) { hipFree(nbuf); HIPCHK(e1); }
      if (d_archive) HIPCHK(hipFree(d_archive));
      d_archive = nbuf;
      arch_cap = cap;
    }
    rc = ensure_arch_idx(ap.size());
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(d_arch_idx, ap.data(), ap.size() * sizeof(int), hipMemcpyHostToDevice, stream));
    k_archive_points<T><<<((int)ap.size() * 12 + 255) / 256, 256, 0, stream>>>(mu(), S(), ld, d_arch_idx, (int)ap.size(),
                                                                             d_archive + have * 12);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(stream));          // `ap` goes out of scope
    arch_real.insert(arch_real.end(), ar.begin(), ar.end());
    return EKF_OK;
  }

  int compact(const std::vector<char>& rm, const std::vector<char>& cv) {
    { int rca = archive_removed(rm); if (rca) return rca; }
    std::vector<int> msrc, mconv, npos, ncoding, nreal, nfind;
    msrc.reserve(n); mconv.reserve(n);
    for (int i = 0; i < camera_dim; ++i) { msrc.push_back(i); mconv.push_back(-1); }
    for (int i = 0; i < N; ++i) {
      if (rm[i]) continue;
      const int fs = coding[i] ? 3 : 6;
      npos.push_back((int)msrc.size());
      nreal.push_back(real_index[i]);
      nfind.push_back(n_find[i]);
      if (cv[i]) {
        for (int e = 0; e < 3; ++e) { msrc.push_back(pos[i]); mconv.push_back(i * 3 + e); }
        ncoding.push_back(1);
      } else {
        for (int e = 0; e < fs; ++e) { msrc.push_back(pos[i] + e); mconv.push_back(-1); }
        ncoding.push_back(coding[i]);
      }
    }
    const int n_new = (int)msrc.size();
    HIPCHK(hipMemcpyAsync(d_map_src, msrc.data(), n_new * sizeof(int), hipMemcpyHostToDevice, stream));
    HIPCHK(hipMemcpyAsync(d_map_conv, mconv.data(), n_new * sizeof(int), hipMemcpyHostToDevice, stream));
    const int dst = 1 - cur, dmu = 1 - cur_mu;
    {
      Scope sc(this, KID_COMPACT);
      dim3 grid((n_new + 1023) / 1024, (n_new + kCompactRows - 1) / kCompactRows);   // four columns per lane, kCompactRows rows per workgroup
      k_compact_transform<T><<<grid, 256, 0, stream>>>(S(), d_S[dst], ld, n_new, d_map_src, d_map_conv, d_Jy);
      k_compact_mu<T><<<(n_new + 255) / 256, 256, 0, stream>>>(mu(), d_mu[dmu], n_new, d_map_src, d_map_conv,
                                                              d_Yxyz);
    }
    HIPCHK(hipGetLastError());
    std::vector<int> keep;
    if (d_patch[0]) {                            // the templates follow their features (patches.erase, vR.cpp:1298)
      for (int i = 0; i < N; ++i)
        if (!rm[i]) keep.push_back(i);
      if (!keep.empty() && (int)keep.size() != N) {
        const int w2 = cfg.window_size * cfg.window_size;
        HIPCHK(hipMemcpyAsync(d_keep, keep.data(), keep.size() * sizeof(int), hipMemcpyHostToDevice, stream));
        k_gather_patches<<<(int)keep.size(), 256, 0, stream>>>(d_patch[cur_patch], d_patch[1 - cur_patch], d_keep, w2);
        k_gather_patches<<<(int)keep.size(), 256, 0, stream>>>(d_mpatch[cur_patch], d_mpatch[1 - cur_patch], d_keep, w2);
        cur_patch = 1 - cur_patch;
      }
    }
    extent[dst] = std::max(extent[dst], n_new);
    int rc = zero_border(dst, n_new);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(stream));        // host map vectors go out of scope
    cur = dst; cur_mu = dmu;
    pos.swap(npos); coding.swap(ncoding);
    real_index.swap(nreal); n_find.swap(nfind);
    N = (int)pos.size();
    n = n_new;
    shard_after_compact(rm);
    layout_dirty = true;
    have_meas = false;
    return EKF_OK;
  }

  int remove_features(const int* idx, int count) override {
    HIPCHK(hipSetDevice(device));
    if (count <= 0) return EKF_OK;
    std::vector<char> rm(N, 0), cv(N, 0);
    for (int k = 0; k < count; ++k) {
      if (idx[k] < 0 || idx[k] >= N) FAIL(EKF_ERR_ARG, "feature index out of range");
      rm[idx[k]] = 1;
    }
    return compact(rm, cv);
  }

  int convert(int index, bool all) override {
    HIPCHK(hipSetDevice(device));
    if (sh_on) return shard_convert(index, all);
    if (!all && (index < 0 || index >= N)) { err = "feature index out of range"; return -EKF_ERR_ARG; }
    if (N == 0) return 0;
    int rc = sync_layout();
    if (rc) return -rc;
    k_linearity<T><<<(N + 127) / 128, 128, 0, stream>>>(mu(), S(), ld, d_pos, d_coding, N, d_cflag, d_Jy, d_Yxyz, 0);
    std::vector<unsigned char> fl(N);
    if (rb_add(fl.data(), d_cflag, N) != EKF_OK || rb_finish(false) != EKF_OK) { err = "linearity flags D2H failed"; return -EKF_ERR_DEVICE; }
    std::vector<char> rm(N, 0), cv(N, 0);
    int cnt = 0;
    for (int i = 0; i < N; ++i) {
      if (!all && i != index) continue;
      if (fl[i] && coding[i] == 0) { cv[i] = 1; ++cnt; }
    }
    if (cnt == 0) return 0;
    rc = compact(rm, cv);
    if (rc) return -rc;
    return cnt;
  }

  // ---- a1-a6 predict --------------------------------------------------------------------
  int launch_measure() {
    int rc = sync_layout();
    if (rc) return rc;
    if (N > 0) {
      Scope sc(this, KID_MEASURE);
      k_measure<T><<<(N + 63) / 64, 64, 0, stream>>>(mu(), d_pos, d_coding, 0, N, cam, d_h, d_Hc, d_Hf, d_flags);
    }
    HIPCHK(hipGetLastError());
    have_meas = true;
    have_sd = false;
    return EKF_OK;
  }

  int predict(const void* tc, const void* rc_, int vcontrol) override {
    HIPCHK(hipSetDevice(device));
    if (sh_on) return shard_predict(tc, rc_, vcontrol);
    MotionArgs a;
    a.dT = dT;
    const T* t = static_cast<const T*>(tc);
    const T* r = static_cast<const T*>(rc_);
    for (int i = 0; i < 3; ++i) { a.t_ctl[i] = t ? double(t[i]) : 0.0; a.r_ctl[i] = r ? double(r[i]) : 0.0; }
    for (int i = 0; i < 6; ++i) a.vdiag[i] = vcontrol ? vmax[i] : double(T(vmax[i]) * T(2));   // vR.cpp:202
    if (opt_fused && !opt_streaming && !(opt_feature_noise > 0.0) && N > 0 && !prof_on(KID_PREDICT_CAMERA) &&
        !prof_on(KID_PROPAGATE_STRIPS) && !prof_on(KID_MEASURE)) {
      // camera step, strip congruence and the per-feature h / H as ONE launch (k_predict_fused): same arithmetic,
      // two launches less per frame
      int rc = sync_layout();
      if (rc) return rc;
      const int nstrip = (2 * n + 255) / 256;
      k_predict_fused<T><<<nstrip + (N + 255) / 256, 256, 0, stream>>>(mu(), d_scr, a, S(), ld, n, nstrip, d_pos, d_coding, N,
                                                                      cam, d_h, d_Hc, d_Hf, d_flags, d_status + 8, d_status);
      HIPCHK(hipGetLastError());
      have_motion = true;
      have_update = false;
      have_meas = true;
      have_sd = false;
      return launch_blur();
    }
    {
      Scope sc(this, KID_PREDICT_CAMERA);
      k_predict_camera<T><<<1, 64, 0, stream>>>(mu(), d_scr, a);
    }
    have_motion = true;
    if (opt_streaming) {
      const int dst = 1 - cur;
      {
        Scope sc(this, KID_PROPAGATE_STREAMING);
        k_propagate_streaming<T><<<n, 256, 0, stream>>>(S(), d_S[dst], ld, n, d_scr + SCR_FT, d_scr + SCR_Q);
      }
      extent[dst] = std::max(extent[dst], n);
      int rc2 = zero_border(dst, n);
      if (rc2) return rc2;
      cur = dst;
    } else {
      Scope sc(this, KID_PROPAGATE_STRIPS);
      k_strip_congruence<T, 13><<<(2 * n + 255) / 256, 256, 0, stream>>>(S(), ld, n, 0, d_scr + SCR_FT, d_scr + SCR_Q);
    }
    if (opt_feature_noise > 0.0 && n > camera_dim)
      k_inflate_diagonal<T><<<(n - camera_dim + 255) / 256, 256, 0, stream>>>(S(), ld, camera_dim, n, T(opt_feature_noise));
    HIPCHK(hipGetLastError());
    have_update = false;
    int rcm = launch_measure();
    if (rcm) return rcm;
    return launch_blur();
  }

  int measure() override {
    HIPCHK(hipSetDevice(device));
    return launch_measure();
  }

  bool have_motion = false;
  int motion_jacobian(void* Ft, void* Q) override {
    HIPCHK(hipSetDevice(device));
    if (!have_motion) FAIL(EKF_ERR_STATE, "no motion Jacobian: call ekf_predict first");
    T buf[2 * 169];
    { int rc = rb_add(buf, d_scr + SCR_FT, sizeof(buf)); if (rc) return rc; rc = rb_finish(false); if (rc) return rc; }   // SCR_FT, SCR_Q adjacent
    for (int b = 0; b < 2; ++b) {
      T* o = static_cast<T*>(b ? Q : Ft);
      if (!o) continue;
      for (int r = 0; r < 13; ++r)
        for (int c = 0; c < 13; ++c) o[c * 13 + r] = buf[b * 169 + r * 13 + c];
    }
    return EKF_OK;
  }

  // 2x2 St blocks on demand (they need the propagated Sigma and the current H; nothing on the device
  // consumes them except the ellipse / RANSAC kernels)
  int ensure_sd() {
    if (have_sd || N == 0) return EKF_OK;
    if (sh_on) {
      // a 2x2 block needs the rows of Sigma of its feature: every rank evaluates the features it owns, then the blocks
      // are all-gathered (4 scalars per feature)
      const int f0 = own_f0(), f1 = own_f1();
      if (f1 > f0)
        k_measure_sd<T><<<(16 * (f1 - f0) + 255) / 256, 256, 0, stream>>>(S(), ld, d_pos, d_coding, f0, f1, T(sigma_pixel_2),
                                                                        d_Hc, d_Hf, d_Sd);
      HIPCHK(hipGetLastError());
      int rc = gather_sd(feature_tab(), nullptr);
      if (rc) return rc;
      have_sd = true;
      return EKF_OK;
    }
    k_measure_sd<T><<<(16 * N + 255) / 256, 256, 0, stream>>>(S(), ld, d_pos, d_coding, 0, N, T(sigma_pixel_2), d_Hc, d_Hf,
                                                             d_Sd);
    HIPCHK(hipGetLastError());
    have_sd = true;
    return EKF_OK;
  }

  int get_predictions(void* h, unsigned char* vis, unsigned char* rem, void* s2, void* hc, void* hf) override {
    HIPCHK(hipSetDevice(device));
    if (!have_meas) FAIL(EKF_ERR_STATE, "no predictions: call ekf_predict / ekf_measure first");
    if (N == 0) return EKF_OK;
    // h, flags and the 2x2 blocks come back through ONE host-mapped pinned buffer the device writes (k_pack_predictions):
    // one launch + one synchronisation per frame for the drop-in caller, instead of three staged copies
    if (!h_pred) {
      const size_t bytes = (size_t)capN * (6 * sizeof(T) + 1) + 64;
      HIPCHK(hipHostMalloc(&h_pred, bytes, hipHostMallocDefault));
    }
    T* ph = reinterpret_cast<T*>(h_pred);
    T* psd = ph + (size_t)2 * capN;
    unsigned char* pfl = reinterpret_cast<unsigned char*>(psd + (size_t)4 * capN);
    std::vector<T> vhc, vhf;
    if (s2 && !have_sd && !sh_on) {              // the 2x2 blocks and the packing in ONE launch
      k_measure_sd<T><<<(16 * N + 255) / 256, 256, 0, stream>>>(S(), ld, d_pos, d_coding, 0, N, T(sigma_pixel_2), d_Hc, d_Hf,
                                                               d_Sd, nullptr, 0, d_h, d_flags, ph, psd, pfl);
      have_sd = true;
    } else {
      if (s2) { int rcs = ensure_sd(); if (rcs) return rcs; }
      k_pack_predictions<T><<<(4 * N + 255) / 256, 256, 0, stream>>>(d_h, s2 ? d_Sd : static_cast<const T*>(nullptr), d_flags, N,
                                                                    ph, psd, pfl);
    }
    HIPCHK(hipGetLastError());
    if (hc) { vhc.resize((size_t)N * 14); HIPCHK(hipMemcpyAsync(vhc.data(), d_Hc, vhc.size() * sizeof(T), hipMemcpyDeviceToHost, stream)); }
    if (hf) { vhf.resize((size_t)N * 12); HIPCHK(hipMemcpyAsync(vhf.data(), d_Hf, vhf.size() * sizeof(T), hipMemcpyDeviceToHost, stream)); }
    HIPCHK(hipStreamSynchronize(stream));
    if (h) memcpy(h, ph, (size_t)N * 2 * sizeof(T));
    for (int i = 0; i < N; ++i) { if (vis) vis[i] = pfl[i] & 1; if (rem) rem[i] = (pfl[i] >> 1) & 1; }
    if (s2) {                                    // row-major 2x2 -> column-major
      T* o = static_cast<T*>(s2);
      const T* sd = psd;
      for (int i = 0; i < N; ++i) { o[4 * i] = sd[4 * i]; o[4 * i + 1] = sd[4 * i + 2]; o[4 * i + 2] = sd[4 * i + 1]; o[4 * i + 3] = sd[4 * i + 3]; }
    }
    if (hc) { T* o = static_cast<T*>(hc); for (int i = 0; i < N; ++i) for (int a = 0; a < 2; ++a) for (int c = 0; c < 7; ++c) o[14 * i + c * 2 + a] = vhc[14 * i + a * 7 + c]; }
    if (hf) { T* o = static_cast<T*>(hf); for (int i = 0; i < N; ++i) for (int a = 0; a < 2; ++a) for (int c = 0; c < 6; ++c) o[12 * i + c * 2 + a] = vhf[12 * i + a * 6 + c]; }
    return EKF_OK;
  }

  // ---- dense tile GEMM dispatch -----------------------------------------------------------
  // C[rows x cols] = beta C + alpha A op(B); rows, cols multiples of the tile.
  // TM x TN: MFMA tile shape (64 or 128 each); the VALU path always uses 64 x 64.
  template <int ROLE, bool BT, int TM = 128, int TN = 128>
  void gemm(const T* A, int lda, const T* B, int ldb, T* C, int ldc, int rows, int cols, int K, T alpha, T beta,
            int tri, int row_off, int col_off, int ktri, int ktile_off = 0, hipStream_t st = nullptr,
            const int* tile_list = nullptr, int ntiles = 0, int zrow = 0, int zcol_end = 0, void* img = nullptr,
            int img_nkc = 0, int img_c0 = 0) {
    GemmArgs g{A, lda, B, ldb, C, ldc, K, double(alpha), double(beta), tri, row_off, col_off, ktri, ktile_off,
               nullptr, 0, nullptr, zrow, zcol_end, (ROLE == ROLE_DOWNDATE) ? 1 : 0};
    g.img = img; g.img_nkc = img_nkc; g.img_c0 = img_c0;
    if (!st) st = stream;
    const bool mf = kIsF32 && opt_mfma;
    dim3 grid(cols / (mf ? TN : 64), rows / (mf ? TM : 64));
    if (tile_list && counter_next + 8 <= kQueueCounters) {
      g.tile_map = tile_list;
      g.ntiles = ntiles;
      g.counter = d_counters + counter_next;
      counter_next += 8;
      // persistent grid: two workgroups per CU the stream may use
      const bool side = (st == stream_b);
      int wgs = 2 * (side ? (num_cus - reserved_cus) : num_cus);
      if (ROLE == ROLE_SOLVE && solve_one_per_cu_now) wgs = num_cus;      // (see the last chunk's solve in update())
      grid = dim3(std::min(ntiles, wgs), 1);
    }
    if constexpr (kIsF32) {
      if (opt_mfma) {
        if constexpr (ROLE == ROLE_SOLVE) {
          ++launch_cnt[solve_s2_now ? EKF_LAUNCH_SOLVE_TWO_GROUPS : EKF_LAUNCH_SOLVE];
          if (solve_s2_now) {
            k_gemm_mfma<ROLE, BT, TM, TN, true><<<grid, 512, 0, st>>>(g);
            return;
          }
        }
        k_gemm_mfma<ROLE, BT, TM, TN><<<grid, 256, 0, st>>>(g);
        return;
      }
    }
    if constexpr (!kIsF32) {
      if (opt_mfma) {                            // fp64 matrix pipe, same 64 x 64 tile contract as the VALU kernel
        k_gemm_mfma_f64<ROLE, BT><<<grid, 256, 0, st>>>(g);
        return;
      }
    }
    k_gemm_valu<T, ROLE, BT><<<grid, 256, 0, st>>>(g);
  }

  // Panel of a chain step: P <- P Linv_jj^T in place (vrows rows, block nb).
  void launch_panel(T* P, const T* Dj, int vrows, hipStream_t st) {
    const int nb = NB();
    if constexpr (kIsF32) {
      if (opt_mfma && nb == 128 && opt_panel_direct) {
        k_panel_direct<<<(vrows + 63) / 64, 256, 0, st>>>(P, ldy, Dj, vrows);
        return;
      }
    }
    if constexpr (!kIsF32) {
      if (opt_mfma && nb == 64 && opt_panel_direct) {
        k_panel_direct_f64<<<(vrows + 63) / 64, 256, 0, st>>>(P, ldy, Dj, vrows);
        return;
      }
    }
    gemm<ROLE_PANEL, false, 64, 128>(P, ldy, Dj, nb, P, ldy, vrows, nb, nb, T(1), T(0), 0, 0, 0, 0, 0, st);
  }

  // Work lists for the queued GEMMs: (1) lower-triangular tiles of an nt x nt grid in 8x8
  // super-tiles; (2) the ntr x ntc tiles of the triangular solve, heaviest (largest bj) first.
  int ensure_tilemap(int nt, int ntr, int ntc) {
    if (tilemap_nt == nt && tilemap_ntc == ntc) return EKF_OK;
    std::vector<int> tm;
    const int SB = 8;
    const int ns = (nt + SB - 1) / SB;
    for (int si = 0; si < ns; ++si)
      for (int sj = 0; sj <= si; ++sj)
        for (int i = si * SB; i < std::min(nt, (si + 1) * SB); ++i)
          for (int j = sj * SB; j < std::min(nt, (sj + 1) * SB); ++j)
            if (j <= i) { tm.push_back(i); tm.push_back(j); }
    tri_count = (int)tm.size() / 2;
    solve_off = (int)tm.size();
    for (int j = ntc - 1; j >= 0; --j)
      for (int i = 0; i < ntr; ++i) { tm.push_back(i); tm.push_back(j); }
    solve64_off = (int)tm.size();                  // the same list for 64-row tiles (narrow chunks: more workgroups)
    for (int j = ntc - 1; j >= 0; --j)
      for (int i = 0; i < 2 * ntr; ++i) { tm.push_back(i); tm.push_back(j); }
    solve6464_off = (int)tm.size();                // ... and for 64 x 64 tiles (small maps)
    for (int j = 2 * ntc - 1; j >= 0; --j)
      for (int i = 0; i < 2 * ntr; ++i) { tm.push_back(i); tm.push_back(j); }
    tri64_off = (int)tm.size();                    // lower-triangular 64 x 64 tiles (small maps: 4x the workgroups)
    {
      const int nt64 = 2 * nt, ns64 = (nt64 + SB - 1) / SB;
      for (int si = 0; si < ns64; ++si)
        for (int sj = 0; sj <= si; ++sj)
          for (int i = si * SB; i < std::min(nt64, (si + 1) * SB); ++i)
            for (int j = sj * SB; j < std::min(nt64, (sj + 1) * SB); ++j)
              if (j <= i) { tm.push_back(i); tm.push_back(j); }
      tri64_count = ((int)tm.size() - tri64_off) / 2;
    }
    tri6_off = (int)tm.size();                     // k_syrk_bf16x6: the diagonal tiles first (their element-wise epilogue is the
    for (int i = 0; i < nt; ++i) { tm.push_back(i); tm.push_back(i); }   // longest: not in the tail of the launch), then the rest
    for (int t = 0; t < tri_count; ++t)
      if (tm[2 * t] != tm[2 * t + 1]) { tm.push_back(tm[2 * t]); tm.push_back(tm[2 * t + 1]); }
    trih_off = (int)tm.size();
    {
      const int ns_ = (opt_split_tail < 0) ? std::min(3 * num_cus / 2, tri_count / 3) : std::min(opt_split_tail, tri_count);
      for (int t = 0; t < tri_count - ns_; ++t) { tm.push_back(tm[2 * t]); tm.push_back(tm[2 * t + 1]); }
      for (int t = tri_count - ns_; t < tri_count; ++t)
        for (int h = 0; h < 2; ++h) { tm.push_back((2 * tm[2 * t] + h) | kHalfTile); tm.push_back(tm[2 * t + 1]); }
      trih_count = ((int)tm.size() - trih_off) / 2;
    }
    HIPCHK(hipStreamSynchronize(stream));
    HIPCHK(hipStreamSynchronize(stream_b));
    if (d_tilemap) HIPCHK(hipFree(d_tilemap));
    d_tilemap = nullptr;
    HIPCHK(hipMalloc(&d_tilemap, tm.size() * sizeof(int)));
    HIPCHK(hipMemcpy(d_tilemap, tm.data(), tm.size() * sizeof(int), hipMemcpyHostToDevice));
    tilemap_nt = nt;
    tilemap_ntc = ntc;
    return EKF_OK;
  }

  // Chunk ends (in block steps) of the factorisation.  One chunk = the plain algorithm (the strip is the
  // whole inverse); several chunks when the chain is long enough to be worth hiding.
  int plan_chunks(int nsteps, int* cend, bool sharded = false) const {
    const bool pipe = (opt_pipeline < 0) ? (nsteps >= 8) : (opt_pipeline != 0);
    if (!pipe || nsteps < 2 || !stream_b) { cend[0] = nsteps; return 1; }
    if (env_nchunks > 0 && env_chunks[env_nchunks - 1] == nsteps) {
      for (int g = 0; g < env_nchunks; ++g) cend[g] = env_chunks[g];
      return env_nchunks;
    }
    if (opt_pipeline < 2) {
      // default: three chunks ending at 3/16, 7/16 (8/16 through round 3) and 1 of the chain: the first chunk is
      // exposed, so it is short; every further chunk re-reads Sigma once in its downdate, so there are few; the chain and
      // the second stream end together
      // Long chains (>= 32 steps; round 3, tools/knob_ab.py: 32 steps 6/16 -> 4/14: -1 %, 63 steps 12/32 -> 5/25: -2.6 %): the chain
      // is hidden there whatever the plan, so the first two chunks shrink in proportion -- less exposed start-up, and a
      // wider last chunk, whose downdate has every CU
      // round 4 (EKF_OPT_W_RECOMPUTE: no W update, the chain as fast as the second stream): 3 / 7 / 16 -- the last chunk's
      // downdate has every CU and the largest K (tools/knob_ab.py: 1.169 ms against 1.187 with 3 / 8 / 16, 1.216 with 3 / 6)
      // round 5 (EKF_OPT_SPLIT_BF16, the downdate 1.5 x faster): the second stream has slack beside the chain, so one more
      // chunk pays -- the exposed first chunk and the exposed last chunk both get shorter (tools/knob_ab.py, ms per step):
      //   16 steps (N = 1000): 1.063 with 3 / 7 / 16; four chunks: a plateau of 1.01-1.04 over {2, 3} x {5 .. 8} x {10 .. 13} (a sweep of 32
      //   plans), 3 / 7 / 11 / 16 at its low end (1.013-1.018 against 1.026-1.034 for 2 / 6 / 11 / 16); five chunks 1.05-1.11
      //   32 steps (N = 2000): 5.28 with 4 / 13 / 32, 5.05 with 4 / 12 / 22 / 32, 5.09 with five chunks
      //   63 steps (N = 4000): 35.96 with 5 / 25 / 63, 34.97 with 4 / 16 / 36 / 63, 34.27 with 4 / 14 / 30 / 46 / 63
      const bool split = kIsF32 && opt_split_bf16 && opt_mfma;
      if (split && nsteps >= 12) {
        // (the sharded step, whose second stream also carries the gathers in series: 2 / 6 / 11 / 16 -- world 1 with forced
        // collectives 1.305 ms against 1.354 with 3 / 7 / 11 / 16)
        static const double f4p[4] = {3.0 / 16, 7.0 / 16, 11.0 / 16, 1.0}, f4s[4] = {2.0 / 16, 6.0 / 16, 11.0 / 16, 1.0};
        const double* f4 = sharded ? f4s : f4p;
        static const double f4l[4] = {4.0 / 32, 12.0 / 32, 22.0 / 32, 1.0},
                            f5[5] = {4.0 / 63, 14.0 / 63, 30.0 / 63, 46.0 / 63, 1.0};
        const int ng = nsteps >= 48 ? 5 : 4;
        const double* fr = nsteps >= 48 ? f5 : (nsteps >= 32 ? f4l : f4);
        int k = 0, prev = 0;
        for (int g = 0; g < ng; ++g) {
          int e = (g == ng - 1) ? nsteps : (int)(fr[g] * nsteps + 0.5);
          if (e > prev) { cend[k++] = e; prev = e; }
        }
        return k;
      }
      static const int kEnd16[3] = {3, 7, 16};
      int k = 0, prev = 0;
      for (int g = 0; g < 3; ++g) {
        int e = (g == 2) ? nsteps : (nsteps * kEnd16[g] + 8) / 16;
        if (nsteps >= 32 && g == 0) e = 3 + (nsteps - 16) / 16;
        if (nsteps >= 32 && g == 1) e = (int)(2.0 + 0.36 * nsteps + 0.5);
        if (e > prev) { cend[k++] = e; prev = e; }
      }
      return k;
    }
    const int want = std::min(std::min(opt_pipeline, 8), nsteps);
    int k = 0, prev = 0;
    for (int g = 0; g < want; ++g) {
      int e = (int)(((long long)nsteps * (g + 1) + want - 1) / want);
      if (g + 1 == want) e = nsteps;
      if (e > prev) { cend[k++] = e; prev = e; }
    }
    return k;
  }

  // W, S (and nu) for a measured set already resident in d_midx / d_z.
  int build_innovation(int M, int plane, bool with_nu, int* m_out, int* m_pad_out, const ChunkTab* tab = nullptr,
                       int strip_rows = 0, bool w_only = false) {
    const int nb = NB();
    const int m = 2 * M + (plane ? 3 : 0);
    const int m_pad = round_up(m, nb);
    const int npad_live = round_up(n, nb);
    T* nu_row = d_W + (size_t)ldy * npad_live;
    // pad rows of W (n..npad_live) and the nu block must be zero; no kernel writes them except row
    // npad_live (nu), so they are cleared only when the layout changed
    if (w_zeroed_n != n) {
      HIPCHK(hipMemsetAsync(d_W + (size_t)n * ldy, 0, (size_t)(npad_live - n + nb) * ldy * sizeof(T), stream));
      w_zeroed_n = n;
    }
    const T* zp = cur_z ? cur_z : d_z;
    const int* ip = cur_midx ? cur_midx : d_midx;
    if (with_nu) counter_next = 0;
    // small problems are latency-bound: fewer rows / features per workgroup so that the grid fills the chip
    const bool small = (size_t)n * m_pad < ((size_t)1 << 22);
    {
      // W = Sigma H^T; with_nu: one more slab of workgroups forms nu = z - h and clears the work-queue heads
      // (k_innovation folded into this launch)
      Scope sc(this, KID_SIGMA_HT);
      const int extra = with_nu ? 1 : 0;
      const T* zq = with_nu ? zp : nullptr;
      T* nuq = with_nu ? nu_row : nullptr;
      if (small) {
        constexpr int RB = 4;
        dim3 grid((m_pad / 2 + 255) / 256, (n + RB - 1) / RB + extra);
        k_sigma_ht<T, RB><<<grid, 256, 0, stream>>>(S(), ld, n, d_Hc, d_Hf, d_pos, d_coding, ip, M, plane, d_W, ldy,
                                                  m_pad, 0, n, N, zq, d_h, mu(), nuq, d_counters, d_status, d_scr + SCR_QOLD);
      } else if constexpr (kIsF32) {
        // 128 slots x 8 rows per workgroup, the row segments staged through LDS (runs of neighbouring inverse-depth
        // features; other stretches of the list take k_sigma_ht's path inside the same launch): bit-identical sums
        constexpr int RB = 8;
        dim3 grid((m_pad / 2 + 127) / 128, (n + RB - 1) / RB + extra);
        k_sigma_ht_fast<RB><<<grid, 256, 0, stream>>>(S(), ld, n, d_Hc, d_Hf, d_pos, d_coding, ip, M, plane, d_W, ldy, m_pad, N,
                                                      0, 0, zq, d_h, mu(), nuq, d_counters, d_status, d_scr + SCR_QOLD);
      } else {
        constexpr int RB = 32;
        dim3 grid((m_pad / 2 + 255) / 256, (n + RB - 1) / RB + extra);
        k_sigma_ht<T, RB><<<grid, 256, 0, stream>>>(S(), ld, n, d_Hc, d_Hf, d_pos, d_coding, ip, M, plane, d_W, ldy,
                                                  m_pad, 0, n, N, zq, d_h, mu(), nuq, d_counters, d_status, d_scr + SCR_QOLD);
      }
    }
    if (!w_only) {                      // (the 1-point RANSAC reads W only)
      Scope sc(this, KID_INNOVATION_COV);
      T* zid = tab ? d_Y + (size_t)m_pad * ldy : nullptr;
      const ChunkTab ct = tab ? *tab : ChunkTab{0, {}};
      if (small) {
        constexpr int KB = 1;
        dim3 grid((m_pad + 255) / 256, std::max(1, (M + KB - 1) / KB) + (m_pad - 2 * M + 7) / 8);
        k_innovation_cov<T, KB><<<grid, 256, 0, stream>>>(d_W, ldy, d_Hc, d_Hf, d_pos, d_coding, ip, M, plane,
                                                        T(sigma_pixel_2), T(0.00001), d_Y, m_pad, 0, M, zid, N, ct, strip_rows);
      } else {
        constexpr int KB = 8;
        dim3 grid((m_pad + 255) / 256, std::max(1, (M + KB - 1) / KB) + (m_pad - 2 * M + 7) / 8);
        k_innovation_cov<T, KB><<<grid, 256, 0, stream>>>(d_W, ldy, d_Hc, d_Hf, d_pos, d_coding, ip, M, plane,
                                                        T(sigma_pixel_2), T(0.00001), d_Y, m_pad, 0, M, zid, N, ct, strip_rows);
      }
    }
    HIPCHK(hipGetLastError());
    *m_out = m;
    *m_pad_out = m_pad;
    return EKF_OK;
  }

  // ---- the chain as ONE look-ahead launch per column chunk (ekf_chain.hpp) ------------------------------------------
  bool chain_persistent_ok() const {
    return kIsF32 && opt_mfma && opt_chain_persistent && (size_t)2 * ldy * ldy * sizeof(T) < ((size_t)1 << 31);
  }
  // Task lists of every launch of the chunk plan (cached until the plan changes), the hand-over words, and this update's epoch.
  int chain_begin_update(int nblk, int nchunks, const int* cend) {
    bool same = chain_plan.nblk == nblk && chain_plan.nchunks == nchunks;
    for (int g = 0; same && g < nchunks; ++g) same = chain_plan.cend[g] == cend[g];
    if (!same) {
      build_chain_plan(chain_plan, nblk, nchunks, cend);
      if (!validate_chain_plan(chain_plan)) {
        chain_plan.nblk = 0;
        FAIL(EKF_ERR_DEVICE, "internal: the task lists of the persistent chain do not complete with one worker per list");
      }
      HIPCHK(hipStreamSynchronize(stream));
      if (stream_b) HIPCHK(hipStreamSynchronize(stream_b));
      if (d_chain_tasks) HIPCHK(hipFree(d_chain_tasks));
      d_chain_tasks = nullptr;
      HIPCHK(hipMalloc(&d_chain_tasks, chain_plan.tasks.size() * sizeof(ChainTask)));
      HIPCHK(hipMemcpy(d_chain_tasks, chain_plan.tasks.data(), chain_plan.tasks.size() * sizeof(ChainTask), hipMemcpyHostToDevice));
      if (chain_plan.nflags > chain_flags_cap) {
        if (d_chain_flags) HIPCHK(hipFree(d_chain_flags));
        d_chain_flags = nullptr;
        chain_flags_cap = chain_plan.nflags + 1024;
        HIPCHK(hipMalloc(&d_chain_flags, (size_t)chain_flags_cap * sizeof(unsigned)));
        chain_epoch = 0;
      }
    }
    if (chain_epoch == 0 || chain_epoch >= (1u << (32 - kChainEpochShift)) - 2) {   // fresh words, or the epoch would wrap
      HIPCHK(hipMemsetAsync(d_chain_flags, 0, (size_t)chain_flags_cap * sizeof(unsigned), stream));
      chain_epoch = 0;
    }
    ++chain_epoch;
    if (d_chain_trace) HIPCHK(hipMemsetAsync(d_chain_trace, 0, 8 * sizeof(unsigned), stream));
    return EKF_OK;
  }
  // Launch gi of the plan: chunk gi's factor, panels and all but its last trailing update (+ the one chunk gi - 1 left).
  // `whole_chip`: nothing else is running (chunk 0): one workgroup per CU; else the CUs the second stream leaves alone.
  int chain_launch(int gi, int m, bool whole_chip, hipStream_t sc_) {
    if (counter_next + 8 > kQueueCounters) FAIL(EKF_ERR_DEVICE, "internal: out of work-queue counters");
    const ChainPlan::Launch& L = chain_plan.launch[gi];
    ChainArgs a{};
    if constexpr (kIsF32) { a.Y = d_Y; a.Dinv = d_Dinv; }
    a.ldy = ldy;
    a.y_bytes = (unsigned)((size_t)2 * ldy * ldy * sizeof(T));
    a.dinv_bytes = (unsigned)((size_t)(ldy / 64) * 128 * 128 * sizeof(T));
    a.status = d_status;
    a.m = m;
    a.s0 = gi ? chain_plan.cend[gi - 1] : 0; a.s1 = chain_plan.cend[gi]; a.deferred = gi > 0 ? 1 : 0;
    a.nblk = chain_plan.nblk; a.rb = chain_plan.rb;
    a.bulk = d_chain_tasks + L.bulk_off; a.nbulk = L.nbulk;
    a.flags = d_chain_flags; a.abort_word = chain_plan.nflags - 1;
    a.epoch = chain_epoch << kChainEpochShift;
    a.counters = d_counters + counter_next;
    a.trace = d_chain_trace; a.trace_cap = kChainTraceCap;
    counter_next += 8;
    const int avail = whole_chip ? num_cus : std::max(2, reserved_cus > 0 ? reserved_cus : 32);
    const int grid = std::max(2, std::min(avail, 1 + L.nbulk));
    Scope sc(this, KID_CHOL_DIAG, sc_);
    ++launch_cnt[EKF_LAUNCH_CHAIN_PERSISTENT];
    if constexpr (kIsF32) k_chain_persistent<<<grid, 1024, kChainLds, sc_>>>(a);
    return EKF_OK;
  }

  // ---- trailing update of step j + diagonal factor of step j + 1 as one launch (k_trail_diag) ---------------------------
  bool trail_diag_ok() const {
    return kIsF32 && opt_mfma && opt_chain_fused_diag && (size_t)2 * ldy * ldy * sizeof(T) < ((size_t)1 << 31);
  }
  // per block step j of the chunk plan: the 128 x 128 blocks of its trailing update except (j + 1, j + 1) -- the S blocks
  // (I >= K > j) and the strip blocks of the step's chunk (row blocks nblk + t, t <= j - s0, columns K in (j, s1))
  int ensure_trail_diag_lists(int nblk, int nchunks, const int* cend) {
    bool same = td_nblk == nblk && td_nchunks == nchunks;
    for (int g = 0; same && g < nchunks; ++g) same = td_cend[g] == cend[g];
    if (same) return EKF_OK;
    std::vector<int> all;
    td_off.assign(nblk, 0);
    td_cnt.assign(nblk, 0);
    for (int g = 0; g < nchunks; ++g) {
      const int s0 = g ? cend[g - 1] : 0, s1 = cend[g];
      for (int j = s0; j < s1; ++j) {
        td_off[j] = (int)all.size();
        for (int K = j + 1; K < nblk; ++K) {
          for (int I = K; I < nblk; ++I) {
            if (I == j + 1) continue;                                   // (j + 1, j + 1): workgroup 0's
            all.push_back(I); all.push_back(K);
          }
          if (K < s1)
            for (int t = 0; t <= j - s0; ++t) { all.push_back(nblk + t); all.push_back(K); }
        }
        td_cnt[j] = ((int)all.size() - td_off[j]) / 2;
      }
    }
    HIPCHK(hipStreamSynchronize(stream));
    if (stream_b) HIPCHK(hipStreamSynchronize(stream_b));
    if (d_td_blocks) HIPCHK(hipFree(d_td_blocks));
    d_td_blocks = nullptr;
    HIPCHK(hipMalloc(&d_td_blocks, std::max<size_t>(all.size(), 2) * sizeof(int)));
    if (!all.empty()) HIPCHK(hipMemcpy(d_td_blocks, all.data(), all.size() * sizeof(int), hipMemcpyHostToDevice));
    td_nblk = nblk; td_nchunks = nchunks;
    for (int g = 0; g < nchunks; ++g) td_cend[g] = cend[g];
    return EKF_OK;
  }
  // true: launched (and the factor of step + 1 is done); false: this step keeps the separate launches
  bool launch_trail_diag(int step, int m, hipStream_t sc_) {
    if constexpr (kIsF32) {
      if (!trail_diag_ok() || NB() != 128 || td_nblk == 0 || step + 1 >= td_nblk || td_cnt[step] < td_min_blocks ||
          td_cnt[step] > td_max_blocks)
        return false;
      TrailDiagArgs a{};
      a.Y = d_Y; a.ldy = ldy; a.y_bytes = (unsigned)((size_t)2 * ldy * ldy * sizeof(T));
      a.Dinv = d_Dinv; a.dinv_bytes = (unsigned)((size_t)(ldy / 64) * 128 * 128 * sizeof(T));
      a.status = d_status; a.m = m; a.j = step;
      a.blocks = d_td_blocks + td_off[step]; a.nblocks = td_cnt[step]; a.do_diag = 1;
      Scope sc(this, KID_CHOL_TRAILING, sc_);
      ++launch_cnt[EKF_LAUNCH_CHAIN_TRAIL_DIAG];
      k_trail_diag<<<a.nblocks + 1, 1024, kChainLds, sc_>>>(a);
      chain_diag_ahead = step + 1;
      return true;
    }
    return false;
  }

  // Block steps [step0, step1) of the serial chain of chunk [c0, c1) on stream sc_: diagonal factor, panel (rows
  // below the block + the chunk's identity-strip rows), trailing update (strip tiles stop at c1).
  // `defer_last`: the trailing update of the chunk's LAST step is left to the next call (it only touches columns >= c1, so
  // the chunk -- its columns of L and its strip -- is complete without it, and the caller records the chunk's event one
  // launch earlier: the second stream starts on the chunk while this update is still running)
  // ---- a whole block step as one launch (ekf_step.hpp): the work lists of a ONE-chunk plan of nblk steps -------------------
  int ensure_step_fused_lists(int nblk) {
    if (sfp.nblk == nblk) return EKF_OK;
    std::vector<int> all;
    sfp.off.assign(nblk, 0);
    sfp.cnt.assign(nblk, 0);
    const int s0 = 0, s1 = nblk;
    for (int j = 0; j < nblk; ++j) {
      sfp.off[j] = (int)all.size();
      const int first = 2 * (j + 1);                        // first 64-row / 64-column block of the trailing matrix
      for (int r = first; r < 2 * nblk; ++r)                // S rows below the diagonal block: lower tiles, the diagonal one writes the panel rows
        for (int c = first; c <= r; ++c) { all.push_back(r); all.push_back(c); all.push_back(SF_TILE | (c == r ? SF_WRITE_PANEL : 0)); }
      for (int r = 2 * nblk; r < 2 * nblk + 2 * (j + 1 - s0); ++r) {   // the strip rows of the chunk: columns up to its end
        if (first < 2 * s1) {
          for (int c = first; c < 2 * s1; ++c) { all.push_back(r); all.push_back(c); all.push_back(SF_TILE | (c == first ? SF_WRITE_PANEL : 0)); }
        } else {
          all.push_back(r); all.push_back(r); all.push_back(SF_WRITE_PANEL);           // no tile: the panel rows only
        }
      }
      sfp.cnt[j] = ((int)all.size() - sfp.off[j]) / 3;
      all[sfp.off[j] + 2] |= SF_WRITE_DIAG;                 // (a step always has a workgroup: the strip rows)
    }
    HIPCHK(hipStreamSynchronize(stream));
    if (d_sf_lists) HIPCHK(hipFree(d_sf_lists));
    d_sf_lists = nullptr;
    HIPCHK(hipMalloc(&d_sf_lists, all.size() * sizeof(int)));
    HIPCHK(hipMemcpy(d_sf_lists, all.data(), all.size() * sizeof(int), hipMemcpyHostToDevice));
    sfp.nblk = nblk; sfp.s1 = s1;
    return EKF_OK;
  }
  bool launch_step_fused(int step, int m, hipStream_t sc_) {
    if constexpr (kIsF32) {
      if (!sf_now || sfp.nblk == 0 || step >= sfp.nblk || sfp.cnt[step] > num_cus / 2 || sc_ != stream) return false;   // (every workgroup resident, with room to spare)
      StepFusedArgs a{};
      a.Y = d_Y; a.ldy = ldy; a.Dj = d_Dinv + (size_t)step * 128 * 128; a.status = d_status; a.m = m; a.j = step;
      a.wl = d_sf_lists + sfp.off[step]; a.nwg = sfp.cnt[step];
      a.gate = reinterpret_cast<unsigned*>(d_status + 9);
      small_gate_total += (unsigned)a.nwg;
      a.gate_target = small_gate_total;
      Scope sc(this, KID_CHOL_TRAILING, sc_);
      ++launch_cnt[EKF_LAUNCH_CHAIN_STEP_FUSED];
      k_chain_step_fused<<<a.nwg, 1024, 0, sc_>>>(a);
      return true;
    }
    return false;
  }
  struct PendingTrailing { int step = -1, c0 = 0, c1 = 0; } chain_pending;
  void chain_trailing(int step, int c0, int c1, int m, int m_pad, hipStream_t sc_, bool allow_fused) {
    const int nb = NB();
    T* Y = d_Y;
    const int j = step * nb, r0 = j + nb;
    const int vrows = m_pad - c0, tcols = m_pad - r0;
    if (r0 >= m_pad || tcols <= 0) return;
    if (allow_fused && launch_trail_diag(step, m, sc_)) return;       // the update of this step and the factor of the next one as one launch
    Scope sc(this, KID_CHOL_TRAILING, sc_);                            // Y[r0.., r0:] -= P P_S^T; strip rows stop at c1
    const T* P = Y + (size_t)r0 * ldy + j;
    T* C = Y + (size_t)r0 * ldy + r0;
    ++launch_cnt[EKF_LAUNCH_CHAIN_STEP];
    gemm<ROLE_TRAILING, false, 64, 64>(P, ldy, P, ldy, C, ldy, vrows, tcols, nb, T(-1), T(1), 1, r0, r0, 0, 0,
                                       sc_, nullptr, 0, m_pad, c1);
  }
  void chain_steps(int step0, int step1, int c0, int c1, int m, int m_pad, hipStream_t sc_, bool skip_panel = false,
                   bool defer_last = false) {
    const int nb = NB();
    T* Y = d_Y;
    if (chain_pending.step >= 0) {
      chain_trailing(chain_pending.step, chain_pending.c0, chain_pending.c1, m, m_pad, sc_, true);
      chain_pending.step = -1;
    }
    for (int step = step0; step < step1; ++step) {
      if (!skip_panel && !defer_last && chain_diag_ahead != step && launch_step_fused(step, m, sc_)) continue;
      const int j = step * nb;
      T* Ajj = Y + (size_t)j * ldy + j;
      T* Dj = d_Dinv + (size_t)step * nb * nb;
      if (chain_diag_ahead != step) {
        Scope sc(this, KID_CHOL_DIAG, sc_);
        ++launch_cnt[EKF_LAUNCH_CHAIN_STEP];
        if (nb == 128) {
          if constexpr (kIsF32)
            k_chol_diag_packed<><<<1, 1024, 0, sc_>>>(Ajj, ldy, Dj, d_status, std::max(1, std::min(8, (m - j + 15) / 16)));
        } else {
          if constexpr (!kIsF32) {
            if (opt_mfma)
              k_chol_diag_packed_f64<<<1, 512, 0, sc_>>>(Ajj, ldy, Dj, d_status, std::max(1, std::min(4, (m - j + 15) / 16)));
            else
              k_chol_diag<T, 64><<<1, 512, diag_lds(64), sc_>>>(Ajj, ldy, Dj, d_status);
          } else {
            k_chol_diag<T, 64><<<1, 512, diag_lds(64), sc_>>>(Ajj, ldy, Dj, d_status);
          }
        }
      }
      // rows that change at this step: S rows below the diagonal block, then strip rows [0, r0 - c0)
      // (Z rows c0..r0 of this chunk): contiguous, m_pad - c0 of them starting at row r0
      const int r0 = j + nb;
      const int vrows = m_pad - c0;
      if (!skip_panel) {
        Scope sc(this, KID_CHOL_PANEL, sc_);                 // P = Y[r0.., j:j+nb] * Linv_jj^T, in place
        T* P = Y + (size_t)r0 * ldy + j;
        ++launch_cnt[EKF_LAUNCH_CHAIN_STEP];
        launch_panel(P, Dj, vrows, sc_);
      }
      if (defer_last && step == step1 - 1 && r0 < m_pad) {
        chain_pending.step = step; chain_pending.c0 = c0; chain_pending.c1 = c1;
      } else {
        chain_trailing(step, c0, c1, m, m_pad, sc_, !skip_panel);
      }
    }
  }
  // ---- a8-a11 update ---------------------------------------------------------------------
  int update(const void* z, const int* idx, int M, int plane, bool on_device) override {
    HIPCHK(hipSetDevice(device));
    if (sh_on) {
      if (on_device) {
        // the ownership split of the list is made on the host: one small D2H of the indices, z stays where it is
        if (M < 0 || M > N) FAIL(EKF_ERR_ARG, "M out of range");
        std::vector<int> hidx((size_t)std::max(M, 0));
        if (M > 0) {
          HIPCHK(hipMemcpyAsync(hidx.data(), idx, (size_t)M * sizeof(int), hipMemcpyDeviceToHost, stream));
          HIPCHK(hipStreamSynchronize(stream));
        }
        return shard_update(z, hidx.data(), M, plane);
      }
      if (M > 0 && z) HIPCHK(hipMemcpyAsync(d_z, z, (size_t)2 * M * sizeof(T), hipMemcpyHostToDevice, stream));
      return shard_update(d_z, idx, M, plane);
    }
    if (M < 0 || M > N) FAIL(EKF_ERR_ARG, "M out of range");
    if (M == 0 && !plane) return EKF_OK;
    if (!have_meas) FAIL(EKF_ERR_STATE, "ekf_update needs the h/H of ekf_predict or ekf_measure");
    if (M > 0 && (!z || !idx)) FAIL(EKF_ERR_ARG, "z / indices are NULL");
    if (!on_device) {
      for (int k = 0; k < M; ++k) {
        if (idx[k] < 0 || idx[k] >= N) FAIL(EKF_ERR_ARG, "feature index out of range");
        if (k > 0 && idx[k - 1] >= idx[k]) FAIL(EKF_ERR_ARG, "measured indices must be strictly ascending");
      }
    }
    cur_z = nullptr;
    cur_midx = nullptr;
    if (M > 0) {
      if (on_device) {              // resident inputs are read in place (they must outlive the step)
        cur_z = static_cast<const T*>(z);
        cur_midx = idx;
      } else {
        // host z / indices go through a small ring of pinned staging slots: the two copies are then truly asynchronous
        // (a copy from pageable memory is staged by the runtime and costs the caller ~10 us each)
        int rcs = stage_inputs(z, idx, M);
        if (rcs) return rcs;
        sh_list.clear();
      }
    }
    const int nb = NB();
    int m = 2 * M + (plane ? 3 : 0), m_pad = round_up(m, nb);
    const int npad_live = round_up(n, nb);
    // Blocked right-looking Cholesky of S in column chunks (a few block steps each).  Chunk g carries its
    // own identity block under S (the strip, rows m_pad..), which the same panel / trailing sweeps turn
    // into Z_gg = L_gg^-T: the inverse of the DIAGONAL chunk only.  As soon as the chain has left chunk g
    // the rest runs on the second stream, off the chain's CUs:
    //   solve       V_g = [W_g; nu_g^T] Z_gg                         (K stops at the diagonal)
    //   W update    [W; nu^T][:, c1:] -= V_g L[c1:, c0:c1]^T         (right-looking, rank = chunk width)
    //   downdate    Sigma -= V_g V_g^T
    // so only the last chunk's solve + downdate are exposed after the chain.
    const int nsteps = m_pad / nb;
    int cend[8];
    const int nchunks = plan_chunks(nsteps, cend);
    ChunkTab tab{nchunks, {}};
    int strip_rows = 0;
    for (int g = 0; g < nchunks; ++g) {
      tab.end[g] = cend[g] * nb;
      strip_rows = std::max(strip_rows, (cend[g] - (g ? cend[g - 1] : 0)) * nb);
    }
    // One diagonal block (2 M + 3 <= 128, the reference's operating point): the chunk inverse is the transposed
    // Linv of the diagonal factor, so the panel launch, the solve and the state update are ONE launch
    // (k_solve_state_oneblock; small maps: the downdate and the normalisation too, k_update_oneblock_small): the step is
    // launch-bound there.
    bool oneblock = false, allinone = false;                // (allinone: downdate + normalisation are in the launch too)
    if constexpr (kIsF32)
      oneblock = opt_fused && opt_mfma && nchunks == 1 && m_pad == 128 && nb == 128 && !prof_on(KID_SOLVE) &&
                 !prof_on(KID_STATE_UPDATE) && !prof_on(KID_CHOL_PANEL);
    const int* ip_list = cur_midx ? cur_midx : d_midx;    // the measured list (device memory), for the re-evaluations of W below
    // Small map (n_pad <= 256: the reference's 20-35 features): W, S, the factor, the solve, the state update, the downdate
    // and the normalisation as ONE launch (ekf_small.hpp) -- every workgroup forms W, S and the factor for itself
    bool onelaunch = false;
    int rc = EKF_OK;
    if constexpr (kIsF32) {
      const int nt64 = (npad_live / 64) * (npad_live / 64 + 1) / 2;
      int small_rc = 0, small_nchunk = 0;
      onelaunch = oneblock && opt_small_onelaunch && npad_live <= kSmallMaxRows && nt64 <= num_cus && ld % 4 == 0 &&
                  small_chunking(n, round_up(m, 4), &small_rc, &small_nchunk) && M <= 62 && !prof_on(KID_SIGMA_HT) &&
                  !prof_on(KID_INNOVATION_COV) && !prof_on(KID_CHOL_DIAG) && !prof_on(KID_DOWNDATE) && !prof_on(KID_NORMALIZE);
      if (onelaunch) {
        if (w_zeroed_n != n) {              // (what build_innovation keeps: pad rows of W and the rows behind nu are zero)
          HIPCHK(hipMemsetAsync(d_W + (size_t)n * ldy, 0, (size_t)(npad_live - n + nb) * ldy * sizeof(T), stream));
          w_zeroed_n = n;
        }
        counter_next = 0;
        SmallUpdateArgs a{};
        a.S = S(); a.ld = ld; a.n = n; a.npad = npad_live;
        a.Hc = d_Hc; a.Hf = d_Hf; a.pos = d_pos; a.coding = d_coding; a.midx = ip_list;
        a.M = M; a.plane = plane; a.nfeat = N;
        a.z = cur_z ? cur_z : d_z; a.h = d_h; a.mu = mu();
        a.r_pix = T(sigma_pixel_2); a.r_plane = T(0.00001);
        a.W = d_W; a.ldw = ldy; a.Y = d_Y; a.ldy = ldy; a.Dinv = d_Dinv; a.V = d_V; a.ldv = ldy;
        a.scr_qn = d_scr + SCR_QN; a.status = d_status;
        a.gate = reinterpret_cast<unsigned*>(d_status + 9);
        small_gate_total += (unsigned)(nt64 + 1);
        a.gate_target = small_gate_total;
        a.ntiles = nt64; a.wp = round_up(m, 4); a.stamps = d_small_stamps;
        a.rc = small_rc; a.nchunk = small_nchunk;
        ++launch_cnt[EKF_LAUNCH_UPDATE_ONELAUNCH];
        k_update_small_onelaunch<<<nt64 + 1, 1024, 0, stream>>>(a);
      }
    }
    if (!onelaunch) rc = build_innovation(M, plane, true, &m, &m_pad, &tab, strip_rows);
    cur_z = nullptr;
    cur_midx = nullptr;
    if (rc) return rc;
    T* Y = d_Y;
    T* Zs = d_Y + (size_t)m_pad * ldy;                     // the strip
    const int tile = (kIsF32 && opt_mfma) ? 128 : 64;
    const int ntr = (npad_live + nb) / tile, ntc = m_pad / tile;
    rc = ensure_tilemap(npad_live / tile, ntr, ntc);
    if (rc) return rc;

    // EKF_OPT_W_RECOMPUTE (fp32 MFMA path, several chunks): the W columns of chunk g + 1 come from the downdated Sigma
    // instead of the right-looking GEMM update (see the option's comment in ekf_monoslam.h)
    const bool recompute = kIsF32 && opt_mfma && opt_wrecompute && nchunks > 1 && tile == 128 && !oneblock;
    int step = 0;
    bool b_inflight = false;
    // the chain of a chunk as ONE look-ahead launch (ekf_chain.hpp) instead of three launches per block step
    const bool pchain = chain_persistent_ok() && nb == 128 && !oneblock && nsteps >= 2;
    if (pchain) { rc = chain_begin_update(nsteps, nchunks, cend); if (rc) return rc; }
    chain_diag_ahead = -1;
    chain_pending.step = -1;
    td_nblk = (td_nblk == nsteps) ? td_nblk : 0;
    if (!pchain && !oneblock && trail_diag_ok() && nb == 128 && nsteps >= 2) { rc = ensure_trail_diag_lists(nsteps, nchunks, cend); if (rc) return rc; }
    sf_now = false;
    if constexpr (kIsF32) {
      if (opt_step_fused && opt_mfma && opt_fused && nb == 128 && nchunks == 1 && !oneblock && !pchain && !prof_on(KID_CHOL_DIAG) &&
          !prof_on(KID_CHOL_PANEL) && !prof_on(KID_CHOL_TRAILING)) {
        rc = ensure_step_fused_lists(nsteps);
        if (rc) return rc;
        sf_now = true;
      }
    }
    if (onelaunch) allinone = true;
    for (int gi = 0; gi < (onelaunch ? 0 : nchunks); ++gi) {
      const int c0 = step * nb, c1 = cend[gi] * nb;
      // chunk 0 has the chip to itself; later chunks run beside the tile GEMMs of stream_b, on the reserved CUs
      hipStream_t sc_ = stream;
      if (pchain) { rc = chain_launch(gi, m, gi == 0, sc_); if (rc) return rc; }
      else chain_steps(step, cend[gi], c0, c1, m, m_pad, sc_, oneblock, opt_chain_defer && gi + 1 < nchunks);
      step = cend[gi];
      const int width = c1 - c0;
      vimg_done = false;
      // the last chunk has nothing left to overlap with: it runs on the main stream, on every CU
      const bool overlap = (stream_b != nullptr) && (gi + 1 < nchunks);
      hipStream_t ss = overlap ? stream_b : stream;
      if (!overlap && b_inflight) {
        if (recompute) {                             // W of this chunk is re-evaluated from Sigma: every earlier downdate first
          HIPCHK(hipEventRecord(ev_b, stream_b));
          HIPCHK(hipStreamWaitEvent(stream, ev_b, 0));
          b_inflight = false;
        } else {                                     // the solve reads W: only the last W update has to be done
          HIPCHK(hipStreamWaitEvent(stream, ev_wu, 0));
        }
      }
      if (overlap) {
        b_inflight = true;
        HIPCHK(hipEventRecord(ev_chain[gi], sc_));
        HIPCHK(hipStreamWaitEvent(stream_b, ev_chain[gi], 0));
      }
      if (oneblock) {
        if constexpr (kIsF32) {
          const int nrb = npad_live / 64, nt64 = nrb * (nrb + 1) / 2;
          // small map: downdate and normalisation congruence in the same launch, one 64 x 64 tile of Sigma per workgroup
          allinone = nt64 <= num_cus && !prof_on(KID_DOWNDATE) && !prof_on(KID_NORMALIZE);
          ++launch_cnt[allinone ? EKF_LAUNCH_UPDATE_ALLINONE : EKF_LAUNCH_UPDATE_ONEBLOCK];
          if (allinone)
            k_update_oneblock_small<<<nt64 + 1, 512, 0, ss>>>(d_W, ldy, d_Dinv, d_V, ldy, npad_live, mu(), n, d_scr + SCR_QOLD,
                                                             d_scr + SCR_QN, Zs, ldy, S(), ld, nt64);
          else
            k_solve_state_oneblock<<<nrb + 1, 256, 0, ss>>>(d_W, ldy, d_Dinv, d_V, ldy, npad_live, mu(), n, d_scr + SCR_QN, Zs,
                                                           ldy);
        }
      } else {
        Scope sc(this, KID_SOLVE, ss);                    // column tiles of the chunk, heaviest first
        solve_s2_now = want_solve_s2(width, npad_live);
        const int wt = width / tile;
        const int slots = 2 * (overlap ? num_cus - reserved_cus : num_cus);
        if (kIsF32 && opt_mfma && 4 * wt * ntr < slots) { // small map: 64 x 64 tiles
          const int* list = d_tilemap + solve6464_off + 2 * (2 * ntc - 2 * wt) * 2 * ntr;
          gemm<ROLE_SOLVE, true, 64, 64>(d_W + c0, ldy, Zs + c0, ldy, d_V + c0, ldy, npad_live + nb, width, width, T(1),
                                         T(0), 0, 0, 0, 1, 0, ss, list, 2 * wt * 2 * ntr);
        } else if (kIsF32 && opt_mfma && !overlap && opt_solve_one_per_cu && wt * ntr < slots && wt * ntr > num_cus) {
          // the last chunk's solve, alone on the chip, between one and two tiles of 128 x 128 per CU (N = 1000: 432 tiles with
          // K = 128 .. 1152): ONE workgroup per CU drawing the tiles heaviest-first -- a workgroup that has its CU to itself
          // runs a K step in 2.0 us against 3.2 us beside a second one, and with at most two per CU the heaviest tile bounds
          // the launch either way; the light tiles then fill the CUs that finish first.  96 -> 86 us, bit-identical
          // (round 4; 128 x 128 tiles on two workgroups per CU: 1.203 ms per step, 64 x 128 tiles: 1.149, this: 1.142)
          solve_one_per_cu_now = true;
          const int* list = d_tilemap + solve_off + 2 * (ntc - wt) * ntr;
          gemm<ROLE_SOLVE, true>(d_W + c0, ldy, Zs + c0, ldy, d_V + c0, ldy, npad_live + nb, width, width, T(1), T(0), 0,
                                 0, 0, 1, 0, ss, list, wt * ntr);
          solve_one_per_cu_now = false;
        } else if (kIsF32 && opt_mfma && wt * ntr < slots) {     // narrow chunk: 64-row tiles fill the chip
          const int* list = d_tilemap + solve64_off + 2 * (ntc - wt) * 2 * ntr;
          // (round 6: when the bf16x6 downdate follows, the tiles also write the plane image of V_g it reads)
          void* vimg = nullptr;
          if constexpr (kIsF32) {
            if (opt_fuse_split && opt_split_bf16 && tile == 128 && tri_count >= num_cus && counter_next + 16 <= kQueueCounters) {
              if (!d_Vimg) HIPCHK(hipMalloc(&d_Vimg, (size_t)(n_pad + 128) * ldy * 6));
              vimg = d_Vimg;
              vimg_done = true;
            }
          }
          gemm<ROLE_SOLVE, true, 64, 128>(d_W + c0, ldy, Zs + c0, ldy, d_V + c0, ldy, npad_live + nb, width, width, T(1),
                                          T(0), 0, 0, 0, 1, 0, ss, list, wt * 2 * ntr, 0, 0, vimg, ldy / 16, c0);
        } else {
          const int* list = d_tilemap + solve_off + 2 * (ntc - wt) * ntr;
          gemm<ROLE_SOLVE, true>(d_W + c0, ldy, Zs + c0, ldy, d_V + c0, ldy, npad_live + nb, width, width, T(1), T(0), 0,
                                 0, 0, 1, 0, ss, list, wt * ntr);
        }
      }
      // W update and downdate of an overlapped chunk go out as ONE queued launch when both run on 128 x 128 tiles:
      // they read the same V_g, and sharing a launch saves one ramp and one partially filled last round.  Not for
      // the chunk before the last: the last solve starts on that chunk's W update, not on its downdate.
      bool fuse = false;
      if constexpr (kIsF32)
        fuse = opt_fuse_wu && (opt_fuse_wu > 1 || recompute || gi + 2 < nchunks) && opt_mfma && overlap && c1 < m_pad &&
               !opt_split_bf16 && tile == 128 && tri_count >= num_cus && counter_next + 8 <= kQueueCounters;
      bool split_now = false;
      if constexpr (kIsF32)
        split_now = opt_split_bf16 && opt_mfma && tile == 128 && tri_count >= num_cus && counter_next + 8 <= kQueueCounters;
      const bool row_rider = split_now && recompute && opt_row_gemv && c1 < m_pad && !fuse;   // rides in the k_syrk_bf16x6 launch below
      if (row_rider) {
        // (nothing here: the row goes with the downdate's launch)
      } else if (c1 < m_pad && !fuse && recompute) {
        // only the innovation row (row npad_live of [W; nu^T]) is updated right-looking: nu^T[c1:] -= y_g^T L[c1:, g]^T
        Scope sc(this, KID_WUPDATE, ss);
        ++launch_cnt[(kIsF32 && opt_row_gemv) ? EKF_LAUNCH_ROW_GEMV : EKF_LAUNCH_ROW_TILE_GEMM];
        if constexpr (kIsF32) {
          if (opt_row_gemv)
            k_innov_row_update<<<(m_pad - c1 + 63) / 64, 64, 0, ss>>>(d_V + (size_t)npad_live * ldy + c0, Y + (size_t)c1 * ldy + c0, ldy,
                                                                      d_W + (size_t)npad_live * ldy + c1, m_pad - c1, width);
        }
        if (!kIsF32 || !opt_row_gemv)
          gemm<ROLE_WUPDATE, false, 64, 128>(d_V + (size_t)npad_live * ldy + c0, ldy, Y + (size_t)c1 * ldy + c0, ldy,
                                             d_W + (size_t)npad_live * ldy + c1, ldy, nb, m_pad - c1, width, T(-1), T(1), 0, 0, 0,
                                             0, 0, ss);
      } else if (c1 < m_pad && !fuse) {
        Scope sc(this, KID_WUPDATE, ss);
        ++launch_cnt[EKF_LAUNCH_W_UPDATE_GEMM];
        const int slots = 2 * (overlap ? num_cus - reserved_cus : num_cus);
        if (kIsF32 && opt_mfma && ((m_pad - c1) / 128) * ntr < slots)
          gemm<ROLE_WUPDATE, false, 64, 128>(d_V + c0, ldy, Y + (size_t)c1 * ldy + c0, ldy, d_W + c1, ldy, npad_live + nb,
                                             m_pad - c1, width, T(-1), T(1), 0, 0, 0, 0, 0, ss);
        else
          gemm<ROLE_WUPDATE, false>(d_V + c0, ldy, Y + (size_t)c1 * ldy + c0, ldy, d_W + c1, ldy, npad_live + nb, m_pad - c1,
                                    width, T(-1), T(1), 0, 0, 0, 0, 0, ss);
      }
      if (overlap && c1 < m_pad && !fuse && !recompute) HIPCHK(hipEventRecord(ev_wu, stream_b));
      if (!overlap && b_inflight) {                  // earlier downdates must be done before Sigma is touched again
        HIPCHK(hipEventRecord(ev_b, stream_b));
        HIPCHK(hipStreamWaitEvent(stream, ev_b, 0));
        b_inflight = false;
      }
      // (round 6: when the last downdate is the bf16x6 kernel, its workgroups take the state update's rows when they run out of
      // tiles -- Syrk6Args::su_*: no launch, no second stream, no events at the end of the step)
      bool su_tail = false;
      if constexpr (kIsF32) su_tail = opt_su_tail && !overlap && nchunks > 1 && split_now;
      if (!overlap && nchunks > 1 && !su_tail) {
        // every column of V and y = L^-1 nu exist now: the state update runs beside the last downdate
        HIPCHK(hipEventRecord(ev_chain[gi], stream));
        HIPCHK(hipStreamWaitEvent(stream_b, ev_chain[gi], 0));
        Scope sc(this, KID_STATE_UPDATE, stream_b);
        k_state_update<T><<<(n + 7) / 8, 512, 0, stream_b>>>(mu(), d_V, ldy, n, d_V + (size_t)npad_live * ldy, m_pad, d_scr + SCR_QN);   // + quaternion normalisation
        b_inflight = true;
      }
      bool split_done = false;
      if constexpr (kIsF32) {
        if (opt_split_bf16 && opt_mfma && tile == 128 && tri_count >= num_cus && counter_next + 8 <= kQueueCounters) {
          // EKF_OPT_SPLIT_BF16: the same contraction on the bf16 matrix pipe at fp32 accuracy (ekf_syrk6.hpp): V_g is split
          // into the plane image (three bf16 per fp32, one 12 KB record per 128 rows x 16 columns), then the lower tiles
          // of Sigma are downdated from LDS-DMA-fed records
          if (!d_Vimg) HIPCHK(hipMalloc(&d_Vimg, (size_t)(n_pad + 128) * ldy * 6));
          if (!vimg_done) {                                 // (the solve's tiles have written the image already)
            Scope sc(this, KID_MISC, ss);
            ++launch_cnt[EKF_LAUNCH_SPLIT_IMAGE];
            dim3 grid(npad_live / 128, width / 16);
            k_split_image<<<grid, 256, 0, ss>>>(d_V, ldy, npad_live, c0, width, d_Vimg, ldy / 16);
          }
          vimg_done = false;
          Scope sc(this, KID_DOWNDATE, ss);
          if (sc.on) prof_work[KID_DOWNDATE] += double(n) * n * (std::min(c1, m) - std::min(c0, m));
          Syrk6Args a{d_Vimg, ldy / 16, c0 / 16, width / 16, S(), ld, d_tilemap + tri6_off, tri_count, d_counters + counter_next,
                      0, 0, INT_MAX};
          a.stag_half = opt_syrk_stag_half; a.stag_mod4 = opt_syrk_stag_mod4;
          if (su_tail) {
            a.su_mu = mu(); a.su_V = d_V; a.su_ldy = ldy; a.su_n = n; a.su_y = d_V + (size_t)npad_live * ldy; a.su_mpad = m_pad;
            a.su_qn = d_scr + SCR_QN; a.su_counter = d_counters + counter_next + 1;
            ++launch_cnt[EKF_LAUNCH_STATE_UPDATE_TAIL];
          }
          if (row_rider) {
            a.ry = d_V + (size_t)npad_live * ldy + c0; a.rL = Y + (size_t)c1 * ldy + c0; a.rldl = ldy;
            a.rnu = d_W + (size_t)npad_live * ldy + c1; a.rcols = m_pad - c1; a.rK = width; a.nrider = (m_pad - c1 + 255) / 256;
          }
          counter_next += 8;
          ++launch_cnt[EKF_LAUNCH_DOWNDATE_BF16X6];
          if (row_rider) ++launch_cnt[EKF_LAUNCH_ROW_RIDER];
          const int wgs = 2 * (overlap ? (num_cus - reserved_cus) : num_cus);
          k_syrk_bf16x6<0><<<a.nrider + std::min(tri_count, wgs), 256, 0, ss>>>(a);
          split_done = true;
        }
      }
      if (fuse) {
        if constexpr (kIsF32) {
          Scope sc(this, KID_DOWNDATE, ss);               // W[:, c1:] -= V_g L[c1:, g]^T, then Sigma -= V_g V_g^T
          // (recompute: of [W; nu^T] only the row tile that holds nu^T -- the rows of W are re-evaluated from Sigma)
          const int nr2 = recompute ? 1 : (npad_live + nb) / 128, n2 = nr2 * ((m_pad - c1) / 128);
          const int row2 = recompute ? npad_live / 128 : 0;
          if (sc.on) {
            const double w = std::min(c1, m) - std::min(c0, m);
            prof_work[KID_DOWNDATE] += double(n) * n * w + 2.0 * (recompute ? 1 : n + 1) * std::max(0, m - c1) * w;
          }
          GemmArgs g{d_V + c0, ldy, d_V + c0, ldy, S(), ld, width, -1.0, 1.0, 2, 0, 0, 0, 0,
                     d_tilemap, n2 + tri_count, d_counters + counter_next, 0, 0, 1,
                     Y + (size_t)c1 * ldy + c0, ldy, d_W + c1, ldy, n2, nr2, row2};
          counter_next += 8;
          ++launch_cnt[EKF_LAUNCH_DOWNDATE_F32_FUSED_WU];
          const int wgs = 2 * (num_cus - reserved_cus);
          k_gemm_mfma<ROLE_DOWNDATE, false><<<std::min(g.ntiles, wgs), 256, 0, ss>>>(g);
          if (!recompute) HIPCHK(hipEventRecord(ev_wu, stream_b));
        }
      } else if (!split_done && !allinone) {
        Scope sc(this, KID_DOWNDATE, ss);                 // Sigma -= V_g V_g^T (lower tiles + mirror)
        if (sc.on) prof_work[KID_DOWNDATE] += double(n) * n * (std::min(c1, m) - std::min(c0, m));   // symmetric half, 2 flop per MAC
        const bool t64 = kIsF32 && opt_mfma && tri_count < num_cus;
        const bool half_tail = !t64 && kIsF32 && opt_mfma && ss != stream_b && trih_count > tri_count &&
                               counter_next + 8 <= kQueueCounters;
        ++launch_cnt[t64 ? EKF_LAUNCH_DOWNDATE_F32_T64 : (half_tail ? EKF_LAUNCH_DOWNDATE_F32_HALF_TAIL : EKF_LAUNCH_DOWNDATE_F32)];
        if (kIsF32 && opt_mfma && tri_count < num_cus)    // small map: 64 x 64 tiles, or most of the chip idles
          gemm<ROLE_DOWNDATE, false, 64, 64>(d_V + c0, ldy, d_V + c0, ldy, S(), ld, npad_live, npad_live, width, T(-1), T(1),
                                             2, 0, 0, 0, 0, ss, d_tilemap + tri64_off, tri64_count);
        else if (kIsF32 && opt_mfma && ss != stream_b)   // half tiles at the end of the list: the launch on every CU only (128 x 128 MFMA kernel)
          gemm<ROLE_DOWNDATE, false>(d_V + c0, ldy, d_V + c0, ldy, S(), ld, npad_live, npad_live, width, T(-1), T(1), 2, 0,
                                     0, 0, 0, ss, d_tilemap + trih_off, trih_count);
        else
          gemm<ROLE_DOWNDATE, false>(d_V + c0, ldy, d_V + c0, ldy, S(), ld, npad_live, npad_live, width, T(-1), T(1), 2, 0,
                                     0, 0, 0, ss, d_tilemap, tri_count);
      }
      if constexpr (kIsF32) {
        if (recompute && gi + 1 < nchunks) {
          // W[:, c1:c2) = Sigma' H^T for the features of the NEXT chunk, Sigma' = Sigma - sum_{g <= gi} V_g V_g^T (stream
          // order: right behind this chunk's downdate, in front of the wait for the chain): the sequential form of the
          // update.  Only these columns of Sigma' are read (a feature's six columns + the camera's), i.e. one more pass
          // over Sigma in all; the right-looking GEMM update W[:, c1:] -= V_g L[c1:, g]^T of every chunk
          // (2 n w_g (m - c1) flop) is not needed
          Scope sc(this, KID_SIGMA_HT, ss);
          ++launch_cnt[EKF_LAUNCH_W_RECOMPUTE];
          const int s0 = c1 / 2, s1 = cend[gi + 1] * nb / 2;
          constexpr int RB = 8;
          dim3 grid((s1 - s0 + 127) / 128, (n + RB - 1) / RB);
          k_sigma_ht_fast<RB><<<grid, 256, 0, ss>>>(S(), ld, n, d_Hc, d_Hf, d_pos, d_coding, ip_list, M, plane, d_W, ldy, m_pad, N,
                                                    s0, s1, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
        }
      }
    }
    if (b_inflight) {
      HIPCHK(hipEventRecord(ev_b, stream_b));
      HIPCHK(hipStreamWaitEvent(stream, ev_b, 0));
      b_inflight = false;
    }
    last_nchunks = nchunks;
    last_recompute = recompute;
    for (int g = 0; g < nchunks; ++g) last_cend[g] = cend[g];
    const T* V = d_V;
    const T* yv = d_V + (size_t)npad_live * ldy;
    if (nchunks == 1 && !oneblock) {
      Scope sc(this, KID_STATE_UPDATE);             // mu += V y, then the quaternion normalisation (same launch)
      k_state_update<T><<<(n + 7) / 8, 512, 0, stream>>>(mu(), V, ldy, n, yv, m_pad, d_scr + SCR_QN);
    }
    if (!allinone) {
      Scope sc(this, KID_NORMALIZE);
      k_strip_congruence<T, 4><<<(2 * n + 255) / 256, 256, 0, stream>>>(S(), ld, n, 3, d_scr + SCR_QN,
                                                                       static_cast<const T*>(nullptr));
    }
    HIPCHK(hipGetLastError());
    last_m = m; last_m_pad = m_pad; last_n = n;
    have_update = true;
    have_meas = false;                                    // h/H belong to the pre-update state
    ++frame_seq;
    return EKF_OK;
  }

  int innovation_covariance(const int* idx, int M, int plane, void* out) override {
    HIPCHK(hipSetDevice(device));
    if (M < 0 || M > N) FAIL(EKF_ERR_ARG, "M out of range");
    if (!have_meas) FAIL(EKF_ERR_STATE, "ekf_innovation_covariance needs ekf_predict / ekf_measure first");
    const int mm = 2 * M + (plane ? 3 : 0);
    if (mm == 0) return EKF_OK;
    for (int k = 0; k < M; ++k)
      if (idx[k] < 0 || idx[k] >= N) FAIL(EKF_ERR_ARG, "feature index out of range");
    if (M > 0) HIPCHK(hipMemcpyAsync(d_midx, idx, (size_t)M * sizeof(int), hipMemcpyHostToDevice, stream));
    sh_list.clear();
    int m = 0, m_pad = 0;
    int rc;
    if (sh_on) {
      rc = check_ascending(idx, M);
      if (rc) return rc;
      rc = shard_build_ws(idx, M, plane, nullptr, &m, &m_pad);       // W rows {camera, own}, own rows of S, "reassemble S"
    } else {
      rc = build_innovation(M, plane, false, &m, &m_pad);
    }
    if (rc) return rc;
    std::vector<T> tmp((size_t)m * m);
    HIPCHK(hipMemcpy2DAsync(tmp.data(), (size_t)m * sizeof(T), d_Y, (size_t)ldy * sizeof(T), (size_t)m * sizeof(T), m,
                            hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    T* o = static_cast<T*>(out);
    for (int r = 0; r < m; ++r)
      for (int c = 0; c < m; ++c) o[(size_t)c * m + r] = tmp[(size_t)r * m + c];
    have_update = false;                                  // Y no longer holds the last factorisation
    return EKF_OK;
  }

  int get_gain(void* out) override {
    HIPCHK(hipSetDevice(device));
    if (!have_update) FAIL(EKF_ERR_STATE, "no update to take the gain from");
    const int m = last_m, nn = last_n, m_pad = last_m_pad;
    const int nb = NB();
    const int npad_live = round_up(nn, nb);
    const size_t need = (size_t)npad_live * m_pad * 2;     // K and the scratch of the back-substitution
    if (need > K_elems) {
      if (d_K) HIPCHK(hipFree(d_K));
      d_K = nullptr;
      HIPCHK(hipMalloc(&d_K, need * sizeof(T)));
      K_elems = need;
    }
    // K = V L^-1 by block back-substitution over the chunks of the factorisation (only the diagonal
    // inverses Z_gg = L_gg^-T exist):  K_g = (V_g - K[:, c1:] L[c1:, c0:c1]) Z_gg^T,  last chunk first.
    // One chunk: K = V Z^T.
    const T* Zs = d_Y + (size_t)m_pad * ldy;
    int wmax = 0;
    for (int g = 0; g < last_nchunks; ++g) wmax = std::max(wmax, (last_cend[g] - (g ? last_cend[g - 1] : 0)) * nb);
    T* Tm = d_K + (size_t)npad_live * m_pad;              // npad_live x wmax scratch behind K
    for (int g = last_nchunks - 1; g >= 0; --g) {
      const int c0 = (g ? last_cend[g - 1] : 0) * nb, c1 = last_cend[g] * nb, w = c1 - c0;
      const T* A = d_V + c0;
      int lda = ldy;
      if (c1 < m_pad) {
        dim3 grid((w + 255) / 256, npad_live);
        k_copy2d<T><<<grid, 256, 0, stream>>>(d_V + c0, ldy, Tm, wmax, npad_live, w);
        gemm<ROLE_GAIN, true>(d_K + c1, m_pad, d_Y + (size_t)c1 * ldy + c0, ldy, Tm, wmax, npad_live, w, m_pad - c1, T(-1),
                              T(1), 0, 0, 0, 0);
        A = Tm;
        lda = wmax;
      }
      gemm<ROLE_GAIN, false>(A, lda, Zs + c0, ldy, d_K + c0, m_pad, npad_live, w, w, T(1), T(0), 0, 0, 0, 0);
    }
    std::vector<T> tmp((size_t)nn * m);
    HIPCHK(hipMemcpy2DAsync(tmp.data(), (size_t)m * sizeof(T), d_K, (size_t)m_pad * sizeof(T), (size_t)m * sizeof(T), nn,
                            hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    T* o = static_cast<T*>(out);
    for (int r = 0; r < nn; ++r)
      for (int c = 0; c < m; ++c) o[(size_t)c * nn + r] = tmp[(size_t)r * m + c];
    return EKF_OK;
  }

  // ---- state / covariance access ------------------------------------------------------------
  int get_state(void* out, int off, int count) override {
    HIPCHK(hipSetDevice(device));
    if (off < 0 || count < 0 || off + count > n) FAIL(EKF_ERR_ARG, "state segment out of range");
    if (count) { int rc = rb_add(out, mu() + off, (size_t)count * sizeof(T)); if (rc) return rc; }
    return rb_finish(true);                              // data and status words in one synchronisation
  }
  int set_state(const void* in, int off, int count) override {
    HIPCHK(hipSetDevice(device));
    if (off < 0 || count < 0 || off + count > n) FAIL(EKF_ERR_ARG, "state segment out of range");
    if (count) HIPCHK(hipMemcpyAsync(mu() + off, in, (size_t)count * sizeof(T), hipMemcpyHostToDevice, stream));
    HIPCHK(hipStreamSynchronize(stream));
    have_meas = false;
    return EKF_OK;
  }
  int get_sigma(void* out, int r0, int c0, int rows, int cols) override {
    HIPCHK(hipSetDevice(device));
    if (r0 < 0 || c0 < 0 || rows < 0 || cols < 0 || r0 + rows > n || c0 + cols > n)
      FAIL(EKF_ERR_ARG, "covariance block out of range");
    if (rows == 0 || cols == 0) return EKF_OK;
    std::vector<T> tmp((size_t)rows * cols);
    int rcs = EKF_OK;
    if (tmp.size() * sizeof(T) <= ((size_t)1 << 22)) {    // small blocks: pinned bounce, status in the same round trip
      int rc = rb_add_2d(tmp.data(), S() + (size_t)r0 * ld + c0, (size_t)ld * sizeof(T), (size_t)cols * sizeof(T), rows);
      if (rc) return rc;
      rcs = rb_finish(true);
    } else {
      HIPCHK(hipMemcpy2DAsync(tmp.data(), (size_t)cols * sizeof(T), S() + (size_t)r0 * ld + c0, (size_t)ld * sizeof(T),
                              (size_t)cols * sizeof(T), rows, hipMemcpyDeviceToHost, stream));
      HIPCHK(hipStreamSynchronize(stream));
      rcs = check_status();
    }
    T* o = static_cast<T*>(out);
    for (int r = 0; r < rows; ++r)
      for (int c = 0; c < cols; ++c) o[(size_t)c * rows + r] = tmp[(size_t)r * cols + c];
    return rcs;
  }
  // diagnostics (tools/determinism_probe_sharded.py): a block of the update's workspace as the last update left it --
  // which = 0: W (the columns of every chunk as the solve read them), 1: V = W L^-T; row-major rows x cols
  int peek_work(int which, void* out, int r0, int c0, int rows, int cols) override {
    HIPCHK(hipSetDevice(device));
    if (which == 2) {
      // the task trace of the persistent chain kernel (EKF_CHAIN_TRACE=1): `rows` records of 8 32-bit words from record
      // r0 on (cols must be 8; record -1 = the header, word 0 = records written); the words are copied as they are
      if (!d_chain_trace) FAIL(EKF_ERR_STATE, "no chain trace (create the filter with EKF_CHAIN_TRACE=1)");
      if (cols != 8 || r0 < -1 || rows < 0 || r0 + rows > kChainTraceCap) FAIL(EKF_ERR_ARG, "trace block out of range");
      HIPCHK(hipDeviceSynchronize());
      HIPCHK(hipMemcpy(out, d_chain_trace + 8 * (size_t)(r0 + 1), (size_t)rows * 8 * sizeof(unsigned), hipMemcpyDeviceToHost));
      return EKF_OK;
    }
    if (which == 3) {
      // the phase stamps of k_update_small_onelaunch (EKF_SMALL_STAMPS=1): 16 64-bit words (rows = 16, cols = 2 32-bit halves)
      if (!d_small_stamps) FAIL(EKF_ERR_STATE, "no stamps (create the filter with EKF_SMALL_STAMPS=1)");
      if (rows != 16 || cols != 2 || r0 != 0 || c0 != 0) FAIL(EKF_ERR_ARG, "stamp block out of range");
      HIPCHK(hipDeviceSynchronize());
      HIPCHK(hipMemcpy(out, d_small_stamps, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
      return EKF_OK;
    }
    if (which < 0 || which > 1) FAIL(EKF_ERR_ARG, "workspace id out of range");
    if (r0 < 0 || c0 < 0 || rows < 0 || cols < 0 || r0 + rows > n_pad + NB() || c0 + cols > ldy)
      FAIL(EKF_ERR_ARG, "workspace block out of range");
    if (rows == 0 || cols == 0) return EKF_OK;
    HIPCHK(hipDeviceSynchronize());
    const T* src = (which == 0 ? d_W : d_V) + (size_t)r0 * ldy + c0;
    HIPCHK(hipMemcpy2D(out, (size_t)cols * sizeof(T), src, (size_t)ldy * sizeof(T), (size_t)cols * sizeof(T), rows, hipMemcpyDeviceToHost));
    return EKF_OK;
  }
  int set_sigma(const void* in, int r0, int c0, int rows, int cols) override {
    HIPCHK(hipSetDevice(device));
    if (r0 < 0 || c0 < 0 || rows < 0 || cols < 0 || r0 + rows > n || c0 + cols > n)
      FAIL(EKF_ERR_ARG, "covariance block out of range");
    if (rows == 0 || cols == 0) return EKF_OK;
    const T* i = static_cast<const T*>(in);
    std::vector<T> tmp((size_t)rows * cols);
    for (int r = 0; r < rows; ++r)
      for (int c = 0; c < cols; ++c) tmp[(size_t)r * cols + c] = i[(size_t)c * rows + r];
    HIPCHK(hipMemcpy2DAsync(S() + (size_t)r0 * ld + c0, (size_t)ld * sizeof(T), tmp.data(), (size_t)cols * sizeof(T),
                            (size_t)cols * sizeof(T), rows, hipMemcpyHostToDevice, stream));
    HIPCHK(hipStreamSynchronize(stream));
    have_meas = false;
    return EKF_OK;
  }
  int covariance_parameter(double* out) override {
    T d[7 * 7];
    int rc = get_sigma(d, 0, 0, 7, 7);
    if (rc) return rc;
    T acc = T(0);
    acc += d[0] + d[8] + d[16];                          // vR.cpp:854
    acc += d[32] + d[40] + d[48] + d[24];                // vR.cpp:855
    *out = double(acc);
    return EKF_OK;
  }
  int check_invariants(double* pad, double* asym, double* big) override {
    HIPCHK(hipSetDevice(device));
    unsigned int* d_out = reinterpret_cast<unsigned int*>(d_status + 4);      // 4 spare ints behind the status words
    HIPCHK(hipMemsetAsync(d_out, 0, 4 * sizeof(int), stream));
    dim3 grid(std::min(16, (ld + 255) / 256), n_pad);
    k_check_invariants<T><<<grid, 256, 0, stream>>>(S(), ld, n, n_pad, d_out);
    unsigned int h[4];
    { int rc = rb_add(h, d_out, sizeof(h)); if (rc) return rc; }
    int rcs = rb_finish(true);
    float f[3];
    memcpy(f, h, sizeof(f));
    if (pad) *pad = f[0];
    if (asym) *asym = f[1];
    if (big) *big = f[2];
    return rcs;
  }
  int feature_xyz(int index, void* xyz, void* cov) override {
    HIPCHK(hipSetDevice(device));
    if (index < 0 || index >= N) FAIL(EKF_ERR_ARG, "feature index out of range");
    { int rcs = shard_sync_diag_blocks(); if (rcs) return rcs; }      // sharded: the owner's block of the feature (collective)
    k_feature_xyz<T><<<1, 64, 0, stream>>>(mu(), S(), ld, pos[index], coding[index], d_tmp);
    T o[12];
    { int rc = rb_add(o, d_tmp, sizeof(o)); if (rc) return rc; rc = rb_finish(false); if (rc) return rc; }
    T* x = static_cast<T*>(xyz);
    T* c = static_cast<T*>(cov);
    if (x) for (int k = 0; k < 3; ++k) x[k] = o[k];
    if (c) for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) c[b * 3 + a] = o[3 + a * 3 + b];
    return EKF_OK;
  }
  // ---- f2 / f1: search ellipses and 1-point RANSAC hypotheses -------------------------------------
  int* d_ibuf = nullptr;                 // 3 N ints (ellipses) / M ints (counts)
  unsigned char* d_rmask = nullptr;      // M x M inlier mask
  size_t rmask_bytes = 0;

  int rescue(const void* cam_before, const void* z, const int* idx, int M, double thr, unsigned char* out) override {
    HIPCHK(hipSetDevice(device));
    if (M <= 0) return EKF_OK;
    if (M > N || !cam_before || !z || !idx || !out) FAIL(EKF_ERR_ARG, "bad arguments");
    for (int k = 0; k < M; ++k)
      if (idx[k] < 0 || idx[k] >= N) FAIL(EKF_ERR_ARG, "feature index out of range");
    int rc = sync_layout();
    if (rc) return rc;
    if (sh_on) { rc = check_ascending(idx, M); if (rc) return rc; }
    rc = stage_inputs(z, idx, M, cam_before);          // pinned staging ring -> d_z / d_midx / d_tmp
    if (rc) return rc;
    sh_list.clear();
    if (!d_ibuf) HIPCHK(hipMalloc(&d_ibuf, (size_t)std::max(capN, 1) * 3 * sizeof(int)));
    // the gate flags go straight into host-mapped pinned memory (one synchronisation, no staged copy)
    if (!h_gate) HIPCHK(hipHostMalloc(&h_gate, (size_t)std::max(capN, 1) + 64, hipHostMallocDefault));
    unsigned char* d_out = static_cast<unsigned char*>(h_gate);
    {
      Scope sc(this, KID_MEASURE);
      k_measure<T><<<(M + 63) / 64, 64, 0, stream>>>(mu(), d_pos, d_coding, 0, N, cam, d_h, d_Hc, d_Hf, d_flags, d_midx, M,
                                                    d_tmp);
      if (sh_on) {
        // S_hi = H Sigma H^T of a listed feature needs its rows of Sigma: owner-computes, then the 2x2 blocks are gathered
        const ShardTab lt = list_tab(idx, M);
        const int k0 = lt.start[sh_rank], kc = lt.count[sh_rank];
        if (kc > 0)
          k_measure_sd<T><<<(16 * kc + 255) / 256, 256, 0, stream>>>(S(), ld, d_pos, d_coding, 0, N, T(0), d_Hc, d_Hf, d_Sd,
                                                                    d_midx + k0, kc);
        HIPCHK(hipGetLastError());
        int rg = gather_sd(lt, d_midx);
        if (rg) return rg;
      } else {
        k_measure_sd<T><<<(16 * M + 255) / 256, 256, 0, stream>>>(S(), ld, d_pos, d_coding, 0, N, T(0), d_Hc, d_Hf, d_Sd,
                                                                 d_midx, M);
      }
      k_chi2_gate<T><<<(M + 127) / 128, 128, 0, stream>>>(d_h, d_Sd, d_z, d_midx, M, T(thr), d_out);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(stream));
    memcpy(out, d_out, (size_t)M);
    have_meas = true;          // the listed features carry fresh h / H (the others keep older ones)
    have_sd = false;           // the 2x2 blocks just written have no measurement noise: not the St blocks
    return EKF_OK;
  }

  T* d_pts = nullptr;
  T* d_tab = nullptr;                    // table of ekf_export_points_table (grow-only)
  size_t tab_rows_cap = 0;
  int export_points(void* out, int convert) override {
    HIPCHK(hipSetDevice(device));
    if (N == 0) return EKF_OK;
    int rc = sync_layout();
    if (rc) return rc;
    if (!d_pts) HIPCHK(hipMalloc(&d_pts, (size_t)std::max(capN, 1) * 12 * sizeof(T)));
    rc = shard_sync_diag_blocks();                                      // sharded: every feature's own block from its owner
    if (rc) return rc;
    const T* scale_ptr = (camera_dim == 14) ? mu() + 13 : nullptr;      // the map scale is read on the device
    k_export_points<T><<<(N + 63) / 64, 64, 0, stream>>>(mu(), S(), ld, d_pos, d_coding, N, T(1), convert, d_pts, nullptr, 0,
                                                        scale_ptr);
    HIPCHK(hipGetLastError());
    { int rcb = rb_add(out, d_pts, (size_t)N * 12 * sizeof(T)); if (rcb) return rcb; }
    return rb_finish(true);                                             // table and status words: one synchronisation
  }

  // RosVSLAM::getPointsFeatures with its real_index row order and the archived patches (RosVSLAMRansac.cpp:340-418)
  int export_points_table(void* out, int max_rows, int* rows_out) override {
    HIPCHK(hipSetDevice(device));
    const int rows = N ? real_index[N - 1] + 1 : 0;             // :350-352
    if (rows_out) *rows_out = rows;
    if (!out || rows == 0) return EKF_OK;
    if (max_rows < rows) FAIL(EKF_ERR_ARG, "ekf_export_points_table: the buffer holds fewer rows than the table");
    int rc = sync_layout();
    if (rc) return rc;
    rc = shard_sync_diag_blocks();                                      // sharded: every feature's own block from its owner
    if (rc) return rc;
    if ((size_t)rows > tab_rows_cap) {                                  // grow-only table buffer (freed with the filter)
      HIPCHK(hipStreamSynchronize(stream));
      if (d_tab) HIPCHK(hipFree(d_tab));
      d_tab = nullptr;
      tab_rows_cap = 0;
      const size_t cap = std::max<size_t>(2 * (size_t)rows, 256);
      HIPCHK(hipMalloc(&d_tab, cap * 12 * sizeof(T)));
      tab_rows_cap = cap;
    }
    HIPCHK(hipMemsetAsync(d_tab, 0, (size_t)rows * 12 * sizeof(T), stream));
    const T* scale_ptr = (camera_dim == 14) ? mu() + 13 : nullptr;      // the map scale is read on the device
    const size_t na = arch_real.size();
    rc = ensure_arch_idx((size_t)N + na);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(d_arch_idx, real_index.data(), (size_t)N * sizeof(int), hipMemcpyHostToDevice, stream));
    if (na) HIPCHK(hipMemcpyAsync(d_arch_idx + N, arch_real.data(), na * sizeof(int), hipMemcpyHostToDevice, stream));
    k_export_points<T><<<(N + 63) / 64, 64, 0, stream>>>(mu(), S(), ld, d_pos, d_coding, N, T(1), 0, d_tab, d_arch_idx, rows,
                                                        scale_ptr);
    if (na)
      k_export_archived<T><<<((int)na * 12 + 255) / 256, 256, 0, stream>>>(d_archive, d_arch_idx + N, (int)na, T(1), d_tab, rows,
                                                                          scale_ptr);
    int rcs = EKF_OK;
    if ((size_t)rows * 12 * sizeof(T) <= ((size_t)1 << 22)) {
      rcs = rb_add(out, d_tab, (size_t)rows * 12 * sizeof(T));
      if (!rcs) rcs = rb_finish(true);
    } else {
      HIPCHK(hipMemcpyAsync(out, d_tab, (size_t)rows * 12 * sizeof(T), hipMemcpyDeviceToHost, stream));
      HIPCHK(hipStreamSynchronize(stream));
      rcs = check_status();
    }
    if (na > 7000) arch_real.clear();                           // :396-404: the archive is emptied once it is that long
    return rcs;
  }
  int feature_ids(int* ri, int* nf) const override {
    for (int i = 0; i < N; ++i) { if (ri) ri[i] = real_index[i]; if (nf) nf[i] = n_find[i]; }
    return EKF_OK;
  }
  int set_feature_meta(int index, int ri, int nf) override {
    if (index < 0 || index >= N) FAIL(EKF_ERR_ARG, "feature index out of range");
    if (ri >= 0) real_index[index] = ri;
    if (nf >= 0) n_find[index] = nf;
    return EKF_OK;
  }
  int num_archived() const override { return (int)arch_real.size(); }

  int search_ellipses(int sigma_size, int* out) override {
    HIPCHK(hipSetDevice(device));
    if (!have_meas) FAIL(EKF_ERR_STATE, "ekf_get_search_ellipses needs ekf_predict / ekf_measure first");
    if (N == 0) return EKF_OK;
    { int rcs = ensure_sd(); if (rcs) return rcs; }
    if (!d_ibuf) HIPCHK(hipMalloc(&d_ibuf, (size_t)std::max(capN, 1) * 3 * sizeof(int)));
    k_search_ellipses<T><<<(N + 127) / 128, 128, 0, stream>>>(d_Sd, N, sigma_size, d_ibuf);
    HIPCHK(hipGetLastError());
    { int rc = rb_add(out, d_ibuf, (size_t)N * 3 * sizeof(int)); if (rc) return rc; }
    return rb_finish(false);
  }

  int ransac(const void* z, const int* idx, int M, double thr, int* counts, unsigned char* inl, int* best) override {
    HIPCHK(hipSetDevice(device));
    if (M <= 0 || M > N || !z || !idx) FAIL(EKF_ERR_ARG, "bad measured set");
    if (!have_meas) FAIL(EKF_ERR_STATE, "ekf_ransac_1point needs ekf_predict / ekf_measure first");
    for (int k = 0; k < M; ++k)
      if (idx[k] < 0 || idx[k] >= N) FAIL(EKF_ERR_ARG, "feature index out of range");
    { int rcs = stage_inputs(z, idx, M); if (rcs) return rcs; }   // pinned staging ring -> d_z / d_midx
    sh_list.clear();
    int m = 0, m_pad = 0;
    ransac_mask_host = nullptr;
    { int rcs = ensure_sd(); if (rcs) return rcs; }
    if (sh_on) return shard_ransac(idx, M, thr, counts, inl, best);
    int rc = build_innovation(M, 0, false, &m, &m_pad, nullptr, 0, true);       // W = Sigma H^T for the listed features
    if (rc) return rc;
    have_update = false;
    if (!d_ibuf) HIPCHK(hipMalloc(&d_ibuf, (size_t)std::max(capN, 1) * 3 * sizeof(int)));
    if ((size_t)M * M > rmask_bytes) {
      if (d_rmask) HIPCHK(hipFree(d_rmask));
      d_rmask = nullptr;
      HIPCHK(hipMalloc(&d_rmask, (size_t)M * M));
      rmask_bytes = (size_t)M * M;
    }
    {
      Scope sc(this, KID_MISC);
      dim3 grid((M + 127) / 128, M);
      k_ransac_eval<T><<<grid, 128, 0, stream>>>(mu(), d_W, ldy, d_Sd, d_h, d_z, d_pos, d_coding, d_midx, M, cam,
                                                T(thr), d_rmask);
      k_ransac_count<<<(M + 127) / 128, 128, 0, stream>>>(d_rmask, M, d_ibuf);
    }
    // ONE read-back: counts, the camera pose and -- up to 64 KiB -- the whole inlier mask through a host-mapped pinned
    // buffer the device writes (k_pack_ransac), one synchronisation
    const size_t counts_bytes = ((size_t)capN * sizeof(int) + 15) / 16 * 16;   // the pose behind the counts stays 16-byte aligned
    if (!h_ransac) HIPCHK(hipHostMalloc(&h_ransac, counts_bytes + 8 * sizeof(T) + kRansacMaskBytes + 64,
                                        hipHostMallocDefault));
    int* pc = reinterpret_cast<int*>(h_ransac);
    T* pcam = reinterpret_cast<T*>(reinterpret_cast<unsigned char*>(h_ransac) + counts_bytes);
    unsigned char* pmask = reinterpret_cast<unsigned char*>(pcam + 8);
    const int with_mask = ((size_t)M * M <= kRansacMaskBytes) ? 1 : 0;
    k_pack_ransac<T><<<with_mask ? std::max(1, std::min(64, (M * M + 255) / 256)) : (M + 255) / 256, 256, 0, stream>>>(
        d_ibuf, M, mu(), d_rmask, with_mask, pc, pcam, pmask);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(stream));
    for (int k = 0; k < 7; ++k) ransac_cam[k] = double(pcam[k]);
    ransac_cam_valid = true;
    if (with_mask) { ransac_mask_host = pmask; ransac_mask_M = M; }
    int b = 0;
    for (int k = 1; k < M; ++k) if (pc[k] > pc[b]) b = k;
    if (counts) for (int k = 0; k < M; ++k) counts[k] = pc[k];
    if (best) *best = b;
    if (inl) {
      if (with_mask) {
        for (int k = 0; k < M; ++k) inl[k] = pmask[(size_t)k * M + b];
      } else {
        int rcf = fetch_mask_column(b, idx, M, inl);
        if (rcf) return rcf;
      }
    }
    return EKF_OK;
  }

  // ---- f1: the whole RANSAC branch of VSlamFilter::update (vR.cpp:964-1130 + 1245-1284) in one call -------------
  // glibc's rand() (random_r TYPE_3: additive feedback r[i] = r[i-3] + r[i-31], seeded by the Lehmer generator
  // 16807 x mod 2^31 - 1, 310 outputs discarded, result >> 1): the reference draws its hypotheses with
  // srand(time(NULL)) / rand() (vR.cpp:970, 989), so a given seed reproduces its draw sequence on a glibc platform.
  struct GlibcRand {
    std::vector<unsigned int> st;
    explicit GlibcRand(unsigned int seed) {
      if (seed == 0) seed = 1;
      st.assign(34, 0);
      st[0] = seed;
      for (int i = 1; i < 31; ++i) {
        long long v = (16807LL * (int)st[i - 1]) % 2147483647LL;
        if (v < 0) v += 2147483647LL;
        st[i] = (unsigned int)v;
      }
      for (int i = 31; i < 34; ++i) st[i] = st[i - 31];
      for (int i = 34; i < 344; ++i) st.push_back(st[i - 31] + st[i - 3]);
    }
    int next() {
      const size_t i = st.size();
      st.push_back(st[i - 31] + st[i - 3]);
      return (int)(st[i] >> 1);
    }
  };

  int update_two_stage(const void* z_, const int* idx, int M, int plane, unsigned int seed, double thr, double chi2,
                       unsigned char* is_li, unsigned char* is_hi, int* hyp_drawn) override {
    HIPCHK(hipSetDevice(device));
    if (M < 0 || M > N) FAIL(EKF_ERR_ARG, "M out of range");
    if (!have_meas) FAIL(EKF_ERR_STATE, "ekf_update_two_stage needs the h / H of ekf_predict");
    const T* z = static_cast<const T*>(z_);
    std::vector<unsigned char> li(M, 0), hi(M, 0);
    int drawn = 0;
    if (M > 0) {
      std::vector<int> counts(M);
      int best = 0;
      int rc = ransac(z, idx, M, thr, counts.data(), nullptr, &best);        // every hypothesis, one device pass
      if (rc) return rc;
      int sel = best;
      if (seed != 0) {
        // the reference's loop (vR.cpp:986-1034): draw without replacement, adapt the number of hypotheses to the best
        // inlier ratio seen; the low-innovation set is that of the LAST hypothesis drawn (its flags are overwritten
        // on every draw, :1022)
        GlibcRand rng(seed);
        std::vector<int> list(M);
        for (int k = 0; k < M; ++k) list[k] = k;
        int nhyp = 10000, num_zli = 0;
        const float p = 0.99f;
        for (int i = 0; i < nhyp && !list.empty(); ++i) {
          const int posr = rng.next() % (int)list.size();
          sel = list[posr];
          list.erase(list.begin() + posr);
          ++drawn;
          if (counts[sel] > num_zli) {
            num_zli = counts[sel];
            nhyp = (int)(std::log(1 - p) / (std::log(1 - (num_zli / (M + 0.0f)))));   // :1030 (-> 0 when all are inliers)
          }
        }
      } else {
        drawn = M;
      }
      rc = fetch_mask_column(sel, idx, M, li.data());
      if (rc) return rc;
    }
    T cam_before[7];
    if (M > 0 && !sh_on && ransac_cam_valid) {                   // came back with the counts (no update since)
      for (int k = 0; k < 7; ++k) cam_before[k] = T(ransac_cam[k]);
    } else {
      HIPCHK(hipMemcpyAsync(cam_before, mu(), sizeof(cam_before), hipMemcpyDeviceToHost, stream));
      HIPCHK(hipStreamSynchronize(stream));
    }
    ransac_cam_valid = false;
    std::vector<T> zl, zr;
    std::vector<int> il, ir, kr;
    for (int k = 0; k < M; ++k) {
      auto& zz = li[k] ? zl : zr;
      (li[k] ? il : ir).push_back(idx[k]);
      if (!li[k]) kr.push_back(k);
      zz.push_back(z[2 * k]);
      zz.push_back(z[2 * k + 1]);
    }
    if (!il.empty()) {                                           // low-innovation update (vR.cpp:1036-1064): no plane rows
      int rc = update(zl.data(), il.data(), (int)il.size(), 0, false);
      if (rc) return rc;
    }
    std::vector<T> zh;
    std::vector<int> ih;
    if (!ir.empty()) {                                           // high-innovation rescue (vR.cpp:1066-1117)
      std::vector<unsigned char> g(ir.size(), 0);
      int rc = rescue(cam_before, zr.data(), ir.data(), (int)ir.size(), chi2, g.data());
      if (rc) return rc;
      for (size_t t = 0; t < ir.size(); ++t)
        if (g[t]) { hi[kr[t]] = 1; ih.push_back(ir[t]); zh.push_back(zr[2 * t]); zh.push_back(zr[2 * t + 1]); }
    }
    if (!ih.empty() || plane) {                                  // second update incl. the plane rows (vR.cpp:1245-1284)
      if (ih.empty()) {                                          // plane rows only: h / H of no feature are needed
        have_meas = true;
      }
      int rc = update(zh.data(), ih.data(), (int)ih.size(), plane, false);
      if (rc) return rc;
    }
    for (int k = 0; k < M; ++k)                                  // update_quality_index (vR.cpp:1297, Patch.cpp:145)
      if (li[k] || hi[k]) n_find[idx[k]] += 1;
    if (is_li) memcpy(is_li, li.data(), M);
    if (is_hi) memcpy(is_hi, hi.data(), M);
    if (hyp_drawn) *hyp_drawn = drawn;
    return EKF_OK;
  }

  // ---- multi-GPU row-panel sharding ------------------------------------------------
  // One process per GPU; every rank holds the same feature list and runs every resize operation, rank g OWNS the
  // contiguous features [sh_fb[g], sh_fb[g+1]) and keeps valid: the rows of Sigma / W / V of those features (all
  // columns) and the camera rows; mu is replicated.  Every exchange is an all-gather of equal-sized staging slots
  // through the host's collective (sh_ag: RCCL via torch.distributed, or a C++ node's own ncclAllGather).
  bool sh_on = false;
  int sh_rank = 0, sh_world = 1;
  std::vector<int> sh_fb;                                  // feature boundaries of the ranks, world + 1 entries
  ekf_allgather_fn sh_ag = nullptr;
  void* sh_ctx = nullptr;
  T *d_stage_send = nullptr, *d_stage_recv = nullptr;
  size_t stage_slot = 0;                                   // scalars per slot the staging buffers hold
  int sh_rebalances = 0;
  bool sh_force = false;                                   // world 1 with a callback and EKF_SHARD_FORCE_COLLECTIVE=1: every exchange still runs (profiling the collective path on one GPU)
  std::vector<int> sh_list;                                // host copy of the measured list resident in d_midx
  double sh_imbalance_limit = 1.125;                       // re-partition when a rank owns > 1.125 x the mean rows

  int own_f0() const { return sh_fb[sh_rank]; }
  int own_f1() const { return sh_fb[sh_rank + 1]; }
  int row_of_feature(int f) const { return f < N ? pos[f] : n; }

  // boundaries that balance the ROWS (6 per inverse-depth, 3 per XYZ feature) of the current map
  void partition_by_rows() {
    sh_fb.assign(sh_world + 1, N);
    sh_fb[0] = 0;
    const long long rows = n - camera_dim;
    int f = 0;
    for (int g = 1; g < sh_world; ++g) {
      const long long target = rows * g / sh_world;
      while (f < N && (long long)(pos[f] - camera_dim) < target) ++f;
      sh_fb[g] = f;
    }
  }

  // ---- distributed chain of the sharded step -----------------------------------------------------------
  // From `shard_dist_min_blocks` block steps on (and two ranks or more) the factorisation of S is no longer replicated: rank r OWNS the 128-row
  // blocks I of S with I % world == r (cyclic: the triangle's work is balanced) and keeps only those rows of the trailing
  // matrix up to date -- plus every diagonal block and the chunk's inverse strip, which every rank updates for itself (one
  // block per remaining step and the strip: cheap, and it saves a broadcast per step).  Block step j:
  //   every rank   factor of the diagonal block (j, j)                       (redundant: 17 us, no exchange)
  //   rank r       panel P(j; I) = S(I, j) Linv_jj^T for its own I > j and for the strip rows
  //   ALL-GATHER   of the own panel blocks: afterwards every rank holds all of column block j of L
  //   rank r       trailing update of its own row blocks, of all diagonal blocks (K, K), K > j, and of the strip
  // Every tile is the plain path's tile (k_panel_direct's product, k_gemm_mfma<TRAILING, 64, 64>'s sums): L, the strips and
  // therefore the whole update are bit-identical to the replicated chain's, whoever computed a tile.
  int opt_shard_dist_chain = 1;                          // EKF_SHARD_DIST_CHAIN=0: the replicated chain at every size
  // EKF_SHARD_DIST_MIN_BLOCKS: 40 block steps (N = 2500).  Per step the distributed form pays ~25 us of stand-alone factor and
  // ~35 us of gather where the replicated one pays the whole trailing update; measured on one rank (profiles/
  // r6_shard_world1_dist_chain.txt) the two meet at 32 steps for 8 ranks (2.1 ms each) and the distributed one wins above
  int shard_dist_min_blocks = 40;
  struct DistPlan {
    int nblk = 0, nchunks = 0, cend[8] = {0, 0, 0, 0, 0, 0, 0, 0}, world = 0, rank = -1;
    std::vector<int> pb_off, pb_own, pb_cnt, tl_off, tl_cnt, slot_blocks;
    std::vector<int> tb_off, tb_cnt;                     // the same update as (I, K) pairs of 128 x 128 blocks, without (j + 1, j + 1): k_trail_diag
  } dist;
  int* d_dist_lists = nullptr;
  int* d_dist_counters = nullptr;
  int dist_counters_cap = 0;
  T *d_dist_send = nullptr, *d_dist_recv = nullptr;
  size_t dist_slot = 0;
  bool dist_chain_ok(int nsteps) const {
    return kIsF32 && opt_mfma && opt_shard_dist_chain && sh_on && exchanges() && NB() == 128 && nsteps >= shard_dist_min_blocks;
  }
  int ensure_dist_plan(int nblk, int nchunks, const int* cend) {
    bool same = dist.nblk == nblk && dist.nchunks == nchunks && dist.world == sh_world && dist.rank == sh_rank;
    for (int g = 0; same && g < nchunks; ++g) same = dist.cend[g] == cend[g];
    if (!same) {
      std::vector<int> all;
      dist.pb_off.assign(nblk, 0); dist.pb_own.assign(nblk, 0); dist.pb_cnt.assign(nblk, 0);
      dist.tl_off.assign(nblk, 0); dist.tl_cnt.assign(nblk, 0); dist.slot_blocks.assign(nblk, 0);
      dist.tb_off.assign(nblk, 0); dist.tb_cnt.assign(nblk, 0);
      int maxslot = 0;
      for (int j = 0; j < nblk; ++j) {
        int s0 = 0, s1 = nblk;
        for (int g = 0; g < nchunks; ++g)
          if (j < cend[g]) { s0 = g ? cend[g - 1] : 0; s1 = cend[g]; break; }
        dist.pb_off[j] = (int)all.size();
        for (int I = j + 1; I < nblk; ++I)
          if (I % sh_world == sh_rank) all.push_back(I);
        dist.pb_own[j] = (int)all.size() - dist.pb_off[j];
        for (int t = 0; t <= j - s0; ++t) all.push_back(nblk + t);
        dist.pb_cnt[j] = (int)all.size() - dist.pb_off[j];
        for (int g = 0; g < sh_world; ++g) {
          const int first = j + 1 + ((g - (j + 1)) % sh_world + sh_world) % sh_world;
          const int cnt = first < nblk ? (nblk - 1 - first) / sh_world + 1 : 0;
          dist.slot_blocks[j] = std::max(dist.slot_blocks[j], cnt);
        }
        maxslot = std::max(maxslot, dist.slot_blocks[j]);
        dist.tl_off[j] = (int)all.size();
        for (int I = j + 1; I < nblk; ++I) {             // 64 x 64 tiles relative to row / column block j + 1
          const int d = I - j - 1;
          if (I % sh_world == sh_rank) {
            for (int bi = 2 * d; bi < 2 * d + 2; ++bi)
              for (int bj = 0; bj <= bi; ++bj) { all.push_back(bi); all.push_back(bj); }
          } else {                                       // somebody else's rows: the diagonal block only
            all.push_back(2 * d); all.push_back(2 * d);
            all.push_back(2 * d + 1); all.push_back(2 * d);
            all.push_back(2 * d + 1); all.push_back(2 * d + 1);
          }
        }
        for (int t = 0; t <= j - s0; ++t) {              // the strip rows of the chunk: columns up to the chunk's end
          const int rb = nblk - j - 1 + t;
          for (int bi = 2 * rb; bi < 2 * rb + 2; ++bi)
            for (int bj = 0; bj < 2 * (s1 - j - 1); ++bj) { all.push_back(bi); all.push_back(bj); }
        }
        dist.tl_cnt[j] = ((int)all.size() - dist.tl_off[j]) / 2;
        dist.tb_off[j] = (int)all.size();
        for (int I = j + 1; I < nblk; ++I) {
          if (I % sh_world == sh_rank) {
            for (int K = j + 1; K <= I; ++K)
              if (!(I == j + 1 && K == j + 1)) { all.push_back(I); all.push_back(K); }
          } else if (I != j + 1) {
            all.push_back(I); all.push_back(I);
          }
        }
        for (int t = 0; t <= j - s0; ++t)
          for (int K = j + 1; K < s1; ++K) { all.push_back(nblk + t); all.push_back(K); }
        dist.tb_cnt[j] = ((int)all.size() - dist.tb_off[j]) / 2;
      }
      HIPCHK(hipStreamSynchronize(stream));
      if (stream_b) HIPCHK(hipStreamSynchronize(stream_b));
      if (d_dist_lists) HIPCHK(hipFree(d_dist_lists));
      d_dist_lists = nullptr;
      HIPCHK(hipMalloc(&d_dist_lists, std::max<size_t>(all.size(), 1) * sizeof(int)));
      HIPCHK(hipMemcpy(d_dist_lists, all.data(), all.size() * sizeof(int), hipMemcpyHostToDevice));
      if (nblk * 8 > dist_counters_cap) {
        if (d_dist_counters) HIPCHK(hipFree(d_dist_counters));
        d_dist_counters = nullptr;
        dist_counters_cap = nblk * 8;
        HIPCHK(hipMalloc(&d_dist_counters, (size_t)dist_counters_cap * sizeof(int)));
      }
      const size_t slot = (size_t)std::max(maxslot, 1) * 128 * 128;
      if (slot > dist_slot) {
        if (d_dist_send) HIPCHK(hipFree(d_dist_send));
        if (d_dist_recv) HIPCHK(hipFree(d_dist_recv));
        d_dist_send = d_dist_recv = nullptr;
        HIPCHK(hipMalloc(&d_dist_send, slot * sizeof(T)));
        HIPCHK(hipMalloc(&d_dist_recv, slot * sizeof(T) * sh_world));
        dist_slot = slot;
      }
      dist.nblk = nblk; dist.nchunks = nchunks; dist.world = sh_world; dist.rank = sh_rank;
      for (int g = 0; g < nchunks; ++g) dist.cend[g] = cend[g];
    }
    HIPCHK(hipMemsetAsync(d_dist_counters, 0, (size_t)nblk * 8 * sizeof(int), stream));
    return EKF_OK;
  }
  // block steps [step0, step1) of the distributed chain on stream st (no deferral: a step ends with its trailing update,
  // which carries the factor of the next step when the rank's blocks fit one round of workgroups)
  int dist_chain_steps(int step0, int step1, int m, int m_pad, hipStream_t st) {
    if constexpr (kIsF32) {
      T* Y = d_Y;
      const int nblk = dist.nblk;
      for (int step = step0; step < step1; ++step) {
        const int j = step * 128, r0 = j + 128;
        T* Dj = d_Dinv + (size_t)step * 128 * 128;
        if (chain_diag_ahead != step) {
          Scope sc(this, KID_CHOL_DIAG, st);
          ++launch_cnt[EKF_LAUNCH_CHAIN_STEP];
          k_chol_diag_packed<><<<1, 1024, 0, st>>>(Y + (size_t)j * ldy + j, ldy, Dj, d_status, std::max(1, std::min(8, (m - j + 15) / 16)));
        }
        const int* pb = d_dist_lists + dist.pb_off[step];
        if (dist.pb_cnt[step] > 0) {
          Scope sc(this, KID_CHOL_PANEL, st);
          ++launch_cnt[EKF_LAUNCH_CHAIN_STEP];
          k_panel_direct_blocks<<<2 * dist.pb_cnt[step], 256, 0, st>>>(Y + j, ldy, Dj, pb);
        }
        if (dist.slot_blocks[step] > 0) {
          Scope sc(this, KID_GATHER_S, st);
          const size_t slot = (size_t)dist.slot_blocks[step] * 128 * 128;
          if (dist.pb_own[step] > 0)
            k_dist_pack_panel<T><<<128 * dist.pb_own[step], 32, 0, st>>>(Y + j, ldy, pb, dist.pb_own[step], d_dist_send);
          ++launch_cnt[EKF_LAUNCH_CHAIN_DIST_GATHER];
          const int rc = sh_ag(sh_ctx, d_dist_send, d_dist_recv, slot * sizeof(T), st);
          if (rc != 0) FAIL(EKF_ERR_DEVICE, "the all-gather callback reported a failure");
          if (sh_world > 1)
            k_dist_unpack_panel<T><<<dim3(128 * dist.slot_blocks[step], 1, sh_world), 32, 0, st>>>(d_dist_recv, slot, Y + j, ldy, step,
                                                                                               nblk, sh_world, sh_rank);
        }
        if (trail_diag_ok() && step + 1 < nblk && dist.tb_cnt[step] >= td_min_blocks && dist.tb_cnt[step] <= std::min(td_max_blocks, num_cus)) {
          // the rank's blocks of the update fit one round of workgroups: the update and the factor of the NEXT step as one
          // launch (k_trail_diag: workgroup 0 updates block (j + 1, j + 1) itself and factors it), as on the plain path
          TrailDiagArgs a{};
          a.Y = d_Y; a.ldy = ldy; a.y_bytes = (unsigned)((size_t)2 * ldy * ldy * sizeof(T));
          a.Dinv = d_Dinv; a.dinv_bytes = (unsigned)((size_t)(ldy / 64) * 128 * 128 * sizeof(T));
          a.status = d_status; a.m = m; a.j = step;
          a.blocks = d_dist_lists + dist.tb_off[step]; a.nblocks = dist.tb_cnt[step]; a.do_diag = 1;
          Scope sc(this, KID_CHOL_TRAILING, st);
          ++launch_cnt[EKF_LAUNCH_CHAIN_TRAIL_DIAG];
          k_trail_diag<<<a.nblocks + 1, 1024, kChainLds, st>>>(a);
          chain_diag_ahead = step + 1;
        } else if (dist.tl_cnt[step] > 0) {
          Scope sc(this, KID_CHOL_TRAILING, st);
          ++launch_cnt[EKF_LAUNCH_CHAIN_STEP];
          const T* P = Y + (size_t)r0 * ldy + j;
          GemmArgs g{P, ldy, P, ldy, Y + (size_t)r0 * ldy + r0, ldy, 128, -1.0, 1.0, 0, r0, r0, 0, 0,
                     d_dist_lists + dist.tl_off[step], dist.tl_cnt[step], d_dist_counters + 8 * step, 0, 0, 0};
          k_gemm_mfma<ROLE_TRAILING, false, 64, 64><<<std::min(dist.tl_cnt[step], 2 * num_cus), 256, 0, st>>>(g);
        }
      }
      HIPCHK(hipGetLastError());
    }
    (void)m_pad;
    return EKF_OK;
  }

  int ensure_stage(size_t slot_elems) {
    if (slot_elems <= stage_slot) return EKF_OK;
    HIPCHK(hipStreamSynchronize(stream));
    if (stream_b) HIPCHK(hipStreamSynchronize(stream_b));
    if (stream_g) HIPCHK(hipStreamSynchronize(stream_g));
    if (d_stage_send) HIPCHK(hipFree(d_stage_send));
    if (d_stage_recv) HIPCHK(hipFree(d_stage_recv));
    d_stage_send = d_stage_recv = nullptr;
    slot_elems = (slot_elems + 63) / 64 * 64;
    HIPCHK(hipMalloc(&d_stage_send, slot_elems * sizeof(T)));
    HIPCHK(hipMalloc(&d_stage_recv, slot_elems * sizeof(T) * sh_world));
    stage_slot = slot_elems;
    return EKF_OK;
  }

  // the collective: slot g of d_stage_recv <- d_stage_send of rank g, ordered on stream `st`
  bool exchanges() const { return sh_world > 1 || sh_force; }
  int all_gather(size_t slot_elems, hipStream_t st) {
    if (!exchanges()) return EKF_OK;
    if (!sh_ag) FAIL(EKF_ERR_STATE, "sharded filter without an all-gather callback (ekf_shard_configure)");
    const int rc = sh_ag(sh_ctx, d_stage_send, d_stage_recv, slot_elems * sizeof(T), st);
    if (rc != 0) FAIL(EKF_ERR_DEVICE, "the all-gather callback reported a failure");
    return EKF_OK;
  }

  // rows [tab.start[g], +tab.count[g]) x columns [col0, col0 + ncols) of `buf` (row stride ldb): own range out,
  // everybody else's in
  int exchange_rows(T* buf, int ldb, const ShardTab& tab, int col0, int ncols, hipStream_t st, int kid) {
    if (!exchanges()) return EKF_OK;
    int maxrows = 0;
    for (int g = 0; g < sh_world; ++g) maxrows = std::max(maxrows, tab.count[g]);
    if (maxrows == 0) return EKF_OK;
    const size_t slot = (size_t)maxrows * ncols;
    int rc = ensure_stage(slot);
    if (rc) return rc;
    Scope sc(this, kid, st);
    const int own = tab.count[sh_rank];
    const int gx = std::max(1, std::min(8, (ncols / (int)(16 / sizeof(T)) + 255) / 256));
    if (own > 0)
      k_pack_rows<T><<<dim3(gx, std::min(own, 65535)), 256, 0, st>>>(buf, ldb, tab.start[sh_rank], own, col0, ncols, d_stage_send);
    rc = all_gather(slot, st);
    if (rc) return rc;
    k_unpack_rows<T><<<dim3(gx, std::min(maxrows, 65535), sh_world), 256, 0, st>>>(d_stage_recv, slot, ncols, buf, ldb, col0, tab);
    HIPCHK(hipGetLastError());
    return EKF_OK;
  }

  ShardTab row_tab() const {                               // state rows owned by every rank
    ShardTab t{sh_world, sh_rank, {}, {}};
    for (int g = 0; g < sh_world; ++g) {
      t.start[g] = row_of_feature(sh_fb[g]);
      t.count[g] = row_of_feature(sh_fb[g + 1]) - t.start[g];
    }
    return t;
  }
  ShardTab feature_tab() const {
    ShardTab t{sh_world, sh_rank, {}, {}};
    for (int g = 0; g < sh_world; ++g) { t.start[g] = sh_fb[g]; t.count[g] = sh_fb[g + 1] - sh_fb[g]; }
    return t;
  }

  // Tile list of a rank's row panel [p0, p0 + prows) x [0, npad_live) for the queued downdate launch (tri = 3): a tile
  // whose rows AND columns lie inside the own rows [r0, r1) is listed once (column tile <= row tile) and mirrored;
  // every other tile -- the ragged first / last row tile of the panel (it holds foreign rows), the columns of other
  // ranks -- is a plain tile.  Row tiles are relative to p0, column tiles absolute.
  int opt_shard_sym = 1;                                   // EKF_SHARD_SYM=0: the plain row panel (A/B)
  int dbg_sync = 0;                                        // EKF_DEBUG_SYNC (bisecting an ordering problem): device synchronisation at 1 the end of
                                                           // the sharded update, 2 the end of every chunk, 4 its start, 8 the end of the sharded predict
  int* d_panel_tiles = nullptr;
  size_t panel_tiles_cap = 0;
  int panel_ntiles = 0;
  std::vector<int> panel_key;
  int ensure_panel_tiles(int p0, int prows, int r0, int r1, int npad_live) {
    std::vector<int> key = {p0, prows, r0, r1, npad_live};
    if (key == panel_key) return EKF_OK;
    std::vector<int> tl;
    const int t0 = p0 / 128, nt = prows / 128, nc = npad_live / 128;
    auto interior = [&](int T_) { return 128 * T_ >= r0 && 128 * T_ + 128 <= r1; };
    for (int P = 0; P < nt; ++P) {
      const int Tg = t0 + P;
      for (int J = 0; J < nc; ++J) {
        if (interior(Tg) && interior(J)) {
          if (J > Tg) continue;                            // the mirror of (J, Tg)
          tl.push_back(J < Tg ? (P | kMirrorTile) : P);
          tl.push_back(J);
        } else {
          tl.push_back(P);
          tl.push_back(J);
        }
      }
    }
    HIPCHK(hipStreamSynchronize(stream));
    if (stream_b) HIPCHK(hipStreamSynchronize(stream_b));
    if (tl.size() > panel_tiles_cap) {
      if (d_panel_tiles) HIPCHK(hipFree(d_panel_tiles));
      d_panel_tiles = nullptr;
      HIPCHK(hipMalloc(&d_panel_tiles, tl.size() * sizeof(int)));
      panel_tiles_cap = tl.size();
    }
    HIPCHK(hipMemcpy(d_panel_tiles, tl.data(), tl.size() * sizeof(int), hipMemcpyHostToDevice));
    panel_ntiles = (int)tl.size() / 2;
    panel_key = key;
    return EKF_OK;
  }

  // Canonical tiles (bi >= bj) of the bf16x6 downdate that touch a row block with a valid row of this rank (the camera
  // block and the blocks of the own rows [r0, r1)), diagonal tiles first, then the plain path's super-tile order.  What
  // each tile reads and stores is decided in the kernel, row by row (k_syrk_bf16x6: cam, v_lo, v_hi).
  int* d_shard_syrk = nullptr;
  size_t shard_syrk_cap = 0;
  int shard_syrk_n = 0;
  std::vector<int> shard_syrk_key;
  int ensure_shard_syrk_list(int r0, int r1, int npad_live) {
    std::vector<int> key = {r0, r1, npad_live, camera_dim};
    if (key == shard_syrk_key) return EKF_OK;
    const int nt = npad_live / 128, SB = 8, ns = (nt + SB - 1) / SB;
    auto touched = [&](int b) { return b * 128 < camera_dim || (r1 > r0 && b * 128 < r1 && b * 128 + 128 > r0); };
    std::vector<int> tl;
    for (int i = 0; i < nt; ++i) if (touched(i)) { tl.push_back(i); tl.push_back(i); }
    for (int si = 0; si < ns; ++si)
      for (int sj = 0; sj <= si; ++sj)
        for (int i = si * SB; i < std::min(nt, (si + 1) * SB); ++i)
          for (int j = sj * SB; j < std::min(nt, (sj + 1) * SB); ++j)
            if (j < i && (touched(i) || touched(j))) { tl.push_back(i); tl.push_back(j); }
    HIPCHK(hipStreamSynchronize(stream));
    if (stream_b) HIPCHK(hipStreamSynchronize(stream_b));
    if (tl.size() > shard_syrk_cap) {
      if (d_shard_syrk) HIPCHK(hipFree(d_shard_syrk));
      d_shard_syrk = nullptr;
      HIPCHK(hipMalloc(&d_shard_syrk, std::max<size_t>(tl.size(), 2) * sizeof(int)));
      shard_syrk_cap = tl.size();
    }
    if (!tl.empty()) HIPCHK(hipMemcpy(d_shard_syrk, tl.data(), tl.size() * sizeof(int), hipMemcpyHostToDevice));
    shard_syrk_n = (int)tl.size() / 2;
    shard_syrk_key = key;
    return EKF_OK;
  }

  // Heaviest-first tile list of the triangular solve on a rank's row panel (64 x 128 tiles; the same order as the plain
  // path's list: column tile ntc - 1 first, every row tile of the panel per column tile).  Round 4: the sharded solve ran
  // as a plain 2-D grid of 128 x 128 tiles, whose static placement pairs the heavy tiles of a column on the same CUs:
  // 0.50 ms of solves per step at N = 1000 / world 1 against 0.16 on the plain path.
  int* d_shard_solve = nullptr;
  int shard_solve_rows = 0;                                // row tiles per column tile of the list
  std::vector<int> shard_solve_key;
  int shard_ntc_max() const { return ldy / 128; }
  // The list covers every column tile the workspace can hold (ldy / 128), last column tile first; a chunk of wt column tiles
  // starts at entry (ntc_max - wt) * rows (the entries are relative to the chunk: bj = wt - 1 .. 0), so the list depends on
  // the rows of the rank only -- it is rebuilt (streams drained) when the panel of the rank moves, never because M crossed a
  // multiple of 64.  Round 5: the 64-row tiles of the camera block (when the panel does not start at row 0)
  // and of the innovation block (rows npad_live ..) are in the SAME list, as row tiles relative to the panel's first row
  // (negative for the camera block): one queued launch per chunk instead of three launches, two of them a single
  // latency-bound tile row (0.31 -> 0.2 ms of solves per step at N = 1000 / world 1).
  int ensure_shard_solve_list(int p0, int prows, int npad_live, bool with_cam) {
    std::vector<int> key = {p0, prows, npad_live, with_cam ? 1 : 0, ldy};
    if (key == shard_solve_key) return EKF_OK;
    std::vector<int> rows;
    if (with_cam) { rows.push_back(-p0 / 64); rows.push_back(-p0 / 64 + 1); }
    for (int i = 0; i < prows / 64; ++i) rows.push_back(i);
    rows.push_back((npad_live - p0) / 64);
    rows.push_back((npad_live - p0) / 64 + 1);
    const int ntc = shard_ntc_max();
    std::vector<int> tl;
    tl.reserve((size_t)2 * rows.size() * ntc);
    for (int j = ntc - 1; j >= 0; --j)
      for (int bi : rows) { tl.push_back(bi); tl.push_back(j); }
    HIPCHK(hipStreamSynchronize(stream));
    if (stream_b) HIPCHK(hipStreamSynchronize(stream_b));
    if (d_shard_solve) HIPCHK(hipFree(d_shard_solve));
    d_shard_solve = nullptr;
    shard_solve_key.clear();
    HIPCHK(hipMalloc(&d_shard_solve, tl.size() * sizeof(int)));
    HIPCHK(hipMemcpy(d_shard_solve, tl.data(), tl.size() * sizeof(int), hipMemcpyHostToDevice));
    shard_solve_rows = (int)rows.size();
    shard_solve_key = key;
    return EKF_OK;
  }

  int check_ascending(const int* idx, int M) {
    for (int k = 1; k < M; ++k)
      if (idx[k - 1] >= idx[k]) FAIL(EKF_ERR_ARG, "sharded filter: measured indices must be strictly ascending");
    return EKF_OK;
  }
  // list positions [start, start + count) of every rank's features in an ascending measured list
  ShardTab list_tab(const int* idx, int M) const {
    ShardTab t{sh_world, sh_rank, {}, {}};
    int k = 0;
    for (int g = 0; g < sh_world; ++g) {
      while (k < M && idx[k] < sh_fb[g]) ++k;
      int e = k;
      while (e < M && idx[e] < sh_fb[g + 1]) ++e;
      t.start[g] = k;
      t.count[g] = e - k;
      k = e;
    }
    return t;
  }
  // all-gather of 2x2 blocks: tab counts FEATURES (list == nullptr: feature indices, else positions of `list`)
  int gather_sd(const ShardTab& tab, const int* list) {
    if (!exchanges()) return EKF_OK;
    int mx = 0;
    for (int g = 0; g < sh_world; ++g) mx = std::max(mx, tab.count[g]);
    if (mx == 0) return EKF_OK;
    const size_t slot = (size_t)4 * mx;
    int rc = ensure_stage(slot);
    if (rc) return rc;
    const int own = tab.count[sh_rank];
    if (own > 0) k_pack_sd<T><<<(4 * own + 255) / 256, 256, 0, stream>>>(d_Sd, list, tab.start[sh_rank], own, d_stage_send);
    rc = all_gather(slot, stream);
    if (rc) return rc;
    k_unpack_sd<T><<<dim3((4 * mx + 255) / 256, sh_world), 256, 0, stream>>>(d_stage_recv, slot, list, d_Sd, tab);
    HIPCHK(hipGetLastError());
    return EKF_OK;
  }
  // Sharded filter: the diagonal block of EVERY feature valid on this rank (a rank's Sigma holds valid rows for the camera
  // and its own features only; the map getters and the removal archive read a feature's own 3 x 3 / 6 x 6 block).  One
  // all-gather of 36 scalars per feature; the foreign blocks are written into the local Sigma (rows nobody else reads,
  // which the next gather of those rows overwrites).  A COLLECTIVE: every rank makes the call.
  int shard_sync_diag_blocks() {
    if (!sh_on || !exchanges() || N == 0) return EKF_OK;
    // repeated map getters between two filter steps cost ONE collective: every entry point that changes mu, Sigma or the
    // layout clears the flag (the C wrappers below), on every rank alike -- the getters stay collective calls all the same
    if (diag_synced) return EKF_OK;
    int rc = sync_layout();
    if (rc) return rc;
    const ShardTab tab = feature_tab();
    int mx = 0;
    for (int g = 0; g < sh_world; ++g) mx = std::max(mx, tab.count[g]);
    const size_t slot = (size_t)36 * mx;
    rc = ensure_stage(slot);
    if (rc) return rc;
    const int own = tab.count[sh_rank];
    if (own > 0)
      k_pack_diag<T><<<(36 * own + 255) / 256, 256, 0, stream>>>(S(), ld, d_pos, d_coding, tab.start[sh_rank], own, d_stage_send);
    rc = all_gather(slot, stream);
    if (rc) return rc;
    k_unpack_diag<T><<<dim3((36 * mx + 255) / 256, sh_world), 256, 0, stream>>>(d_stage_recv, slot, S(), ld, d_pos, d_coding, tab);
    HIPCHK(hipGetLastError());
    diag_synced = true;
    return EKF_OK;
  }
  // raw bytes of every rank (own_bytes <= slot_bytes each) -> host buffer of world x slot_bytes
  int gather_bytes_to_host(const void* d_own, size_t own_bytes, size_t slot_bytes, std::vector<unsigned char>& host) {
    const size_t slot = (slot_bytes + sizeof(T) - 1) / sizeof(T);
    int rc = ensure_stage(slot);
    if (rc) return rc;
    if (own_bytes) HIPCHK(hipMemcpyAsync(d_stage_send, d_own, own_bytes, hipMemcpyDeviceToDevice, stream));
    host.assign((size_t)sh_world * stage_bytes_of(slot), 0);
    if (exchanges()) {
      rc = all_gather(slot, stream);
      if (rc) return rc;
      HIPCHK(hipMemcpyAsync(host.data(), d_stage_recv, (size_t)sh_world * stage_bytes_of(slot), hipMemcpyDeviceToHost, stream));
    } else {
      HIPCHK(hipMemcpyAsync(host.data(), d_stage_send, stage_bytes_of(slot), hipMemcpyDeviceToHost, stream));
    }
    HIPCHK(hipStreamSynchronize(stream));
    return EKF_OK;
  }
  static size_t stage_bytes_of(size_t slot_elems) { return slot_elems * sizeof(T); }

  // W rows {camera, own} for the measured list resident in d_midx, the own rows of S (+ plane / padding rows), and the
  // all-gather of the rows of S ("reassemble S"); d_zz != nullptr: the innovation nu and the queue heads as well
  int shard_build_ws(const int* idx, int M, int plane, const T* d_zz, int* m_out, int* m_pad_out) {
    const int nb = NB();
    const int m = 2 * M + (plane ? 3 : 0), m_pad = round_up(m, nb);
    const int npad_live = round_up(n, nb);
    T* nu_row = d_W + (size_t)ldy * npad_live;
    if (w_zeroed_n != n) {
      HIPCHK(hipMemsetAsync(d_W + (size_t)n * ldy, 0, (size_t)(npad_live - n + nb) * ldy * sizeof(T), stream));
      // the pad rows [n, npad_live) of V are the B operand of every rank's downdate, but only the rank whose tile-padded
      // panel reaches them ever solves them: everywhere else they must be cleared when n shrinks, or rows gathered in
      // earlier frames would be subtracted into the zero padding of Sigma
      if (npad_live > n)
        HIPCHK(hipMemsetAsync(d_V + (size_t)n * ldy, 0, (size_t)(npad_live - n) * ldy * sizeof(T), stream));
      w_zeroed_n = n;
    }
    const int f0 = own_f0(), f1 = own_f1();
    const int r0 = row_of_feature(f0), r1 = row_of_feature(f1);
    ShardTab stab = list_tab(idx, M);
    const int k0 = stab.start[sh_rank], k1 = k0 + stab.count[sh_rank];
    for (int g = 0; g < sh_world; ++g) { stab.start[g] *= 2; stab.count[g] *= 2; }      // rows of S
    if (d_zz) {
      Scope sc(this, KID_INNOVATION);
      k_innovation<T><<<(std::max(m_pad, 64) + 255) / 256, 256, 0, stream>>>(d_zz, d_h, d_midx, M, plane, mu(), nu_row, m_pad,
                                                                          d_counters, N, d_status);
      counter_next = 0;
    }
    {
      Scope sc(this, KID_SIGMA_HT);                        // W rows {camera, own}
      constexpr int RB = 32;
      dim3 g1((m_pad / 2 + 255) / 256, (camera_dim + RB - 1) / RB);
      k_sigma_ht<T, RB><<<g1, 256, 0, stream>>>(S(), ld, n, d_Hc, d_Hf, d_pos, d_coding, d_midx, M, plane, d_W, ldy,
                                              m_pad, 0, camera_dim, N);
      if (r1 > r0) {
        bool fastk = false;
        if constexpr (kIsF32) {
          if ((size_t)n * m_pad >= ((size_t)1 << 22)) {        // the LDS-staged kernel of the plain path on the own rows (same sums)
            constexpr int RBf = 8;
            dim3 g2((m_pad / 2 + 127) / 128, (r1 - r0 + RBf - 1) / RBf);
            k_sigma_ht_fast<RBf><<<g2, 256, 0, stream>>>(S(), ld, r1, d_Hc, d_Hf, d_pos, d_coding, d_midx, M, plane, d_W, ldy, m_pad, N,
                                                         0, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, r0);
            fastk = true;
          }
        }
        if (!fastk) {
          dim3 g2((m_pad / 2 + 255) / 256, (r1 - r0 + RB - 1) / RB);
          k_sigma_ht<T, RB><<<g2, 256, 0, stream>>>(S(), ld, n, d_Hc, d_Hf, d_pos, d_coding, d_midx, M, plane, d_W, ldy,
                                                  m_pad, r0, r1, N);
        }
      }
    }
    {
      Scope sc(this, KID_INNOVATION_COV);                  // S rows of the own measured features + the plane / padding rows
      constexpr int KB = 8;
      dim3 grid((m_pad + 255) / 256, std::max(1, (k1 - k0 + KB - 1) / KB) + (m_pad - 2 * M + 7) / 8);
      k_innovation_cov<T, KB><<<grid, 256, 0, stream>>>(d_W, ldy, d_Hc, d_Hf, d_pos, d_coding, d_midx, M, plane,
                                                      T(sigma_pixel_2), T(0.00001), d_Y, m_pad, k0, k1,
                                                      static_cast<T*>(nullptr), N);
    }
    HIPCHK(hipGetLastError());
    int rc = exchange_rows(d_Y, ldy, stab, 0, m_pad, stream, KID_GATHER_S);     // "reassemble S"
    if (rc) return rc;
    *m_out = m;
    *m_pad_out = m_pad;
    return EKF_OK;
  }

  // 1-point RANSAC under sharding (vR.cpp:986-1034): hypothesis k needs W[:, 2k:2k+2] on the rows of the feature it
  // re-projects -- a rank evaluates EVERY hypothesis on the listed features it owns (its rows of W), the partial inlier
  // counts are all-gathered and summed; the inlier column of one hypothesis is gathered the same way (fetch_mask_column)
  int shard_ransac(const int* idx, int M, double thr, int* counts, unsigned char* inl, int* best) {
    int rc = check_ascending(idx, M);
    if (rc) return rc;
    int m = 0, m_pad = 0;
    {
      // W rows {camera, own} only (no S): the sigma_ht half of shard_build_ws
      const int nb = NB();
      m = 2 * M; m_pad = round_up(m, nb);
      const int npad_live = round_up(n, nb);
      if (w_zeroed_n != n) {
        HIPCHK(hipMemsetAsync(d_W + (size_t)n * ldy, 0, (size_t)(npad_live - n + nb) * ldy * sizeof(T), stream));
        if (npad_live > n)
          HIPCHK(hipMemsetAsync(d_V + (size_t)n * ldy, 0, (size_t)(npad_live - n) * ldy * sizeof(T), stream));
        w_zeroed_n = n;
      }
      const int r0 = row_of_feature(own_f0()), r1 = row_of_feature(own_f1());
      Scope sc(this, KID_SIGMA_HT);
      constexpr int RB = 32;
      dim3 g1((m_pad / 2 + 255) / 256, (camera_dim + RB - 1) / RB);
      k_sigma_ht<T, RB><<<g1, 256, 0, stream>>>(S(), ld, n, d_Hc, d_Hf, d_pos, d_coding, d_midx, M, 0, d_W, ldy, m_pad, 0,
                                              camera_dim, N);
      if (r1 > r0) {
        dim3 g2((m_pad / 2 + 255) / 256, (r1 - r0 + RB - 1) / RB);
        k_sigma_ht<T, RB><<<g2, 256, 0, stream>>>(S(), ld, n, d_Hc, d_Hf, d_pos, d_coding, d_midx, M, 0, d_W, ldy, m_pad,
                                                r0, r1, N);
      }
    }
    have_update = false;
    if (!d_ibuf) HIPCHK(hipMalloc(&d_ibuf, (size_t)std::max(capN, 1) * 3 * sizeof(int)));
    if ((size_t)M * M > rmask_bytes) {
      if (d_rmask) HIPCHK(hipFree(d_rmask));
      d_rmask = nullptr;
      HIPCHK(hipMalloc(&d_rmask, (size_t)M * M));
      rmask_bytes = (size_t)M * M;
    }
    const ShardTab lt = list_tab(idx, M);
    const int k0 = lt.start[sh_rank], kc = lt.count[sh_rank];
    {
      Scope sc(this, KID_MISC);
      if (kc > 0) {
        dim3 grid((M + 127) / 128, kc);
        k_ransac_eval<T><<<grid, 128, 0, stream>>>(mu(), d_W, ldy, d_Sd, d_h, d_z, d_pos, d_coding, d_midx, M, cam, T(thr),
                                                  d_rmask, k0);
      }
      k_ransac_count<<<(M + 127) / 128, 128, 0, stream>>>(d_rmask, M, d_ibuf, k0, k0 + kc);
    }
    HIPCHK(hipGetLastError());
    std::vector<unsigned char> host;
    rc = gather_bytes_to_host(d_ibuf, (size_t)M * sizeof(int), (size_t)M * sizeof(int), host);
    if (rc) return rc;
    const size_t slot_b = host.size() / sh_world;
    std::vector<int> cnt(M, 0);
    const int parts = exchanges() ? sh_world : 1;
    for (int g = 0; g < parts; ++g) {
      const int* p = reinterpret_cast<const int*>(host.data() + (size_t)g * slot_b);
      for (int k = 0; k < M; ++k) cnt[k] += p[k];
    }
    int b = 0;
    for (int k = 1; k < M; ++k) if (cnt[k] > cnt[b]) b = k;
    if (counts) for (int k = 0; k < M; ++k) counts[k] = cnt[k];
    if (best) *best = b;
    if (inl) return fetch_mask_column(b, idx, M, inl);
    return EKF_OK;
  }

  // column `sel` of the inlier mask of the last ekf_ransac_1point (rows = list positions); under sharding every rank
  // contributes the rows of its own listed features
  int fetch_mask_column(int sel, const int* idx, int M, unsigned char* out) {
    if (!sh_on && ransac_mask_host && ransac_mask_M == M) {       // it came back with the counts
      for (int k = 0; k < M; ++k) out[k] = ransac_mask_host[(size_t)k * M + sel];
      return EKF_OK;
    }
    if (!sh_on) {
      // a large mask stayed on the device: its column is packed there (a strided 2-D copy of M one-byte rows takes
      // milliseconds at M = 1000) and comes back through pinned memory
      if (!h_gate) HIPCHK(hipHostMalloc(&h_gate, (size_t)std::max(capN, 1) + 64, hipHostMallocDefault));
      k_pack_mask_col<<<(M + 255) / 256, 256, 0, stream>>>(d_rmask, M, sel, 0, M, static_cast<unsigned char*>(h_gate));
      HIPCHK(hipGetLastError());
      HIPCHK(hipStreamSynchronize(stream));
      memcpy(out, h_gate, (size_t)M);
      return EKF_OK;
    }
    const ShardTab lt = list_tab(idx, M);
    int mx = 0;
    for (int g = 0; g < sh_world; ++g) mx = std::max(mx, lt.count[g]);
    const int k0 = lt.start[sh_rank], kc = lt.count[sh_rank];
    const size_t slot_bytes = (size_t)std::max(mx, 1);
    int rc = ensure_stage((slot_bytes + sizeof(T) - 1) / sizeof(T));
    if (rc) return rc;
    unsigned char* d_col = reinterpret_cast<unsigned char*>(d_ibuf) + (size_t)M * sizeof(int);   // behind the counts
    if (kc > 0) k_pack_mask_col<<<(kc + 255) / 256, 256, 0, stream>>>(d_rmask, M, sel, k0, kc, d_col);
    HIPCHK(hipGetLastError());
    std::vector<unsigned char> host;
    rc = gather_bytes_to_host(d_col, (size_t)kc, slot_bytes, host);
    if (rc) return rc;
    const size_t slot_b = host.size() / sh_world;
    if (exchanges()) {
      for (int g = 0; g < sh_world; ++g) memcpy(out + lt.start[g], host.data() + (size_t)g * slot_b, (size_t)lt.count[g]);
    } else {
      memcpy(out, host.data(), (size_t)M);
    }
    return EKF_OK;
  }

  int shard_configure(int rank, int world, ekf_allgather_fn fn, void* ctx) override {
    if (world < 1 || world > kMaxWorld || rank < 0 || rank >= world) FAIL(EKF_ERR_ARG, "bad rank / world");
    if (world > 1 && !fn) FAIL(EKF_ERR_ARG, "world > 1 needs an all-gather callback");
    HIPCHK(hipSetDevice(device));
    sh_rank = rank; sh_world = world; sh_ag = fn; sh_ctx = ctx;
    sh_on = true;
    if (const char* e = getenv("EKF_SHARD_FORCE_COLLECTIVE")) sh_force = (atoi(e) != 0) && fn != nullptr;
    if (const char* e = getenv("EKF_SHARD_SYM")) opt_shard_sym = atoi(e);
    if (const char* e = getenv("EKF_DEBUG_SYNC")) dbg_sync = atoi(e);
    // From four ranks on the REPLICATED chain, not the rank's share of the GEMMs, is what a step of a long list waits for
    // (DESIGN 6): steps with many rounds of trailing blocks then go back to the 64 x 64 tile kernel, whose occupancy makes the
    // chain's own kernel time 21 % shorter at N = 4000 (13.6 -> 10.7 ms at world 1, profiles/r6_shard_world1_nccl_n4000*.json) at
    // the price of the exposed factor -- on one or two ranks the chain is hidden beside the downdate and the fused launch wins.
    if (world >= 4 && !getenv("EKF_TD_MAX_BLOCKS")) td_max_blocks = 256;
    if (!stream_g) {
      HIPCHK(hipStreamCreateWithFlags(&stream_g, hipStreamNonBlocking));
      for (auto& e : ev_gath) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      HIPCHK(hipEventCreateWithFlags(&ev_g, hipEventDisableTiming));
    }
    // every rank was built by the same calls, so all of Sigma is valid everywhere right now: any partition will do
    partition_by_rows();
    have_meas = false;
    return EKF_OK;
  }

  int shard_info(ekf_shard_info* o) override {
    if (!o) FAIL(EKF_ERR_ARG, "null info");
    memset(o, 0, sizeof(*o));
    o->rank = sh_rank; o->world = sh_world; o->N = N; o->state_dim = n;
    if (sh_on) {
      o->f_begin = own_f0(); o->f_end = own_f1();
      o->row_begin = row_of_feature(own_f0()); o->row_end = row_of_feature(own_f1());
      int mx = 0;
      for (int g = 0; g < sh_world; ++g) mx = std::max(mx, row_of_feature(sh_fb[g + 1]) - row_of_feature(sh_fb[g]));
      o->max_rows_any_rank = mx;
    } else {
      o->f_end = N; o->row_begin = camera_dim; o->row_end = n; o->max_rows_any_rank = n - camera_dim;
    }
    o->rebalances = sh_rebalances;
    return EKF_OK;
  }

  // ownership follows the resize operations every rank executes identically
  void shard_after_add() { if (sh_on) sh_fb[sh_world] = N; }
  void shard_after_compact(const std::vector<char>& rm) {
    if (!sh_on) return;
    std::vector<int> kept_before(rm.size() + 1, 0);
    for (size_t i = 0; i < rm.size(); ++i) kept_before[i + 1] = kept_before[i] + (rm[i] ? 0 : 1);
    for (int g = 0; g <= sh_world; ++g) sh_fb[g] = kept_before[std::min<size_t>(sh_fb[g], rm.size())];
  }

  // Makes every row of Sigma valid on every rank (all-gather of the row panels, in column blocks) and re-partitions
  // the features so that the rows are balanced again.  Also the sync point before reading foreign rows of Sigma.
  int shard_rebalance() override {
    HIPCHK(hipSetDevice(device));
    if (!sh_on) FAIL(EKF_ERR_STATE, "ekf_shard_configure first");
    if (exchanges() && N > 0) {
      const ShardTab tab = row_tab();
      const int cw = 4096;                                 // column block: bounds the staging buffers
      const int vec = 16 / (int)sizeof(T);
      for (int c0 = 0; c0 < n; c0 += cw) {
        const int nc = round_up(std::min(cw, n - c0), vec);
        int rc = exchange_rows(S(), ld, tab, c0, nc, stream, KID_GATHER_SIGMA);
        if (rc) return rc;
      }
    }
    partition_by_rows();
    ++sh_rebalances;
    have_meas = false;
    return EKF_OK;
  }

  bool shard_needs_rebalance() const {
    if (sh_world == 1 || N == 0) return false;
    int mx = 0;
    for (int g = 0; g < sh_world; ++g) mx = std::max(mx, row_of_feature(sh_fb[g + 1]) - row_of_feature(sh_fb[g]));
    const double mean = double(n - camera_dim) / sh_world;
    return mx > sh_imbalance_limit * mean + 6.0;
  }

  int shard_predict(const void* tc, const void* rc_, int vcontrol) {
    if (shard_needs_rebalance()) { int rr = shard_rebalance(); if (rr) return rr; }
    MotionArgs a;
    a.dT = dT;
    const T* t = static_cast<const T*>(tc);
    const T* r = static_cast<const T*>(rc_);
    for (int i = 0; i < 3; ++i) { a.t_ctl[i] = t ? double(t[i]) : 0.0; a.r_ctl[i] = r ? double(r[i]) : 0.0; }
    for (int i = 0; i < 6; ++i) a.vdiag[i] = vcontrol ? vmax[i] : double(T(vmax[i]) * T(2));
    int rc = sync_layout();
    if (rc) return rc;
    { Scope sc(this, KID_PREDICT_CAMERA); k_predict_camera<T><<<1, 64, 0, stream>>>(mu(), d_scr, a); }
    have_motion = true;
    {
      // rows 0..12 are replicated; the column strip of foreign rows works on stale data nobody reads
      Scope sc(this, KID_PROPAGATE_STRIPS);
      k_strip_congruence<T, 13><<<(2 * n + 255) / 256, 256, 0, stream>>>(S(), ld, n, 0, d_scr + SCR_FT, d_scr + SCR_Q);
    }
    if (opt_feature_noise > 0.0 && n > camera_dim)
      k_inflate_diagonal<T><<<(n - camera_dim + 255) / 256, 256, 0, stream>>>(S(), ld, camera_dim, n, T(opt_feature_noise));
    const int f0 = own_f0(), f1 = own_f1();
    if (f1 > f0) {
      Scope sc(this, KID_MEASURE);
      k_measure<T><<<(f1 - f0 + 63) / 64, 64, 0, stream>>>(mu(), d_pos, d_coding, f0, f1, cam, d_h, d_Hc, d_Hf, d_flags);
    }
    HIPCHK(hipGetLastError());
    if (exchanges() && N > 0) {                            // "reassemble H": h, compact Jacobians, flags of every feature
      const ShardTab tab = feature_tab();
      int mx = 0;
      for (int g = 0; g < sh_world; ++g) mx = std::max(mx, tab.count[g]);
      const size_t slot = (size_t)mx * kFeatRec;
      rc = ensure_stage(slot);
      if (rc) return rc;
      Scope sc(this, KID_GATHER_H);
      if (f1 > f0)
        k_pack_features<T><<<((f1 - f0) * kFeatRec + 255) / 256, 256, 0, stream>>>(d_h, d_Hc, d_Hf, d_flags, f0, f1 - f0, d_stage_send);
      rc = all_gather(slot, stream);
      if (rc) return rc;
      k_unpack_features<T><<<dim3((mx * kFeatRec + 255) / 256, sh_world), 256, 0, stream>>>(d_stage_recv, slot, d_h, d_Hc, d_Hf, d_flags, tab);
      HIPCHK(hipGetLastError());
    }
    have_update = false;
    have_meas = true;
    have_sd = false;
    if (dbg_sync & 8) HIPCHK(hipDeviceSynchronize());
    return launch_blur();                                  // predicted blur of every template (replicated, like the templates)
  }

  // The sharded EKF update block for the measured list `idx` (host, strictly ascending), z resident on the device.
  int shard_update(const void* dz, const int* idx, int M, int plane) override {
    HIPCHK(hipSetDevice(device));
    if (!sh_on) FAIL(EKF_ERR_STATE, "ekf_shard_configure first");
    if (M < 0 || M > N) FAIL(EKF_ERR_ARG, "M out of range");
    if (M == 0 && !plane) return EKF_OK;
    if (!have_meas) FAIL(EKF_ERR_STATE, "the sharded update needs the h / H of the sharded predict");
    if (M > 0 && (!dz || !idx)) FAIL(EKF_ERR_ARG, "z / indices are NULL");
    for (int k = 0; k < M; ++k) {
      if (idx[k] < 0 || idx[k] >= N) FAIL(EKF_ERR_ARG, "feature index out of range");
      if (k > 0 && idx[k - 1] >= idx[k]) FAIL(EKF_ERR_ARG, "measured indices must be strictly ascending");
    }
    if (M > 0 && ((int)sh_list.size() != M || memcmp(sh_list.data(), idx, (size_t)M * sizeof(int)) != 0)) {
      sh_list.assign(idx, idx + M);
      HIPCHK(hipMemcpyAsync(d_midx, sh_list.data(), (size_t)M * sizeof(int), hipMemcpyHostToDevice, stream));
    }
    const T* d_zz = static_cast<const T*>(dz);
    const int nb = NB();
    int m = 0, m_pad = 0;
    const int npad_live = round_up(n, nb);
    if (dbg_sync & 4) HIPCHK(hipDeviceSynchronize());
    // nu (replicated), W rows {camera, own}, own rows of S, "reassemble S"
    int rc = shard_build_ws(idx, M, plane, d_zz, &m, &m_pad);
    if (rc) return rc;
    // own state rows, and the tile-padded panel [p0, p0 + prows) the tile GEMMs run on: it covers the own rows and,
    // at its ends, a few foreign ones (whose results nobody reads and the next gather overwrites); it never reaches
    // past the padded live block (row npad_live of W / V is the nu / y row)
    const int f0 = own_f0(), f1 = own_f1();
    const int r0 = row_of_feature(f0), r1 = row_of_feature(f1);
    // (round 3: the panel starts on a tile boundary, so that the tiles that lie INSIDE the own rows on both sides -- the
    // own x own block of Sigma minus its ragged ends -- are computed once, as lower tiles, and mirrored: both rows of a
    // mirrored pair are then owned.  Per rank the downdate is rows x m x (2 n - rows) flop instead of 2 rows n m.)
    int p0 = 0, prows = 0;
    if (r1 > r0) {
      p0 = r0 / nb * nb;
      prows = round_up(r1, nb) - p0;
    }
    struct Rows { int r0, count; };
    const Rows ranges[3] = {{0, p0 > 0 || prows == 0 ? nb : 0}, {p0, prows}, {npad_live, nb}};
    bool sym_panel = false;
    if constexpr (kIsF32) sym_panel = opt_mfma && nb == 128 && prows > 0 && opt_shard_sym;
    if (sym_panel) { rc = ensure_panel_tiles(p0, prows, r0, r1, npad_live); if (rc) return rc; }

    // Replicated chain in column chunks; the rank's share of every chunk beside it:
    //   second stream (CU-masked):  solve V_g rows, W update rows, ... downdate of the PREVIOUS chunk
    //   gather stream:              all-gather of the own rows of V_g (needs the solve; the downdate needs it)
    // so a gather travels while the next chunk is being solved, and only the last chunk's solve -> gather -> downdate
    // is exposed after the chain.  With 1 / world of the GEMM work per rank many narrow chunks are affordable: the
    // exposed tail shrinks with the width of the last one.
    const int nsteps = m_pad / nb;
    int cend[8];
    int nchunks;
    {
      const bool pipe = (opt_pipeline < 0) ? (nsteps >= 4) : (opt_pipeline != 0);
      if (!pipe || !stream_b || !stream_g) {
        cend[0] = nsteps;
        nchunks = 1;
      } else if (env_nchunks > 0 && env_chunks[env_nchunks - 1] == nsteps) {
        for (int g = 0; g < env_nchunks; ++g) cend[g] = env_chunks[g];
        nchunks = env_nchunks;
      } else if (sh_world <= 1 && nsteps >= 8) {
        nchunks = plan_chunks(nsteps, cend, true);           // one rank: the plain path's plan with an earlier first cut
      } else {
        // chunk count by world size: every chunk costs a pass over the remaining columns of W and re-reads the Sigma
        // panel, and a rank's share of that work is 1 / world -- two ranks afford 4 chunks, four and more 8
        const int cap = sh_world <= 1 ? 3 : (sh_world <= 2 ? 4 : 8);
        const int want = std::min(cap, std::max(2, (nsteps + 1) / 2));
        nchunks = 0;
        int prev = 0;
        for (int g = 0; g < want; ++g) {
          int e = (int)(((long long)nsteps * (g + 1) + want - 1) / want);
          if (g + 1 == want) e = nsteps;
          if (e > prev) { cend[nchunks++] = e; prev = e; }
        }
      }
    }
    ChunkTab tab{nchunks, {}};
    int strip_rows = 0;
    for (int g = 0; g < nchunks; ++g) {
      tab.end[g] = cend[g] * nb;
      strip_rows = std::max(strip_rows, (cend[g] - (g ? cend[g - 1] : 0)) * nb);
    }
    T* Y = d_Y;
    T* Zs = d_Y + (size_t)m_pad * ldy;
    { Scope sc(this, KID_MISC);
      dim3 grid((m_pad + 255) / 256, strip_rows);
      k_set_identity_strip<T><<<grid, 256, 0, stream>>>(Zs, ldy, m_pad, tab); }
    const ShardTab rtab = row_tab();
    bool shard_split = false;
    if constexpr (kIsF32)                                  // (the plain path's rule: the lower tiles of the WHOLE matrix fill the chip)
      shard_split = opt_split_bf16 && opt_mfma && nb == 128 && (npad_live / 128) * (npad_live / 128 + 1) / 2 >= num_cus;
    if (shard_split) { rc = ensure_shard_syrk_list(r0, r1, npad_live); if (rc) return rc; }
    bool sh_row_pending = false;                           // (sequential form) the innovation row still waits for chunk [c0, c1)
    auto row_update_alone = [&](int c0, int c1, hipStream_t ss) {
      if constexpr (kIsF32) {
        Scope sc(this, KID_WUPDATE, ss);
        ++launch_cnt[EKF_LAUNCH_ROW_GEMV];
        k_innov_row_update<<<(m_pad - c1 + 63) / 64, 64, 0, ss>>>(d_V + (size_t)npad_live * ldy + c0, d_Y + (size_t)c1 * ldy + c0, ldy,
                                                                  d_W + (size_t)npad_live * ldy + c1, m_pad - c1, c1 - c0);
      }
    };
    auto downdate_chunk = [&](int c0, int c1, hipStream_t ss) -> int {
      if (sh_row_pending && !(kIsF32 && shard_split && counter_next + 8 <= kQueueCounters)) {
        row_update_alone(c0, c1, ss);
        sh_row_pending = false;
      }
      if constexpr (kIsF32) {
        if (shard_split && counter_next + 8 <= kQueueCounters) {
          // EKF_OPT_SPLIT_BF16: V_g (every row: the gather is done) -> plane image, then ONE launch over the canonical tiles
          // that touch the camera block or an own block; each element pair is the same sum as on the plain path
          if (!d_Vimg) HIPCHK(hipMalloc(&d_Vimg, (size_t)(n_pad + 128) * ldy * 6));
          {
            Scope sc(this, KID_MISC, ss);
            dim3 grid(npad_live / 128, (c1 - c0) / 16);
            k_split_image<<<grid, 256, 0, ss>>>(d_V, ldy, npad_live, c0, c1 - c0, d_Vimg, ldy / 16);
          }
          if (shard_syrk_n > 0) {
            Scope sc(this, KID_DOWNDATE, ss);
            if (sc.on) prof_work[KID_DOWNDATE] += 2.0 * 128 * 128 * shard_syrk_n * double(std::min(c1, m) - std::min(c0, m));
            Syrk6Args a{d_Vimg, ldy / 16, c0 / 16, (c1 - c0) / 16, S(), ld, d_shard_syrk, shard_syrk_n, d_counters + counter_next,
                        camera_dim, r0, r1};
            a.stag_half = opt_syrk_stag_half; a.stag_mod4 = opt_syrk_stag_mod4;
            if (sh_row_pending) {
              a.ry = d_V + (size_t)npad_live * ldy + c0; a.rL = d_Y + (size_t)c1 * ldy + c0; a.rldl = ldy;
              a.rnu = d_W + (size_t)npad_live * ldy + c1; a.rcols = m_pad - c1; a.rK = c1 - c0; a.nrider = (m_pad - c1 + 255) / 256;
              sh_row_pending = false;
              ++launch_cnt[EKF_LAUNCH_ROW_RIDER];
            }
            ++launch_cnt[EKF_LAUNCH_DOWNDATE_BF16X6];
            counter_next += 8;
            const int wgs = 2 * ((ss == stream_b) ? (num_cus - reserved_cus) : num_cus);
            k_syrk_bf16x6<0><<<a.nrider + std::min(shard_syrk_n, wgs), 256, 0, ss>>>(a);
          }
          if (sh_row_pending) { row_update_alone(c0, c1, ss); sh_row_pending = false; }
          return EKF_OK;
        }
      }
      for (int q = 0; q < 2; ++q) {                        // Sigma[rows, :] -= V_g[rows] V_g^T: camera tile, own panel
        const Rows& rr = ranges[q];
        if (rr.count == 0) continue;
        Scope sc(this, KID_DOWNDATE, ss);
        ++launch_cnt[EKF_LAUNCH_DOWNDATE_F32];
        if (q == 1 && sym_panel && counter_next + 8 <= kQueueCounters) {
          // the own panel as ONE queued launch over the listed tiles: interior x interior lower tiles + mirror, the rest plain
          if (sc.on) prof_work[KID_DOWNDATE] += 2.0 * 128 * 128 * panel_ntiles * double(std::min(c1, m) - std::min(c0, m));
          gemm<ROLE_DOWNDATE, false>(d_V + (size_t)rr.r0 * ldy + c0, ldy, d_V + c0, ldy, S() + (size_t)rr.r0 * ld, ld, rr.count,
                                     npad_live, c1 - c0, T(-1), T(1), 3, rr.r0, 0, 0, 0, ss, d_panel_tiles, panel_ntiles);
          continue;
        }
        if (sc.on) prof_work[KID_DOWNDATE] += 2.0 * rr.count * double(n) * (std::min(c1, m) - std::min(c0, m));
        gemm<ROLE_DOWNDATE, false>(d_V + (size_t)rr.r0 * ldy + c0, ldy, d_V + c0, ldy, S() + (size_t)rr.r0 * ld, ld, rr.count,
                                   npad_live, c1 - c0, T(-1), T(1), 0, 0, 0, 0, 0, ss);
      }
      return EKF_OK;
    };
    // Round 5: the SEQUENTIAL form of the chunked update on a rank too (EKF_OPT_W_RECOMPUTE): after the downdate of chunk g
    // the rank re-evaluates ITS rows of W for chunk g + 1 from its downdated rows of Sigma (local: a rank holds every
    // column of its rows), instead of the right-looking GEMM update of all later columns (0.40 of 1.56 ms per step at
    // N = 1000 / world 1, 8.7 of 44 ms at N = 4000).  Per chunk the rank's second stream then runs solve -> gather of V_g
    // -> downdate -> W' in series; that order is off the critical path as soon as the replicated chain is what a step
    // waits for, which it is from two ranks on.
    bool sh_rec = false;
    if constexpr (kIsF32) sh_rec = opt_mfma && opt_wrecompute && nb == 128 && nchunks > 1;
    int step = 0;
    bool side_busy = false;
    sf_now = false;                                        // (the fused block step is the plain path's)
    const bool dchain = dist_chain_ok(nsteps);             // the factorisation distributed over the ranks (see dist_chain_steps)
    const bool pchain = !dchain && chain_persistent_ok() && nb == 128 && nsteps >= 2;
    if (pchain) { rc = chain_begin_update(nsteps, nchunks, cend); if (rc) return rc; }
    if (dchain) { rc = ensure_dist_plan(nsteps, nchunks, cend); if (rc) return rc; }
    chain_diag_ahead = -1;
    chain_pending.step = -1;
    td_nblk = (td_nblk == nsteps) ? td_nblk : 0;
    if (!pchain && !dchain && trail_diag_ok() && nb == 128 && nsteps >= 2) { rc = ensure_trail_diag_lists(nsteps, nchunks, cend); if (rc) return rc; }
    int pend_c0 = -1, pend_c1 = -1, pend_g = -1;           // overlapped chunk whose downdate is still to be issued
    for (int gi = 0; gi < nchunks; ++gi) {
      const int c0 = step * nb, c1 = cend[gi] * nb, width = c1 - c0;
      if (pchain) { rc = chain_launch(gi, m, gi == 0, stream); if (rc) return rc; }
      else if (dchain) { rc = dist_chain_steps(step, cend[gi], m, m_pad, stream); if (rc) return rc; }
      else chain_steps(step, cend[gi], c0, c1, m, m_pad, stream, false, opt_chain_defer && gi + 1 < nchunks);
      step = cend[gi];
      const bool overlap = (gi + 1 < nchunks);
      hipStream_t ss = overlap ? stream_b : stream;
      if (overlap) {
        HIPCHK(hipEventRecord(ev_chain[gi], stream));
        HIPCHK(hipStreamWaitEvent(stream_b, ev_chain[gi], 0));
        side_busy = true;
      } else if (side_busy) {
        // the last chunk runs on the main stream: it needs every earlier W update (second stream) first
        HIPCHK(hipEventRecord(ev_b, stream_b));
        HIPCHK(hipStreamWaitEvent(stream, ev_b, 0));
      }
      bool solved = false;
      if constexpr (kIsF32) {
        if (opt_mfma && nb == 128 && prows > 0 && counter_next + 8 <= kQueueCounters) {
          // camera block, own panel and innovation block as ONE queued launch of 64 x 128 tiles, heaviest column tiles first
          // (row tiles relative to the panel's first row; any tile shape adds the same terms in the same order)
          Scope sc(this, KID_SOLVE, ss);
          solve_s2_now = want_solve_s2(width, npad_live);
          rc = ensure_shard_solve_list(p0, prows, npad_live, ranges[0].count > 0);
          if (rc) return rc;
          const int wt = width / 128;
          const size_t off = (size_t)p0 * ldy;
          const int* list = d_shard_solve + 2 * (shard_ntc_max() - wt) * shard_solve_rows;
          gemm<ROLE_SOLVE, true, 64, 128>(d_W + off + c0, ldy, Zs + c0, ldy, d_V + off + c0, ldy, prows, width, width, T(1),
                                          T(0), 0, 0, 0, 1, 0, ss, list, wt * shard_solve_rows);
          solved = true;
        }
      }
      for (int q = 0; q < 3 && !solved; ++q) {
        const Rows& rr = ranges[q];
        if (rr.count == 0) continue;
        const size_t off = (size_t)rr.r0 * ldy;
        Scope sc(this, KID_SOLVE, ss);
        solve_s2_now = want_solve_s2(width, npad_live);
        gemm<ROLE_SOLVE, true>(d_W + off + c0, ldy, Zs + c0, ldy, d_V + off + c0, ldy, rr.count, width, width, T(1), T(0),
                               0, 0, 0, 1, 0, ss);
      }
      if (sh_rec) {
        if constexpr (kIsF32) {
          // solve -> gather -> innovation row -> downdate -> W' of the next chunk, in this order on the chunk's stream
          if (overlap) {
            HIPCHK(hipEventRecord(ev_solve[gi], stream_b));
            HIPCHK(hipStreamWaitEvent(stream_g, ev_solve[gi], 0));
            rc = exchange_rows(d_V, ldy, rtab, c0, width, stream_g, KID_GATHER_V);
            if (rc) return rc;
            HIPCHK(hipEventRecord(ev_gath[gi], stream_g));
            HIPCHK(hipStreamWaitEvent(stream_b, ev_gath[gi], 0));
          } else {
            if (side_busy) {
              HIPCHK(hipEventRecord(ev_g, stream_g));
              HIPCHK(hipStreamWaitEvent(stream, ev_g, 0));
            }
            rc = exchange_rows(d_V, ldy, rtab, c0, width, stream, KID_GATHER_V);
            if (rc) return rc;
          }
          // nu^T[c1:] -= y_g^T L[c1:, g]^T (replicated, every rank): rides in the downdate's launch when there is one
          sh_row_pending = c1 < m_pad;
          rc = downdate_chunk(c0, c1, ss);
          if (rc) return rc;
          if (gi + 1 < nchunks) {
            Scope sc(this, KID_SIGMA_HT, ss);                // W'[rows, c1:c2) = Sigma'[rows, :] H^T, rows = camera + own
            ++launch_cnt[EKF_LAUNCH_W_RECOMPUTE];
            const int s0 = c1 / 2, s1 = cend[gi + 1] * nb / 2;
            constexpr int RB = 8;
            dim3 g1((s1 - s0 + 127) / 128, (camera_dim + RB - 1) / RB);
            k_sigma_ht_fast<RB><<<g1, 256, 0, ss>>>(S(), ld, camera_dim, d_Hc, d_Hf, d_pos, d_coding, d_midx, M, plane, d_W, ldy, m_pad, N,
                                                    s0, s1, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0);
            if (r1 > r0) {
              dim3 g2((s1 - s0 + 127) / 128, (r1 - r0 + RB - 1) / RB);
              k_sigma_ht_fast<RB><<<g2, 256, 0, ss>>>(S(), ld, r1, d_Hc, d_Hf, d_pos, d_coding, d_midx, M, plane, d_W, ldy, m_pad, N,
                                                      s0, s1, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, r0);
            }
          }
          if (dbg_sync & 2) HIPCHK(hipDeviceSynchronize());
          continue;
        }
      }
      if (overlap) {
        // own rows of V_g are final: their gather starts now, on the gather stream ...
        HIPCHK(hipEventRecord(ev_solve[gi], stream_b));
        HIPCHK(hipStreamWaitEvent(stream_g, ev_solve[gi], 0));
        rc = exchange_rows(d_V, ldy, rtab, c0, width, stream_g, KID_GATHER_V);
        if (rc) return rc;
        HIPCHK(hipEventRecord(ev_gath[gi], stream_g));
      }
      if (c1 < m_pad) {                                    // ... while the W update (needs the OWN rows of V_g only) goes on
        for (const Rows& rr : ranges) {
          if (rr.count == 0) continue;
          const size_t off = (size_t)rr.r0 * ldy;
          Scope sc(this, KID_WUPDATE, ss);
          ++launch_cnt[EKF_LAUNCH_W_UPDATE_GEMM];
          gemm<ROLE_WUPDATE, false>(d_V + off + c0, ldy, Y + (size_t)c1 * ldy + c0, ldy, d_W + off + c1, ldy, rr.count,
                                    m_pad - c1, width, T(-1), T(1), 0, 0, 0, 0, 0, ss);
        }
      }
      if (pend_g >= 0) {                                   // the downdate of the previous chunk, behind this chunk's solve
        HIPCHK(hipStreamWaitEvent(stream_b, ev_gath[pend_g], 0));
        rc = downdate_chunk(pend_c0, pend_c1, stream_b);
        if (rc) return rc;
        pend_g = -1;
      }
      if (overlap) {
        pend_g = gi; pend_c0 = c0; pend_c1 = c1;
      } else {
        // last chunk, everything on the main stream: gather (the staging buffers are free once the gather stream has
        // drained), then its downdate behind every earlier one
        if (side_busy) {
          HIPCHK(hipEventRecord(ev_g, stream_g));
          HIPCHK(hipStreamWaitEvent(stream, ev_g, 0));
        }
        rc = exchange_rows(d_V, ldy, rtab, c0, width, stream, KID_GATHER_V);
        if (rc) return rc;
        if (side_busy) {
          HIPCHK(hipEventRecord(ev_b, stream_b));
          HIPCHK(hipStreamWaitEvent(stream, ev_b, 0));
        }
        rc = downdate_chunk(c0, c1, stream);
        if (rc) return rc;
      }
    }
    HIPCHK(hipGetLastError());
    last_nchunks = nchunks;
    last_recompute = sh_rec;
    for (int g = 0; g < nchunks; ++g) last_cend[g] = cend[g];
    {
      Scope sc(this, KID_STATE_UPDATE);                    // mu is replicated: every rank adds V y over all rows
      k_state_update<T><<<(n + 7) / 8, 512, 0, stream>>>(mu(), d_V, ldy, n, d_V + (size_t)npad_live * ldy, m_pad, d_scr + SCR_QN);
    }
    {
      Scope sc(this, KID_NORMALIZE);
      k_strip_congruence<T, 4><<<(2 * n + 255) / 256, 256, 0, stream>>>(S(), ld, n, 3, d_scr + SCR_QN,
                                                                       static_cast<const T*>(nullptr));
    }
    HIPCHK(hipGetLastError());
    last_m = m; last_m_pad = m_pad; last_n = n;
    have_update = true;
    have_meas = false;
    ++frame_seq;
    if (dbg_sync & 1) HIPCHK(hipDeviceSynchronize());
    return EKF_OK;
  }

  // convert2XYZ_ifLinear(All) under sharding: the linearity index needs Sigma(rho, rho) -- known to the owner --
  // so every rank tests its own features and the flags are gathered before the (identical) compaction pass
  int shard_convert(int index, bool all) {
    if (!all && (index < 0 || index >= N)) { err = "feature index out of range"; return -EKF_ERR_ARG; }
    if (N == 0) return 0;
    int rc = sync_layout();
    if (rc) return -rc;
    k_linearity<T><<<(N + 127) / 128, 128, 0, stream>>>(mu(), S(), ld, d_pos, d_coding, N, d_cflag, d_Jy, d_Yxyz, 0);
    if (exchanges()) {
      const ShardTab tab = feature_tab();
      int mx = 0;
      for (int g = 0; g < sh_world; ++g) mx = std::max(mx, tab.count[g]);
      const size_t slot = (size_t)mx;
      rc = ensure_stage(slot);
      if (rc) return -rc;
      const int cnt = own_f1() - own_f0();
      if (cnt > 0) k_pack_flags<T><<<(cnt + 255) / 256, 256, 0, stream>>>(d_cflag, own_f0(), cnt, d_stage_send);
      rc = all_gather(slot, stream);
      if (rc) return -rc;
      k_unpack_flags<T><<<dim3((mx + 255) / 256, sh_world), 256, 0, stream>>>(d_stage_recv, slot, d_cflag, tab);
    }
    std::vector<unsigned char> fl(N);
    if (rb_add(fl.data(), d_cflag, N) != EKF_OK || rb_finish(false) != EKF_OK) { err = "linearity flags D2H failed"; return -EKF_ERR_DEVICE; }
    std::vector<char> rm(N, 0), cv(N, 0);
    int cnt = 0;
    for (int i = 0; i < N; ++i) {
      if (!all && i != index) continue;
      if (fl[i] && coding[i] == 0) { cv[i] = 1; ++cnt; }
    }
    if (cnt == 0) return 0;
    rc = compact(rm, cv);
    if (rc) return -rc;
    return cnt;
  }

  int profile_read(int kid, double* ms, long long* cnt) override {
    if (kid < 0 || kid >= KID_COUNT) FAIL(EKF_ERR_ARG, "kernel id out of range");
    hipSetDevice(device);
    resolve_profile();
    if (ms) *ms = prof_ms[kid];
    if (cnt) *cnt = prof_cnt[kid];
    return EKF_OK;
  }
  bool last_recompute = false;
  int chunk_plan(int* ends, int max_chunks, int* block, int* wrec) override {
    if (block) *block = NB();
    if (wrec) *wrec = last_recompute ? 1 : 0;
    if (last_cend[0] == 0) return 0;                          // no update yet
    for (int g = 0; g < last_nchunks && g < max_chunks; ++g) if (ends) ends[g] = last_cend[g];
    return last_nchunks;
  }
  int profile_work(int kid, double* flop) override {
    if (kid < 0 || kid >= KID_COUNT) FAIL(EKF_ERR_ARG, "kernel id out of range");
    if (flop) *flop = prof_work[kid];
    return EKF_OK;
  }
  int profile_reset() override {
    hipSetDevice(device);
    resolve_profile();
    frame_seq = 0;
    memset(prof_ms, 0, sizeof(prof_ms));
    memset(prof_cnt, 0, sizeof(prof_cnt));
    memset(prof_work, 0, sizeof(prof_work));
    memset(launch_cnt, 0, sizeof(launch_cnt));
    return EKF_OK;
  }
};

}  // namespace ekf

// =============================================================================================
// C ABI
// =============================================================================================
using ekf::FilterBase;

struct ekf_filter { FilterBase* impl; };

extern "C" {

void ekf_config_default(ekf_config* c) {
  if (!c) return;
  memset(c, 0, sizeof(*c));
  c->sigma_vx = c->sigma_vy = c->sigma_vz = 0.01f;                 // ConfigVSLAM.cpp:27-28
  c->sigma_wx = c->sigma_wy = c->sigma_wz = 0.01f;
  c->window_size = 21; c->sigma_pixel = 2;                          // :30-31
  c->rho_0 = 0.1f; c->sigma_rho_0 = 0.25f;                          // :33-34
  c->scale = 1; c->T_camera = 0.5f; c->sigma_size = 2;              // :36-40
  c->nInitFeatures = 5; c->min_features = 30; c->max_features = 100; c->forsePlane = 0;   // :42-47
  c->kernel_size = 0;
  c->fx = 592.2860f; c->fy = 584.9968f; c->u0 = 362.1059f; c->v0 = 275.9642f;            // camModel.hpp:22-31
  c->k1 = -0.3954f; c->k2 = 0.5521f; c->k3 = 0.f; c->p1 = -0.0075f; c->p2 = 0.0140f;
  c->image_width = 640; c->image_height = 480;
}

int ekf_abi_version(void) { return EKF_ABI_VERSION; }

int ekf_create(const ekf_config* cfg, int camera_dim, int capacity_features, int dtype, int device,
               ekf_filter** out) {
  if (!out) return EKF_ERR_ARG;
  *out = nullptr;
  if (!cfg || (camera_dim != 13 && camera_dim != 14) || capacity_features < 0 ||
      (dtype != EKF_F32 && dtype != EKF_F64)) {
    ekf::g_create_error = "ekf_create: bad argument";
    return EKF_ERR_ARG;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
    ekf::g_create_error = "ekf_create: no HIP device (this library has no CPU fallback)";
    return EKF_ERR_DEVICE;
  }
  FilterBase* impl = nullptr;
  int rc;
  if (dtype == EKF_F32) {
    auto* f = new ekf::Filter<float>();
    rc = f->init(cfg, camera_dim, capacity_features, device);
    impl = f;
  } else {
    auto* f = new ekf::Filter<double>();
    rc = f->init(cfg, camera_dim, capacity_features, device);
    impl = f;
  }
  if (rc != EKF_OK) {
    ekf::g_create_error = impl->err;
    delete impl;
    return rc;
  }
  *out = new ekf_filter{impl};
  return EKF_OK;
}

void ekf_destroy(ekf_filter* f) {
  if (!f) return;
  delete f->impl;
  delete f;
}

const char* ekf_last_error(const ekf_filter* f) {
  if (!f) return ekf::g_create_error.c_str();
  return f->impl->err.c_str();
}

#define IMPL_OR_ARG(f) \
  if (!(f)) return EKF_ERR_ARG
// entry points that change mu, Sigma, the layout or the sharding: the gathered diagonal blocks are stale afterwards
#define MUTATES(f) (f)->impl->diag_synced = false

int ekf_set_dt(ekf_filter* f, double dT) { IMPL_OR_ARG(f); return f->impl->set_dt(dT); }
double ekf_get_dt(const ekf_filter* f) { return f ? f->impl->get_dt() : 0.0; }
int ekf_set_stream(ekf_filter* f, void* s) { IMPL_OR_ARG(f); return f->impl->set_stream(s); }
int ekf_set_option(ekf_filter* f, int o, int v) { IMPL_OR_ARG(f); MUTATES(f); return f->impl->set_option(o, v); }
int ekf_synchronize(ekf_filter* f) { IMPL_OR_ARG(f); return f->impl->synchronize(); }

int ekf_add_feature(ekf_filter* f, double u, double v) { if (!f) return -EKF_ERR_ARG; MUTATES(f); return f->impl->add_feature(u, v); }
int ekf_remove_feature(ekf_filter* f, int index) { IMPL_OR_ARG(f); MUTATES(f); return f->impl->remove_features(&index, 1); }
int ekf_remove_features(ekf_filter* f, const int* idx, int count) {
  IMPL_OR_ARG(f); MUTATES(f);
  if (count > 0 && !idx) return EKF_ERR_ARG;
  return f->impl->remove_features(idx, count);
}

int ekf_predict(ekf_filter* f, const void* t, const void* r, int vc) { IMPL_OR_ARG(f); MUTATES(f); return f->impl->predict(t, r, vc); }
int ekf_measure(ekf_filter* f) { IMPL_OR_ARG(f); return f->impl->measure(); }
int ekf_get_motion_jacobian(ekf_filter* f, void* Ft, void* Q) { IMPL_OR_ARG(f); return f->impl->motion_jacobian(Ft, Q); }
int ekf_get_predictions(ekf_filter* f, void* h, unsigned char* vis, unsigned char* rem, void* s2, void* hc, void* hf) {
  IMPL_OR_ARG(f);
  return f->impl->get_predictions(h, vis, rem, s2, hc, hf);
}
int ekf_update(ekf_filter* f, const void* z, const int* idx, int M, int plane) {
  IMPL_OR_ARG(f); MUTATES(f);
  return f->impl->update(z, idx, M, plane, false);
}
int ekf_update_device(ekf_filter* f, const void* dz, const int* didx, int M, int plane) {
  IMPL_OR_ARG(f); MUTATES(f);
  return f->impl->update(dz, didx, M, plane, true);
}
int ekf_innovation_covariance(ekf_filter* f, const int* idx, int M, int plane, void* out) {
  IMPL_OR_ARG(f);
  if (!out) return EKF_ERR_ARG;
  return f->impl->innovation_covariance(idx, M, plane, out);
}
int ekf_get_gain(ekf_filter* f, void* out) { IMPL_OR_ARG(f); if (!out) return EKF_ERR_ARG; return f->impl->get_gain(out); }
int ekf_last_measurement_rows(const ekf_filter* f) { return f ? f->impl->last_rows() : 0; }

int ekf_convert_xyz_if_linear(ekf_filter* f, int index) { if (!f) return -EKF_ERR_ARG; MUTATES(f); return f->impl->convert(index, false); }
int ekf_convert_xyz_if_linear_all(ekf_filter* f) { if (!f) return -EKF_ERR_ARG; MUTATES(f); return f->impl->convert(0, true); }

int ekf_num_features(const ekf_filter* f) { return f ? f->impl->num_features() : 0; }
int ekf_state_dim(const ekf_filter* f) { return f ? f->impl->state_dim() : 0; }
int ekf_get_feature_layout(const ekf_filter* f, int* p, int* c) { IMPL_OR_ARG(f); return f->impl->get_layout(p, c); }

int ekf_get_state(ekf_filter* f, void* out, int off, int cnt) { IMPL_OR_ARG(f); if (cnt > 0 && !out) return EKF_ERR_ARG; return f->impl->get_state(out, off, cnt); }
int ekf_set_state(ekf_filter* f, const void* in, int off, int cnt) { IMPL_OR_ARG(f); MUTATES(f); if (cnt > 0 && !in) return EKF_ERR_ARG; return f->impl->set_state(in, off, cnt); }
int ekf_peek_workspace(ekf_filter* f, int which, void* out, int r0, int c0, int rows, int cols) {
  IMPL_OR_ARG(f);
  return f->impl->peek_work(which, out, r0, c0, rows, cols);
}
int ekf_get_sigma_block(ekf_filter* f, void* out, int r0, int c0, int rows, int cols) {
  IMPL_OR_ARG(f);
  if (!out) return EKF_ERR_ARG;
  return f->impl->get_sigma(out, r0, c0, rows, cols);
}
int ekf_set_sigma_block(ekf_filter* f, const void* in, int r0, int c0, int rows, int cols) {
  IMPL_OR_ARG(f); MUTATES(f);
  if (!in) return EKF_ERR_ARG;
  return f->impl->set_sigma(in, r0, c0, rows, cols);
}
int ekf_covariance_parameter(ekf_filter* f, double* out) { IMPL_OR_ARG(f); if (!out) return EKF_ERR_ARG; return f->impl->covariance_parameter(out); }
int ekf_check_invariants(ekf_filter* f, double* pad, double* asym, double* big) { IMPL_OR_ARG(f); return f->impl->check_invariants(pad, asym, big); }
int ekf_feature_xyz(ekf_filter* f, int index, void* xyz, void* cov) { IMPL_OR_ARG(f); return f->impl->feature_xyz(index, xyz, cov); }

int ekf_profile_kernels(void) { return ekf::KID_COUNT; }
const char* ekf_profile_kernel_name(int kid) { return (kid >= 0 && kid < ekf::KID_COUNT) ? ekf::kKernelNames[kid] : ""; }
int ekf_profile_read(ekf_filter* f, int kid, double* ms, long long* cnt) { IMPL_OR_ARG(f); return f->impl->profile_read(kid, ms, cnt); }
int ekf_profile_reset(ekf_filter* f) { IMPL_OR_ARG(f); return f->impl->profile_reset(); }
int ekf_launch_kinds(void) { return EKF_LAUNCH_KINDS; }
const char* ekf_launch_kind_name(int kind) { return (kind >= 0 && kind < EKF_LAUNCH_KINDS) ? ekf::kLaunchNames[kind] : ""; }
int ekf_launch_count(ekf_filter* f, int kind, long long* launches) {
  IMPL_OR_ARG(f);
  if (kind < 0 || kind >= EKF_LAUNCH_KINDS) { f->impl->err = "launch kind out of range"; return EKF_ERR_ARG; }
  if (launches) *launches = f->impl->launch_cnt[kind];
  return EKF_OK;
}
int ekf_profile_work(ekf_filter* f, int kid, double* flop) { IMPL_OR_ARG(f); return f->impl->profile_work(kid, flop); }
int ekf_get_chunk_plan(ekf_filter* f, int* ends, int max_chunks, int* block, int* w_recompute) {
  if (!f || !f->impl) return 0;
  return f->impl->chunk_plan(ends, max_chunks, block, w_recompute);
}

int ekf_rescue_high_innovation(ekf_filter* f, const void* cam, const void* z, const int* idx, int M, double thr,
                               unsigned char* out) {
  IMPL_OR_ARG(f);
  return f->impl->rescue(cam, z, idx, M, thr, out);
}
int ekf_set_frame(ekf_filter* f, const unsigned char* gray, int width, int height, int stride) {
  IMPL_OR_ARG(f);
  return f->impl->set_frame(gray, width, height, stride);
}
int ekf_set_patch(ekf_filter* f, int index, const unsigned char* pixels) { IMPL_OR_ARG(f); return f->impl->set_patch(index, pixels); }
int ekf_get_patch(ekf_filter* f, int index, int matching, unsigned char* out) {
  IMPL_OR_ARG(f);
  return f->impl->get_patch(index, matching, out);
}
int ekf_get_blur_predictions(ekf_filter* f, void* hb) { IMPL_OR_ARG(f); return f->impl->blur_predictions(hb); }
int ekf_find_matches(ekf_filter* f, double threshold, void* z, unsigned char* found, float* score) {
  IMPL_OR_ARG(f);
  return f->impl->find_matches(threshold, z, found, score);
}
int ekf_export_points(ekf_filter* f, void* out, int conv) { IMPL_OR_ARG(f); if (!out) return EKF_ERR_ARG; return f->impl->export_points(out, conv); }
int ekf_export_points_table(ekf_filter* f, void* out, int max_rows, int* rows) { IMPL_OR_ARG(f); return f->impl->export_points_table(out, max_rows, rows); }
int ekf_get_feature_ids(const ekf_filter* f, int* real_index, int* n_find) { IMPL_OR_ARG(f); return f->impl->feature_ids(real_index, n_find); }
int ekf_set_feature_meta(ekf_filter* f, int index, int real_index, int n_find) { IMPL_OR_ARG(f); return f->impl->set_feature_meta(index, real_index, n_find); }
int ekf_num_archived(const ekf_filter* f) { return f ? f->impl->num_archived() : 0; }
int ekf_get_search_ellipses(ekf_filter* f, int sigma_size, int* out) { IMPL_OR_ARG(f); if (!out) return EKF_ERR_ARG; return f->impl->search_ellipses(sigma_size, out); }
int ekf_ransac_1point(ekf_filter* f, const void* z, const int* idx, int M, double thr, int* counts,
                      unsigned char* inl, int* best) {
  IMPL_OR_ARG(f);
  return f->impl->ransac(z, idx, M, thr, counts, inl, best);
}
int ekf_update_two_stage(ekf_filter* f, const void* z, const int* idx, int M, int plane, unsigned int seed,
                         double thr, double chi2, unsigned char* is_li, unsigned char* is_hi, int* drawn) {
  IMPL_OR_ARG(f); MUTATES(f);
  if (M > 0 && (!z || !idx)) return EKF_ERR_ARG;
  return f->impl->update_two_stage(z, idx, M, plane, seed, thr, chi2, is_li, is_hi, drawn);
}
int ekf_shard_configure(ekf_filter* f, int rank, int world, ekf_allgather_fn fn, void* ctx) {
  IMPL_OR_ARG(f); MUTATES(f);
  return f->impl->shard_configure(rank, world, fn, ctx);
}
int ekf_shard_get_info(ekf_filter* f, ekf_shard_info* out) { IMPL_OR_ARG(f); return f->impl->shard_info(out); }
int ekf_shard_update(ekf_filter* f, const void* dz, const int* idx, int M, int plane) {
  IMPL_OR_ARG(f); MUTATES(f);
  return f->impl->shard_update(dz, idx, M, plane);
}
int ekf_shard_rebalance(ekf_filter* f) { IMPL_OR_ARG(f); MUTATES(f); return f->impl->shard_rebalance(); }

void* ekf_device_mu(ekf_filter* f) { return f ? f->impl->dev_mu() : nullptr; }
void* ekf_device_sigma(ekf_filter* f, int* ld) { return f ? f->impl->dev_sigma(ld) : nullptr; }

}  // extern "C"
