// ba_solver.hip -- host side of the BA C-ABI (include/theia_hip.h): problem
// flattening into wave tiles, device residency, and the Levenberg-Marquardt
// control loop that replaces ceres::Solve as configured by
//   BundleAdjuster::SetSolverOptions / Optimize
//   (src/theia/sfm/bundle_adjustment/bundle_adjuster.cc:63-89,315-355).
// LM rules restated from Ceres 2.2 (trust_region_minimizer.cc,
// levenberg_marquardt_strategy.cc): see DESIGN.md "LM control".
// All state stays in HBM, including the trust-region step control (k_lm_control):
// the host enqueues several LM iterations back to back and synchronises once
// per chunk to read the LM state.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <limits>
#include <map>
#include <memory>
#include <new>
#include <string>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <unistd.h>
#include <pthread.h>
#include <thread>
#include <vector>

#include "ba_kernels.h"
#include "theia_hip.h"
#include "theia_hip_internal.h"
#include "pools.h"
#include "host_team.h"

namespace thip {

thread_local std::string g_last_error;

int set_error(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

#define HIP_TRY(expr)                                                                   \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess)                                                               \
      return set_error(e_ == hipErrorOutOfMemory ? THEIA_HIP_ERR_OUT_OF_MEMORY         \
                                                 : THEIA_HIP_ERR_NO_DEVICE,            \
                       "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

static double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// Small pageable sources (the ~60 index / flag vectors of a create()) are copied into pinned blocks that live until the
// end of the call, so that their uploads queue behind the big ones instead of each waiting for the stream: a create() at
// 3 M observations spent ~5 ms in those waits -- the DMA of the 72 MB of sorted observations in front of them -- while the
// host had the K3 plan and the gather lists still to build.  The arena belongs to the handle: create() returns without waiting
// for the uploads (everything else on the handle queues behind them on its stream); the blocks go back to the host cache at the
// end of the first run() -- which has waited for the stream -- or with the handle, whose destructor waits for it.
struct StageArena {
  std::vector<std::unique_ptr<HBuf<char>>> blocks;
  size_t used = 0;
  hipStream_t stream = nullptr;
  void release() { blocks.clear(); used = 0; }   // (after a synchronisation of the stream)
  void* reserve(size_t bytes) {   // room for `bytes` in a pinned block (the caller fills it); nullptr: the caller uploads from its source and waits
    if (getenv("THEIA_HIP_NO_PINNED")) return nullptr;   // (test switch: a host that refuses to pin memory)
    const size_t al = (bytes + 63) & ~(size_t)63;
    if (blocks.empty() || used + al > blocks.back()->cap) {
      std::unique_ptr<HBuf<char>> b(new HBuf<char>);
      if (!b->reserve(std::max<size_t>(al, (size_t)4 << 20))) return nullptr;
      blocks.push_back(std::move(b)); used = 0;
    }
    void* dst = blocks.back()->p + used;
    used += al;
    return dst;
  }
  void* put(const void* src, size_t bytes) {
    void* dst = reserve(bytes);
    if (dst) std::memcpy(dst, src, bytes);
    return dst;
  }
};
inline StageArena*& stage_arena() { static thread_local StageArena* a = nullptr; return a; }
struct StageScope {   // the arena (the handle's: it lives until the uploads are known to be done) serves this thread's uploads
  StageArena* prev;
  explicit StageScope(StageArena* a) : prev(stage_arena()) { stage_arena() = a; }
  ~StageScope() { stage_arena() = prev; }
  StageScope(const StageScope&) = delete;
  StageScope& operator=(const StageScope&) = delete;
};

template <typename T>
struct DevBuf {
  // Blocks come from the library's device cache (pools.h): creating a handle makes ~80 allocations, and at a million
  // observations hipMalloc + hipFree were ~15 ms of a 100 ms create().  A cached block is handed out without a device
  // synchronisation, so the owner makes sure no kernel still uses a buffer when it goes back (the handle's destructor
  // waits for its stream; a re-allocation waits for the device).
  T* p = nullptr;
  size_t n = 0, bytes = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { if (p) dev_pool().give(p, bytes); }
  int alloc(size_t count) {
    if (p) { (void)hipDeviceSynchronize(); dev_pool().give(p, bytes); p = nullptr; bytes = 0; }
    n = count;
    if (count == 0) return 0;
    size_t got = 0;
    p = static_cast<T*>(dev_pool().take(std::max<size_t>(count * sizeof(T), 256), &got));
    if (!p) { n = 0; return set_error(THEIA_HIP_ERR_OUT_OF_MEMORY, "hipMalloc(%zu) failed", count * sizeof(T)); }
    bytes = got;
    return 0;
  }
  int upload(const std::vector<T>& h, hipStream_t st) { return upload(h.data(), h.size(), st, false); }
  // pinned = the source is a pinned block that outlives the copy (the caller synchronises the stream before it lets go)
  int upload(const T* src, size_t count, hipStream_t st, bool pinned) {
    int rc = alloc(count);
    if (rc) return rc;
    // A pageable source (usually a temporary vector) must have been read before upload() returns.
    if (count) {
      StageArena* a = stage_arena();
      if (!pinned && a && a->stream == st && count * sizeof(T) <= ((size_t)32 << 20))
        if (const void* staged = a->put(src, count * sizeof(T))) { src = static_cast<const T*>(staged); pinned = true; }
      HIP_TRY(hipMemcpyAsync(p, src, count * sizeof(T), hipMemcpyHostToDevice, st));
      if (!pinned) HIP_TRY(hipStreamSynchronize(st));
    }
    return 0;
  }
};

}  // namespace thip

using namespace thip;

struct theia_ba_handle_s {
  theia_ba_options opt;
  void* idh = nullptr;   // inverse-depth problems (THEIA_BA_FLAG_INVERSE_DEPTH) live in their own object (ba_invdepth.hip): create / reset / run / download only
  int nc = 0, ng = 0, np = 0, ncv = 0, n = 0, pd = 3;
  int64_t nobs = 0, nobs_main = 0;
  int ntiles_main = 0, ntiles_eval = 0, ntiles_all = 0;  // linearize tiles < + long-track eval tiles < + fixed tiles
  int long_nobs = 0, long_ntracks = 0;
  DevBuf<int> long_obs_index, long_obs_slot, long_track_start, long_track_pt;
  DevBuf<double> long_scratch;
  hipStream_t stream = nullptr;
  static constexpr int kMaxChunk = 8;   // LM iterations enqueued per host synchronisation
  hipEvent_t ev[kMaxChunk][6] = {};
  DevBuf<char> lm_state;                // LmState (device): radius, cost, counters, termination
  DevBuf<char> lm_ctl;                  // LmCtl (device): per-run tolerances, caps, trace pointers
  hipGraph_t graph = nullptr;           // one captured LM iteration (no all-reduce callback, no phase timing)
  hipGraphExec_t graph_exec = nullptr;
  bool graph_failed = false;
  void drop_graph() {
    if (graph_exec) { (void)hipGraphExecDestroy(graph_exec); graph_exec = nullptr; }
    if (graph) { (void)hipGraphDestroy(graph); graph = nullptr; }
  }
  DevBuf<double> tr_cost, tr_g, tr_step, tr_radius;
  DevBuf<int> tr_acc;
  // host-side bookkeeping
  HBuf<int64_t> perm;              // sorted obs index -> original obs index (a block of the pinned host cache: no page faults)
  std::vector<int> cam_red, grp_red, grp_k;
  // cameras that take part in the fused Schur assembly: a variable extrinsics block OR (fused_bw > 0) a variable intrinsics
  // group.  cam_part[c] = index among them in camera order (-1 = none), part_cam = inverse.  Without variable intrinsics
  // this is cam_red.
  std::vector<int> cam_part, part_cam;
  int ncp = 0, fused_bw = 0, n_sum_items2 = 0;
  unsigned fused_kmask = 0;
  std::vector<unsigned> grp_free;
  std::vector<uint8_t> cam_mask, pt_const;
  int ni = 0, ngv = 0;
  // device buffers
  DevBuf<double> snap_cam, snap_pts, snap_intr;   // theia_hip_ba_snapshot_parameters
  DevBuf<double> xnorm_part;
  bool has_snapshot = false;
  DevBuf<double> cam[2], pts[2], intr[2], scale_c, scale_p, ones_c, ones_p, colsq_c0, colsq_p0;
  DevBuf<double> scale_i, ones_i, colsq_i0, scale_red;
  DevBuf<int> d_grp_red, d_grp_k;
  DevBuf<unsigned> d_grp_free, d_red_free;
  int intr_rows = 10;            // intrinsics rows per gather record (ba_kernels.hip RecI): 10, or 4 compact rows
  DevBuf<int> group_model, cam_group, d_cam_red, obs_cam, obs_pt, tile_start, tile_count, f2s, fmaxflag;
  DevBuf<uint8_t> d_cam_mask, d_pt_const;
  DevBuf<double2> obs_uv, obs_si;
  DevBuf<uint8_t> obs_kind;
  DevBuf<double> reduce, Vinv, gp, tile_part, red_part, scalB, chol_work, stop_flag;
  DevBuf<double> rec;                       // per-observation records of the gather-based Schur assembly
  DevBuf<int> diag_items, cam_obs, blk_items, slot_obs, slot_pt;
  DevBuf<int> prior_cam, prior_kind;        // camera priors in use (compact list)
  // inner iterations (ba_inner.hip): observation lists by camera / group / track, a third parameter buffer the sweep
  // works on, its scalars {step^2, |x|^2, cost, invalid}, the gate flag
  bool inner = false;
  DevBuf<int> in_cam_off, in_cam_idx, in_grp_off, in_grp_idx, in_trk_off, in_gate, in_grp_bar;
  DevBuf<double> in_grp_part;   // partial sums of the intrinsics sweep, inner_group_wgs() workgroups per group
  DevBuf<double> in_cam, in_pts, in_intr, in_scal, in_part;
  // inner iterations of a SHARDED solve (theia_hip_ba_set_inner_global): every rank sweeps all cameras and intrinsics groups
  // over the FULL observation set (the same sums on every rank: no exchange of their results), its own tracks afterwards
  bool inner_global = false;
  int g_np = 0, g_npriors = 0;
  int64_t g_nobs = 0;
  DevBuf<double2> g_uv, g_si;
  DevBuf<int> g_cam, g_pt, g_cam_off, g_cam_idx, g_grp_off, g_grp_idx, g_pidx, g_prior_cam, g_prior_kind;
  DevBuf<uint8_t> g_kind;
  DevBuf<double> g_pts, g_prior_vec, g_prior_info, g_stage;
  int in_ntracks = 0;
  DevBuf<double> prior_vec, prior_info;
  int n_priors = 0;
  DevBuf<int2> blk_pairs;
  DevBuf<int> pt_sum_slot;   // [np] pseudo-record of a track's summed intrinsics fields, -1 = none (build_gather_lists_intr)
  DevBuf<uint8_t> slot_in_sum;   // [#records] the observation's track is summed
  DevBuf<uint8_t> pt_sum_cnt;    // [np] number of summed groups (pseudo-records) of a track
  DevBuf<int> sum_group;         // [#pseudo-records] reduced group index
  int sum_base = 0;              // first pseudo-record slot
  int n_trk_sums = 0;
  int n_diag_items = 0, n_blk_items = 0;
  // fused linearise + Schur plan (ba_fused.hip)
  bool use_fused = false;
  unsigned model_mask = 0xffu;          // camera models present in the problem
  DevBuf<FusedRun> fruns;
  DevBuf<int> frun_cams, frun_stage, tile_trk_end, sum_items, sum_src, frun_order, frun_next;
  DevBuf<unsigned short> frun_tgt;
  DevBuf<uint8_t> obs_lc, obs_tl;
  DevBuf<double> fpart, camrot, camrot_cand, camdir;
  int n_fruns = 0, n_sum_items = 0;
  double* h_scal = nullptr;  // pinned: [scalA(16) | scalB(16) | stop flag out / in (2) | spare]
  char* h_state = nullptr;   // pinned: LmState read-back
  int cur = 0;
  bool have_scale = false;
  bool camrot_valid = false;     // P.camrot holds the per-camera blocks of the current state (k_lm_accept keeps it so on accepted steps)
  double fixed_cost = 0.0;
  theia_allreduce_fn allreduce = nullptr;
  void* allreduce_ctx = nullptr;
  ReduceBuf rb;
  DevProblem P;
  // K3 schedule: tile co-visibility of the reduced system (this rank's tracks;
  // OR-ed over the ranks before the first distributed solve) and its plan
  std::vector<uint8_t> tile_adj;
  CholPlan* plan = nullptr;
  bool plan_is_global = true;
  // multi-rank: only the structurally non-zero lower 64x64 tiles of S travel through the all-reduce
  DevBuf<int2> pack_tiles;
  DevBuf<double> pack_buf;
  int shard_rank = -1, shard_world = 0;   // theia_hip_ba_set_shard
  // distributed K3 of a sharded solve (sync_plan): every rank factors the tile columns only its own tracks touch before the
  // all-reduce, which then carries the shared tiles only; tile_cls: 0 shared, 1 this rank's, 2 another rank's
  bool dist_k3 = false;
  std::vector<uint8_t> tile_adj_local, tile_cls, tile_touch;   // tile_touch: this rank's observations / priors write into the tile column
  DevBuf<uint8_t> d_tile_cls;
  int n_pack_tiles = 0;

  StageArena stage;                     // pinned staging of create()'s small uploads (released after the first run)

  ~theia_ba_handle_s() {
    if (idh) thip::id_handle_destroy(idh);
    if (stream) (void)hipStreamSynchronize(stream);   // the buffers below go back to the device cache, not to hipFree
    drop_graph();
    if (plan) chol_plan_destroy(plan);
    for (auto& row : ev) for (auto& e : row) if (e) (void)hipEventDestroy(e);
    if (h_scal) (void)hipHostFree(h_scal);
    if (h_state) (void)hipHostFree(h_state);
    if (stream) (void)hipStreamDestroy(stream);
  }
};

namespace {
// Host-side loops over independent index ranges on a few threads (handle creation at millions of observations).
// THEIA_HIP_HOST_THREADS caps the count (default min(hardware threads, 32); 1 = serial).
unsigned host_thread_cap() {   // read per call (a handful per create()): tests switch it inside one process
  const char* e = getenv("THEIA_HIP_HOST_THREADS");
  const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
  return e ? (unsigned)std::max(1, atoi(e)) : std::min(hw, 32u);
}
thread_local bool g_in_host_region = false;
// fn(k) for the parts k = 0 .. nparts-1 of a fixed partition (the result must not depend on who runs which part)
template <class F>
void host_parts(int nparts, bool threaded, F&& fn) {
  bool& in_region = g_in_host_region;   // a region started inside a region runs on its caller alone
  const unsigned cap = (threaded && !in_region) ? std::min<unsigned>(host_thread_cap(), (unsigned)nparts) : 1u;
  if (cap <= 1) { for (int k = 0; k < nparts; ++k) fn(k); return; }
  struct Flag { bool& f; explicit Flag(bool& x) : f(x) { f = true; } ~Flag() { f = false; } };
  const std::function<void(int)> job = [&fn](int k) { Flag g(g_in_host_region); fn(k); };
  if (host_team().run(nparts, cap, job)) return;
  std::vector<std::thread> th;
  for (unsigned t = 1; t < cap; ++t) th.emplace_back([&job, t, cap, nparts] { for (int k = (int)t; k < nparts; k += (int)cap) job(k); });
  for (int k = 0; k < nparts; k += (int)cap) job(k);
  for (auto& x : th) x.join();
}
template <class F>
void host_chunks(int64_t n, F&& fn) {
  const unsigned cap = host_thread_cap();
  // (THEIA_HIP_HOST_CHUNK_MIN: test switch -- small problems through the threaded passes)
  const char* cm = getenv("THEIA_HIP_HOST_CHUNK_MIN");
  const int64_t min_n = cm ? std::max(1, atoi(cm)) : 262144;
  if (n < min_n || cap <= 1) { fn((int64_t)0, n); return; }
  const int64_t per = (n + cap - 1) / cap;
  host_parts((int)cap, true, [&](int t) {
    const int64_t a = std::min<int64_t>(n, (int64_t)t * per), b = std::min<int64_t>(n, a + per);
    if (a < b) fn(a, b);
  });
}


// std::vector without value-initialisation of trivially constructible elements (resize() leaves them uninitialised)
template <class T>
struct NoInitAlloc : std::allocator<T> {
  template <class U> struct rebind { using other = NoInitAlloc<U>; };
  template <class U, class... A>
  void construct(U* ptr, A&&... args) {
    if constexpr (sizeof...(A) == 0) ::new (static_cast<void*>(ptr)) U; else ::new (static_cast<void*>(ptr)) U(std::forward<A>(args)...);
  }
};

// number of parts of a threaded pass over n items, `grain` items per part at least (the test switch lowers the grain)
int host_part_count(int64_t n, int64_t grain) {
  const char* cm = getenv("THEIA_HIP_HOST_CHUNK_MIN");
  if (cm) grain = std::max<int64_t>(1, std::min<int64_t>(grain, atoi(cm)));
  return (int)std::max<int64_t>(1, std::min<int64_t>(host_thread_cap(), n / grain));
}

// A plain uninitialised array for the per-observation / per-track temporaries of create() that are written in full by the
// pass that fills them: value-initialising 15 MB of std::vectors was a millisecond of single-threaded memset per create().
template <class T>
struct RawArray {
  std::unique_ptr<T[]> p;
  explicit RawArray(size_t n) : p(new T[std::max<size_t>(1, n)]) {}
  T& operator[](size_t i) { return p[i]; }
  const T& operator[](size_t i) const { return p[i]; }
  T* data() { return p.get(); }
};

// One stable counting pass on host threads: `to` = `from` ordered by digit (0 <= digit < nb), ties in input order.  Per-part
// histograms, offsets taken in (bucket, part) order; the result does not depend on the number of parts.
template <class T, class Digit>
void counting_pass(const T* from, T* to, int64_t n, int nb, Digit&& digit) {
  int parts = host_part_count(n, 32768);
  if (n < 393216 && !getenv("THEIA_HIP_HOST_CHUNK_MIN")) parts = 1;   // (two regions of the thread team cost more than a serial pass over a few hundred thousand entries)
  if ((int64_t)parts * nb > ((int64_t)1 << 24)) parts = 1;
  const int64_t per = (n + parts - 1) / parts;
  std::vector<int> head((size_t)parts * nb, 0);
  host_parts(parts, true, [&](int t) {
    int* hh = head.data() + (size_t)t * nb;
    for (int64_t i = t * per; i < std::min<int64_t>(n, (t + 1) * per); ++i) hh[digit(from[i])]++;
  });
  int run = 0;
  for (int b = 0; b < nb; ++b)
    for (int t = 0; t < parts; ++t) { int& c = head[(size_t)t * nb + b]; const int cnt = c; c = run; run += cnt; }
  host_parts(parts, true, [&](int t) {
    int* hh = head.data() + (size_t)t * nb;
    for (int64_t i = t * per; i < std::min<int64_t>(n, (t + 1) * per); ++i) to[hh[digit(from[i])]++] = from[i];
  });
}

// Free intrinsics of a model under an OptimizeIntrinsicsType mask
// (GetSubsetFromOptimizeIntrinsicsType of every *_camera_model.cc, e.g.
// pinhole_camera_model.cc:132-162): bit q = parameter q is optimised.
unsigned intrinsics_free_mask(int model, int opt) {
  const bool noskew = (model == THEIA_CAM_FOV || model == THEIA_CAM_DIVISION_UNDISTORTION);
  unsigned m = 0;
  if (opt & THEIA_INTR_FOCAL_LENGTH) m |= 1u << 0;
  if (opt & THEIA_INTR_ASPECT_RATIO) m |= 1u << 1;
  if ((opt & THEIA_INTR_SKEW) && !noskew) m |= 1u << 2;
  if (opt & THEIA_INTR_PRINCIPAL_POINTS) m |= noskew ? (3u << 2) : (3u << 3);
  if (opt & THEIA_INTR_RADIAL_DISTORTION) {
    switch (model) {
      case THEIA_CAM_PINHOLE: case THEIA_CAM_DOUBLE_SPHERE: case THEIA_CAM_EXTENDED_UNIFIED: case THEIA_CAM_ORTHOGRAPHIC: m |= 3u << 5; break;
      case THEIA_CAM_PINHOLE_RADIAL_TANGENTIAL: m |= 7u << 5; break;
      case THEIA_CAM_FISHEYE: m |= 15u << 5; break;
      case THEIA_CAM_FOV: case THEIA_CAM_DIVISION_UNDISTORTION: m |= 1u << 4; break;
    }
  }
  if ((opt & THEIA_INTR_TANGENTIAL_DISTORTION) && model == THEIA_CAM_PINHOLE_RADIAL_TANGENTIAL) m |= 3u << 8;
  return m;
}
int intrinsics_size(int model) {
  static const int K[8] = {7, 10, 9, 5, 5, 7, 7, 7};  // kIntrinsicsSize of the eight models
  return (model >= 0 && model < 8) ? K[model] : 0;
}
// bundle_adjuster.cc:406-427 parameter bounds (applied to the initial point as Ceres does)
void project_intrinsics_to_bounds(int model, double* k) {
  if (k[0] < 1.0) k[0] = 1.0;
  if (model == THEIA_CAM_DOUBLE_SPHERE) { k[5] = std::min(1.0, std::max(-1.0, k[5])); k[6] = std::min(1.0, std::max(0.0, k[6])); }
  if (model == THEIA_CAM_EXTENDED_UNIFIED) { k[5] = std::min(1.0, std::max(0.0, k[5])); k[6] = std::max(0.1, k[6]); }
}

enum { SB_COST = 0, SB_MCC = 1, SB_STEPSQ = 2, SB_XNORMSQ = 3, SB_INVALID = 4, SB_STEPSQ_CAM = 8, SB_XNORMSQ_CAM = 9 };

// ------------------------------------------------------------ LM step control
// Trust-region bookkeeping of one solve, resident on the device so that several
// iterations can be enqueued without a host round trip.
struct LmState {
  double radius, decrease_factor, x_cost, x_norm, gmax, minimum_cost, initial_cost;
  int step_successful, iter, invalid_steps, term, done, first, accepted, num_successful, trace_size, pending_grad,
      fail_at_first, bodies;
  int inner_enabled;   // inner iterations still running (they switch themselves off: inner_iteration_tolerance)
  int use_inner;       // this body's candidate is the point the inner iterations ended at (k_lm_accept copies that one)
};
struct LmCtl {   // per-run control block, device resident so that a captured graph of the iteration stays valid
  int max_iterations, trace_capacity;
  double function_tolerance, gradient_tolerance, parameter_tolerance, max_radius, fixed_cost;
  double *tc, *tg, *ts, *tr;
  int* ta;
  const double* inner_scal;   // [4] = {|x - x_inner|^2, |x_inner|^2, cost at x_inner, invalid}, null = no inner iterations
};

__device__ void lm_trace(LmState* st, const LmCtl& c, double cost, double g, double step, double radius, int acc) {
  if (!c.tc || st->trace_size >= c.trace_capacity) return;
  const int k = st->trace_size++;
  c.tc[k] = cost; c.tg[k] = g; c.ts[k] = step; c.tr[k] = radius; c.ta[k] = acc;
}

// One pass of the TrustRegionMinimizer loop body (ceres trust_region_minimizer.cc; the
// same rules the host loop of the first versions applied after each read-back):
// sa = scalars of the linearisation at x, sb = scalars of the trial step.
__device__ void lm_control_body(LmState* st, const double* __restrict__ sa, const double* __restrict__ sb,
                                const LmCtl* __restrict__ cp) {
  const LmCtl c = *cp;
  double* tg = c.tg; double* tc = c.tc;
  st->accepted = 0; st->use_inner = 0;
  if (st->done) return;
  st->bodies++;
  const double x_cost = sa[SC_COST];
  const double gmax = sa[SC_GMAX];
  st->x_cost = x_cost; st->gmax = gmax;
  if (st->pending_grad >= 0 && tg) tg[st->pending_grad] = gmax;
  st->pending_grad = -1;
  if (st->first) {
    st->first = 0;
    st->initial_cost = x_cost + c.fixed_cost;
    st->minimum_cost = x_cost;
    if (sa[SC_INVALID] > 0.0 || !isfinite(x_cost)) { st->term = THEIA_TERM_FAILURE; st->fail_at_first = 1; st->done = 1; return; }
    lm_trace(st, c, x_cost + c.fixed_cost, gmax, 0.0, st->radius, 1);
  }
  if (st->iter >= c.max_iterations) { st->term = THEIA_TERM_NO_CONVERGENCE; st->done = 1; return; }
  if (st->step_successful && gmax <= c.gradient_tolerance) { st->term = THEIA_TERM_CONVERGENCE; st->done = 1; return; }
  if (st->radius <= 1e-32) { st->term = THEIA_TERM_CONVERGENCE; st->done = 1; return; }
  st->iter++;
  double mcc = sb[SB_MCC];
  double stepsq = sb[SB_STEPSQ] + sb[SB_STEPSQ_CAM];
  const bool solved = sa[SC_NOTPD] == 0.0 && isfinite(mcc) && isfinite(stepsq);
  if (!(solved && mcc > 0.0)) {
    if (++st->invalid_steps >= 5) { st->term = THEIA_TERM_FAILURE; st->done = 1; return; }
    st->radius /= st->decrease_factor; st->decrease_factor *= 2.0; st->step_successful = 0;
    lm_trace(st, c, x_cost + c.fixed_cost, gmax, 0.0, st->radius, 0);
    return;
  }
  st->invalid_steps = 0;
  double cand_cost = sb[SB_COST];
  if (sb[SB_INVALID] > 0.0 || !isfinite(cand_cost)) cand_cost = DBL_MAX;
  // TrustRegionMinimizer::DoInnerIterationsIfNeeded: the sweep ran on a copy of the candidate (k_inner_gate said so
  // from the same quantities); its result replaces the candidate unless the evaluation there failed
  bool inner_useful = false;
  double xnormsq = sb[SB_XNORMSQ] + sb[SB_XNORMSQ_CAM];
  if (c.inner_scal && st->inner_enabled && cand_cost < DBL_MAX) {
    const double inner_cost = c.inner_scal[2];
    if (c.inner_scal[3] == 0.0 && isfinite(inner_cost)) {
      st->use_inner = 1;
      mcc += cand_cost - inner_cost;
      inner_useful = inner_cost < x_cost;
      st->inner_enabled = (1.0 - inner_cost / cand_cost) > 1e-3;   // inner_iteration_tolerance
      cand_cost = inner_cost;
      stepsq = c.inner_scal[0]; xnormsq = c.inner_scal[1];
    }
  }
  const double step_norm = sqrt(stepsq);
  if (step_norm <= c.parameter_tolerance * (st->x_norm + c.parameter_tolerance)) {
    lm_trace(st, c, cand_cost + c.fixed_cost, gmax, step_norm, st->radius, 0);
    st->term = THEIA_TERM_CONVERGENCE; st->done = 1; return;
  }
  const double cost_change = x_cost - cand_cost;
  if (fabs(cost_change) <= c.function_tolerance * x_cost) {
    lm_trace(st, c, cand_cost + c.fixed_cost, gmax, step_norm, st->radius, 0);
    st->term = THEIA_TERM_CONVERGENCE; st->done = 1; return;
  }
  const double rho = cost_change / mcc;
  if (inner_useful || rho > 1e-3) {   // IsStepSuccessful
    st->accepted = 1;   // k_lm_accept copies the candidate buffers over the state
    st->x_norm = sqrt(xnormsq);
    st->radius = st->radius / fmax(1.0 / 3.0, 1.0 - pow(2.0 * rho - 1.0, 3));
    st->radius = fmin(c.max_radius, st->radius);
    st->decrease_factor = 2.0; st->step_successful = 1;
    st->num_successful++;
    if (cand_cost < st->minimum_cost) st->minimum_cost = cand_cost;
    if (tc && st->trace_size < c.trace_capacity) st->pending_grad = st->trace_size;
    lm_trace(st, c, cand_cost + c.fixed_cost, -1.0, step_norm, st->radius, 1);
  } else {
    st->radius /= st->decrease_factor; st->decrease_factor *= 2.0; st->step_successful = 0;
    lm_trace(st, c, cand_cost + c.fixed_cost, gmax, step_norm, st->radius, 0);
  }
  // the iteration cap is known now: no further pass is needed to detect it
  if (st->iter >= c.max_iterations) { st->term = THEIA_TERM_NO_CONVERGENCE; st->done = 1; }
}

__global__ void k_lm_control(LmState* st, const double* __restrict__ sa, const double* __restrict__ sb,
                             const LmCtl* __restrict__ cp) {
  lm_control_body(st, sa, sb, cp);
}
// The tile reduction of the trial step (launch_reduce_tiles cfg 1 -> scalB) and the step control in one launch.
__global__ __launch_bounds__(1024) void k_reduce_control(int ntiles, const double* __restrict__ part,
                                                         const int* __restrict__ f2s, const int* __restrict__ fmaxflag,
                                                         LmState* st, const double* __restrict__ sa, double* __restrict__ sb,
                                                         const LmCtl* __restrict__ cp) {
  __shared__ double sm[8][16];
  thip::reduce_tiles_body(ntiles, part, 5, f2s, fmaxflag, sb, sm);
  if (threadIdx.x == 0) lm_control_body(st, sa, sb, cp);
}

// |x| over the variable parameter blocks (TrustRegionMinimizer's x_norm at the start):
// out2[0] = points (per track shard), out2[1] = cameras + intrinsics.  kXnormBlocks workgroups write their
// partial sums to part[b][2]; k_xnorm_reduce adds them in block order (no atomics: the norm is reproducible).
constexpr int kXnormBlocks = 64;
constexpr int kMaxShardSlots = 1024;   // ranks whose MAX scalar fits in the packed SUM all-reduce
__global__ __launch_bounds__(256) void k_xnorm_partial(DevProblem P, const double* __restrict__ cam,
                                                       const double* __restrict__ pts, const double* __restrict__ intr,
                                                       double* __restrict__ part) {
  __shared__ double s1[256], s2[256];
  double sp = 0.0, sc = 0.0;
  const int t0 = blockIdx.x * 256 + threadIdx.x, stride = gridDim.x * 256;
  for (int p = t0; p < P.np; p += stride) {
    const double4 x = reinterpret_cast<const double4*>(pts)[p];   // unconditional load, masked sum
    const double m = P.pt_const[p] ? 0.0 : 1.0;
    sp += m * (((x.x * x.x + x.y * x.y) + x.z * x.z) + x.w * x.w);
  }
  if (blockIdx.x == 0) {
    for (int c = threadIdx.x; c < P.nc; c += 256)
      if (P.cam_red[c] >= 0) for (int q = 0; q < 6; ++q) sc += cam[6 * c + q] * cam[6 * c + q];
    if (P.ni)
      for (int g = threadIdx.x; g < P.ng_total; g += 256)
        if (P.grp_red[g] >= 0) for (int q = 0; q < P.grp_k[g]; ++q) sc += intr[(size_t)g * THEIA_MAX_INTRINSICS + q] * intr[(size_t)g * THEIA_MAX_INTRINSICS + q];
  }
  s1[threadIdx.x] = sp; s2[threadIdx.x] = sc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) { s1[threadIdx.x] += s1[threadIdx.x + s]; s2[threadIdx.x] += s2[threadIdx.x + s]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { part[2 * blockIdx.x] = s1[0]; part[2 * blockIdx.x + 1] = s2[0]; }
}
__global__ void k_xnorm_reduce(const double* __restrict__ part, int nblocks, double* __restrict__ out2) {
  double a = 0.0, b = 0.0;
  for (int k = 0; k < nblocks; ++k) { a += part[2 * k]; b += part[2 * k + 1]; }
  out2[0] = a; out2[1] = b;
}
__global__ void k_xnorm_set(LmState* st, const double* __restrict__ in2) { st->x_norm = sqrt(in2[0] + in2[1]); }
__global__ void k_fill_value(double* x, size_t n, double v) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) x[i] = v;
}
__global__ void k_lm_init_state(LmState* dst, LmState v) { *dst = v; }
__global__ void k_lm_init_ctl(LmCtl* dst, LmCtl v) { *dst = v; }

// Packed all-reduce buffer: [tiles (64x64, zero padded) | rhs | colsq | g_c | 8 scalars].  S outside the
// structurally non-zero tiles is zero on every rank, so only those tiles are summed across ranks
// (C2: 39 tiles = 1.3 MB instead of the 11.5 MB dense matrix).
__global__ __launch_bounds__(256) void k_pack_rcs(int n, const double* __restrict__ base, const int2* __restrict__ tiles,
                                                  int ntiles, double* __restrict__ pack, int to_pack, int rank,
                                                  int world) {
  const int b = blockIdx.x;
  if (b < ntiles) {
    const int r0 = tiles[b].x * 64, c0 = tiles[b].y * 64;
    double* pk = pack + (size_t)b * 4096;
    for (int e = threadIdx.x; e < 4096; e += 256) {
      const int r = r0 + (e >> 6), c = c0 + (e & 63);
      if (r < n && c < n) {
        double* s = const_cast<double*>(base) + (size_t)r * n + c;
        if (to_pack) pk[e] = *s; else *s = pk[e];
      } else if (to_pack) pk[e] = 0.0;
    }
    return;
  }
  // tail: everything after S in the reduce buffer up to the 8 sum-reduced scalars
  const size_t tail = (size_t)3 * n + 8;
  double* src = const_cast<double*>(base) + (size_t)n * n;
  double* pk = pack + (size_t)ntiles * 4096;
  for (size_t e = (size_t)(b - ntiles) * 256 + threadIdx.x; e < tail; e += (size_t)(gridDim.x - ntiles) * 256) {
    if (to_pack) pk[e] = src[e]; else src[e] = pk[e];
  }
  // the MAX-reduced scalar (gradient max-norm, >= 0) in one slot per rank of the SUM all-reduce
  if (world > 0 && b == ntiles && threadIdx.x == 0) {
    double* slots = pk + tail;
    double* gmax = src + (size_t)3 * n + SC_GMAX;   // scal = [.. | colsq | gc | scal[16]]
    if (to_pack) {
      for (int r = 0; r < world; ++r) slots[r] = (r == rank) ? *gmax : 0.0;
    } else {
      double m = 0.0;
      for (int r = 0; r < world; ++r) m = fmax(m, slots[r]);
      *gmax = m;
    }
  }
}

// distributed K3: after the back-substitution a rank holds the step of its own private columns and of the shared ones; the
// columns of other ranks' private tiles are zeroed, the shared ones kept on rank 0 only, and one SUM all-reduce of the n
// doubles (x + 0 is exact) gives every rank the whole camera step
__global__ void k_y_own(int n, double* __restrict__ y, const uint8_t* __restrict__ tile_cls, int rank) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= n) return;
  const int c = tile_cls[d >> 6];
  if (!(c == 1 || (c == 0 && rank == 0))) y[d] = 0.0;
}

// accepted step: the candidate parameters become the state
// Will this body's control pass run inner iterations?  The same conditions lm_control_body applies, from the same inputs.
__global__ void k_inner_gate(const LmState* __restrict__ st, const double* __restrict__ sa, const double* __restrict__ sb,
                             int* __restrict__ gate) {
  const double mcc = sb[SB_MCC], stepsq = sb[SB_STEPSQ] + sb[SB_STEPSQ_CAM], cand = sb[SB_COST];
  const bool valid = sa[SC_NOTPD] == 0.0 && isfinite(mcc) && isfinite(stepsq) && mcc > 0.0;
  *gate = (!st->done && st->inner_enabled && valid && sb[SB_INVALID] == 0.0 && isfinite(cand)) ? 1 : 0;
}

// use_inner: the accepted point is the one the inner iterations ended at (in_*), not the trust-region candidate
__global__ void k_lm_accept(const LmState* __restrict__ st, double* __restrict__ cam, const double* __restrict__ cand_cam, size_t ncam,
                            double* __restrict__ pts, const double* __restrict__ cand_pts, size_t npts,
                            double* __restrict__ intr, const double* __restrict__ cand_intr, size_t nintr,
                            const double* __restrict__ in_cam = nullptr, const double* __restrict__ in_pts = nullptr,
                            const double* __restrict__ in_intr = nullptr, double* __restrict__ camrot = nullptr,
                            const double* __restrict__ camrot_cand = nullptr, size_t ncamrot = 0) {
  if (!st->accepted) return;
  if (st->use_inner) { cand_cam = in_cam; cand_pts = in_pts; if (nintr) cand_intr = in_intr; }
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  // the candidate's per-camera blocks become the state's (enqueue_linearize then skips k_cam_prep); never with inner iterations
  if (camrot && !st->use_inner)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < ncamrot; i += stride) camrot[i] = camrot_cand[i];
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npts; i += stride) pts[i] = cand_pts[i];
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < ncam; i += stride) cam[i] = cand_cam[i];
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nintr; i += stride) intr[i] = cand_intr[i];
}

int supported_model(int m) { return m >= THEIA_CAM_PINHOLE && m <= THEIA_CAM_ORTHOGRAPHIC; }

int validate(const theia_ba_problem* p, const theia_ba_options* o) {
  if (!p || !o) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "null problem/options");
  if (p->num_cameras < 0 || p->num_points < 0 || p->num_obs < 0 || p->num_groups < 0)
    return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "negative sizes");
  if (p->num_obs >= (int64_t)1 << 31) return set_error(THEIA_HIP_ERR_UNSUPPORTED, "num_obs >= 2^31");
  if (p->num_obs > 0 && (!p->cam_ext || !p->intrinsics || !p->group_model || !p->cam_group || !p->points ||
                         !p->obs_uv || !p->obs_cam || !p->obs_pt))
    return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "null array in problem");
  for (int c = 0; c < p->num_cameras; ++c)
    if (p->cam_group[c] < 0 || p->cam_group[c] >= p->num_groups)
      return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "cam_group[%d] out of range", c);
  for (int g = 0; g < p->num_groups; ++g)
    if (!supported_model(p->group_model[g]))
      return set_error(THEIA_HIP_ERR_UNSUPPORTED, "camera model %d of group %d has no HIP kernel yet", p->group_model[g], g);
  {   // (on host threads at millions of observations; the first offender is reported whoever finds it)
    std::atomic<int64_t> first_bad{std::numeric_limits<int64_t>::max()};
    host_chunks(p->num_obs, [&](int64_t i0, int64_t i1) {
      for (int64_t i = i0; i < i1; ++i)
        if (p->obs_cam[i] < 0 || p->obs_cam[i] >= p->num_cameras || p->obs_pt[i] < 0 || p->obs_pt[i] >= p->num_points) {
          int64_t cur = first_bad.load();
          while (i < cur && !first_bad.compare_exchange_weak(cur, i)) {}
          return;
        }
    });
    if (first_bad.load() != std::numeric_limits<int64_t>::max())
      return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "observation %lld indexes out of range", (long long)first_bad.load());
  }
  if (o->intrinsics_to_optimize < 0 || o->intrinsics_to_optimize > THEIA_INTR_ALL)
    return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "invalid intrinsics_to_optimize bit mask");
  if (o->prior_mask < 0 || o->prior_mask > 7) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "invalid prior_mask");
  if (p->obs_kind) {
    bool any = false;
    for (int64_t i = 0; i < p->num_obs; ++i) {
      if (p->obs_kind[i] > THEIA_OBS_DEPTH_PRIOR) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "obs_kind[%lld] is not a THEIA_OBS_* value", (long long)i);
      any = any || p->obs_kind[i] == THEIA_OBS_DEPTH_PRIOR;
    }
    if (any && !p->obs_sqrt_info)
      return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "depth-prior rows need obs_sqrt_info (1 / sqrt(depth_prior_variance))");
    if (any && !(o->robust_loss_width_depth_prior > 0.0))
      return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "robust_loss_width_depth_prior must be positive");
  }
  if (o->loss_function_type < 0 || o->loss_function_type > THEIA_LOSS_TRUNCATED)
    return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "invalid loss function type");  // reference: LOG(FATAL)
  return 0;
}

void fill_devproblem(theia_ba_handle_s* h) {
  h->camrot_valid = false;   // whoever rebuilds the device view may have touched cam[] / the scales: the next linearisation runs k_cam_prep
  DevProblem& P = h->P;
  P.nc = h->nc; P.np = h->np; P.ncv = h->ncv; P.ntiles = h->ntiles_main; P.nobs = h->nobs_main;
  P.n = h->n; P.pd = h->pd; P.loss_type = h->opt.loss_function_type; P.loss_width = h->opt.robust_loss_width;
  P.intr = h->intr[h->cur].p; P.intr_cand = h->intr[1 - h->cur].p;
  P.group_model = h->group_model.p; P.cam_group = h->cam_group.p;
  P.ni = h->ni; P.ng_total = h->ng; P.grp_red = h->d_grp_red.p; P.grp_free = h->d_grp_free.p; P.grp_k = h->d_grp_k.p;
  P.red_free = h->d_red_free.p; P.intr_rows = h->intr_rows;
  P.scale_i = h->scale_i.p; P.scale_red = h->scale_red.p;
  P.cam_red = h->d_cam_red.p; P.cam_mask = h->d_cam_mask.p; P.pt_const = h->d_pt_const.p;
  P.obs_uv = h->obs_uv.p; P.obs_si = h->obs_si.p; P.obs_cam = h->obs_cam.p; P.obs_pt = h->obs_pt.p;
  P.obs_kind = h->obs_kind.n ? h->obs_kind.p : nullptr; P.loss_width_depth = h->opt.robust_loss_width_depth_prior;
  P.tile_start = h->tile_start.p; P.tile_count = h->tile_count.p;
  P.scale_c = h->scale_c.p; P.scale_p = h->scale_p.p;
  P.long_nobs = h->long_nobs; P.long_ntracks = h->long_ntracks;
  P.long_obs_index = h->long_obs_index.p; P.long_obs_slot = h->long_obs_slot.p;
  P.long_track_start = h->long_track_start.p; P.long_track_pt = h->long_track_pt.p;
  P.n_priors = h->n_priors; P.prior_cam = h->prior_cam.p; P.prior_kind = h->prior_kind.p;
  P.prior_vec = h->prior_vec.p; P.prior_info = h->prior_info.p;
  P.rec = h->rec.p; P.n_diag_items = h->n_diag_items; P.n_blk_items = h->n_blk_items;
  P.n_fruns = h->use_fused ? h->n_fruns : 0; P.fruns = h->fruns.p; P.frun_order = h->frun_order.p; P.frun_next = h->frun_next.p; P.frun_cams = h->frun_cams.p; P.frun_stage = h->frun_stage.p; P.frun_tgt = h->frun_tgt.p;
  P.obs_lc = h->obs_lc.p; P.obs_tl = h->obs_tl.p; P.tile_trk_end = h->tile_trk_end.p; P.fpart = h->fpart.p; P.camrot = h->camrot.p; P.camrot_cand = h->camrot_cand.p; P.camdir = h->camdir.p;
  { const char* dbg = getenv("THEIA_HIP_FUSED_DBG"); P.fused_dbg = dbg ? atoi(dbg) : 0; }
  P.model_mask = h->model_mask;
  P.n_sum_items = h->n_sum_items; P.sum_items = h->sum_items.p; P.sum_src = h->sum_src.p;
  P.fused_bw = h->use_fused ? h->fused_bw : 0; P.n_sum_items2 = h->n_sum_items2; P.fused_kmask = h->fused_kmask;
  P.diag_items = h->diag_items.p; P.rec_slot = h->cam_obs.p; P.slot_obs = h->slot_obs.p; P.slot_pt = h->slot_pt.p; P.blk_items = h->blk_items.p; P.blk_pairs = h->blk_pairs.p;
  P.pt_sum_slot = h->n_trk_sums ? h->pt_sum_slot.p : nullptr; P.slot_in_sum = h->n_trk_sums ? h->slot_in_sum.p : nullptr;
  P.pt_sum_cnt = h->n_trk_sums ? h->pt_sum_cnt.p : nullptr; P.sum_group = h->n_trk_sums ? h->sum_group.p : nullptr; P.sum_base = h->sum_base;
}

// create()'s staged uploads sit in pinned blocks of the handle's arena until the stream is known to have passed them: every
// entry point lets them go as soon as the stream is idle (a handle used only for evaluation, covariances or reset never
// reaches the end of a run(), which is where they were released before)
void release_stage_if_idle(theia_ba_handle_s* h) {
  if (h->stage.blocks.empty()) return;
  if (hipStreamQuery(h->stream) == hipSuccess) h->stage.release();
  else (void)hipGetLastError();   // hipErrorNotReady: still copying
}

int upload_parameters(theia_ba_handle_s* h, const theia_ba_problem* p) {
  // Inside create() (a staging arena on this stream): the caller's arrays are copied into pinned blocks on host threads and
  // uploaded from there, nothing waits.  Otherwise (reset_parameters): pageable sources, the stream is waited for.
  StageArena* a = stage_arena();
  const bool staged = a && a->stream == h->stream;
  bool must_wait = false;
  auto up = [&](double* dst0, double* dst1, const double* src, size_t count) -> int {
    if (!count) return 0;
    const double* from = src;
    // (staged up to 64 MB per array -- C4's points are 16 MB; larger arrays go from the caller's memory and are waited for,
    // like DevBuf::upload's cap: pinned blocks count against the 2 GiB pinned cache until the arena is released)
    const bool fits = count * sizeof(double) <= ((size_t)64 << 20);
    double* st = (staged && fits) ? static_cast<double*>(a->reserve(count * sizeof(double))) : nullptr;
    if (st) {
      host_chunks((int64_t)count, [&](int64_t i0, int64_t i1) { std::memcpy(st + i0, src + i0, sizeof(double) * (size_t)(i1 - i0)); });
      from = st;
    } else {
      if (staged && fits) (void)hipGetLastError();   // the host does not pin that much: the caller's array is the source, and is waited for
      must_wait = true;
    }
    HIP_TRY(hipMemcpyAsync(dst0, from, sizeof(double) * count, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(dst1, dst0, sizeof(double) * count, hipMemcpyDeviceToDevice, h->stream));
    return 0;
  };
  int rc;
  if ((rc = up(h->cam[0].p, h->cam[1].p, p->cam_ext, (size_t)6 * h->nc))) return rc;
  if ((rc = up(h->pts[0].p, h->pts[1].p, p->points, (size_t)4 * h->np))) return rc;
  if (h->ng) {
    std::vector<double> hk(p->intrinsics, p->intrinsics + (size_t)THEIA_MAX_INTRINSICS * h->ng);
    for (int g = 0; g < h->ng; ++g) if (h->grp_red[g] >= 0) project_intrinsics_to_bounds(p->group_model[g], &hk[(size_t)g * THEIA_MAX_INTRINSICS]);
    if ((rc = up(h->intr[0].p, h->intr[1].p, hk.data(), hk.size()))) return rc;
    if (must_wait) HIP_TRY(hipStreamSynchronize(h->stream));   // (hk is a local)
  }
  if (must_wait) HIP_TRY(hipStreamSynchronize(h->stream));
  h->cur = 0;
  h->P.intr = h->intr[0].p; h->P.intr_cand = h->intr[1].p;
  h->have_scale = false; h->camrot_valid = false;
  return 0;
}

// reduce_tiles configurations, uploaded once at create():
//   cfg 0 (linearize): {cost, gmax(max), invalid, notpd} -> rb.scal
//   cfg 1 (backsub)  : {cost, mcc, stepsq, xnormsq, invalid} -> scalB
//   cfg 2 (cost only): {cost, invalid} -> scalB
const int kCfgF2S[3][8] = {{SC_COST, SC_GMAX, SC_INVALID, SC_NOTPD, 0, 0, 0, 0},
                           {SB_COST, SB_MCC, SB_STEPSQ, SB_XNORMSQ, SB_INVALID, 0, 0, 0},
                           {SB_COST, SB_INVALID, 0, 0, 0, 0, 0, 0}};
const int kCfgMax[3][8] = {{0, 1, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0}};

int do_allreduce(theia_ba_handle_s* h, double* buf, size_t count, int op) {
  if (!h->allreduce || count == 0) return 0;
  const int rc = h->allreduce(h->allreduce_ctx, buf, count, op, (void*)h->stream);
  if (rc) return set_error(THEIA_HIP_ERR_INTERNAL, "allreduce callback failed (%d)", rc);
  return 0;
}

// cost of a tile range at given parameters (deterministic reduction)
int cost_of_tiles(theia_ba_handle_s* h, int tile0, int ntiles, const double* cam, const double* pts, double* cost, double* invalid) {
  *cost = 0.0; *invalid = 0.0;
  if (ntiles == 0) return 0;
  DevProblem Q = h->P;
  Q.tile_start = h->tile_start.p + tile0; Q.tile_count = h->tile_count.p + tile0; Q.ntiles = ntiles;
  HIP_TRY(hipMemsetAsync(h->scalB.p, 0, sizeof(double) * 16, h->stream));
  launch_cost_only(Q, cam, pts, h->tile_part.p, h->scalB.p, h->stream);
  launch_reduce_tiles(ntiles, h->tile_part.p, 2, h->f2s.p + 16, h->fmaxflag.p + 16, h->scalB.p, h->stream);
  HIP_TRY(hipMemcpyAsync(h->h_scal + 16, h->scalB.p, sizeof(double) * 16, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  *cost = h->h_scal[16 + SB_COST]; *invalid = h->h_scal[16 + SB_INVALID];
  return 0;
}

// Jacobi scaling 1/(1+sqrt(colnorm^2)) from the unscaled Jacobian at the
// current point (ceres trust_region_minimizer.cc, computed once per solve).
// With several ranks the camera column norms are summed first: cameras are
// shared by all track shards.
int compute_scale(theia_ba_handle_s* h) {
  DevProblem Q = h->P;
  Q.scale_c = h->ones_c.p; Q.scale_p = h->ones_p.p;
  if (h->colsq_c0.n) HIP_TRY(hipMemsetAsync(h->colsq_c0.p, 0, sizeof(double) * h->colsq_c0.n, h->stream));
  if (h->colsq_p0.n) HIP_TRY(hipMemsetAsync(h->colsq_p0.p, 0, sizeof(double) * h->colsq_p0.n, h->stream));
  if (h->colsq_i0.n) HIP_TRY(hipMemsetAsync(h->colsq_i0.p, 0, sizeof(double) * h->colsq_i0.n, h->stream));
  Q.scale_i = h->ones_i.p;
  Q.intr = h->intr[h->cur].p;
  launch_colnorm(Q, h->cam[h->cur].p, h->pts[h->cur].p, h->colsq_c0.p, h->colsq_p0.p, h->colsq_i0.p, h->stream);
  if (Q.fused_bw > 0 && Q.n_fruns > 0) {
    // camera and intrinsics columns: one pass of the fused kernel over unit scales -- its per-(camera, row) lanes already
    // sum the squared column norms, k_sum_items leaves them by reduced index, the group columns summed over the group's
    // cameras (fixed order, no atomics).  The reduced system it writes on the way is cleared by the first linearisation.
    Q.fused_dbg |= 8;   // (no pair products in this pass)
    launch_linearize_fused_intr(Q, h->cam[h->cur].p, h->pts[h->cur].p, h->ones_c.p /* radius 1 */, h->rb, h->Vinv.p, h->tile_part.p, h->stream);
    launch_scatter_colsq(Q, h->rb.colsq, h->colsq_c0.p, h->colsq_i0.p, h->stream);
  }
  launch_long_colnorm(Q, h->cam[h->cur].p, h->pts[h->cur].p, h->colsq_c0.p, h->colsq_p0.p, h->long_scratch.p, h->stream, h->colsq_i0.p);
  launch_cam_priors(Q, PRIOR_COLNORM, h->cam[h->cur].p, nullptr, nullptr, nullptr, h->colsq_c0.p, nullptr, nullptr, h->stream);
  int rc = do_allreduce(h, h->colsq_c0.p, h->colsq_c0.n, THEIA_REDUCE_SUM);
  if (!rc && h->ni) rc = do_allreduce(h, h->colsq_i0.p, h->colsq_i0.n, THEIA_REDUCE_SUM);
  if (rc) return rc;
  launch_make_scale((int)h->colsq_c0.n, h->colsq_c0.p, h->scale_c.p, h->stream);
  launch_make_scale((int)h->colsq_p0.n, h->colsq_p0.p, h->scale_p.p, h->stream);
  launch_make_scale((int)h->colsq_i0.n, h->colsq_i0.p, h->scale_i.p, h->stream);
  launch_build_scale_red(h->P, h->scale_red.p, h->stream);
  h->have_scale = true; h->camrot_valid = false;   // (the blocks carry the Jacobi scaling of the extrinsics columns)
  return 0;
}

// enqueue: clear, linearize + Schur, tile reduction, (all-reduce), LM diagonal.
// The trust-region radius is read from the device-resident LM state.
// slot < 0: no phase-timing events (graph capture, or timing not requested)
int enqueue_linearize(theia_ba_handle_s* h, int slot = 0) {
  const double* radius = &reinterpret_cast<const LmState*>(h->lm_state.p)->radius;
  h->P.intr = h->intr[h->cur].p; h->P.intr_cand = h->intr[1 - h->cur].p;
  // clear the reduced system: the tiles the K3 plan knows (assembly + fill) and the vector tail; everything else in
  // the n x n buffer is never written (it was zeroed once at create())
  if (h->n > 0 && !(h->allreduce && h->n_pack_tiles == 0) &&
      chol_plan_clear(h->plan, h->rb.S, h->n, h->stream, h->rb.rhs, h->reduce.n - (size_t)h->n * h->n)) {
  } else
    HIP_TRY(hipMemsetAsync(h->reduce.p, 0, sizeof(double) * h->reduce.n, h->stream));
  if (slot >= 0) HIP_TRY(hipEventRecord(h->ev[slot][4], h->stream));
  // the state's per-camera blocks (k_cam_prep) are still current after the first body: an accepted step copies the candidate's
  // blocks over them (k_lm_accept), a rejected one leaves the state where it was.  Not with inner iterations (the accepted
  // point may be the swept one, and the sweep reuses the candidate's blocks) and not with free intrinsics.
  const bool keep_blocks = h->use_fused && h->ni == 0 && !h->inner && !getenv("THEIA_HIP_CAM_PREP_ALWAYS");
  h->P.camrot_current = keep_blocks && h->camrot_valid;
  launch_linearize(h->P, h->cam[h->cur].p, h->pts[h->cur].p, radius, h->rb, h->Vinv.p, h->gp.p, h->tile_part.p, h->stream);
  h->camrot_valid = keep_blocks;
  if (slot >= 0) HIP_TRY(hipEventRecord(h->ev[slot][5], h->stream));
  // without an all-reduce (and without phase timing) the tile reduction rides in k_finalize_rcs: one launch less
  const bool fuse_reduce = !h->allreduce && slot < 0 && h->ntiles_main > 0;
  if (h->ntiles_main && !fuse_reduce) launch_reduce_tiles(h->ntiles_main, h->tile_part.p, 4, h->f2s.p, h->fmaxflag.p, h->rb.scal, h->stream, h->red_part.p);
  launch_long_linearize(h->P, h->cam[h->cur].p, h->pts[h->cur].p, radius, h->rb, h->Vinv.p, h->gp.p, h->long_scratch.p, h->stream);
  launch_cam_priors(h->P, PRIOR_LINEARIZE, h->cam[h->cur].p, nullptr, nullptr, &h->rb, nullptr, h->rb.scal + SC_COST, nullptr, h->stream);
  // one SUM all-reduce of [S | rhs | colsq | gc | scal[0,8)], one MAX of scal[8,16) (folded into the SUM as
  // per-rank slots when the shard geometry is known)
  int rc = 0;
  bool max_done = false;
  if (h->dist_k3) {
    // the rank's private columns: LM diagonal, then their factorisation from this rank's sums alone (their Schur
    // complement lands in the shared tiles and the shared rows of the rhs, which the all-reduce sums next)
    launch_finalize_rcs(h->P, radius, h->rb, h->stream, 0, nullptr, nullptr, nullptr, h->d_tile_cls.p, 1);
    chol_plan_solve_phase(h->plan, 0, h->rb.S, h->n, h->rb.rhs, h->chol_work.p, h->rb.scal + SC_NOTPD, h->stream);
  }
  if (h->allreduce && h->n_pack_tiles > 0 && h->n > 0) {
    const int tail_blocks = (int)((3 * (size_t)h->n + 8 + 255) / 256);
    const int grid = h->n_pack_tiles + std::max(1, std::min(tail_blocks, 64));
    const int world = (h->shard_world > 0 && h->shard_rank >= 0 && h->shard_rank < h->shard_world && h->shard_world <= kMaxShardSlots)
                          ? h->shard_world : 0;
    k_pack_rcs<<<grid, 256, 0, h->stream>>>(h->n, h->reduce.p, h->pack_tiles.p, h->n_pack_tiles, h->pack_buf.p, 1, h->shard_rank, world);
    rc = do_allreduce(h, h->pack_buf.p, (size_t)h->n_pack_tiles * 4096 + 3 * (size_t)h->n + 8 + world, THEIA_REDUCE_SUM);
    k_pack_rcs<<<grid, 256, 0, h->stream>>>(h->n, h->reduce.p, h->pack_tiles.p, h->n_pack_tiles, h->pack_buf.p, 0, h->shard_rank, world);
    max_done = world > 0;
  } else {
    rc = do_allreduce(h, h->reduce.p, (size_t)h->n * h->n + 3 * (size_t)h->n + 8, THEIA_REDUCE_SUM);
  }
  if (!rc && !max_done) rc = do_allreduce(h, h->rb.scal + 8, 8, THEIA_REDUCE_MAX);
  if (rc) return rc;
  if (fuse_reduce && h->ntiles_main > 4 * kReduceBlocks) {   // two stages: one workgroup over 50k tile rows costs 45 us
    launch_reduce_tiles_stage1(h->ntiles_main, h->tile_part.p, 4, h->fmaxflag.p, h->red_part.p, h->stream);
    launch_finalize_rcs(h->P, radius, h->rb, h->stream, kReduceBlocks, h->red_part.p, h->f2s.p, h->fmaxflag.p);
  } else if (fuse_reduce) launch_finalize_rcs(h->P, radius, h->rb, h->stream, h->ntiles_main, h->tile_part.p, h->f2s.p, h->fmaxflag.p);
  else if (h->dist_k3) launch_finalize_rcs(h->P, radius, h->rb, h->stream, 0, nullptr, nullptr, nullptr, h->d_tile_cls.p, 0);   // the shared columns
  else launch_finalize_rcs(h->P, radius, h->rb, h->stream);
  return 0;
}

// enqueue: dense solve, candidate cameras, back-substitution + trial cost.
// defer_reduce: the tile reduction of the trial step is left to k_reduce_control (no all-reduce in between).
int enqueue_solve_and_backsub(theia_ba_handle_s* h, int slot = 0, bool defer_reduce = false) {
  double* yc = h->rb.rhs;  // the solution overwrites the rhs row
  if (h->dist_k3) {
    chol_plan_solve_phase(h->plan, 1, h->rb.S, h->n, h->rb.rhs, h->chol_work.p, h->rb.scal + SC_NOTPD, h->stream);
    k_y_own<<<(h->n + 255) / 256, 256, 0, h->stream>>>(h->n, yc, h->d_tile_cls.p, h->shard_rank);
    const int rcy = do_allreduce(h, yc, (size_t)h->n, THEIA_REDUCE_SUM);
    if (rcy) return rcy;
  } else
  chol_plan_solve(h->plan, h->rb.S, h->n, h->rb.rhs, h->chol_work.p, h->rb.scal + SC_NOTPD, h->stream);
  if (slot >= 0) HIP_TRY(hipEventRecord(h->ev[slot][2], h->stream));
  const int nxt = 1 - h->cur;
  launch_cam_update(h->P, h->cam[h->cur].p, yc, h->cam[nxt].p, h->ni ? h->intr[nxt].p : nullptr,
                    h->scalB.p + SB_STEPSQ_CAM, h->scalB.p + SB_XNORMSQ_CAM, h->stream, h->scalB.p);
  launch_backsub(h->P, h->cam[h->cur].p, h->pts[h->cur].p, h->cam[nxt].p, h->pts[nxt].p, yc, h->Vinv.p, h->tile_part.p, h->scalB.p, h->stream);
  if (h->ntiles_main && !defer_reduce) launch_reduce_tiles(h->ntiles_main, h->tile_part.p, 5, h->f2s.p + 8, h->fmaxflag.p + 8, h->scalB.p, h->stream, h->red_part.p);
  launch_long_backsub(h->P, h->cam[h->cur].p, h->pts[h->cur].p, h->cam[nxt].p, h->pts[nxt].p, yc, h->Vinv.p, h->long_scratch.p, h->scalB.p, h->stream);
  launch_cam_priors(h->P, PRIOR_TRIAL, h->cam[h->cur].p, h->cam[nxt].p, yc, nullptr, nullptr, h->scalB.p + SB_COST, h->scalB.p + SB_MCC, h->stream);
  return do_allreduce(h, h->scalB.p, 8, THEIA_REDUCE_SUM);
}

// The all-reduced S has the union of the ranks' tile structures: OR the tile
// co-visibility over the ranks (MAX all-reduce) and rebuild the K3 schedule.
int sync_plan(theia_ba_handle_s* h) {
  const int nt = (h->n + 63) / 64;
  const size_t cnt = (size_t)nt * nt;
  h->dist_k3 = false;
  if (cnt == 0 || !h->allreduce) { h->plan_is_global = true; return 0; }
  if (h->tile_adj_local.size() != cnt) h->tile_adj_local = h->tile_adj;   // this rank's own structure (tile_adj becomes the union)
  std::vector<double> a(cnt);
  for (size_t i = 0; i < cnt; ++i) a[i] = h->tile_adj_local[i];
  HIP_TRY(hipMemcpyAsync(h->reduce.p, a.data(), sizeof(double) * cnt, hipMemcpyHostToDevice, h->stream));
  int rc = do_allreduce(h, h->reduce.p, cnt, THEIA_REDUCE_MAX);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(a.data(), h->reduce.p, sizeof(double) * cnt, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  for (size_t i = 0; i < cnt; ++i) h->tile_adj[i] = a[i] != 0.0 ? 1 : 0;
  h->drop_graph();
  if (h->plan) chol_plan_destroy(h->plan);
  h->plan = nullptr;
  // Distributed K3 (round 4): with the shard geometry declared (set_shard / set_rccl), no intrinsics columns and more than one
  // rank, a tile column that only ONE rank's tracks touch is factored by that rank alone, before the all-reduce, which then
  // carries the shared tiles only.  Every rank takes the same decision from the same all-reduced numbers.
  const bool geom = h->shard_world > 1 && h->shard_rank >= 0 && h->shard_rank < h->shard_world && h->shard_world <= kMaxShardSlots;
  if (geom && h->ni == 0 && nt > 2 && !getenv("THEIA_HIP_K3_REPLICATED")) {
    std::vector<double> t(nt + 1, 0.0);
    for (int i = 0; i < nt; ++i) t[i] = (i < (int)h->tile_touch.size() && h->tile_touch[i]) ? 1.0 : 0.0;
    HIP_TRY(hipMemcpyAsync(h->reduce.p, t.data(), sizeof(double) * nt, hipMemcpyHostToDevice, h->stream));
    if ((rc = do_allreduce(h, h->reduce.p, (size_t)nt, THEIA_REDUCE_SUM))) return rc;
    std::vector<double> c(nt);
    HIP_TRY(hipMemcpyAsync(c.data(), h->reduce.p, sizeof(double) * nt, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->tile_cls.assign(nt, 0);
    int nmine = 0;
    for (int i = 0; i < nt; ++i)
      if (c[i] == 1.0) { h->tile_cls[i] = t[i] != 0.0 ? 1 : 2; nmine += t[i] != 0.0; }
    CholPlan* pl = chol_plan_create_sharded(h->n, h->tile_adj.data(), h->tile_cls.data());
    double bad = pl ? 0.0 : 1.0;      // agreed between the ranks: one rank without a level schedule keeps everybody replicated
    HIP_TRY(hipMemcpyAsync(h->reduce.p, &bad, sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if ((rc = do_allreduce(h, h->reduce.p, 1, THEIA_REDUCE_MAX))) { if (pl) chol_plan_destroy(pl); return rc; }
    HIP_TRY(hipMemcpyAsync(&bad, h->reduce.p, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (bad == 0.0) {
      h->plan = pl; h->dist_k3 = true;
      if ((rc = h->d_tile_cls.upload(h->tile_cls, h->stream))) return rc;
      if (getenv("THEIA_HIP_CREATE_TIMING"))
        fprintf(stderr, "theia_hip distributed K3: rank %d of %d, %d of %d tiles private, %d levels before the all-reduce, %zu shared tiles in it\n",
                h->shard_rank, h->shard_world, nmine, nt, chol_plan_split_level(pl), chol_plan_shared_tiles(pl).size() / 2);
    } else {
      if (getenv("THEIA_HIP_CREATE_TIMING"))
        fprintf(stderr, "theia_hip distributed K3: rank %d of %d keeps the replicated plan (%d of %d tiles private here; %s)\n", h->shard_rank,
                h->shard_world, nmine, nt, pl ? "another rank has no level schedule" : "no level schedule for this rank's structure");
      if (pl) chol_plan_destroy(pl);
    }
  } else if (getenv("THEIA_HIP_CREATE_TIMING") && h->shard_world > 1)
    fprintf(stderr, "theia_hip distributed K3: not taken (shard geometry %d/%d, %d intrinsics columns, %d tiles)\n", h->shard_rank, h->shard_world, h->ni, nt);
  if (!h->plan) h->plan = chol_plan_create(h->n, h->tile_adj.data());
  h->plan_is_global = true;
  HIP_TRY(hipMemsetAsync(h->reduce.p, 0, sizeof(double) * h->reduce.n, h->stream));   // tiles only the old plan touched
  {
    std::vector<int2> tiles;
    if (h->dist_k3) {
      const std::vector<int>& st = chol_plan_shared_tiles(h->plan);
      for (size_t k = 0; k + 1 < st.size(); k += 2) tiles.push_back(make_int2(st[k], st[k + 1]));
      if (tiles.empty()) tiles.push_back(make_int2(0, 0));   // (the packed path needs one)
    } else
    for (int i = 0; i < nt; ++i)
      for (int j = 0; j <= i; ++j)
        if (i == j || h->tile_adj[(size_t)i * nt + j]) tiles.push_back(make_int2(i, j));
    h->n_pack_tiles = (int)tiles.size();
    int rc2 = h->pack_tiles.upload(tiles, h->stream);
    if (!rc2) rc2 = h->pack_buf.alloc((size_t)tiles.size() * 4096 + 3 * (size_t)h->n + 8 + kMaxShardSlots);
    if (rc2) return rc2;
    HIP_TRY(hipStreamSynchronize(h->stream));
  }
  return 0;
}

}  // namespace

extern "C" {

void theia_ba_options_default(theia_ba_options* o) {
  std::memset(o, 0, sizeof(*o));
  o->loss_function_type = THEIA_LOSS_TRIVIAL;
  o->robust_loss_width = 2.0;
  o->intrinsics_to_optimize = THEIA_INTR_NONE;
  o->max_num_iterations = 100;
  o->use_homogeneous_point_parametrization = 1;
  o->use_inner_iterations = 1;
  o->function_tolerance = 1e-6;
  o->gradient_tolerance = 1e-10;
  o->parameter_tolerance = 1e-8;
  o->max_trust_region_radius = 1e12;
  o->max_solver_time_in_seconds = 3600.0;
  o->robust_loss_width_depth_prior = 0.01;   // bundle_adjustment.h:94
}

#define UP(buf, vec) do { rc = h->buf.upload(vec, st); if (rc) return rc; } while (0)
#define AL(buf, cnt) do { rc = h->buf.alloc(cnt); if (rc) return rc; } while (0)
// Static gather lists of the Schur assembly without intrinsics (k_lin_obs / k_schur); see create().
// with_pairs = false (fused Schur assembly): only the per-camera observation lists the column-norm pass uses.
int build_gather_lists(theia_ba_handle_s* h, const int* ocam, const int* opt,
                       const std::vector<int>& l_obs, bool with_pairs = true) {
  int rc = 0;
  hipStream_t st = h->stream;
  // Static gather lists of the Schur assembly (k_schur_diag / k_schur_blocks):
  // per reduced camera its observations, per camera pair (ri > rj) the
  // (observation of ri, observation of rj) pairs of their common variable
  // tracks.  Tracks of the slow path (> 64 observations) assemble themselves.
  const int64_t nm = h->nobs_main;
  std::vector<char> is_long(l_obs.empty() ? 0 : nm, 0);
  for (int s2 : l_obs) is_long[s2] = 1;
  constexpr int kChunk = 2048;
  if (!with_pairs) {
    // Fused Schur assembly: only the cameras' observation lists are needed (column norms of the camera blocks,
    // k_colnorm_gather) -- a stable counting sort by reduced camera over a fixed partition of the observations, on host
    // threads, written into a pinned block.
    constexpr int kParts = 32;
    auto red_of = [&](int64_t s) { return (!is_long.empty() && is_long[s]) ? -1 : h->cam_red[ocam[s]]; };
    std::vector<std::vector<int>> fill(kParts, std::vector<int>(std::max(1, h->ncv), 0));
    const bool threaded = nm >= 262144;
    host_parts(kParts, threaded, [&](int k) {
      for (int64_t s = nm * k / kParts; s < nm * (k + 1) / kParts; ++s) { const int r = red_of(s); if (r >= 0) fill[k][r]++; }
    });
    std::vector<int> dbeg(h->ncv + 1, 0);
    for (int c = 0; c < h->ncv; ++c) {
      int at = dbeg[c];
      for (int k = 0; k < kParts; ++k) { const int cnt = fill[k][c]; fill[k][c] = at; at += cnt; }
      dbeg[c + 1] = at;
    }
    HBuf<int> sobs;
    const size_t nrec = (size_t)std::max(1, dbeg[h->ncv]);
    if (!sobs.resize(nrec, true)) return set_error(THEIA_HIP_ERR_OUT_OF_MEMORY, "host staging of %zu records failed", nrec);
    sobs[0] = 0;
    host_parts(kParts, threaded, [&](int k) {
      for (int64_t s = nm * k / kParts; s < nm * (k + 1) / kParts; ++s) { const int r = red_of(s); if (r >= 0) sobs[fill[k][r]++] = (int)s; }
    });
    std::vector<int> ditems;
    for (int c = 0; c < h->ncv; ++c) {
      const int nchunk = (dbeg[c + 1] - dbeg[c] + kChunk - 1) / kChunk;
      for (int k = 0; k < nchunk; ++k) {
        ditems.push_back(c); ditems.push_back(dbeg[c] + k * kChunk);
        ditems.push_back(std::min(dbeg[c + 1], dbeg[c] + (k + 1) * kChunk)); ditems.push_back(nchunk > 1 ? 1 : 0);
      }
    }
    h->n_diag_items = (int)ditems.size() / 4; h->n_blk_items = 0;
    if ((rc = h->slot_obs.upload(sobs.data(), nrec, st, sobs.pinned()))) return rc;
    UP(diag_items, ditems);   // pageable: synchronises the stream, the pinned block above is free after it
    return 0;
  }
  std::vector<int> red(nm);
  for (int64_t s = 0; s < nm; ++s) red[s] = (!is_long.empty() && is_long[s]) ? -1 : h->cam_red[ocam[s]];
  std::vector<int> dbeg(h->ncv + 1, 0);
  for (int64_t s = 0; s < nm; ++s) if (red[s] >= 0) dbeg[red[s] + 1]++;
  for (int c = 0; c < h->ncv; ++c) dbeg[c + 1] += dbeg[c];
  // records are stored camera-major: slot of observation s = its rank in its camera's list
  std::vector<int> cam_obs(nm, -1);   // = rec_slot
  {
    std::vector<int> f(dbeg.begin(), dbeg.end() - 1);
    for (int64_t s = 0; s < nm; ++s) if (red[s] >= 0) cam_obs[s] = f[red[s]]++;
  }
  std::vector<int> ditems;
  for (int c = 0; c < h->ncv; ++c) {
    const int nchunk = (dbeg[c + 1] - dbeg[c] + kChunk - 1) / kChunk;
    for (int k = 0; k < nchunk; ++k) {
      ditems.push_back(c); ditems.push_back(dbeg[c] + k * kChunk);
      ditems.push_back(std::min(dbeg[c + 1], dbeg[c] + (k + 1) * kChunk)); ditems.push_back(nchunk > 1 ? 1 : 0);
    }
  }
  // pairs, bucketed by row camera then sorted by column camera
  std::vector<int64_t> rbeg(h->ncv + 1, 0);
  auto for_each_pair = [&](auto&& fn) {
    for (int64_t s0 = 0; s0 < nm;) {
      int64_t s1 = s0 + 1;
      while (s1 < nm && opt[s1] == opt[s0]) ++s1;
      if ((is_long.empty() || !is_long[s0]) && !h->pt_const[opt[s0]])
        for (int64_t a = s0; a < s1; ++a) {
          if (red[a] < 0) continue;
          for (int64_t b = s0; b < s1; ++b)
            if (red[b] >= 0 && (red[a] > red[b] || (red[a] == red[b] && a != b))) fn((int)a, (int)b);
        }
      s0 = s1;
    }
  };
  for_each_pair([&](int a, int) { rbeg[red[a] + 1]++; });
  for (int c = 0; c < h->ncv; ++c) rbeg[c + 1] += rbeg[c];
  if (rbeg[h->ncv] > (int64_t)std::numeric_limits<int>::max() - 64)
    return set_error(THEIA_HIP_ERR_UNSUPPORTED, "too many camera pairs for 32-bit pair lists");
  std::vector<int2> pairs(rbeg[h->ncv]);
  {
    std::vector<int64_t> f(rbeg.begin(), rbeg.end() - 1);
    for_each_pair([&](int a, int b) { pairs[f[red[a]]++] = make_int2(a, b); });
  }
  std::vector<int> bitems;
  // each row is ordered by (column camera, a, b).  The pairs of a row were generated in ascending (a, b)
  // (tracks are contiguous and visited in order), so a STABLE counting sort on the column camera is enough.
  {
    std::vector<int64_t> cnt(h->ncv + 1);
    std::vector<int2> tmp;
    for (int c = 0; c < h->ncv; ++c) {
      const int64_t b0 = rbeg[c], b1 = rbeg[c + 1];
      if (b1 - b0 < 2) continue;
      std::fill(cnt.begin(), cnt.end(), 0);
      for (int64_t q = b0; q < b1; ++q) cnt[red[pairs[q].y] + 1]++;
      for (int k = 0; k < h->ncv; ++k) cnt[k + 1] += cnt[k];
      tmp.assign(pairs.begin() + b0, pairs.begin() + b1);
      for (const int2& pr : tmp) pairs[b0 + cnt[red[pr.y]]++] = pr;
    }
  }
  for (int c = 0; c < h->ncv; ++c) {
    for (int64_t q = rbeg[c]; q < rbeg[c + 1];) {
      int64_t e = q + 1;
      const int rj = red[pairs[q].y];
      while (e < rbeg[c + 1] && red[pairs[e].y] == rj) ++e;
      const int nchunk = (int)((e - q + kChunk - 1) / kChunk);
      for (int k = 0; k < nchunk; ++k) {
        bitems.push_back(c); bitems.push_back(rj); bitems.push_back((int)(q + (int64_t)k * kChunk));
        bitems.push_back((int)std::min<int64_t>(e, q + (int64_t)(k + 1) * kChunk));
        bitems.push_back((nchunk > 1 || rj == c) ? 1 : 0);
      }
      q = e;
    }
  }
  // a camera that sees a track twice also has a (c, c) pair list: both kinds of items then ADD
  // into the diagonal block (they run in one launch, unordered)
  {
    std::vector<char> self(h->ncv, 0);
    for (size_t k = 0; k + 4 < bitems.size() + 1; k += 5) if (bitems[k] == bitems[k + 1]) self[bitems[k]] = 1;
    for (size_t k = 0; k + 3 < ditems.size() + 1; k += 4) if (self[ditems[k]]) ditems[k + 3] = 1;
  }
  // XCD-aware order: workgroup b runs on XCD b % 8 (observed dispatch order, a speed matter only).  All items whose
  // ROW camera is c are placed on XCD c % 8, so that camera's records are fetched into one L2 and re-used by its
  // block items and its diagonal item instead of being pulled into all eight.
  if (!getenv("THEIA_HIP_NO_XCD_ORDER")) {
    auto reorder = [&](std::vector<int>& items, int stride, int first_wg) {
      const int n = (int)items.size() / stride;
      std::vector<std::vector<int>> bucket(8);
      for (int k = 0; k < n; ++k) bucket[items[(size_t)k * stride] & 7].push_back(k);
      std::vector<size_t> head(8, 0);
      std::vector<int> out;
      out.reserve(items.size());
      for (int pos = 0; pos < n; ++pos) {
        int x = (first_wg + pos) & 7;
        if (head[x] >= bucket[x].size()) {   // that XCD's list is exhausted: take from the fullest one
          size_t best = 0;
          for (int y = 0; y < 8; ++y) { const size_t left = bucket[y].size() - head[y]; if (left > best) { best = left; x = y; } }
        }
        const int k = bucket[x][head[x]++];
        out.insert(out.end(), items.begin() + (size_t)k * stride, items.begin() + (size_t)(k + 1) * stride);
      }
      items.swap(out);
    };
    reorder(bitems, 5, 0);
    reorder(ditems, 4, (int)bitems.size() / 5);
  }
  h->n_diag_items = (int)ditems.size() / 4; h->n_blk_items = (int)bitems.size() / 5;
  for (auto& pr : pairs) { pr.x = cam_obs[pr.x]; pr.y = cam_obs[pr.y]; }
  {
    std::vector<int> sobs(std::max(1, dbeg[h->ncv]), 0);
    for (int64_t s2 = 0; s2 < nm; ++s2) if (cam_obs[s2] >= 0) sobs[cam_obs[s2]] = (int)s2;
    UP(slot_obs, sobs);
    std::vector<int> spt(sobs.size(), 0);
    for (int64_t s2 = 0; s2 < nm; ++s2) if (cam_obs[s2] >= 0) spt[cam_obs[s2]] = opt[s2];
    UP(slot_pt, spt);
  }
  UP(diag_items, ditems); UP(cam_obs, cam_obs); UP(blk_items, bitems); UP(blk_pairs, pairs);
  AL(rec, (size_t)std::max(1, dbeg[h->ncv]) * (6 * h->pd + 14));
  return 0;
}

// The gather lists when intrinsics are optimised (k_lin_obs_intr / k_schur_intr); see create().
int build_gather_lists_intr(theia_ba_handle_s* h, const theia_ba_problem* p, const int* ocam,
                            const int* opt, const std::vector<int>& l_obs) {
  const bool itiming = getenv("THEIA_HIP_CREATE_TIMING") != nullptr;
  double it0 = now_s();
  auto itick = [&](const char* what) {
    if (!itiming) return;
    const double t = now_s();
    fprintf(stderr, "theia_hip create:     intrinsics lists: %-24s %8.2f ms\n", what, 1e3 * (t - it0));
    it0 = t;
  };
  int rc = 0;
  hipStream_t st = h->stream;
  // Gather lists with intrinsics (k_lin_obs_intr / k_schur_intr): records for every observation whose camera OR
  // intrinsics group is variable, stored (group, camera)-major; item = {type, row0, col0, beg, end, flags}.
  enum { IT_CC = 0, IT_CG = 1, IT_GG0 = 2, IT_GG1 = 3, IT_CD = 4, IT_CGD = 5, IT_GD0 = 6, IT_GD1 = 7, IT_GV = 8 };
  const int64_t nm = h->nobs_main;
  constexpr int kChunk = 2048;
  std::vector<int> red(nm), grd(nm);
  for (int64_t s = 0; s < nm; ++s) { red[s] = h->cam_red[ocam[s]]; grd[s] = h->grp_red[p->cam_group[ocam[s]]]; }
  for (int s2 : l_obs) { red[s2] = -1; grd[s2] = -1; }   // tracks of the slow path (k_long_*) assemble themselves
  // slots: sort the observations that need a record by (group, camera)
  std::vector<int> order;
  for (int64_t s = 0; s < nm; ++s) if (red[s] >= 0 || grd[s] >= 0) order.push_back((int)s);
  {   // stable sort by (group, camera): two counting passes, camera first (keys start at -1)
    std::vector<int> tmp(order.size());
    for (int pass = 0; pass < 2; ++pass) {
      const std::vector<int>& key = pass == 0 ? red : grd;
      std::vector<size_t> cnt((size_t)(pass == 0 ? h->ncv : h->ngv) + 2, 0);
      for (int x : order) cnt[(size_t)(key[x] + 1) + 1]++;
      for (size_t k = 0; k + 1 < cnt.size(); ++k) cnt[k + 1] += cnt[k];
      for (int x : order) tmp[cnt[(size_t)(key[x] + 1)]++] = x;
      order.swap(tmp);
    }
  }
  std::vector<int> slot(nm, -1);
  for (size_t k = 0; k < order.size(); ++k) slot[order[k]] = (int)k;
  itick("record order");
  std::vector<int> items;
  auto push_item = [&](int type, int row0, int col0, int64_t beg, int64_t end, int flags) {
    const int nchunk = (int)((end - beg + kChunk - 1) / kChunk);
    for (int k = 0; k < nchunk; ++k) {
      items.push_back(type); items.push_back(row0); items.push_back(col0);
      items.push_back((int)(beg + (int64_t)k * kChunk)); items.push_back((int)std::min<int64_t>(end, beg + (int64_t)(k + 1) * kChunk));
      items.push_back(flags | (nchunk > 1 ? 1 : 0));
    }
  };
  // ---- pair lists: entries (key, slot a, slot b) sorted by key; one item (or two halves) per key
  // Camera x camera blocks take every ordered pair of observations of a track.  For the blocks with an intrinsics group
  // on one or both sides, a track that sees few variable groups is represented by the SUMS of its observations'
  // intrinsics fields, one per group (pseudo-records behind the real ones, written by k_lin_obs_intr from segmented wave sums):
  // Sum_b T_a WI_b^T = T_a (Sum_b WI_b)^T, so the track gives L (observation, sum) pairs and one (sum, sum) pair instead
  // of 2 L (L - 1) ordered pairs.  The sum includes b == a, so the per-observation items leave that term out for the
  // observations of such a track (slot_in_sum).  Tracks that see several variable groups keep their explicit pairs.
  // THEIA_HIP_INTR_PAIRS=1 keeps the lists of the first version (no sums).
  struct PairE { uint64_t key; int a, b; };
  const bool track_sums = !getenv("THEIA_HIP_INTR_PAIRS");
  const int nslots = (int)order.size();
  // The tracks are walked twice on host threads, in a fixed number of ranges of consecutive tracks: the first walk counts a
  // range's entries and pseudo-records, the second writes them at the offsets the counts give -- the lists come out in track
  // order whatever the number of threads (one thread pushing ~20 M entries into growing vectors took 145 ms at 1000 views).
  std::vector<int64_t> tbeg;   // first sorted observation of every track with non-fixed observations, then nm
  for (int64_t s0 = 0; s0 < nm;) {
    int64_t s1 = s0 + 1;
    while (s1 < nm && opt[s1] == opt[s0]) ++s1;
    tbeg.push_back(s0);
    s0 = s1;
  }
  tbeg.push_back(nm);
  const int64_t ntrk = (int64_t)tbeg.size() - 1;
  constexpr int kTrackParts = 64;
  struct PartCount { size_t cc = 0, cg = 0, gg = 0; int sums = 0; };
  std::vector<PartCount> pc(kTrackParts + 1);
  std::vector<int> pt_sum(track_sums ? h->np : 0, -1);   // pseudo-record slot of a track, -1 = none
  std::vector<uint8_t> slot_sum(track_sums ? std::max(1, nslots) : 0, 0);
  std::vector<uint8_t> pt_cnt(track_sums ? h->np : 0, 0);   // number of summed groups of a track
  std::vector<int> sum_group;                               // group of a pseudo-record
  PairE* ccp = nullptr; PairE* cgp = nullptr; PairE* ggp = nullptr;
  // one track: sink_cc / sink_cg / sink_gg receive its entries in the order of the one-thread loop; `ps` is its first pseudo-record
  auto walk_track = [&](int64_t s0, int64_t s1, int ps, bool write, PartCount& n) {
    if (h->pt_const[opt[s0]]) return;
    // the track's variable groups in order of first appearance (long tracks carry -1 everywhere: none).  A track is
    // summed per group when that shortens its lists: one group, or fewer groups (at most kMaxSumGroups) than
    // observations of variable groups
    constexpr int kMaxSumGroups = 4;
    int tgs[kMaxSumGroups], ntg = 0, lg = 0;
    bool many = false;
    for (int64_t b2 = s0; b2 < s1; ++b2) {
      if (grd[b2] < 0) continue;
      ++lg;
      bool seen = false;
      for (int k = 0; k < ntg; ++k) seen |= tgs[k] == grd[b2];
      if (seen) continue;
      if (ntg == kMaxSumGroups) { many = true; break; }
      tgs[ntg++] = grd[b2];
    }
    const bool sum_mode = track_sums && !many && ntg >= 1 && (ntg == 1 || ntg < lg);
    auto put = [&](PairE* base, size_t& at, uint64_t key, int x, int y) { if (write) base[at] = PairE{key, x, y}; ++at; };
    for (int64_t x = s0; x < s1; ++x)
      for (int64_t y = s0; y < s1; ++y) {
        if (x == y) continue;
        if (red[x] >= 0 && red[y] >= 0 && red[x] >= red[y]) put(ccp, n.cc, ((uint64_t)red[x] << 32) | (uint32_t)red[y], slot[x], slot[y]);
        if (sum_mode) continue;
        if (red[x] >= 0 && grd[y] >= 0) put(cgp, n.cg, ((uint64_t)red[x] << 32) | (uint32_t)grd[y], slot[x], slot[y]);
        if (grd[x] >= 0 && grd[y] >= 0 && grd[x] >= grd[y]) put(ggp, n.gg, ((uint64_t)grd[x] << 32) | (uint32_t)grd[y], slot[x], slot[y]);
      }
    if (sum_mode) {
      const int my = ps + n.sums;   // pseudo-records my .. my + ntg - 1, one per group
      if (write) {
        pt_sum[opt[s0]] = my; pt_cnt[opt[s0]] = (uint8_t)ntg;
        for (int k = 0; k < ntg; ++k) sum_group[(size_t)(my - nslots + k)] = tgs[k];
      }
      for (int64_t x = s0; x < s1; ++x) {
        if (write && slot[x] >= 0) slot_sum[slot[x]] = 1;
        if (red[x] >= 0)
          for (int k = 0; k < ntg; ++k) put(cgp, n.cg, ((uint64_t)red[x] << 32) | (uint32_t)tgs[k], slot[x], my + k);
      }
      for (int k = 0; k < ntg; ++k)
        for (int k2 = 0; k2 < ntg; ++k2)
          if (tgs[k] >= tgs[k2]) put(ggp, n.gg, ((uint64_t)tgs[k] << 32) | (uint32_t)tgs[k2], my + k, my + k2);
      n.sums += ntg;
    }
  };
  auto part_range = [&](int k, int64_t* t0, int64_t* t1) { *t0 = ntrk * k / kTrackParts; *t1 = ntrk * (k + 1) / kTrackParts; };
  host_parts(kTrackParts, ntrk >= 4096, [&](int k) {
    int64_t t0, t1; part_range(k, &t0, &t1);
    PartCount n;
    for (int64_t t = t0; t < t1; ++t) walk_track(tbeg[t], tbeg[t + 1], 0, false, n);
    pc[k + 1] = n;
  });
  for (int k = 0; k < kTrackParts; ++k) { pc[k + 1].cc += pc[k].cc; pc[k + 1].cg += pc[k].cg; pc[k + 1].gg += pc[k].gg; pc[k + 1].sums += pc[k].sums; }
  const int nsums = pc[kTrackParts].sums;
  std::vector<PairE, NoInitAlloc<PairE>> cc, cg, gg;   // (resize() does not touch the 100s of MB)
  {
    cc.resize(pc[kTrackParts].cc); cg.resize(pc[kTrackParts].cg); gg.resize(pc[kTrackParts].gg);
    sum_group.assign((size_t)nsums, 0);
    ccp = cc.data(); cgp = cg.data(); ggp = gg.data();
  }
  host_parts(kTrackParts, ntrk >= 4096, [&](int k) {
    int64_t t0, t1; part_range(k, &t0, &t1);
    PartCount n = pc[k];
    for (int64_t t = t0; t < t1; ++t) walk_track(tbeg[t], tbeg[t + 1], nslots, true, n);
  });
  itick("pair entries");
  // (the pair list is written into a block of the pinned host cache and uploaded from there)
  HBuf<int2> pairs;
  size_t npairs = 0;
  if (!pairs.resize(std::max<size_t>(1, cc.size() + cg.size() + gg.size()), true))
    return set_error(THEIA_HIP_ERR_OUT_OF_MEMORY, "host staging of %zu pairs failed", cc.size() + cg.size() + gg.size());
  // entries are generated in ascending (a, b): two stable counting passes (low, then high half of the key)
  // order them by (key, a, b) without a comparison sort
  const size_t nbucket = (size_t)std::max(h->ncv, h->ngv) + 2;
  // (one uninitialised scratch array for the three lists: a value-initialised vector of the 10 - 25 M entries was a 100+ MB memset,
  // and a fresh one per list two more rounds of page faults)
  RawArray<PairE> tmp_raw(std::max(cc.size(), std::max(cg.size(), gg.size())));
  auto emit_pairs = [&](auto& v, auto&& per_key) {
    const size_t nv = v.size();
    PairE* a = v.data();
    PairE* b = tmp_raw.data();
    {   // each pass over a fixed 32-way partition of the entries with per-part histograms, on host threads (stable)
      constexpr int kParts = 32;
      const bool threaded = nv >= 262144;
      std::vector<std::vector<size_t>> cnt(kParts, std::vector<size_t>(nbucket));
      for (int pass = 0; pass < 2; ++pass) {
        const int sh = pass == 0 ? 0 : 32;
        host_parts(kParts, threaded, [&](int k) {
          std::fill(cnt[k].begin(), cnt[k].end(), 0);
          for (size_t i = nv * k / kParts; i < nv * (k + 1) / kParts; ++i) cnt[k][(size_t)((a[i].key >> sh) & 0xffffffffu)]++;
        });
        size_t at = 0;
        for (size_t bk = 0; bk < nbucket; ++bk)
          for (int k = 0; k < kParts; ++k) { const size_t c = cnt[k][bk]; cnt[k][bk] = at; at += c; }
        host_parts(kParts, threaded, [&](int k) {
          for (size_t i = nv * k / kParts; i < nv * (k + 1) / kParts; ++i) b[cnt[k][(size_t)((a[i].key >> sh) & 0xffffffffu)]++] = a[i];
        });
        std::swap(a, b);
      }
    }
    // (two passes: the sorted entries are back in v) -- the pair list is their (a, b) columns, copied on host threads; the key
    // boundaries come from one scan
    const int64_t base = (int64_t)npairs;
    npairs += nv;
    int2* out = pairs.data() + base;
    constexpr int kScanParts = 32;
    std::vector<std::vector<size_t>> starts(kScanParts);   // first entries of the keys, per range of the scan
    host_parts(kScanParts, nv >= 262144, [&](int k) {
      for (size_t i = nv * k / kScanParts; i < nv * (k + 1) / kScanParts; ++i) {
        out[i] = make_int2(a[i].a, a[i].b);
        if (i == 0 || a[i].key != a[i - 1].key) starts[k].push_back(i);
      }
    });
    std::vector<size_t> first;
    for (const auto& v2 : starts) first.insert(first.end(), v2.begin(), v2.end());
    first.push_back(nv);
    for (size_t k = 0; k + 1 < first.size(); ++k) {
      const size_t q = first[k], e = first[k + 1];
      per_key((int)(a[q].key >> 32), (int)(a[q].key & 0xffffffffu), base + (int64_t)q, base + (int64_t)e);
    }
  };
  // cameras seen twice by a track give (c, c) lists: the camera block is then fed by two kinds of items
  std::vector<char> self(h->ncv, 0);
  for (const PairE& e : cc) if ((e.key >> 32) == (e.key & 0xffffffffu)) self[e.key >> 32] = 1;
  emit_pairs(cc, [&](int ra, int rb, int64_t beg, int64_t end) {
    push_item(IT_CC, h->ni + 6 * ra, h->ni + 6 * rb, beg, end, ra == rb ? 3 : 0);
  });
  emit_pairs(cg, [&](int ra, int gb, int64_t beg, int64_t end) { push_item(IT_CG, h->ni + 6 * ra, 10 * gb, beg, end, 1); });
  const bool compact = h->intr_rows == 4;   // four compact intrinsics rows: one GG / GD item instead of the 4 + 6 split
  emit_pairs(gg, [&](int ga, int gb, int64_t beg, int64_t end) {
    push_item(IT_GG0, 10 * ga, 10 * gb, beg, end, 1 | (ga == gb ? 2 : 0));
    if (!compact) push_item(IT_GG1, 10 * ga, 10 * gb, beg, end, 1 | (ga == gb ? 2 : 0));
  });
  // (the CG / GG targets also receive the per-observation diagonal items below: always atomic)
  itick("sorted pair lists");
  if (npairs > (size_t)std::numeric_limits<int>::max() - 64)
    return set_error(THEIA_HIP_ERR_UNSUPPORTED, "too many observation pairs for 32-bit pair lists");
  // ---- per-observation (diagonal) items over contiguous slot ranges
  for (size_t q = 0; q < order.size();) {   // per camera (inside its group)
    size_t e = q + 1;
    while (e < order.size() && red[order[e]] == red[order[q]] && grd[order[e]] == grd[order[q]]) ++e;
    const int rc = red[order[q]], gr = grd[order[q]];
    if (rc >= 0) {
      push_item(IT_CD, h->ni + 6 * rc, h->ni + 6 * rc, (int64_t)q, (int64_t)e, self[rc] ? 1 : 0);
      if (gr >= 0) push_item(IT_CGD, h->ni + 6 * rc, 10 * gr, (int64_t)q, (int64_t)e, 1);
    }
    q = e;
  }
  for (size_t q = 0; q < order.size();) {   // per group
    size_t e = q + 1;
    while (e < order.size() && grd[order[e]] == grd[order[q]]) ++e;
    const int gr = grd[order[q]];
    if (gr >= 0) {
      push_item(IT_GD0, 10 * gr, 10 * gr, (int64_t)q, (int64_t)e, 1);
      if (!compact) push_item(IT_GD1, 10 * gr, 10 * gr, (int64_t)q, (int64_t)e, 1);
      push_item(IT_GV, 10 * gr, 10 * gr, (int64_t)q, (int64_t)e, 0);
    }
    q = e;
  }
  h->n_diag_items = 0; h->n_blk_items = (int)items.size() / 6;
  {
    std::vector<int> sobs(order.begin(), order.end());
    if (sobs.empty()) sobs.push_back(0);
    UP(slot_obs, sobs);
  }
  UP(cam_obs, slot); UP(blk_items, items);
  if ((rc = h->blk_pairs.upload(pairs.data(), std::max<size_t>(1, npairs), st, pairs.pinned()))) return rc;
  itick("items + uploads");
  h->n_trk_sums = nsums;
  h->sum_base = nslots;
  if (nsums) { UP(pt_sum_slot, pt_sum); UP(slot_in_sum, slot_sum); UP(pt_sum_cnt, pt_cnt); UP(sum_group, sum_group); }
  AL(rec, (std::max<size_t>(1, order.size()) + (size_t)h->n_trk_sums) * (12 * h->pd + 20 + 2 * h->intr_rows * h->pd + 3 * h->intr_rows));
  return 0;
}

// ---------------------------------------------------------------------------------------------------------
// Plan of the fused linearise + Schur kernel (ba_fused.hip): wave tiles, runs, local camera tables, target
// blocks, and the per-S-block lists of partial sums.  Replaces build_tiles() for the main tiles.
//   off    : [np + 1] offsets of the tracks (by rank) into the sorted observation arrays
//   sred   : [nobs_main] reduced camera index of a sorted observation (-1 = constant camera)
//   skey   : [np] ordering key of a track rank (its first variable camera)
// Tracks that do not fit the fused kernel (more than 64 observations, more than kFusedMaxCams variable cameras,
// a camera seen twice) go to the per-observation slow path (k_long_*), like the > 64 ones before.
// local cameras a run of the fused plan may hold: its target blocks (pairs, diagonal included) times the lanes per block fit the
// 256 threads of a workgroup -- 22 cameras / 253 blocks of one lane; compound blocks of width bw: three rows per lane
inline int fused_run_cameras(int bw) {
  if (bw == 0) return kFusedMaxCams;
  if (bw == 9) return kFusedMaxCamsIntr;   // 78 blocks x 3 lanes
  if (bw == 10) return 10;                 // 55 x 4
  return 10;                               // 13 / 16 rows: 55 x 4
}
// lanes of a target block (ba_fused_intr.hip: lanes_per_target): three rows per lane up to width 10, four at 13
inline int fused_lanes_per_target(int bw) { return bw == 0 ? 1 : (bw <= 10 ? (bw + 2) / 3 : (bw + 3) / 4); }

struct FusedHost {
  std::vector<FusedRun> runs;
  std::vector<int> cams, stage, tile_trk_end, sum_items, sum_src;
  std::vector<unsigned short> tgts;
  std::vector<uint8_t> obs_lc, obs_tl;
  std::vector<uint8_t> tile_adj;   // [nt][nt] 64-wide tiles of S coupled by a variable track (the K3 plan's input)
  size_t part_doubles = 0;
};
// One contiguous range [q_begin, q_end) of track ranks -> the tiles / runs / long tracks of that range, with tile, camera-table,
// target and partial-sum offsets relative to the segment (merge_fused_segments rebases them).  obs_lc / obs_tl are indexed
// by observation: segments write disjoint parts of the shared arrays.
struct FusedSegment {
  std::vector<int> tstart, tcount, tkey, l_obs, l_slot, l_start, l_pt;
  FusedHost fp;
};
void build_fused_segment(const theia_ba_handle_s* h, const std::vector<int64_t>& off, const std::vector<int>& porder,
                         const int* sred, const int* ocam, const std::vector<int>& skey, int q_begin, int q_end,
                         uint8_t* obs_lc, uint8_t* obs_tl, FusedSegment& seg) {
  std::vector<int>& tstart = seg.tstart; std::vector<int>& tcount = seg.tcount; std::vector<int>& tkey = seg.tkey;
  std::vector<int>& l_obs = seg.l_obs; std::vector<int>& l_slot = seg.l_slot; std::vector<int>& l_start = seg.l_start;
  std::vector<int>& l_pt = seg.l_pt;
  FusedHost& fp = seg.fp;
  const int64_t nm = h->nobs_main;
  const int tps = fused_tiles_per_subchunk(h->pd);
  // geometry of the consumer lanes: lanes per target block, rows per local camera, partial-sum doubles (ba_fused.hip:
  // one lane per 6 x 6 block; ba_fused_intr.hip: 3 or 4 lanes per compound block, stored 10 x 10)
  const int bw = h->fused_bw;
  const size_t lanes_tgt = bw == 0 ? 1 : (size_t)fused_lanes_per_target(bw), rows_cam = bw == 0 ? 6 : (size_t)bw;
  const size_t part_tgt = bw == 0 ? 36 : (size_t)bw * bw, part_cam = bw == 0 ? 18 : (size_t)3 * bw;   // compound blocks: BW x BW per target, BW x 3 per camera
  const int max_cams = fused_run_cameras(bw);
  const size_t max_tgts = bw == 0 ? 253 : 256 / lanes_tgt;
  // track slices per consumer wave for a run of ntgt target blocks over W cameras (0 = needs more than one wave)
  auto packing = [&](size_t ntgt, size_t W) -> int {
    if (lanes_tgt * ntgt > 64 || rows_cam * W > 64) return 0;
    static const int cand[6] = {10, 6, 4, 3, 2, 1};
    for (int c : cand) if ((size_t)(64 / c) >= lanes_tgt * ntgt) return c;
    return 1;
  };
  const char* rm = getenv("THEIA_HIP_FUSED_RUN_OBS");
  const int64_t run_max = rm ? std::max(64, atoi(rm)) : std::max<int64_t>(256, std::min<int64_t>(2048, nm / 600));   // 3.0 M observations, round-5 kernel (cameras staged per run, queue popped one run ahead): 2048 0.364 ms, 1664 0.365, 1344 0.372, 1024 0.39 (round 4: 1344)
  // current tile / run
  int64_t t_start = 0, t_len = 0;
  int t_tracks = 0, sc_tracks = 0;
  int run_tile0 = 0, run_ntiles = 0, run_key0 = -1;
  int64_t run_obs = 0;
  // cameras / co-visible camera pairs of the open run: membership through per-run stamps (a sorted-set union per track
  // cost 64 ms at 500k tracks), the lists are sorted once when the run closes
  std::vector<int> run_cams;           // unique, sorted by finalize_run
  std::vector<int64_t> run_pairs;      // unique keys hi * 2^32 | lo of co-visible reduced cameras (hi >= lo), sorted by finalize_run
  std::vector<int> tc;
  std::vector<int64_t> tp;
  int serial = 1, prev_serial = 0;
  std::vector<int> prev_tc;
  std::vector<int> cam_stamp((size_t)std::max(1, h->ncp), 0);
  std::vector<uint8_t> cam_local((size_t)std::max(1, h->ncp), 0);   // camera -> index in the closing run's sorted table
  // CONSTANT cameras the open run's tracks see (the fused kernels stage their blocks in LDS behind the local cameras'), in order
  // of appearance: obs_lc = 0x80 | index
  const int max_const = bw == 0 ? kFusedMaxConst : kFusedMaxConstIntr;
  static_assert(kFusedMaxConst <= 0x7f && kFusedMaxConstIntr <= 0x7f, "obs_lc keeps the constant-camera index in seven bits");
  std::vector<int> run_ccams, tcc;
  std::vector<int> ccam_stamp((size_t)std::max(1, h->nc), 0);
  std::vector<uint8_t> ccam_local((size_t)std::max(1, h->nc), 0);
  constexpr int kPairSlots = 2048;     // > 4 x 253
  std::vector<int64_t> pair_key(kPairSlots, 0);
  std::vector<int> pair_stamp(kPairSlots, 0);
  auto pair_slot = [&](int64_t key) -> int {   // slot holding `key` in this run, or the free slot where it would go
    unsigned hsh = ((unsigned)(key >> 32) * 0x9E3779B1u) ^ ((unsigned)key * 0x85EBCA77u);
    int sl = (int)(hsh >> 21) & (kPairSlots - 1);
    while (pair_stamp[sl] == serial && pair_key[sl] != key) sl = (sl + 1) & (kPairSlots - 1);
    return sl;
  };
  const int adj_nt = (h->n + 63) / 64;
  fp.tile_adj.assign((size_t)adj_nt * adj_nt, 0);
  std::vector<int> tl;
  auto mark_tiles = [&](int q) {   // tile co-visibility of a variable track (tc: its variable cameras, ascending)
    if (h->pt_const[porder[q]]) return;
    tl.clear();
    for (int pcam : tc) {
      const int rcam = h->cam_red[h->part_cam[pcam]];   // (participating camera -> reduced camera; constant: no rows in S)
      if (rcam < 0) continue;
      const int s0 = h->ni + 6 * rcam;
      if (tl.empty() || tl.back() != s0 / 64) tl.push_back(s0 / 64);
      if ((s0 + 5) / 64 != s0 / 64) tl.push_back((s0 + 5) / 64);
    }
    for (int a : tl) for (int b : tl) fp.tile_adj[(size_t)a * adj_nt + b] = 1;
  };

  auto close_tile = [&](int q_end) {
    if (!t_len) return;
    tstart.push_back((int)t_start); tcount.push_back((int)t_len); tkey.push_back(run_key0 < 0 ? 0 : run_key0);
    (void)q_end;
    fp.tile_trk_end.push_back(sc_tracks);
    run_ntiles++;
    t_len = 0; t_tracks = 0;
  };
  auto finalize_run = [&]() {
    if (!run_ntiles) { run_cams.clear(); run_pairs.clear(); run_ccams.clear(); run_obs = 0; run_key0 = -1; ++serial; return; }
    std::sort(run_cams.begin(), run_cams.end());
    std::sort(run_pairs.begin(), run_pairs.end());
    FusedRun r;
    r.tile0 = run_tile0; r.ntiles = run_ntiles;
    r.cam_off = (int)fp.cams.size(); r.W = (int)run_cams.size();
    fp.cams.insert(fp.cams.end(), run_cams.begin(), run_cams.end());
    r.tgt_off = (int)fp.tgts.size(); r.ntgt = (int)run_pairs.size();
    for (size_t i = 0; i < run_cams.size(); ++i) cam_local[run_cams[i]] = (uint8_t)i;   // local index by table, not by search
    for (int64_t key : run_pairs) {   // ascending (hi, lo) -> ascending (la, lb)
      const int la = cam_local[(int)(key >> 32)], lb = cam_local[(int)(key & 0xffffffff)];
      fp.tgts.push_back((unsigned short)(la | (lb << 8)));
    }
    const int need = (int)std::max(lanes_tgt * r.ntgt, rows_cam * r.W);
    const int G = need <= 64 ? 1 : (need <= 128 ? 2 : 4);
    r.gp = G | ((G == 1 ? std::max(1, packing((size_t)r.ntgt, (size_t)r.W)) : 1) << 8);
    r.part_off = (int)fp.part_doubles;
    fp.part_doubles += (size_t)r.ntgt * part_tgt + (size_t)r.W * part_cam;
    r.stage_off = (int)fp.stage.size();
    for (int pc : run_cams) fp.stage.push_back(h->part_cam[pc]);
    fp.stage.insert(fp.stage.end(), run_ccams.begin(), run_ccams.end());
    r.nstage = (int)(run_cams.size() + run_ccams.size());
    for (int t = run_tile0; t < run_tile0 + run_ntiles; ++t)
      for (int s = tstart[t]; s < tstart[t] + tcount[t]; ++s) {
        if (sred[s] >= 0) obs_lc[s] = cam_local[sred[s]];
        else obs_lc[s] = (uint8_t)(0x80 | ccam_local[ocam[s]]);
      }
    fp.runs.push_back(r);
    run_tile0 += run_ntiles; run_ntiles = 0; run_obs = 0; run_key0 = -1;
    run_cams.clear(); run_pairs.clear(); run_ccams.clear(); ++serial;
  };
  auto push_long = [&](int q) {
    const int slot = (int)l_pt.size();
    l_pt.push_back(porder[q]);
    for (int64_t c = off[q]; c < off[q + 1]; ++c) { l_obs.push_back((int)c); l_slot.push_back(slot); }
    l_start.push_back((int)l_obs.size());
  };
  run_tile0 = (int)tstart.size();
  for (int q = q_begin; q < q_end; ++q) {
    const int64_t L = off[q + 1] - off[q];
    if (L == 0) continue;
    // the track's variable cameras
    tc.clear(); tcc.clear();
    for (int64_t s = off[q]; s < off[q + 1]; ++s) {
      if (sred[s] >= 0) tc.push_back(sred[s]);
      else tcc.push_back(ocam[s]);
    }
    std::sort(tc.begin(), tc.end());
    const bool dup = std::adjacent_find(tc.begin(), tc.end()) != tc.end();
    if (!tcc.empty()) { std::sort(tcc.begin(), tcc.end()); tcc.erase(std::unique(tcc.begin(), tcc.end()), tcc.end()); }
    mark_tiles(q);
    if (L > 64 || dup || (int)tc.size() > max_cams || (int)tcc.size() > max_const) {
      close_tile(q);            // tiles are contiguous observation ranges
      push_long(q);
      continue;
    }
    // the camera set of the previous track of this run again (tracks are ordered by first camera: common): nothing new
    const bool same_set = serial == prev_serial && tc == prev_tc;
    size_t ucams = run_cams.size(), upairs = run_pairs.size();
    if (!same_set) {
      tp.clear();
      for (size_t a = 0; a < tc.size(); ++a)
        for (size_t b = 0; b <= a; ++b) tp.push_back(((int64_t)tc[a] << 32) | (uint32_t)tc[b]);
      // would the run still fit?  sizes of the unions with the run's sets
      for (int c : tc) ucams += cam_stamp[c] != serial;
      for (int64_t key : tp) upairs += pair_stamp[pair_slot(key)] != serial;
    }
    bool new_run = (int)ucams > max_cams || upairs > max_tgts;
    if (!tcc.empty()) {
      size_t uc = run_ccams.size();
      for (int c : tcc) uc += ccam_stamp[c] != serial;
      if (uc > (size_t)max_const) new_run = true;
    }
    // a run keeps its packing level (track slices per wave) once it has some work, and stays inside one
    // first-camera key once it is large enough
    // (compound blocks: the level is the number of track slices a workgroup walks in parallel -- 4 / G, or 4 PS with one
    // wave per slice -- so that a run of short tracks, two slices, does not absorb the long tracks of the same cameras)
    auto level = [&](size_t ntgt, size_t W) -> int {
      const int pk = packing(ntgt, W);
      if (bw == 0 || pk > 0) return bw == 0 ? pk : 4;   // (one wave per slice: four slices or more, all the same to this rule)
      const size_t need = std::max(lanes_tgt * ntgt, rows_cam * W);
      return need <= 128 ? 2 : 1;
    };
    if (!new_run && run_obs >= 64 && level(upairs, ucams) < level(run_pairs.size(), run_cams.size())) new_run = true;
    if (!new_run && run_obs >= run_max / 4 && skey[q] != run_key0) new_run = true;
    if (new_run) {
      close_tile(q);
      finalize_run();
    }
    if (t_len + L > 64 || t_tracks >= kFusedTileTracks) {
      close_tile(q);
      // runs that need several waves per track slice walk their tracks (almost) serially: keep them short, so that
      // many workgroups share that work instead of a few long ones setting the kernel's duration
      const size_t need = std::max(lanes_tgt * run_pairs.size(), rows_cam * run_cams.size());
      // (compound blocks: nearly every run needs the whole workgroup per track slice, and a run's partial blocks are 44 KB --
      // one run per sub-chunk wrote 420 MB of them per iteration at 1000 views / 500k tracks)
      const int64_t cap = bw ? run_max : (need <= 64 ? run_max : (need <= 128 ? run_max / 4 : 1));
      if (run_ntiles % tps == 0 && run_obs >= cap) finalize_run();
    }
    if (t_len == 0) {
      t_start = off[q];
      if (run_ntiles % tps == 0) sc_tracks = 0;
      if (run_key0 < 0) run_key0 = skey[q];
    }
    for (int64_t s = off[q]; s < off[q + 1]; ++s) obs_tl[s] = (uint8_t)sc_tracks;
    sc_tracks++; t_tracks++; t_len += L; run_obs += L;
    // the track joins the open run
    if (!(same_set && serial == prev_serial)) {   // (a run closed above: the sets are empty again and tp may be stale)
      if (same_set) {
        tp.clear();
        for (size_t a = 0; a < tc.size(); ++a)
          for (size_t b = 0; b <= a; ++b) tp.push_back(((int64_t)tc[a] << 32) | (uint32_t)tc[b]);
      }
      for (int c : tc) if (cam_stamp[c] != serial) { cam_stamp[c] = serial; run_cams.push_back(c); }
      for (int64_t key : tp) {
        const int sl = pair_slot(key);
        if (pair_stamp[sl] != serial) { pair_stamp[sl] = serial; pair_key[sl] = key; run_pairs.push_back(key); }
      }
      prev_tc = tc; prev_serial = serial;
    }
    for (int c : tcc) if (ccam_stamp[c] != serial) { ccam_stamp[c] = serial; ccam_local[c] = (uint8_t)run_ccams.size(); run_ccams.push_back(c); }
  }
  close_tile(q_end);
  finalize_run();
}

// Sum lists of the fused assembly with intrinsics (k_sum_items, ba_fused_intr.hip): per block of S (camera x camera,
// camera x group, group x group) and per vector block (rhs / gradient / column norms of a camera or a group) the pieces of
// the runs' partial blocks that feed it, in run order.  A run's target (la, lb) holds the compound block
// [cam_a | intr_a] x [cam_b | intr_b] as if the two cameras owned their intrinsics; the intrinsics rows / columns of every
// camera of a group land on the group's (bundle_adjuster.cc:463-475: the cameras of a group share ONE parameter block).
// Lists longer than 2 x chunk go through intermediate sums (SK_CHUNK items) and a second-level item.
int build_sum_items_intr(theia_ba_handle_s* h, const int* cam_group, FusedHost& fp, int* n_items1, int* n_items2) {
  constexpr int SK_BLOCK = 0, SK_LOWER = 1, SK_VEC = 2, SK_CHUNK = 3;
  const int KI = h->fused_bw - 6, ni = h->ni, BW = h->fused_bw;   // a run's partial blocks: BW x BW per target (row stride BW), BW x 3 per camera
  struct Ent { int64_t key; int off, code, dims; };   // dims = nr | nc << 4 | kind << 8 | rgrp << 12 | cgrp << 13
  auto src_code = [](int r0, int c0, int tr, int stride) { return r0 | (c0 << 4) | (tr << 8) | (stride << 16); };
  auto dims = [](int nr, int nc, int kind, int rg, int cg) { return nr | (nc << 4) | (kind << 8) | (rg << 12) | (cg << 13); };
  auto key_blk = [](int row0, int col0) { return ((int64_t)row0 << 30) | (int64_t)col0; };
  auto key_vec = [](int row0) { return ((int64_t)1 << 60) | ((int64_t)row0 << 30); };
  // The entries of fixed ranges of runs on host threads, in run order: one pass counts, one writes.  They live in blocks of
  // the pinned host cache -- three fresh 18 MB vectors cost more in page faults than the sort that fills them.
  constexpr int kRunParts = 32;
  const int nruns = (int)fp.runs.size();
  auto emit = [&](int ir, auto&& sink) {
    const FusedRun& r = fp.runs[ir];
    auto cam_of = [&](int l) { return h->part_cam[fp.cams[r.cam_off + l]]; };
    for (int k = 0; k < r.ntgt; ++k) {
      const unsigned us = fp.tgts[r.tgt_off + k];
      const int la = us & 0xff, lb = us >> 8;
      const int ca = cam_of(la), cb = cam_of(lb);
      const int rca = h->cam_red[ca], rcb = h->cam_red[cb], ga = h->grp_red[cam_group[ca]], gb = h->grp_red[cam_group[cb]];
      const int base = r.part_off + BW * BW * k;
      if (rca >= 0 && rcb >= 0)
        sink(Ent{key_blk(ni + 6 * rca, ni + 6 * rcb), base, src_code(0, 0, 0, BW), dims(6, 6, la == lb ? SK_LOWER : SK_BLOCK, 0, 0)});
      if (rca >= 0 && gb >= 0)
        sink(Ent{key_blk(ni + 6 * rca, 10 * gb), base, src_code(0, 6, 0, BW), dims(6, KI, SK_BLOCK, 0, 1)});
      if (la != lb && rcb >= 0 && ga >= 0)
        sink(Ent{key_blk(ni + 6 * rcb, 10 * ga), base, src_code(6, 0, 1, BW), dims(6, KI, SK_BLOCK, 0, 1)});
      if (ga >= 0 && gb >= 0) {
        // (an item is one wave: nr x nc <= 64 elements.  Up to seven compact rows a group x group block is one item; the 10 x 10
        // blocks of the 16-row plan go as two items of five compact rows, item row0 = 10 g + first compact row)
        const int nh = KI > 7 ? 2 : 1, hr = KI > 7 ? KI / 2 : KI;
        for (int hh = 0; hh < nh; ++hh) {
          const int r0 = hh * hr, nr = (hh == nh - 1) ? KI - r0 : hr;
          if (ga > gb) sink(Ent{key_blk(10 * ga + r0, 10 * gb), base, src_code(6 + r0, 6, 0, BW), dims(nr, KI, SK_BLOCK, 1, 1)});
          else if (ga < gb) sink(Ent{key_blk(10 * gb + r0, 10 * ga), base, src_code(6, 6 + r0, 1, BW), dims(nr, KI, SK_BLOCK, 1, 1)});
          else {
            sink(Ent{key_blk(10 * ga + r0, 10 * ga), base, src_code(6 + r0, 6, 0, BW), dims(nr, KI, SK_LOWER, 1, 1)});
            if (la != lb) sink(Ent{key_blk(10 * ga + r0, 10 * ga), base, src_code(6, 6 + r0, 1, BW), dims(nr, KI, SK_LOWER, 1, 1)});
          }
        }
      }
    }
    for (int l = 0; l < r.W; ++l) {
      const int c = cam_of(l), rc = h->cam_red[c], gr = h->grp_red[cam_group[c]];
      const int base = r.part_off + BW * BW * r.ntgt + 3 * BW * l;
      if (rc >= 0) sink(Ent{key_vec(ni + 6 * rc), base, src_code(0, 0, 0, 3), dims(6, 3, SK_VEC, 0, 0)});
      if (gr >= 0) sink(Ent{key_vec(10 * gr), base, src_code(6, 0, 0, 3), dims(KI, 3, SK_VEC, 1, 0)});
    }
  };
  std::vector<size_t> at(kRunParts + 1, 0);
  host_parts(kRunParts, nruns >= 256, [&](int part) {
    size_t cnt = 0;
    for (int ir = (int)((int64_t)nruns * part / kRunParts); ir < (int)((int64_t)nruns * (part + 1) / kRunParts); ++ir) emit(ir, [&](const Ent&) { ++cnt; });
    at[part + 1] = cnt;
  });
  for (int k = 0; k < kRunParts; ++k) at[k + 1] += at[k];
  const size_t nent = at[kRunParts];
  HBuf<Ent> ents_b, tmp_b;
  if (!ents_b.resize(std::max<size_t>(1, nent), true) || !tmp_b.resize(std::max<size_t>(1, nent), true)) return set_error(THEIA_HIP_ERR_OUT_OF_MEMORY, "host staging of %zu sum-list entries failed", nent);
  Ent* const ents = ents_b.data();
  host_parts(kRunParts, nruns >= 256, [&](int part) {
    Ent* out = ents + at[part];
    for (int ir = (int)((int64_t)nruns * part / kRunParts); ir < (int)((int64_t)nruns * (part + 1) / kRunParts); ++ir) emit(ir, [&](const Ent& e) { *out++ = e; });
  });
  {   // ascending key = (vector blocks last, row, column), ties in run order: two stable counting passes over the row / column
      // offsets (a std::stable_sort of the 0.8 M entries of the 1000-view configuration took 30 ms of the create())
    const int nb = h->n + 2;
    counting_pass(ents, tmp_b.data(), (int64_t)nent, nb, [](const Ent& e) { return (int)(e.key & 0x3fffffff); });
    counting_pass(tmp_b.data(), ents, (int64_t)nent, 2 * nb, [nb](const Ent& e) { return (int)((e.key >> 30) & 0x3fffffff) + ((e.key >> 60) ? nb : 0); });
  }
  std::vector<int> items1, items2, src2;
  fp.sum_src.clear();
  fp.sum_src.reserve(2 * nent);
  for (size_t k = 0; k < nent; ++k) { fp.sum_src.push_back(ents[k].off); fp.sum_src.push_back(ents[k].code); }
  size_t chunk_off = fp.part_doubles;
  auto push = [](std::vector<int>& v, int row0, int col0, int code, int beg, int end, int dst) {
    v.push_back(row0); v.push_back(col0); v.push_back(code); v.push_back(beg); v.push_back(end); v.push_back(dst);
  };
  const int nsrc1 = (int)nent;
  for (size_t q = 0; q < nent;) {
    size_t e = q;
    while (e < nent && ents[e].key == ents[q].key) ++e;
    const int row0 = (int)((ents[q].key >> 30) & 0x3fffffff), col0 = (int)(ents[q].key & 0x3fffffff);
    const int code = ents[q].dims, nr = code & 15, nc = (code >> 4) & 15;
    const size_t cnt = e - q;
    size_t ch = 96;   // sources per first-level chunk: a wave adds them with 64 / (nr nc) lane groups, eight loads in flight each
    while (ch * ch < cnt) ++ch;
    if (cnt <= 2 * ch) {
      push(items1, row0, col0, code, (int)q, (int)e, 0);
    } else {
      const int beg2 = nsrc1 + (int)src2.size() / 2;
      for (size_t c0 = q; c0 < e; c0 += ch) {
        push(items1, 0, 0, (code & 0xff) | (SK_CHUNK << 8), (int)c0, (int)std::min(e, c0 + ch), (int)chunk_off);
        src2.push_back((int)chunk_off); src2.push_back(nc << 16);
        chunk_off += (size_t)nr * nc;
      }
      push(items2, row0, col0, code, beg2, nsrc1 + (int)src2.size() / 2, 0);
    }
    q = e;
  }
  fp.sum_src.insert(fp.sum_src.end(), src2.begin(), src2.end());
  if (getenv("THEIA_HIP_CREATE_TIMING")) {
    size_t longest = 0, nchunk = 0;
    for (size_t k = 0; k < items1.size(); k += 6) {
      longest = std::max<size_t>(longest, (size_t)(items1[k + 4] - items1[k + 3]));
      nchunk += ((items1[k + 2] >> 8) & 15) == SK_CHUNK;
    }
    fprintf(stderr, "theia_hip sum lists: %zu first-level items (%zu chunks), %zu second-level, %d + %zu sources, longest list %zu, "
            "%zu partial doubles + %zu chunk doubles\n", items1.size() / 6, nchunk, items2.size() / 6, nsrc1, src2.size() / 2, longest,
            fp.part_doubles, chunk_off - fp.part_doubles);
  }
  fp.part_doubles = chunk_off;
  *n_items1 = (int)items1.size() / 6; *n_items2 = (int)items2.size() / 6;
  fp.sum_items = items1;
  fp.sum_items.insert(fp.sum_items.end(), items2.begin(), items2.end());
  return 0;
}

// The segments in order -> one plan (offsets rebased), then per S block the partial sums that feed it, in run order.
void merge_fused_segments(const theia_ba_handle_s* h, std::vector<FusedSegment>& segs, std::vector<int>& tstart, std::vector<int>& tcount,
                          std::vector<int>& tkey, std::vector<int>& l_obs, std::vector<int>& l_slot, std::vector<int>& l_start,
                          std::vector<int>& l_pt, FusedHost& fp) {
  const int adj_nt = (h->n + 63) / 64;
  fp.tile_adj.assign((size_t)adj_nt * adj_nt, 0);
  for (FusedSegment& sg : segs) {
    const int tile0 = (int)tstart.size(), cam0 = (int)fp.cams.size(), tgt0 = (int)fp.tgts.size(), stage0 = (int)fp.stage.size();
    const int slot0 = (int)l_pt.size(), lobs0 = (int)l_obs.size();
    tstart.insert(tstart.end(), sg.tstart.begin(), sg.tstart.end());
    tcount.insert(tcount.end(), sg.tcount.begin(), sg.tcount.end());
    tkey.insert(tkey.end(), sg.tkey.begin(), sg.tkey.end());
    for (FusedRun r : sg.fp.runs) {
      r.tile0 += tile0; r.cam_off += cam0; r.tgt_off += tgt0; r.part_off += (int)fp.part_doubles; r.stage_off += stage0;
      fp.runs.push_back(r);
    }
    fp.part_doubles += sg.fp.part_doubles;
    fp.cams.insert(fp.cams.end(), sg.fp.cams.begin(), sg.fp.cams.end());
    fp.stage.insert(fp.stage.end(), sg.fp.stage.begin(), sg.fp.stage.end());
    fp.tgts.insert(fp.tgts.end(), sg.fp.tgts.begin(), sg.fp.tgts.end());
    fp.tile_trk_end.insert(fp.tile_trk_end.end(), sg.fp.tile_trk_end.begin(), sg.fp.tile_trk_end.end());
    l_obs.insert(l_obs.end(), sg.l_obs.begin(), sg.l_obs.end());
    for (int v : sg.l_slot) l_slot.push_back(v + slot0);
    for (int v : sg.l_start) l_start.push_back(v + lobs0);
    l_pt.insert(l_pt.end(), sg.l_pt.begin(), sg.l_pt.end());
    for (size_t i = 0; i < fp.tile_adj.size(); ++i) fp.tile_adj[i] |= sg.fp.tile_adj[i];
  }
  const bool ptiming = getenv("THEIA_HIP_CREATE_TIMING") != nullptr;
  auto pt0 = std::chrono::steady_clock::now();
  auto ptick = [&](const char* what) {
    if (!ptiming) return;
    const auto t = std::chrono::steady_clock::now();
    fprintf(stderr, "theia_hip create:     fused plan: %-18s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t - pt0).count());
    pt0 = t;
  };
  ptick("(since runs built)");
  if (h->fused_bw) return;   // compound blocks: build_sum_items_intr (create())
  // per S block: the partial sums that feed it, in run order
  struct Ent { int64_t key; int src; int isd; };
  std::vector<Ent> ents;
  for (const FusedRun& r : fp.runs) {
    for (int k = 0; k < r.ntgt; ++k) {
      const unsigned us = fp.tgts[r.tgt_off + k];
      const int ri = fp.cams[r.cam_off + (us & 0xff)], rj = fp.cams[r.cam_off + (us >> 8)];
      ents.push_back({((int64_t)ri << 32) | (uint32_t)rj, r.part_off + 36 * k, 0});
    }
    for (int lc = 0; lc < r.W; ++lc) {
      const int ri = fp.cams[r.cam_off + lc];
      ents.push_back({((int64_t)ri << 32) | (uint32_t)ri, r.part_off + 36 * r.ntgt + 18 * lc, 1});
    }
  }
  {   // order (row camera, column camera, block pieces before diagonal pieces, run order): three stable counting passes, least
      // significant key first (std::stable_sort on the 230 k entries of the 1000-view configuration took 2.3 ms of the create())
    const int nb = std::max(2, h->ncp) + 1;
    std::vector<Ent> tmp(ents.size());
    const int64_t ne = (int64_t)ents.size();
    counting_pass(ents.data(), tmp.data(), ne, nb, [](const Ent& en) { return en.isd; });
    counting_pass(tmp.data(), ents.data(), ne, nb, [](const Ent& en) { return (int)(en.key & 0xffffffff); });
    counting_pass(ents.data(), tmp.data(), ne, nb, [](const Ent& en) { return (int)(en.key >> 32); });
    ents.swap(tmp);
  }
  for (size_t q = 0; q < ents.size();) {
    size_t e = q;
    while (e < ents.size() && ents[e].key == ents[q].key) ++e;
    size_t m = q;
    while (m < e && !ents[m].isd) ++m;
    fp.sum_items.push_back((int)(ents[q].key >> 32)); fp.sum_items.push_back((int)(ents[q].key & 0xffffffff));
    fp.sum_items.push_back((int)q); fp.sum_items.push_back((int)m);
    fp.sum_items.push_back((int)m); fp.sum_items.push_back((int)e);
    q = e;
  }
  fp.sum_src.reserve(ents.size());
  for (const Ent& en : ents) fp.sum_src.push_back(en.src);
  ptick("sum items");
}

// An error left behind by an earlier, deliberately ignored HIP call of this host thread (a refused cooperative launch, an
// event of a destroyed stream, another library) must not be mistaken for a failure of THIS call: run() asks hipGetLastError()
// after it has enqueued its kernels.  The entry points therefore start from a clean slate; THEIA_HIP_DEBUG_STICKY=1 reports
// what was discarded (one box in six showed a stale "operation not permitted when stream is capturing" here, origin unknown).
static void debug_sticky(const char* where) {
  static const bool report = getenv("THEIA_HIP_DEBUG_STICKY") != nullptr;
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess && report) std::fprintf(stderr, "theia_hip: stale HIP error discarded at %s: %s\n", where, hipGetErrorString(e));
}
#undef UP
#undef AL
// (private return code of ba_create_impl: the fused intrinsics plan does not fit this problem, build it again on the gather lists)
constexpr int kRetryWithoutFusedIntr = 0x7a11;
static int ba_create_impl(const theia_ba_problem* p, const theia_ba_options* o, theia_ba_handle* out, bool allow_fused_intr);
int theia_hip_ba_create(const theia_ba_problem* p, const theia_ba_options* o, theia_ba_handle* out) {
  debug_sticky("create entry");
  int rc = ba_create_impl(p, o, out, true);
  if (rc == kRetryWithoutFusedIntr) rc = ba_create_impl(p, o, out, false);
  debug_sticky("create exit");
  return rc;
}
static int ba_create_impl(const theia_ba_problem* p, const theia_ba_options* o, theia_ba_handle* out, bool allow_fused_intr) {
  if (!out) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "null handle pointer");
  *out = nullptr;
  int rc = validate(p, o);
  if (rc) return rc;
  if (p->flags & THEIA_BA_FLAG_INVERSE_DEPTH) {
    std::unique_ptr<theia_ba_handle_s> hi(new theia_ba_handle_s());
    hi->opt = *o;
    hi->nc = p->num_cameras; hi->ng = p->num_groups; hi->np = p->num_points; hi->nobs = p->num_obs;
    if ((rc = thip::id_handle_create(p, o, &hi->idh))) return rc;
    *out = hi.release();
    return 0;
  }
  rc = thip::ensure_device();
  if (rc) return rc;
  theia_ba_handle_s* h = new theia_ba_handle_s();
  std::unique_ptr<theia_ba_handle_s> guard(h);
  h->opt = *o;
  h->nc = p->num_cameras; h->ng = p->num_groups; h->np = p->num_points; h->nobs = p->num_obs;
  h->pd = o->use_homogeneous_point_parametrization ? 3 : 4;
  HIP_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  h->stage.stream = h->stream;
  StageScope stage_scope(&h->stage);
  PoolStreamScope pool_scope(h->stream);   // blocks that go back to the caches inside this call are tagged with an event on it
  for (auto& row : h->ev) for (auto& e : row) HIP_TRY(hipEventCreate(&e));
  HIP_TRY(hipHostMalloc((void**)&h->h_scal, sizeof(double) * 40, hipHostMallocDefault));
  HIP_TRY(hipHostMalloc((void**)&h->h_state, 1024, hipHostMallocDefault));
  static_assert(sizeof(LmState) <= 1024, "pinned read-back block too small");

  // THEIA_HIP_CREATE_TIMING=1: wall time of the create() stages on stderr
  const bool ctiming = getenv("THEIA_HIP_CREATE_TIMING") != nullptr;
  auto ct0 = std::chrono::steady_clock::now();
  auto tick = [&](const char* what) {
    if (!ctiming) return;
    const auto t = std::chrono::steady_clock::now();
    fprintf(stderr, "theia_hip create: %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t - ct0).count());
    ct0 = t;
  };
  tick("stream + events");
  // --- problem structure (bundle_adjuster.cc:116-221,357-380,477-527) ---
  std::vector<uint8_t> cam_used(h->nc, 0);
  RawArray<uint8_t> pt_used((size_t)h->np);
  std::vector<uint8_t> grp_used(h->ng, 0);
  std::vector<int> toff((size_t)h->np + 1, 0);   // the track CSR (below)
  std::atomic<int> unsorted{0};
  {   // cameras with observations: flags per host thread, merged (the tracks' flags come with the key pass below)
    std::mutex mu;
    host_chunks(h->nobs, [&](int64_t i0, int64_t i1) {
      std::vector<uint8_t> mine(h->nc, 0);
      bool disorder = false;   // (the same pass: does the input come track by track?  see the track CSR below)
      for (int64_t i = i0; i < i1; ++i) {
        mine[p->obs_cam[i]] = 1;
        const int q1 = p->obs_pt[i], q0 = i ? p->obs_pt[i - 1] : -1;
        disorder |= q1 < q0;
        for (int q = q0 + 1; q <= q1; ++q) toff[q] = (int)i;   // toff[q] = first observation of a track >= q (meaningful if no disorder)
      }
      if (disorder) unsorted.store(1, std::memory_order_relaxed);
      std::lock_guard<std::mutex> lk(mu);
      for (int c = 0; c < h->nc; ++c) cam_used[c] |= mine[c];
    });
    for (int c = 0; c < h->nc; ++c) if (cam_used[c]) grp_used[p->cam_group[c]] = 1;
  }
  h->cam_red.assign(h->nc, -1); h->cam_mask.assign(h->nc, 0x3f); h->pt_const.assign(h->np, 1);
  h->ncv = 0;
  for (int c = 0; c < h->nc; ++c) {
    unsigned m = p->cam_const ? p->cam_const[c] : 0;
    if (o->constant_camera_orientation) m |= THEIA_CAM_CONST_ORIENTATION;
    if (o->constant_camera_position) m |= THEIA_CAM_CONST_POSITION;
    if (o->orthographic_camera) m |= THEIA_CAM_CONST_TZ;
    unsigned cols = 0;
    if (m & THEIA_CAM_CONST_POSITION) cols |= 0x07;
    if (m & THEIA_CAM_CONST_ORIENTATION) cols |= 0x38;
    if (m & THEIA_CAM_CONST_TZ) cols |= 0x04;
    if (cols != 0x3f && (cam_used[c] || (p->flags & THEIA_BA_FLAG_KEEP_UNOBSERVED_CAMERAS))) { h->cam_red[c] = h->ncv++; h->cam_mask[c] = (uint8_t)cols; }
  }
  // intrinsics blocks (bundle_adjuster.cc:382-460): constant when nothing is optimised
  // or the caller marked the group constant, otherwise a subset manifold
  h->grp_red.assign(h->ng, -1); h->grp_free.assign(h->ng, 0u); h->grp_k.assign(h->ng, 0);
  h->ngv = 0;
  for (int g = 0; g < h->ng; ++g) {
    h->grp_k[g] = intrinsics_size(p->group_model[g]);
    const unsigned fm = intrinsics_free_mask(p->group_model[g], o->intrinsics_to_optimize);
    const bool gconst = (p->group_const && p->group_const[g]) || fm == 0 ||
                        (!grp_used[g] && !(p->flags & THEIA_BA_FLAG_KEEP_UNOBSERVED_CAMERAS));
    if (!gconst) { h->grp_red[g] = h->ngv++; h->grp_free[g] = fm; }
  }
  h->ni = THEIA_MAX_INTRINSICS * h->ngv;
  h->n = h->ni + 6 * h->ncv;
  {   // fused assembly with intrinsics (ba_fused_intr.hip): compact rows, at most four free parameters per group
    int most = 0;
    for (int g = 0; g < h->ng; ++g) if (h->grp_red[g] >= 0) most = std::max(most, __builtin_popcount(h->grp_free[g]));
    const char* force = getenv("THEIA_HIP_INTR_ROWS");
    // (block width 6 + rows: 9, 10, 13 for five to seven free parameters -- every intrinsic of the pinhole model --, 16 for up to
    // the ten of the radial-tangential model)
    h->fused_bw = (allow_fused_intr && h->ni > 0 && !(force && atoi(force) == 10) && !getenv("THEIA_HIP_INTR_GATHER")) ? (most <= 3 ? 9 : (most <= 4 ? 10 : (most <= 7 ? 13 : 16))) : 0;
    h->fused_kmask = 0;
    {
      bool first = true, same = true;
      for (int g = 0; g < h->ng; ++g) if (h->grp_red[g] >= 0) { if (first) { h->fused_kmask = h->grp_free[g]; first = false; } else if (h->grp_free[g] != h->fused_kmask) same = false; }
      if (!same) h->fused_kmask = 0;
    }
    h->cam_part.assign(h->nc, -1);
    h->ncp = 0;
    for (int c = 0; c < h->nc; ++c)
      if (h->cam_red[c] >= 0 || (h->fused_bw && cam_used[c] && h->grp_red[p->cam_group[c]] >= 0)) { h->cam_part[c] = h->ncp++; h->part_cam.push_back(c); }
  }

  // Tracks are visited in the order of their first (lowest) variable camera of
  // the reduced ordering, so that a workgroup's tile range touches a short
  // window of cameras (LDS accumulation in k_linearize).  Observations are
  // sorted by that track order; residual blocks whose blocks are all constant
  // are evaluated once ("fixed cost", ceres reduced program).
  RawArray<uint8_t> fixed((size_t)h->nobs);   // (these four are written by the per-track pass below)
  RawArray<int> pkey((size_t)h->np);
  RawArray<int> nvar((size_t)h->np);   // variable cameras of a track
  // The input's observations grouped by track (CSR): toff[q] .. toff[q + 1] are track q's entries of tobs, in input order.
  // Input that already comes track by track (obs_pt non-decreasing: what a flattened reconstruction looks like) needs no
  // index array; anything else is counted, scattered with atomic cursors and put back into input order per track.
  HBuf<int> tobs_b;
  const int* tobs = nullptr;   // nullptr: the identity
  {
    if (!unsorted.load()) {   // (the offsets were written by the first pass over the observations, above)
      for (int q = h->nobs ? p->obs_pt[h->nobs - 1] + 1 : 0; q <= h->np; ++q) toff[q] = (int)h->nobs;
    } else {
      std::fill(toff.begin(), toff.end(), 0);
      if (!tobs_b.resize((size_t)std::max<int64_t>(1, h->nobs), true)) return set_error(THEIA_HIP_ERR_OUT_OF_MEMORY, "host staging of %lld observations failed", (long long)h->nobs);
      int* const tb = tobs_b.data();
      host_chunks(h->nobs, [&](int64_t i0, int64_t i1) { for (int64_t i = i0; i < i1; ++i) __atomic_fetch_add(&toff[(size_t)p->obs_pt[i] + 1], 1, __ATOMIC_RELAXED); });
      for (int q = 0; q < h->np; ++q) toff[q + 1] += toff[q];
      std::vector<int> cur(toff.begin(), toff.end() - 1);
      host_chunks(h->nobs, [&](int64_t i0, int64_t i1) { for (int64_t i = i0; i < i1; ++i) tb[__atomic_fetch_add(&cur[p->obs_pt[i]], 1, __ATOMIC_RELAXED)] = (int)i; });
      host_chunks(h->np, [&](int64_t q0, int64_t q1) { for (int64_t q = q0; q < q1; ++q) std::sort(tb + toff[q], tb + toff[q + 1]); });
      tobs = tb;
    }
  }
  RawArray<int> nfix((size_t)h->np);   // residual blocks of a track whose blocks are all constant
  host_chunks(h->np, [&](int64_t q0, int64_t q1) {   // per-track sums: tracks are independent
    for (int64_t q = q0; q < q1; ++q) {
      pt_used[q] = toff[q + 1] > toff[q];
      const bool qconst = p->point_const && p->point_const[q];
      int nv = 0, pk = std::numeric_limits<int>::max(), nf = 0;
      for (int k = toff[q]; k < toff[q + 1]; ++k) {
        const int i = tobs ? tobs[k] : k;
        const int c = p->obs_cam[i];
        const int rc = h->cam_part[c];
        if (rc >= 0) { nv++; if (rc < pk) pk = rc; }
        // a residual block whose blocks are all constant (an observed track is constant iff the caller marked it)
        fixed[i] = (qconst && h->cam_red[c] < 0 && h->grp_red[p->cam_group[c]] < 0) ? 1 : 0;
        nf += fixed[i];
      }
      nvar[q] = nv; pkey[q] = pk; nfix[q] = nf;
      h->pt_const[q] = ((p->point_const && p->point_const[q]) || !pt_used[q]) ? 1 : 0;
    }
  });
  std::vector<int> porder(h->np);
  // Inside one first-camera key, short tracks come first (classes by number of variable cameras): the fused Schur
  // kernel packs several short tracks into one wave step when a run of tracks touches few target blocks.
  RawArray<int> skey_pt((size_t)h->np);
  std::atomic<int> maxkey_all{-1};
  host_chunks(h->np, [&](int64_t q0, int64_t q1) {
    int mk = -1;
    for (int64_t q = q0; q < q1; ++q) {
      porder[q] = (int)q;
      // measured at 1k views / 500k tracks (K1 + K2 launch group): {<= 7 | >= 8} 0.575 ms, {<= 6 | >= 7} 0.599, {<= 5 | >= 6} 0.670,
      // {<= 8 | >= 9} 0.646, {<= 3 | 4..6 | >= 7} 0.636, one class 0.593
      static const int ncls = getenv("THEIA_HIP_FUSED_CLASSES") ? atoi(getenv("THEIA_HIP_FUSED_CLASSES")) : 2;
      static const int cut0 = getenv("THEIA_HIP_FUSED_CUT0") ? atoi(getenv("THEIA_HIP_FUSED_CUT0")) : 7;
      int cls = ncls == 2 ? (nvar[q] <= cut0 ? 0 : 2) : (nvar[q] <= 3 ? 0 : (nvar[q] <= 6 ? 1 : 2));
      {   // development: THEIA_HIP_FUSED_CUTS="a,b,c" -> classes {<= a | <= b | <= c | more}
        static const std::vector<int> cuts = [] { std::vector<int> v; const char* e = getenv("THEIA_HIP_FUSED_CUTS"); if (e) { for (const char* c = e; *c;) { v.push_back(atoi(c)); while (*c && *c != ',') ++c; if (*c) ++c; } } return v; }();
        if (!cuts.empty()) { cls = 0; for (int cu : cuts) if (nvar[q] > cu) ++cls; cls = std::min(cls, 3); }
      }
      // compound blocks (three or four lanes per target): four classes, by the number of track slices a workgroup can walk
      // in parallel -- <= 3 cameras and 4 .. 6: four slices; 7: two; more: one
      if (h->fused_bw && !getenv("THEIA_HIP_FUSED_CLASSES")) cls = nvar[q] <= 3 ? 0 : (nvar[q] <= 6 ? 1 : (nvar[q] <= 7 ? 2 : 3));
      static const bool noclass = getenv("THEIA_HIP_FUSED_NOCLASS") != nullptr;
      skey_pt[q] = pkey[q] == std::numeric_limits<int>::max() ? pkey[q] : (((h->ni == 0 || h->fused_bw) && !noclass) ? pkey[q] * 4 + cls : pkey[q]);
      if (skey_pt[q] != std::numeric_limits<int>::max()) mk = std::max(mk, skey_pt[q]);
    }
    int cur = maxkey_all.load();
    while (mk > cur && !maxkey_all.compare_exchange_weak(cur, mk)) {}
  });
  tick("  structure: masks, keys");
  // Stable counting sort of the tracks by key (keys are < 4 * (#variable cameras) + 4, or INT_MAX = no variable camera: last
  // bucket) on host threads -- per-part histograms, offsets in (bucket, part) order -- and, in the same scatter, what the track
  // of every RANK brings: its first-camera key and its fixed / non-fixed residual blocks.
  std::vector<int64_t> cnt_main(h->np + 1, 0), cnt_fix(h->np + 1, 0);   // (offsets indexed by track rank, after the running sums)
  std::vector<int> skey(h->np);                                         // run boundaries of the fused plan follow the first-camera key
  {
    const int maxkey = maxkey_all.load();
    struct Ranked { int q, key, nfix, len; };   // one 16-byte record per rank: the scatter is ONE random write per track
    RawArray<Ranked> rk((size_t)h->np);
    auto place = [&](int64_t r, int q) { rk[(size_t)r] = Ranked{q, pkey[q], nfix[q], toff[q + 1] - toff[q]}; };
    auto unpack = [&]() {   // ... and the arrays the later passes read come out of a sequential pass
      host_chunks(h->np, [&](int64_t r0, int64_t r1) {
        for (int64_t r = r0; r < r1; ++r) {
          const Ranked& e = rk[(size_t)r];
          porder[r] = e.q; skey[r] = e.key; cnt_fix[r + 1] = e.nfix; cnt_main[r + 1] = e.len - e.nfix;
        }
      });
    };
    if (maxkey >= 0 && (int64_t)maxkey < 8 * (int64_t)h->np + 1024) {
      const int nb = maxkey + 2;
      int parts = host_part_count(h->np, 32768);
      if ((int64_t)parts * nb > ((int64_t)1 << 24)) parts = 1;
      const int64_t per = ((int64_t)h->np + parts - 1) / parts;
      std::vector<int> head((size_t)parts * nb, 0);
      auto bucket = [&](int q) { return skey_pt[q] == std::numeric_limits<int>::max() ? nb - 1 : skey_pt[q]; };
      host_parts(parts, true, [&](int t) {
        int* hh = head.data() + (size_t)t * nb;
        for (int64_t q = t * per; q < std::min<int64_t>(h->np, (t + 1) * per); ++q) hh[bucket((int)q)]++;
      });
      int run = 0;
      for (int bk = 0; bk < nb; ++bk)
        for (int t = 0; t < parts; ++t) { int& c = head[(size_t)t * nb + bk]; const int n = c; c = run; run += n; }
      host_parts(parts, true, [&](int t) {
        int* hh = head.data() + (size_t)t * nb;
        for (int64_t q = t * per; q < std::min<int64_t>(h->np, (t + 1) * per); ++q) place(hh[bucket((int)q)]++, (int)q);
      });
    } else {
      std::stable_sort(porder.begin(), porder.end(), [&](int x, int y) { return skey_pt[x] < skey_pt[y]; });
      for (int r = 0; r < h->np; ++r) place(r, porder[r]);
    }
    unpack();
  }
  tick("  structure: sort tracks");
  for (int r = 0; r < h->np; ++r) { cnt_fix[r + 1] += cnt_fix[r]; cnt_main[r + 1] += cnt_main[r]; }   // ... as running sums
  tick("    permutation: counts");
  h->nobs_main = cnt_main[h->np];
  if (!h->perm.resize((size_t)std::max<int64_t>(1, h->nobs), true)) return set_error(THEIA_HIP_ERR_OUT_OF_MEMORY, "host staging of %lld observations failed", (long long)h->nobs);
  // The sorted observation arrays are staged in pinned blocks of the library's host cache: 24 bytes per observation of
  // fresh pageable vectors cost more in page faults than the gather itself, and the copies below run as plain DMA.  They
  // are filled in the pass that lays out the permutation -- a track's observations are read where they lie in the input
  // (one run of it when the input comes track by track) instead of through 3 M random reads of a separate gather pass.
  HBuf<double2> uv, si;
  HBuf<int> ocam_b, opt_b, sred_b;
  if (!uv.resize((size_t)std::max<int64_t>(1, h->nobs), true) || !ocam_b.resize((size_t)std::max<int64_t>(1, h->nobs), true) ||
      !opt_b.resize((size_t)std::max<int64_t>(1, h->nobs), true) || !sred_b.resize((size_t)std::max<int64_t>(1, h->nobs_main), true) ||
      (p->obs_sqrt_info && !si.resize((size_t)std::max<int64_t>(1, h->nobs), true)))
    return set_error(THEIA_HIP_ERR_OUT_OF_MEMORY, "pinned staging of %lld observations failed", (long long)h->nobs);
  if (!p->obs_sqrt_info) si.n = 0;
  int* const ocam = ocam_b.data();
  int* const opt = opt_b.data();
  int* const sred = sred_b.data();   // participating camera of a non-fixed observation (-1: constant camera and group)
  host_chunks(h->np, [&](int64_t r0, int64_t r1) {   // observations of a track keep their input order
    constexpr int kAhead = 12;   // the tracks of consecutive ranks lie anywhere in the input: their lines are requested ahead
    for (int64_t r = r0; r < r1; ++r) {
      if (r + 2 * kAhead < r1) __builtin_prefetch(&toff[porder[r + 2 * kAhead]]);
      if (r + kAhead < r1) {
        const int qn = porder[r + kAhead];
        if (tobs) __builtin_prefetch(&tobs[toff[qn]]);
        else {
          const size_t i0 = (size_t)toff[qn];
          __builtin_prefetch(p->obs_uv + 2 * i0); __builtin_prefetch(p->obs_uv + 2 * i0 + 8);
          __builtin_prefetch(p->obs_cam + i0); __builtin_prefetch(&fixed[i0]);
          if (p->obs_sqrt_info) { __builtin_prefetch(p->obs_sqrt_info + 2 * i0); __builtin_prefetch(p->obs_sqrt_info + 2 * i0 + 8); }
        }
      }
      const int q = porder[r];
      int64_t m = cnt_main[r], f = h->nobs_main + cnt_fix[r];
      for (int k = toff[q]; k < toff[q + 1]; ++k) {
        const int i = tobs ? tobs[k] : k;
        const int64_t s2 = fixed[i] ? f++ : m++;
        h->perm[s2] = i;
        uv[s2] = make_double2(p->obs_uv[2 * (size_t)i], p->obs_uv[2 * (size_t)i + 1]);
        if (p->obs_sqrt_info) si[s2] = make_double2(p->obs_sqrt_info[2 * (size_t)i], p->obs_sqrt_info[2 * (size_t)i + 1]);
        ocam[s2] = p->obs_cam[i]; opt[s2] = q;
        if (!fixed[i]) sred[s2] = h->cam_part[p->obs_cam[i]];
      }
    }
  });
  tick("    permutation: fill");
  // their uploads start now and run under the rest of the plan construction (pinned sources: nothing waits here; a block that
  // goes back to the host cache on an early return is tagged with an event on this stream, see pool_scope above)
#define UPP(buf, src, cnt, pin) do { rc = h->buf.upload(src, cnt, h->stream, pin); if (rc) return rc; } while (0)
  UPP(obs_uv, uv.data(), (size_t)h->nobs, uv.pinned()); UPP(obs_si, si.data(), si.n, si.pinned());
  UPP(obs_cam, ocam, (size_t)h->nobs, ocam_b.pinned()); UPP(obs_pt, opt, (size_t)h->nobs, opt_b.pinned());
#undef UPP
  // wave tiles: <= 64 observations, never splitting a track
  std::vector<int> tstart, tcount, tkey;
  std::vector<int> l_obs, l_slot, l_start(1, 0), l_pt;  // long tracks (> 64 observations): slow path
  auto build_tiles = [&](const std::vector<int64_t>& off, int64_t base, bool allow_long) -> int {
    int64_t cur0 = 0, curlen = 0;
    int curkey = 0;
    for (int q = 0; q < h->np; ++q) {
      const int64_t L = off[q + 1] - off[q];
      if (L == 0) continue;
      if (L > 64) {
        if (!allow_long) {  // fixed (all-constant) blocks: any split is fine, no per-track sums needed
          if (curlen) { tstart.push_back((int)(base + cur0)); tcount.push_back((int)curlen); tkey.push_back(curkey); curlen = 0; }
          for (int64_t c = 0; c < L; c += 64) { tstart.push_back((int)(base + off[q] + c)); tcount.push_back((int)std::min<int64_t>(64, L - c)); tkey.push_back(0); }
          continue;
        }
        // tiles are contiguous observation ranges: close the open tile before skipping this track
        if (curlen) { tstart.push_back((int)(base + cur0)); tcount.push_back((int)curlen); tkey.push_back(curkey); curlen = 0; }
        const int slot = (int)l_pt.size();
        l_pt.push_back(porder[q]);
        for (int64_t c = 0; c < L; ++c) { l_obs.push_back((int)(base + off[q] + c)); l_slot.push_back(slot); }
        l_start.push_back((int)l_obs.size());
        continue;
      }
      if (curlen + L > 64) { tstart.push_back((int)(base + cur0)); tcount.push_back((int)curlen); tkey.push_back(curkey); curlen = 0; }
      if (curlen == 0) { cur0 = off[q]; curkey = pkey[porder[q]]; }
      curlen += L;
    }
    if (curlen) { tstart.push_back((int)(base + cur0)); tcount.push_back((int)curlen); tkey.push_back(curkey); }
    return 0;
  };
  // Schur assembly without intrinsics: the fused kernel (ba_fused.hip) unless most of the problem would not fit it
  // (tracks that see a camera twice -- e.g. depth-prior rows -- or more than kFusedMaxCams cameras take the
  // per-observation slow path there); THEIA_HIP_SCHUR_GATHER=1 selects the first-generation gather kernels.
  FusedHost fplan;
  tick("  structure: permutation");
  h->use_fused = (h->ni == 0 || h->fused_bw) && h->nobs_main > 0 && !getenv("THEIA_HIP_SCHUR_GATHER");
  const int fused_max_cams = fused_run_cameras(h->fused_bw);
  if (h->use_fused) {
    std::atomic<long long> misfit{0};
    host_chunks(h->np, [&](int64_t q0, int64_t q1) {   // tracks are independent
      std::vector<int> tc;
      long long mine = 0;
      for (int64_t q = q0; q < q1; ++q) {
        const int64_t L = cnt_main[q + 1] - cnt_main[q];
        if (L < 2 || L > 64) continue;
        tc.clear();
        for (int64_t s = cnt_main[q]; s < cnt_main[q + 1]; ++s) if (sred[s] >= 0) tc.push_back(sred[s]);
        std::sort(tc.begin(), tc.end());
        if ((int)tc.size() > fused_max_cams || std::adjacent_find(tc.begin(), tc.end()) != tc.end()) mine += L;
      }
      misfit += mine;
    });
    if (misfit.load() * 20 > h->nobs_main) h->use_fused = false;
    if (!h->use_fused && h->fused_bw) return kRetryWithoutFusedIntr;   // the keys above speak of participating cameras: start over
  }
  tick("  structure: fit check");
  if (h->use_fused) {
    // 32 segments of tracks, cut where the first-camera key changes, built on host threads and merged in order (the
    // segment count is fixed: the plan -- and with it the summation order of S -- does not depend on the machine)
    static const int kSegs = getenv("THEIA_HIP_PLAN_SEGS") ? std::max(1, atoi(getenv("THEIA_HIP_PLAN_SEGS"))) : 32;   // (the switch: measurement only)
    std::vector<int> cut{0};
    if (h->np >= (getenv("THEIA_HIP_HOST_CHUNK_MIN") ? 64 : 65536))
      for (int k = 1; k < kSegs; ++k) {
        int q = (int)((int64_t)h->np * k / kSegs);
        while (q < h->np && q > 0 && skey[q] == skey[q - 1]) ++q;
        if (q > cut.back() && q < h->np) cut.push_back(q);
      }
    cut.push_back(h->np);
    std::vector<FusedSegment> segs(cut.size() - 1);
    tick("    fused plan: keys + cuts");
    fplan.obs_lc.assign((size_t)std::max<int64_t>(1, h->nobs_main), 0xff);
    fplan.obs_tl.assign((size_t)std::max<int64_t>(1, h->nobs_main), 0);
    host_parts((int)segs.size(), true, [&](int k) {
      build_fused_segment(h, cnt_main, porder, sred, ocam, skey, cut[k], cut[k + 1], fplan.obs_lc.data(), fplan.obs_tl.data(), segs[k]);
    });
    tick("    fused plan: segments built");
    merge_fused_segments(h, segs, tstart, tcount, tkey, l_obs, l_slot, l_start, l_pt, fplan);
    tick("    fused plan: merged");
    if (h->fused_bw) {
      int n1 = 0, n2 = 0;
      if ((rc = build_sum_items_intr(h, p->cam_group, fplan, &n1, &n2))) return rc;
      h->n_sum_items2 = n2;
      tick("  structure: sum lists (intrinsics)");
    }
    if (fplan.part_doubles > (size_t)std::numeric_limits<int>::max() / 2)
      return set_error(THEIA_HIP_ERR_UNSUPPORTED, "partial-sum buffer of the fused Schur assembly exceeds 32-bit offsets");
  } else {
    build_tiles(cnt_main, 0, true);
  }
  tick("  structure: fused plan / tiles");
  h->ntiles_main = (int)tstart.size();
  tick("    (tiles)");
  // evaluation-only tiles over the long tracks' observations (no per-track sums there)
  h->long_nobs = (int)l_obs.size(); h->long_ntracks = (int)l_pt.size();
  for (int s2 = 0; s2 < h->long_ntracks; ++s2)
    for (int c = l_start[s2]; c < l_start[s2 + 1]; c += 64) {
      tstart.push_back(l_obs[c]); tcount.push_back(std::min(64, l_start[s2 + 1] - c)); tkey.push_back(0);
    }
  h->ntiles_eval = (int)tstart.size();
  build_tiles(cnt_fix, h->nobs_main, false);
  h->ntiles_all = (int)tstart.size();
  hipStream_t st = h->stream;
#define UP(buf, vec) do { rc = h->buf.upload(vec, st); if (rc) return rc; } while (0)
#define AL(buf, cnt) do { rc = h->buf.alloc(cnt); if (rc) return rc; } while (0)
  tick("structure, sort, tiles");
  h->inner = h->opt.use_inner_iterations != 0 && h->nobs_main > 0;
  if (h->inner) {
    // residual blocks that depend on a block: the camera's / the group's / the track's observations among the
    // non-fixed ones [0, nobs_main) of the sorted arrays (depth-prior rows do not depend on the intrinsics)
    const int64_t nm = h->nobs_main;
    // stable counting sorts by camera / by group on host threads: per-part histograms, offsets in (key, part) order
    // (the index lists go into blocks of the pinned host cache and are uploaded from there: no zero-filled 12 MB vectors,
    // no staging copy; a block that goes back to the cache at the end of this scope is tagged with an event on the stream)
    std::vector<int> coff(h->nc + 1, 0), goff(h->ng + 1, 0), toff;
    HBuf<int> cidx, gidx;
    if (!cidx.resize((size_t)std::max<int64_t>(1, nm), true) || !gidx.resize((size_t)std::max<int64_t>(1, nm), true))
      return set_error(THEIA_HIP_ERR_OUT_OF_MEMORY, "host staging of %lld observations failed", (long long)nm);
    gidx[0] = 0; gidx.n = 1;
    auto is_depth = [&](int64_t s) { return p->obs_kind && p->obs_kind[h->perm[s]]; };
    auto bucket_sort = [&](int nkeys, std::vector<int>& off, HBuf<int>& idx, auto&& key_of) {   // key < 0: not listed
      const int parts = host_part_count(nm, 65536);
      const int64_t per = (nm + parts - 1) / parts;
      std::vector<int> hist((size_t)parts * nkeys, 0);
      host_parts(parts, true, [&](int t) {
        int* hh = hist.data() + (size_t)t * nkeys;
        for (int64_t s = t * per; s < std::min<int64_t>(nm, (t + 1) * per); ++s) { const int k = key_of(s); if (k >= 0) hh[k]++; }
      });
      int run = 0;
      for (int k = 0; k < nkeys; ++k) {
        off[k] = run;
        for (int t = 0; t < parts; ++t) { const int c = hist[(size_t)t * nkeys + k]; hist[(size_t)t * nkeys + k] = run; run += c; }
      }
      off[nkeys] = run;
      idx.n = (size_t)std::max(1, run);
      host_parts(parts, true, [&](int t) {
        int* hh = hist.data() + (size_t)t * nkeys;
        for (int64_t s = t * per; s < std::min<int64_t>(nm, (t + 1) * per); ++s) { const int k = key_of(s); if (k >= 0) idx[hh[k]++] = (int)s; }
      });
    };
    bucket_sort(h->nc, coff, cidx, [&](int64_t s) { return ocam[s]; });
    if (h->ni) bucket_sort(h->ng, goff, gidx, [&](int64_t s) { return is_depth(s) ? -1 : p->cam_group[ocam[s]]; });
    toff.reserve((size_t)h->np + 1);   // the sorted observations of a track are one range: its non-fixed count
    for (int r = 0; r < h->np; ++r) if (cnt_main[r + 1] > cnt_main[r]) toff.push_back((int)cnt_main[r]);
    toff.push_back((int)nm);
    h->in_ntracks = (int)toff.size() - 1;
    UP(in_cam_off, coff); UP(in_grp_off, goff); UP(in_trk_off, toff);
    if ((rc = h->in_cam_idx.upload(cidx.data(), cidx.n, st, cidx.pinned())) || (rc = h->in_grp_idx.upload(gidx.data(), gidx.n, st, gidx.pinned()))) return rc;
    AL(in_cam, (size_t)6 * std::max(1, h->nc)); AL(in_pts, (size_t)4 * std::max(1, h->np));
    AL(in_intr, (size_t)THEIA_MAX_INTRINSICS * std::max(1, h->ng));
    AL(in_scal, 8); AL(in_part, 4 * (size_t)kInnerCostBlocks); AL(in_gate, 4);
    if (!h->use_fused) AL(camrot_cand, (size_t)40 * std::max(1, h->nc));   // the track sweep reads the cameras as k_cam_prep-style blocks (k_inner_cam_blocks)
    if (h->ni && inner_group_wgs(h->ng) > 1) { AL(in_grp_part, (size_t)h->ng * 2 * inner_group_wgs(h->ng) * kInnerGroupSums); AL(in_grp_bar, 2 * (size_t)std::max(1, h->ng) + 2); }
  }
  if (p->obs_kind) {   // depth-prior rows (sorted like the other observation arrays)
    std::vector<uint8_t> okind(h->nobs);
    for (int64_t s = 0; s < h->nobs; ++s) okind[s] = p->obs_kind[h->perm[s]];
    UP(obs_kind, okind);
  }
  UP(tile_start, tstart); UP(tile_count, tcount);
  UP(long_obs_index, l_obs); UP(long_obs_slot, l_slot); UP(long_track_start, l_start); UP(long_track_pt, l_pt);
  AL(long_scratch, (size_t)14 * std::max(1, h->long_ntracks));
  UP(d_cam_red, h->cam_red); UP(d_cam_mask, h->cam_mask); UP(d_pt_const, h->pt_const);
  std::vector<int> gm(p->group_model, p->group_model + h->ng), cg(p->cam_group, p->cam_group + h->nc);
  h->model_mask = 0u;
  for (int g = 0; g < h->ng; ++g) h->model_mask |= 1u << p->group_model[g];
  UP(group_model, gm); UP(cam_group, cg);
  for (int k = 0; k < 2; ++k) { AL(cam[k], (size_t)6 * h->nc); AL(pts[k], (size_t)4 * h->np); AL(intr[k], (size_t)THEIA_MAX_INTRINSICS * h->ng); }
  UP(d_grp_red, h->grp_red); UP(d_grp_free, h->grp_free); UP(d_grp_k, h->grp_k);
  {
    std::vector<unsigned> rf((size_t)std::max(1, h->ngv), 0u);
    int most = 0;
    for (int g = 0; g < h->ng; ++g)
      if (h->grp_red[g] >= 0) { rf[h->grp_red[g]] = h->grp_free[g]; most = std::max(most, __builtin_popcount(h->grp_free[g])); }
    UP(d_red_free, rf);
    const char* force = getenv("THEIA_HIP_INTR_ROWS");
    h->intr_rows = (most <= 4 && !(force && atoi(force) == 10)) ? 4 : 10;
  }
  {
    std::vector<double> ones_i((size_t)THEIA_MAX_INTRINSICS * h->ng, 1.0);
    UP(ones_i, ones_i); UP(scale_i, ones_i);
  }
  AL(colsq_i0, (size_t)THEIA_MAX_INTRINSICS * h->ng); AL(scale_red, (size_t)std::max(1, h->n));
  AL(ones_c, (size_t)6 * h->nc); AL(ones_p, (size_t)h->pd * h->np); AL(scale_c, (size_t)6 * h->nc); AL(scale_p, (size_t)h->pd * h->np);
  for (DevBuf<double>* b : {&h->ones_c, &h->ones_p, &h->scale_c, &h->scale_p})   // filled on the device
    if (b->n) k_fill_value<<<(unsigned)std::min<size_t>(1024, (b->n + 255) / 256), 256, 0, st>>>(b->p, b->n, 1.0);
  AL(colsq_c0, (size_t)6 * h->nc); AL(colsq_p0, (size_t)h->pd * h->np);
  {
    std::vector<int> a(&kCfgF2S[0][0], &kCfgF2S[0][0] + 24), b(&kCfgMax[0][0], &kCfgMax[0][0] + 24);
    UP(f2s, a); UP(fmaxflag, b);
  }
  const size_t nn = (size_t)h->n * h->n;
  AL(reduce, nn + 3 * (size_t)h->n + SC_COUNT);
  if (h->reduce.n) HIP_TRY(hipMemsetAsync(h->reduce.p, 0, sizeof(double) * h->reduce.n, st));
  h->rb.base = h->reduce.p; h->rb.count = h->reduce.n;
  h->rb.S = h->reduce.p; h->rb.rhs = h->rb.S + nn; h->rb.colsq = h->rb.rhs + h->n; h->rb.gc = h->rb.colsq + h->n;
  h->rb.scal = h->rb.gc + h->n;
  AL(Vinv, (size_t)(h->pd * (h->pd + 1) / 2) * h->np); AL(gp, (size_t)h->pd * h->np);
  // constant points are never written: the Schur readers rebuild T = W V^-1 from these arrays and need zeros there
  if (h->Vinv.n) HIP_TRY(hipMemsetAsync(h->Vinv.p, 0, sizeof(double) * h->Vinv.n, st));
  if (h->gp.n) HIP_TRY(hipMemsetAsync(h->gp.p, 0, sizeof(double) * h->gp.n, st));
  AL(tile_part, (size_t)5 * std::max(1, h->ntiles_all)); AL(scalB, 16); AL(red_part, (size_t)8 * kReduceBlocks);
  AL(chol_work, dense_cholesky_workspace(h->n));
  AL(lm_state, sizeof(LmState)); AL(lm_ctl, sizeof(LmCtl));
  tick("allocations + uploads");
  {
    // camera priors in use: the camera's bit AND the option's bit (bundle_adjuster.cc:159-172,291-313)
    std::vector<int> pc, pk;
    std::vector<double> pv, pi;
    if (p->cam_prior_mask && o->prior_mask) {
      const double* vecs[3] = {p->cam_position_prior, p->cam_gravity_prior, p->cam_orientation_prior};
      const double* infos[3] = {p->cam_position_prior_sqrt_info, p->cam_gravity_prior_sqrt_info, p->cam_orientation_prior_sqrt_info};
      for (int c = 0; c < h->nc; ++c)
        for (int k = 0; k < 3; ++k) {
          const int bit = 1 << k;
          if (!(p->cam_prior_mask[c] & bit) || !(o->prior_mask & bit) || !vecs[k] || !infos[k]) continue;
          pc.push_back(c); pk.push_back(bit);
          pv.insert(pv.end(), vecs[k] + 3 * (size_t)c, vecs[k] + 3 * (size_t)c + 3);
          pi.insert(pi.end(), infos[k] + 9 * (size_t)c, infos[k] + 9 * (size_t)c + 9);
        }
    }
    h->n_priors = (int)pc.size();
    UP(prior_cam, pc); UP(prior_kind, pk); UP(prior_vec, pv); UP(prior_info, pi);
  }
  {
    // tile co-visibility: two 64-wide tiles of S couple iff a variable track is
    // seen by cameras of both (the Schur complement's block structure)
    const int nt = (h->n + 63) / 64;
    h->tile_adj.assign((size_t)nt * nt, 0);
    if (h->use_fused && fplan.tile_adj.size() == (size_t)nt * nt) {
      h->tile_adj = fplan.tile_adj;   // marked track by track while the fused plan was built (same rule as below)
    } else {
    std::vector<int64_t> off(h->np + 1, 0);
    for (int64_t i = 0; i < h->nobs; ++i)
      if (h->cam_red[p->obs_cam[i]] >= 0 && !h->pt_const[p->obs_pt[i]]) off[p->obs_pt[i] + 1]++;
    for (int q = 0; q < h->np; ++q) off[q + 1] += off[q];
    std::vector<int> rcs(off[h->np]);
    {
      std::vector<int64_t> fill(off.begin(), off.end() - 1);
      for (int64_t i = 0; i < h->nobs; ++i) {
        const int rcam = h->cam_red[p->obs_cam[i]];
        if (rcam >= 0 && !h->pt_const[p->obs_pt[i]]) rcs[fill[p->obs_pt[i]]++] = rcam;
      }
    }
    std::vector<int> tl;
    for (int q = 0; q < h->np; ++q) {
      tl.clear();
      for (int64_t k = off[q]; k < off[q + 1]; ++k) {
        const int s0 = h->ni + 6 * rcs[k];
        tl.push_back(s0 / 64);
        if ((s0 + 5) / 64 != s0 / 64) tl.push_back((s0 + 5) / 64);
      }
      std::sort(tl.begin(), tl.end());
      tl.erase(std::unique(tl.begin(), tl.end()), tl.end());
      for (int a : tl) for (int b : tl) h->tile_adj[(size_t)a * nt + b] = 1;
    }
    }
    // a variable camera's own 6 x 6 block is written whether or not any of its tracks is variable (F^T F, priors):
    // its tile(s), and the off-diagonal tile when its rows straddle a 64-row boundary, always belong to the plan
    for (int rcam = 0; rcam < h->ncv; ++rcam) {
      const int s0 = h->ni + 6 * rcam, a = s0 / 64, b = (s0 + 5) / 64;
      h->tile_adj[(size_t)a * nt + a] = 1;
      h->tile_adj[(size_t)b * nt + b] = 1;
      h->tile_adj[(size_t)a * nt + b] = h->tile_adj[(size_t)b * nt + a] = 1;
    }
    // which tile columns THIS problem (a rank's shard) writes into: cameras it observes (any point, constant ones included:
    // their F^T F lands on the diagonal) or holds a prior for.  The distributed K3 of a sharded solve asks for it (sync_plan).
    h->tile_touch.assign(nt, 0);
    for (int c = 0; c < h->nc; ++c) {
      const bool prior = p->cam_prior_mask && o->prior_mask && (p->cam_prior_mask[c] & o->prior_mask);
      if ((!cam_used[c] && !prior) || h->cam_red[c] < 0) continue;
      const int s0 = h->ni + 6 * h->cam_red[c];
      h->tile_touch[s0 / 64] = 1; h->tile_touch[(s0 + 5) / 64] = 1;
    }
    // shared intrinsics couple with every camera of their group: treat as dense
    for (int a = 0; a < (h->ni + 63) / 64; ++a)
      for (int b = 0; b < nt; ++b) h->tile_adj[(size_t)a * nt + b] = h->tile_adj[(size_t)b * nt + a] = 1;
    tick("priors + tile adjacency");
    h->plan = chol_plan_create(h->n, h->tile_adj.data());
    tick("K3 plan");
  }
  if (h->ni == 0 && h->ntiles_main > 0 && (rc = build_gather_lists(h, ocam, opt, l_obs, !h->use_fused))) return rc;
  if (h->use_fused) {
    h->n_fruns = (int)fplan.runs.size(); h->n_sum_items = (int)fplan.sum_items.size() / 6 - h->n_sum_items2;
    fplan.tile_trk_end.resize(std::max<size_t>(1, fplan.tile_trk_end.size()));
    if (ctiming) {   // THEIA_HIP_CREATE_TIMING: shape of the fused plan
      std::map<int, std::pair<int, int>> by;   // gp -> (runs, sub-chunks)
      long long nsc = 0;
      for (const FusedRun& r : fplan.runs) { auto& e = by[r.gp]; e.first++; e.second += (r.ntiles + 3) / 4; nsc += (r.ntiles + 3) / 4; }
      fprintf(stderr, "theia_hip fused plan: %zu runs, %lld sub-chunks (%.1f obs each), %zu partial doubles, %d sum items\n",
              fplan.runs.size(), nsc, nsc ? (double)h->nobs_main / nsc : 0.0, fplan.part_doubles, h->n_sum_items);
      for (auto& kv : by) fprintf(stderr, "  G=%d slices/wave=%d: %d runs, %d sub-chunks\n", kv.first & 0xff, kv.first >> 8, kv.second.first, kv.second.second);
    }
    if (fplan.runs.empty()) fplan.runs.push_back(FusedRun{0, 0, 0, 0, 0, 0, 0, 1 | (1 << 8), 0, 0});
    {   // the workgroups of k_lin_schur take runs from a queue, the most expensive first (cost ~ wave tiles, weighted by the
        // target blocks a wave step covers): the kernel ends when the last run does, and with ~4 runs per workgroup a
        // static round robin left workgroups with one run more than others waiting for them
      std::vector<int> order(fplan.runs.size());
      for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
      const int lanes_tgt = fused_lanes_per_target(h->fused_bw);
      auto cost = [&](int i) { const FusedRun& r = fplan.runs[i]; return (long long)r.ntiles * (64 + lanes_tgt * r.ntgt); };
      std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cost(a) > cost(b); });
      UP(frun_order, order);
      AL(frun_next, 2);
    }
    fplan.stage.resize(std::max<size_t>(1, fplan.stage.size()));
    UP(fruns, fplan.runs); UP(frun_cams, fplan.cams); UP(frun_stage, fplan.stage); UP(frun_tgt, fplan.tgts); UP(obs_lc, fplan.obs_lc); UP(obs_tl, fplan.obs_tl);
    UP(tile_trk_end, fplan.tile_trk_end); UP(sum_items, fplan.sum_items); UP(sum_src, fplan.sum_src);
    AL(fpart, std::max<size_t>(1, fplan.part_doubles));
    AL(camrot, (size_t)40 * std::max(1, h->nc)); AL(camrot_cand, (size_t)40 * std::max(1, h->nc)); AL(camdir, (size_t)12 * std::max(1, h->nc));
  }
  if (h->ni > 0 && !h->use_fused && h->ntiles_main > 0 && (rc = build_gather_lists_intr(h, p, ocam, opt, l_obs))) return rc;
#undef UP
#undef AL
  tick("gather lists");
  fill_devproblem(h);
  rc = upload_parameters(h, p);
  if (rc) return rc;
  tick("parameter upload");
  // fixed cost
  double fc = 0.0, inv = 0.0;
  rc = cost_of_tiles(h, h->ntiles_eval, h->ntiles_all - h->ntiles_eval, h->cam[0].p, h->pts[0].p, &fc, &inv);
  if (rc) return rc;
  if (h->n_priors) {   // priors on constant cameras: residual blocks without variable parameters
    double pf = 0.0;
    HIP_TRY(hipMemsetAsync(h->scalB.p, 0, sizeof(double) * 16, h->stream));
    launch_cam_priors(h->P, PRIOR_FIXED, h->cam[0].p, nullptr, nullptr, nullptr, nullptr, h->scalB.p, nullptr, h->stream);
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy(&pf, h->scalB.p, sizeof(double), hipMemcpyDeviceToHost));
    fc += pf;
  }
  h->fixed_cost = fc;
  tick("fixed cost");
  *out = guard.release();
  return 0;
}

int theia_hip_ba_reset_parameters(theia_ba_handle h, const theia_ba_problem* p) {
  if (!h || !p) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "null argument");
  if (h->idh) return thip::id_handle_reset(h->idh, p);
  if (p->num_cameras != h->nc || p->num_points != h->np || p->num_groups != h->ng)
    return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "problem shape differs from the handle's");
  release_stage_if_idle(h);
  return upload_parameters(h, p);
}

int theia_hip_ba_set_shard(theia_ba_handle h, int32_t rank, int32_t world_size) {
  if (!h) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "null handle");
  if (world_size < 1 || rank < 0 || rank >= world_size) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "rank / world_size out of range");
  h->shard_rank = rank; h->shard_world = world_size;
  h->have_scale = false; h->camrot_valid = false;   // the Jacobi scales are all-reduced over the new geometry: recomputed, and the camera blocks with them
  return 0;
}

int theia_hip_ba_snapshot_parameters(theia_ba_handle h) {
  if (h && h->idh) return thip::id_handle_snapshot(h->idh);
  if (!h) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "null handle");
  release_stage_if_idle(h);
  int rc;
  if ((rc = h->snap_cam.alloc(h->cam[0].n)) || (rc = h->snap_pts.alloc(h->pts[0].n)) || (rc = h->snap_intr.alloc(h->intr[0].n))) return rc;
  const int c = h->cur;
  if (h->cam[c].n) HIP_TRY(hipMemcpyAsync(h->snap_cam.p, h->cam[c].p, sizeof(double) * h->cam[c].n, hipMemcpyDeviceToDevice, h->stream));
  if (h->pts[c].n) HIP_TRY(hipMemcpyAsync(h->snap_pts.p, h->pts[c].p, sizeof(double) * h->pts[c].n, hipMemcpyDeviceToDevice, h->stream));
  if (h->intr[c].n) HIP_TRY(hipMemcpyAsync(h->snap_intr.p, h->intr[c].p, sizeof(double) * h->intr[c].n, hipMemcpyDeviceToDevice, h->stream));
  h->has_snapshot = true;
  return 0;
}

int theia_hip_ba_restore_parameters(theia_ba_handle h) {
  if (h && h->idh) return thip::id_handle_restore(h->idh);
  if (!h) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "null handle");
  if (!h->has_snapshot) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "no snapshot taken on this handle");
  release_stage_if_idle(h);
  for (int k = 0; k < 2; ++k) {
    if (h->cam[k].n) HIP_TRY(hipMemcpyAsync(h->cam[k].p, h->snap_cam.p, sizeof(double) * h->cam[k].n, hipMemcpyDeviceToDevice, h->stream));
    if (h->pts[k].n) HIP_TRY(hipMemcpyAsync(h->pts[k].p, h->snap_pts.p, sizeof(double) * h->pts[k].n, hipMemcpyDeviceToDevice, h->stream));
    if (h->intr[k].n) HIP_TRY(hipMemcpyAsync(h->intr[k].p, h->snap_intr.p, sizeof(double) * h->intr[k].n, hipMemcpyDeviceToDevice, h->stream));
  }
  h->cur = 0;
  h->P.intr = h->intr[0].p; h->P.intr_cand = h->intr[1].p;
  h->have_scale = false; h->camrot_valid = false;
  return 0;
}

int theia_hip_ba_set_options(theia_ba_handle h, const theia_ba_options* o) {
  if (!h || !o) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "null argument");
  if (h->idh) {   // the run checks the structural options against the ones the object was created with
    if (o->intrinsics_to_optimize != h->opt.intrinsics_to_optimize || o->constant_camera_position != h->opt.constant_camera_position ||
        o->constant_camera_orientation != h->opt.constant_camera_orientation || o->prior_mask != h->opt.prior_mask)
      return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "structural options differ from the ones the handle was created with");
    h->opt = *o;
    return 0;
  }
  const theia_ba_options& c = h->opt;
  if (o->use_homogeneous_point_parametrization != c.use_homogeneous_point_parametrization ||
      o->constant_camera_orientation != c.constant_camera_orientation ||
      o->constant_camera_position != c.constant_camera_position || o->orthographic_camera != c.orthographic_camera ||
      o->intrinsics_to_optimize != c.intrinsics_to_optimize || o->prior_mask != c.prior_mask)
    return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "structural options differ from the ones the handle was created with");
  if (o->use_inner_iterations && !h->inner && h->nobs_main > 0)
    return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "use_inner_iterations: the handle was created without the inner-iteration lists");
  if (o->loss_function_type < 0 || o->loss_function_type > THEIA_LOSS_TRUNCATED)
    return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "invalid loss function type");
  if (o->loss_function_type != h->opt.loss_function_type || o->robust_loss_width != h->opt.robust_loss_width ||
      o->robust_loss_width_depth_prior != h->opt.robust_loss_width_depth_prior)
    h->drop_graph();   // the loss is baked into the captured kernel arguments
  h->opt = *o;
  h->P.loss_type = o->loss_function_type;
  h->P.loss_width = o->robust_loss_width;
  h->P.loss_width_depth = o->robust_loss_width_depth_prior;
  return 0;
}

int theia_hip_ba_set_allreduce(theia_ba_handle h, theia_allreduce_fn fn, void* ctx) {
  if (h && h->idh) return set_error(THEIA_HIP_ERR_UNSUPPORTED, "inverse-depth handle: sharding is not built in this mode");
  if (!h) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "null handle");
  h->allreduce = fn; h->allreduce_ctx = ctx;
  h->plan_is_global = (fn == nullptr);   // the K3 schedule must cover every rank's tracks
  return 0;
}

int theia_hip_ba_set_inner_global(theia_ba_handle h, const theia_ba_problem* full, const int64_t* point_global_index) {
  if (!h || !full || (h->np > 0 && !point_global_index)) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "null argument");
  if (h->idh) return set_error(THEIA_HIP_ERR_UNSUPPORTED, "inverse-depth handle: sharding is not built in this mode");
  if (!h->inner) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "the handle was created without use_inner_iterations");
  if (full->num_cameras != h->nc || full->num_groups != h->ng)
    return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "the full problem must carry the shard's cameras and intrinsics groups");
  if (full->num_obs > 0 && (!full->obs_uv || !full->obs_cam || !full->obs_pt)) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "null observation array");
  const int64_t nobs = full->num_obs;
  const int gnp = full->num_points;
  if (nobs >= (int64_t)1 << 31) return set_error(THEIA_HIP_ERR_UNSUPPORTED, "more than 2^31 observations");
  std::vector<int> gp(h->np);
  for (int i = 0; i < h->np; ++i) {
    if (point_global_index[i] < 0 || point_global_index[i] >= gnp) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "point_global_index out of range");
    gp[i] = (int)point_global_index[i];
  }
  for (int64_t i = 0; i < nobs; ++i)
    if (full->obs_cam[i] < 0 || full->obs_cam[i] >= h->nc || full->obs_pt[i] < 0 || full->obs_pt[i] >= gnp)
      return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "observation index out of range");
  hipStream_t st = h->stream;
  int rc = 0;
  // observation lists by camera and by intrinsics group (depth-prior rows do not depend on the intrinsics), in the caller's order
  std::vector<int> coff(h->nc + 1, 0), cidx(nobs), goff(h->ng + 1, 0), gidx;
  for (int64_t i = 0; i < nobs; ++i) coff[full->obs_cam[i] + 1]++;
  for (int c = 0; c < h->nc; ++c) coff[c + 1] += coff[c];
  { std::vector<int> fill(coff.begin(), coff.end() - 1); for (int64_t i = 0; i < nobs; ++i) cidx[fill[full->obs_cam[i]]++] = (int)i; }
  if (h->ni) {
    auto is_depth = [&](int64_t i) { return full->obs_kind && full->obs_kind[i]; };
    for (int64_t i = 0; i < nobs; ++i) if (!is_depth(i)) goff[full->cam_group[full->obs_cam[i]] + 1]++;
    for (int g = 0; g < h->ng; ++g) goff[g + 1] += goff[g];
    gidx.resize(goff[h->ng]);
    std::vector<int> fill(goff.begin(), goff.end() - 1);
    for (int64_t i = 0; i < nobs; ++i) if (!is_depth(i)) gidx[fill[full->cam_group[full->obs_cam[i]]]++] = (int)i;
  }
  if (gidx.empty()) gidx.push_back(0);
  std::vector<int> pc, pk;
  std::vector<double> pv, pi;
  if (full->cam_prior_mask && h->opt.prior_mask) {   // the priors of ALL cameras: the shards carry them on one rank only
    const double* vecs[3] = {full->cam_position_prior, full->cam_gravity_prior, full->cam_orientation_prior};
    const double* infos[3] = {full->cam_position_prior_sqrt_info, full->cam_gravity_prior_sqrt_info, full->cam_orientation_prior_sqrt_info};
    for (int c = 0; c < h->nc; ++c)
      for (int k = 0; k < 3; ++k) {
        const int bit = 1 << k;
        if (!(full->cam_prior_mask[c] & bit) || !(h->opt.prior_mask & bit) || !vecs[k] || !infos[k]) continue;
        pc.push_back(c); pk.push_back(bit);
        pv.insert(pv.end(), vecs[k] + 3 * (size_t)c, vecs[k] + 3 * (size_t)c + 3);
        pi.insert(pi.end(), infos[k] + 9 * (size_t)c, infos[k] + 9 * (size_t)c + 9);
      }
  }
  h->g_npriors = (int)pc.size(); h->g_np = gnp; h->g_nobs = nobs;
  std::vector<int> oc(full->obs_cam, full->obs_cam + nobs), op(full->obs_pt, full->obs_pt + nobs);
  if ((rc = h->g_uv.upload(reinterpret_cast<const double2*>(full->obs_uv), (size_t)nobs, st, false)) ||
      (rc = h->g_cam.upload(oc, st)) || (rc = h->g_pt.upload(op, st)) || (rc = h->g_cam_off.upload(coff, st)) || (rc = h->g_cam_idx.upload(cidx, st)) ||
      (rc = h->g_grp_off.upload(goff, st)) || (rc = h->g_grp_idx.upload(gidx, st)) || (rc = h->g_pidx.upload(gp, st)) ||
      (rc = h->g_prior_cam.upload(pc, st)) || (rc = h->g_prior_kind.upload(pk, st)) || (rc = h->g_prior_vec.upload(pv, st)) ||
      (rc = h->g_prior_info.upload(pi, st)) || (rc = h->g_pts.alloc((size_t)4 * std::max(1, gnp))) || (rc = h->g_stage.alloc(8)))
    return rc;
  if (full->obs_sqrt_info) { if ((rc = h->g_si.upload(reinterpret_cast<const double2*>(full->obs_sqrt_info), (size_t)nobs, st, false))) return rc; }
  else if ((rc = h->g_si.alloc(0))) return rc;
  if (full->obs_kind) { if ((rc = h->g_kind.upload(full->obs_kind, (size_t)nobs, st, false))) return rc; }
  else if ((rc = h->g_kind.alloc(0))) return rc;
  HIP_TRY(hipStreamSynchronize(st));   // the sources above are local vectors and the caller's arrays
  h->inner_global = true;
  return 0;
}

int theia_hip_ba_plan_info(theia_ba_handle h, int32_t* n, int32_t* k3_levels, double* k3_flops, int32_t* fused_runs,
                           int32_t* slow_path_tracks) {
  if (h && h->idh) {   // (no fused runs / slow-path tracks in this mode: the per-track kernels of ba_invdepth.hip)
    thip::id_handle_plan_info(h->idh, n, k3_levels, k3_flops);
    if (fused_runs) *fused_runs = 0;
    if (slow_path_tracks) *slow_path_tracks = 0;
    return 0;
  }
  if (!h) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "null handle");
  if (n) *n = h->n;
  if (k3_levels) *k3_levels = chol_plan_levels(h->plan);
  if (k3_flops) *k3_flops = chol_plan_flops(h->plan);
  if (fused_runs) *fused_runs = h->use_fused ? h->n_fruns : 0;
  if (slow_path_tracks) *slow_path_tracks = h->long_ntracks;
  return 0;
}

int theia_hip_ba_download(theia_ba_handle h, theia_ba_problem* p) {
  if (!h || !p) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "null argument");
  if (h->idh) return thip::id_handle_download(h->idh, p);
  // caller-owned (pageable) destinations: drain the stream, then blocking copies
  HIP_TRY(hipStreamSynchronize(h->stream));
  h->stage.release();
  if (h->nc) HIP_TRY(hipMemcpy(p->cam_ext, h->cam[h->cur].p, sizeof(double) * 6 * h->nc, hipMemcpyDeviceToHost));
  if (h->np) HIP_TRY(hipMemcpy(p->points, h->pts[h->cur].p, sizeof(double) * 4 * h->np, hipMemcpyDeviceToHost));
  if (h->ng && h->ni) HIP_TRY(hipMemcpy(p->intrinsics, h->intr[h->cur].p, sizeof(double) * THEIA_MAX_INTRINSICS * h->ng, hipMemcpyDeviceToHost));
  return 0;
}

int theia_hip_ba_destroy(theia_ba_handle h) {
  debug_sticky("destroy entry");
  delete h;
  debug_sticky("destroy exit");
  return 0;
}

int theia_hip_ba_run(theia_ba_handle h, theia_ba_summary* S) {
  if (!h || !S) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "null argument");
  if (h->idh) return thip::id_handle_run(h->idh, &h->opt, S);
  debug_sticky("run entry");
  release_stage_if_idle(h);   // (a previous run() that returned early with an error did not reach its release)
  const theia_ba_options& O = h->opt;
  const double t_start = now_s();
  S->trace_size = 0; S->success = 0; S->num_iterations = 0; S->num_successful_steps = 0;
  S->time_linearize = S->time_solve_reduced = S->time_backsub = 0.0;
  S->time_kernel_linearize = 0.0; S->num_linearize_launches = 0;
  S->setup_time_in_seconds = 0.0;
  int rc = h->plan_is_global ? 0 : sync_plan(h);
  if (rc) return rc;
  h->cur = 0;   // state = buffer 0, candidate = buffer 1 (accepted steps are copied back on the device)
  rc = compute_scale(h);
  if (rc) return rc;
  // device-resident LM state, control block and trace
  LmState st;
  std::memset(&st, 0, sizeof(st));
  st.radius = 1e4; st.decrease_factor = 2.0; st.step_successful = 1; st.first = 1;
  st.term = THEIA_TERM_NO_CONVERGENCE; st.pending_grad = -1;
  // inner iterations need every residual block of a camera on this rank: a sharded solve that asks for them (the
  // reference's default, bundle_adjustment.h:144) is refused instead of silently walking another trajectory
  if (h->allreduce && O.use_inner_iterations != 0 && h->nobs > 0 && !h->inner_global)
    return set_error(THEIA_HIP_ERR_UNSUPPORTED, "use_inner_iterations in a sharded solve needs theia_hip_ba_set_inner_global on every rank (or set it to 0 on every rank)");
  const bool inner = h->inner && O.use_inner_iterations != 0;
  st.inner_enabled = inner ? 1 : 0;
  LmState* dst = reinterpret_cast<LmState*>(h->lm_state.p);
  // the initial state and the control block travel as kernel arguments (k_lm_init): no host buffer whose
  // lifetime would need a synchronisation before the first iteration
  k_lm_init_state<<<1, 1, 0, h->stream>>>(dst, st);
  // |x| of the variable blocks at the start: summed on the device (points per shard, all-reduced)
  HIP_TRY(hipMemsetAsync(h->scalB.p, 0, sizeof(double) * 16, h->stream));
  if (h->xnorm_part.n < 2 * (size_t)kXnormBlocks && (rc = h->xnorm_part.alloc(2 * kXnormBlocks))) return rc;
  k_xnorm_partial<<<kXnormBlocks, 256, 0, h->stream>>>(h->P, h->cam[0].p, h->pts[0].p, h->intr[0].p, h->xnorm_part.p);
  k_xnorm_reduce<<<1, 1, 0, h->stream>>>(h->xnorm_part.p, kXnormBlocks, h->scalB.p);
  rc = do_allreduce(h, h->scalB.p, 1, THEIA_REDUCE_SUM);
  if (rc) return rc;
  k_xnorm_set<<<1, 1, 0, h->stream>>>(dst, h->scalB.p);
  LmCtl ctl;
  ctl.max_iterations = O.max_num_iterations;
  ctl.trace_capacity = S->trace_cost ? S->trace_capacity : 0;
  ctl.function_tolerance = O.function_tolerance; ctl.gradient_tolerance = O.gradient_tolerance;
  ctl.parameter_tolerance = O.parameter_tolerance; ctl.max_radius = O.max_trust_region_radius; ctl.fixed_cost = h->fixed_cost;
  const size_t tcap = (size_t)std::max(1, ctl.trace_capacity);
  if (h->tr_cost.n < tcap) {
    if ((rc = h->tr_cost.alloc(tcap)) || (rc = h->tr_g.alloc(tcap)) || (rc = h->tr_step.alloc(tcap)) ||
        (rc = h->tr_radius.alloc(tcap)) || (rc = h->tr_acc.alloc(tcap)))
      return rc;
  }
  ctl.tc = ctl.trace_capacity ? h->tr_cost.p : nullptr;
  ctl.tg = h->tr_g.p; ctl.ts = h->tr_step.p; ctl.tr = h->tr_radius.p; ctl.ta = h->tr_acc.p;
  ctl.inner_scal = inner ? h->in_scal.p : nullptr;
  k_lm_init_ctl<<<1, 1, 0, h->stream>>>(reinterpret_cast<LmCtl*>(h->lm_ctl.p), ctl);
  const LmCtl* dctl = reinterpret_cast<const LmCtl*>(h->lm_ctl.p);
  const int nxt = 1;
  // one LM iteration ("body"): linearise + Schur, solve, trial step, step control, accept
  auto enqueue_body = [&](int slot) -> int {
    int r;
    if (slot >= 0) HIP_TRY(hipEventRecord(h->ev[slot][0], h->stream));
    if ((r = enqueue_linearize(h, slot))) return r;
    if (slot >= 0) HIP_TRY(hipEventRecord(h->ev[slot][1], h->stream));
    const bool fuse = !h->allreduce && !inner && slot < 0 && h->ntiles_main > 0;   // tile reduction inside the control kernel
    if ((r = enqueue_solve_and_backsub(h, slot, fuse))) return r;
    if (inner) {
      // DoInnerIterationsIfNeeded: one sweep of block coordinate descent on a copy of the candidate, its cost and its
      // distance from x; every kernel returns at once when the gate is closed (step invalid, inner iterations off, done)
      k_inner_gate<<<1, 1, 0, h->stream>>>(dst, h->rb.scal, h->scalB.p, h->in_gate.p);
      HIP_TRY(hipMemcpyAsync(h->in_cam.p, h->cam[nxt].p, sizeof(double) * 6 * h->nc, hipMemcpyDeviceToDevice, h->stream));
      HIP_TRY(hipMemcpyAsync(h->in_pts.p, h->pts[nxt].p, sizeof(double) * 4 * h->np, hipMemcpyDeviceToDevice, h->stream));
      HIP_TRY(hipMemcpyAsync(h->in_intr.p, (h->ni ? h->intr[nxt].p : h->intr[0].p), sizeof(double) * THEIA_MAX_INTRINSICS * h->ng, hipMemcpyDeviceToDevice, h->stream));
      InnerArgs IA;
      IA.P = h->P;
      IA.cam_obs_off = h->in_cam_off.p; IA.cam_obs_idx = h->in_cam_idx.p; IA.grp_obs_off = h->in_grp_off.p; IA.grp_obs_idx = h->in_grp_idx.p;
      IA.trk_off = h->in_trk_off.p; IA.ntracks = h->in_ntracks; IA.nobs = h->nobs_main;
      IA.cam = h->in_cam.p; IA.pts = h->in_pts.p; IA.intr = h->in_intr.p; IA.gate = h->in_gate.p;
      IA.grp_part = h->in_grp_part.p; IA.grp_bar = h->in_grp_bar.p; IA.grp_wgs = h->in_grp_part.p ? inner_group_wgs(h->ng) : 1;
      if (h->allreduce && h->inner_global) {
        // the full candidate point set: every shard's points at their global indices, summed over the ranks
        HIP_TRY(hipMemsetAsync(h->g_pts.p, 0, sizeof(double) * 4 * (size_t)h->g_np, h->stream));
        launch_inner_scatter_points(h->np, h->in_pts.p, h->g_pidx.p, h->g_pts.p, h->stream);
        if ((r = do_allreduce(h, h->g_pts.p, (size_t)4 * h->g_np, THEIA_REDUCE_SUM))) return r;
        InnerArgs IG = IA;      // cameras and groups over the full observation set, identically on every rank
        IG.P.obs_uv = h->g_uv.p; IG.P.obs_si = h->g_si.n ? h->g_si.p : nullptr; IG.P.obs_cam = h->g_cam.p; IG.P.obs_pt = h->g_pt.p;
        IG.P.obs_kind = h->g_kind.n ? h->g_kind.p : nullptr; IG.P.np = h->g_np; IG.P.nobs = h->g_nobs;
        IG.P.n_priors = h->g_npriors; IG.P.prior_cam = h->g_prior_cam.p; IG.P.prior_kind = h->g_prior_kind.p;
        IG.P.prior_vec = h->g_prior_vec.p; IG.P.prior_info = h->g_prior_info.p;
        IG.cam_obs_off = h->g_cam_off.p; IG.cam_obs_idx = h->g_cam_idx.p; IG.grp_obs_off = h->g_grp_off.p; IG.grp_obs_idx = h->g_grp_idx.p;
        IG.pts = h->g_pts.p; IG.ntracks = 0; IG.nobs = h->g_nobs;
        // with the shard geometry known the cameras (then the groups) are dealt to the ranks by index and the results summed
        // (the non-owned entries zeroed: x + 0 is exact, every rank ends with the same bits); otherwise every rank sweeps all
        const bool deal = h->shard_world > 1 && h->shard_rank >= 0 && h->shard_rank < h->shard_world;
        if (deal) { IG.own_rank = h->shard_rank; IG.own_world = h->shard_world; }
        launch_inner_sweep(IG, h->stream, 1);
        if (deal) {
          launch_inner_keep_owned(h->in_cam.p, h->nc, 6, h->shard_rank, h->shard_world, h->stream);
          if ((r = do_allreduce(h, h->in_cam.p, (size_t)6 * h->nc, THEIA_REDUCE_SUM))) return r;
        }
        if (h->ni > 0) {
          launch_inner_sweep(IG, h->stream, 2);
          if (deal) {
            launch_inner_keep_owned(h->in_intr.p, h->ng, THEIA_MAX_INTRINSICS, h->shard_rank, h->shard_world, h->stream);
            if ((r = do_allreduce(h, h->in_intr.p, (size_t)THEIA_MAX_INTRINSICS * h->ng, THEIA_REDUCE_SUM))) return r;
          }
        }
        launch_inner_sweep(IA, h->stream, 4);   // this shard's tracks against the swept cameras
        launch_inner_norms(IA, h->cam[0].p, h->pts[0].p, h->intr[0].p, h->g_stage.p, h->in_part.p, h->stream, 1);
        launch_inner_cost(IA, h->in_part.p, h->g_stage.p + 2, h->stream);
        if ((r = do_allreduce(h, h->g_stage.p, 4, THEIA_REDUCE_SUM))) return r;
        launch_inner_norms(IA, h->cam[0].p, h->pts[0].p, h->intr[0].p, h->g_stage.p + 4, h->in_part.p, h->stream, 2);
        launch_inner_combine(h->g_stage.p, h->g_stage.p + 4, h->in_scal.p, h->stream);
      } else {
      launch_inner_sweep(IA, h->stream);
      launch_inner_norms_cost(IA, h->cam[0].p, h->pts[0].p, h->intr[0].p, h->in_scal.p, h->in_part.p, h->stream);   // step norms + cost, two launches
      }
    }
    if (slot >= 0) HIP_TRY(hipEventRecord(h->ev[slot][3], h->stream));
    if (fuse && h->ntiles_main > 4 * kReduceBlocks) {
      launch_reduce_tiles_stage1(h->ntiles_main, h->tile_part.p, 5, h->fmaxflag.p + 8, h->red_part.p, h->stream);
      k_reduce_control<<<1, 1024, 0, h->stream>>>(kReduceBlocks, h->red_part.p, h->f2s.p + 8, h->fmaxflag.p + 8, dst, h->rb.scal, h->scalB.p, dctl);
    } else if (fuse) k_reduce_control<<<1, 1024, 0, h->stream>>>(h->ntiles_main, h->tile_part.p, h->f2s.p + 8, h->fmaxflag.p + 8, dst, h->rb.scal, h->scalB.p, dctl);
    else k_lm_control<<<1, 1, 0, h->stream>>>(dst, h->rb.scal, h->scalB.p, dctl);
    k_lm_accept<<<256, 256, 0, h->stream>>>(dst, h->cam[0].p, h->cam[nxt].p, (size_t)6 * h->nc, h->pts[0].p, h->pts[nxt].p,
                                            (size_t)4 * h->np, h->intr[0].p, h->intr[nxt].p, h->ni ? (size_t)THEIA_MAX_INTRINSICS * h->ng : 0,
                                            h->in_cam.p, h->in_pts.p, h->in_intr.p,
                                            (h->use_fused && h->ni == 0 && !h->inner) ? h->camrot.p : nullptr, h->camrot_cand.p, (size_t)40 * h->nc);
    return 0;
  };
  // Phase timing (HIP events around the kernel groups) is opt-in: THEIA_HIP_PHASE_TIMING=1.
  // Otherwise, and without an all-reduce callback, the body is captured ONCE into a hipGraph
  // and replayed: ~45 launches per iteration cost more host time than the GPU needs to run them.
  const bool timing = getenv("THEIA_HIP_PHASE_TIMING") != nullptr;
  const char* genv = getenv("THEIA_HIP_LM_GRAPH");
  // hipGraph replay is opt-in (THEIA_HIP_LM_GRAPH=1): it measured no faster than direct launches (the iteration
  // is bounded by the dependent kernels, not by host enqueue time), and a stream capture in one host thread makes
  // legacy-stream calls of other threads fail ("would make the legacy stream depend on a capturing blocking
  // stream") -- the entry points must stay callable concurrently from a thread pool.
  const bool want_graph = !timing && !h->allreduce && !inner && !h->graph_failed && (genv && genv[0] == '1');
  if (want_graph && !h->graph_exec) {
    bool ok = hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
    if (ok) {
      const int r = enqueue_body(-1);
      hipGraph_t g = nullptr;
      const hipError_t e = hipStreamEndCapture(h->stream, &g);
      ok = (r == 0) && e == hipSuccess && g != nullptr;
      if (ok) { h->graph = g; ok = hipGraphInstantiate(&h->graph_exec, g, nullptr, nullptr, 0) == hipSuccess; }
      else if (g) (void)hipGraphDestroy(g);
    }
    if (!ok) { h->drop_graph(); h->graph_failed = true; (void)hipGetLastError(); }
  }
  const bool use_graph = want_graph && h->graph_exec;
  // Several bodies are enqueued per synchronisation; bodies after termination are no-ops for
  // the state (k_lm_control returns at once) -- their kernels run on unchanged data.
  const char* chunk_env = getenv("THEIA_HIP_LM_CHUNK");
  int chunk = chunk_env ? atoi(chunk_env) : 4;
  chunk = std::max(1, std::min(chunk, (int)theia_ba_handle_s::kMaxChunk));
  const long long bodies_max = std::max(1, O.max_num_iterations);
  if (bodies_max <= theia_ba_handle_s::kMaxChunk && !chunk_env) chunk = (int)bodies_max;
  long long bodies_enqueued = 0;
  while (true) {
    {
      // max_solver_time_in_seconds: every other LM decision comes from all-reduced device state, so in a sharded solve
      // this one is made collective too (MAX over the ranks' own clocks) -- a rank that stopped alone would leave the
      // others waiting in their next all-reduce.
      double expired = (now_s() - t_start >= O.max_solver_time_in_seconds && bodies_enqueued > 0) ? 1.0 : 0.0;
      if (h->allreduce && bodies_enqueued > 0) {
        if (h->stop_flag.n < 8 && (rc = h->stop_flag.alloc(8))) return rc;
        h->h_scal[32] = expired;
        HIP_TRY(hipMemcpyAsync(h->stop_flag.p, h->h_scal + 32, sizeof(double), hipMemcpyHostToDevice, h->stream));
        if ((rc = do_allreduce(h, h->stop_flag.p, 1, THEIA_REDUCE_MAX))) return rc;
        HIP_TRY(hipMemcpyAsync(h->h_scal + 33, h->stop_flag.p, sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        expired = h->h_scal[33];
      }
      if (expired != 0.0) { st.term = THEIA_TERM_NO_CONVERGENCE; break; }
    }
    const int nb = (int)std::min<long long>(chunk, bodies_max - bodies_enqueued);
    if (nb <= 0) break;
    for (int b = 0; b < nb; ++b) {
      if (use_graph) HIP_TRY(hipGraphLaunch(h->graph_exec, h->stream));
      else if ((rc = enqueue_body(timing ? b : -1))) return rc;
    }
    bodies_enqueued += nb;
    HIP_TRY(hipGetLastError());   // a rejected launch configuration would otherwise go unnoticed
    // read-back through the handle's pinned block: an asynchronous copy needs a peer that outlives the call (HIP may
    // pin pageable pages and run the DMA later; stack temporaries were the round-1 corruption)
    HIP_TRY(hipMemcpyAsync(h->h_state, dst, sizeof(st), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    std::memcpy(&st, h->h_state, sizeof(st));
    const int ran = (int)std::min<long long>(nb, std::max<long long>(0, (long long)st.bodies - (bodies_enqueued - nb)));
    for (int b = 0; timing && b < ran; ++b) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, h->ev[b][0], h->ev[b][1]) == hipSuccess) S->time_linearize += ms * 1e-3;
      if (hipEventElapsedTime(&ms, h->ev[b][1], h->ev[b][2]) == hipSuccess) S->time_solve_reduced += ms * 1e-3;
      if (hipEventElapsedTime(&ms, h->ev[b][2], h->ev[b][3]) == hipSuccess) S->time_backsub += ms * 1e-3;
      if (hipEventElapsedTime(&ms, h->ev[b][4], h->ev[b][5]) == hipSuccess) S->time_kernel_linearize += ms * 1e-3;
      S->num_linearize_launches++;
    }
    if (st.done) break;
  }
  if (ctl.trace_capacity) {
    const int k = std::min(st.trace_size, S->trace_capacity);
    S->trace_size = k;
    if (k) {
      HIP_TRY(hipMemcpy(S->trace_cost, h->tr_cost.p, sizeof(double) * k, hipMemcpyDeviceToHost));
      if (S->trace_gradient_max_norm) HIP_TRY(hipMemcpy(S->trace_gradient_max_norm, h->tr_g.p, sizeof(double) * k, hipMemcpyDeviceToHost));
      if (S->trace_step_norm) HIP_TRY(hipMemcpy(S->trace_step_norm, h->tr_step.p, sizeof(double) * k, hipMemcpyDeviceToHost));
      if (S->trace_radius) HIP_TRY(hipMemcpy(S->trace_radius, h->tr_radius.p, sizeof(double) * k, hipMemcpyDeviceToHost));
      if (S->trace_accepted) HIP_TRY(hipMemcpy(S->trace_accepted, h->tr_acc.p, sizeof(int) * k, hipMemcpyDeviceToHost));
    }
  }
  if (O.verbose)
    for (int k = 0; k < S->trace_size; ++k)
      std::fprintf(stderr, "[theia_hip] %3d cost %.6e |g| %.3e |step| %.3e radius %.3e %s\n", k, S->trace_cost[k],
                   S->trace_gradient_max_norm ? S->trace_gradient_max_norm[k] : 0.0, S->trace_step_norm ? S->trace_step_norm[k] : 0.0,
                   S->trace_radius ? S->trace_radius[k] : 0.0, (S->trace_accepted && S->trace_accepted[k]) ? "ok" : "rej");
  S->num_iterations = st.iter;
  S->num_successful_steps = st.num_successful;
  S->termination_type = st.term;
  S->success = st.term != THEIA_TERM_FAILURE;
  S->initial_cost = st.initial_cost;
  S->final_cost = st.fail_at_first ? st.initial_cost : st.minimum_cost + h->fixed_cost;
  h->stage.release();   // (the stream has been waited for: create()'s staged uploads are done)
  S->solve_time_in_seconds = now_s() - t_start;
  return 0;
}

int theia_hip_ba_solve(const theia_ba_problem* problem, const theia_ba_options* options, theia_ba_summary* summary) {
  if (!summary) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "null summary");
  if (problem && options && (problem->flags & THEIA_BA_FLAG_INVERSE_DEPTH)) {
    const int vrc = validate(problem, options);
    return vrc ? vrc : ba_solve_inverse_depth(problem, options, summary);
  }
  const double t0 = now_s();
  theia_ba_handle h = nullptr;
  int rc = theia_hip_ba_create(problem, options, &h);
  if (rc) return rc;
  const double t1 = now_s();
  rc = theia_hip_ba_run(h, summary);
  const double t2 = now_s();
  if (!rc) rc = theia_hip_ba_download(h, const_cast<theia_ba_problem*>(problem));
  const double t3 = now_s();
  summary->setup_time_in_seconds = t1 - t0;
  theia_hip_ba_destroy(h);
  if (getenv("THEIA_HIP_CREATE_TIMING"))
    fprintf(stderr, "theia_hip ba_solve: create %.2f ms, run %.2f ms, download %.2f ms, destroy %.2f ms\n", 1e3 * (t1 - t0), 1e3 * (t2 - t1),
            1e3 * (t3 - t2), 1e3 * (now_s() - t3));
  return rc;
}

int theia_hip_ba_evaluate(theia_ba_handle h, double* cost, double* residuals, double* jac_cam, double* jac_pt, uint8_t* valid) {
  return theia_hip_ba_evaluate_ex(h, cost, residuals, jac_cam, jac_pt, nullptr, valid);
}

int theia_hip_ba_evaluate_ex(theia_ba_handle h, double* cost, double* residuals, double* jac_cam, double* jac_pt,
                             double* jac_intr, uint8_t* valid) {
  if (h && h->idh) return set_error(THEIA_HIP_ERR_UNSUPPORTED, "inverse-depth handle: evaluate is not built in this mode");
  if (!h) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "null handle");
  release_stage_if_idle(h);
  const int pd = h->pd;
  DevBuf<double> dr, djc, djp, dji; DevBuf<uint8_t> dv;
  int rc;
  const size_t nm = (size_t)h->nobs_main;
  if ((rc = dr.alloc(2 * nm)) || (rc = djc.alloc(12 * nm)) || (rc = djp.alloc(2 * pd * nm)) || (rc = dv.alloc(nm))) return rc;
  const bool want_ji = jac_intr && h->ni;
  if (want_ji && (rc = dji.alloc(20 * nm))) return rc;
  DevProblem Q = h->P;
  Q.scale_c = h->ones_c.p; Q.scale_p = h->ones_p.p;
  Q.ntiles = h->ntiles_eval;
  Q.scale_i = h->ones_i.p; Q.intr = h->intr[h->cur].p;
  HIP_TRY(hipMemsetAsync(h->scalB.p, 0, sizeof(double) * 16, h->stream));
  launch_evaluate(Q, h->cam[h->cur].p, h->pts[h->cur].p, dr.p, djc.p, djp.p, dv.p, h->tile_part.p, h->stream, want_ji ? dji.p : nullptr);
  if (h->ntiles_eval) launch_reduce_tiles(h->ntiles_eval, h->tile_part.p, 2, h->f2s.p + 16, h->fmaxflag.p + 16, h->scalB.p, h->stream);
  launch_cam_priors(Q, PRIOR_COST, h->cam[h->cur].p, nullptr, nullptr, nullptr, nullptr, h->scalB.p + SB_COST, nullptr, h->stream);
  std::vector<double> hr(2 * nm), hjc(12 * nm), hjp(2 * pd * nm); std::vector<uint8_t> hv(nm);
  if (nm) {
    HIP_TRY(hipMemcpyAsync(hr.data(), dr.p, sizeof(double) * hr.size(), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(hjc.data(), djc.p, sizeof(double) * hjc.size(), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(hjp.data(), djp.p, sizeof(double) * hjp.size(), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(hv.data(), dv.p, hv.size(), hipMemcpyDeviceToHost, h->stream));
  }
  std::vector<double> hji;
  if (want_ji && nm) { hji.resize(20 * nm); HIP_TRY(hipMemcpyAsync(hji.data(), dji.p, sizeof(double) * hji.size(), hipMemcpyDeviceToHost, h->stream)); }
  HIP_TRY(hipMemcpyAsync(h->h_scal + 16, h->scalB.p, sizeof(double) * 16, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (cost) *cost = h->h_scal[16 + SB_COST] + h->fixed_cost;
  if (jac_intr) std::fill(jac_intr, jac_intr + 20 * h->nobs, 0.0);
  if (want_ji) for (size_t s2 = 0; s2 < nm; ++s2) std::copy(&hji[20 * s2], &hji[20 * s2] + 20, jac_intr + 20 * h->perm[s2]);
  if (residuals) std::fill(residuals, residuals + 2 * h->nobs, 0.0);
  if (jac_cam) std::fill(jac_cam, jac_cam + 12 * h->nobs, 0.0);
  if (jac_pt) std::fill(jac_pt, jac_pt + 2 * pd * h->nobs, 0.0);
  if (valid) std::fill(valid, valid + h->nobs, (uint8_t)1);
  for (size_t s = 0; s < nm; ++s) {
    const int64_t i = h->perm[s];
    if (residuals) { residuals[2 * i] = hr[2 * s]; residuals[2 * i + 1] = hr[2 * s + 1]; }
    if (jac_cam) std::copy(&hjc[12 * s], &hjc[12 * s] + 12, jac_cam + 12 * i);
    if (jac_pt) std::copy(&hjp[2 * pd * s], &hjp[2 * pd * s] + 2 * pd, jac_pt + 2 * pd * i);
    if (valid) valid[i] = hv[s];
  }
  release_stage_if_idle(h);
  return 0;
}

// Covariance blocks of the two block-diagonal cases the reference exposes (GetCovarianceFor{Track,Tracks,View,Views},
// bundle_adjuster.cc:660-773, behind the *WithCov entry points bundle_adjustment.cc:288-386,420-499): tracks against
// constant cameras (3x3 / 4x4 in the point's tangent space) and views against constant tracks (6x6).  In both the
// normal matrix J'J is block diagonal, so ceres::Covariance's (J'J)^-1 is the inverse of each block.  J is the
// loss-corrected, unscaled Jacobian at the current state (Covariance::Options::apply_loss_function = true).
int theia_hip_ba_covariance(theia_ba_handle h, double* point_cov, double* cam_cov) {
  if (h && h->idh) return set_error(THEIA_HIP_ERR_UNSUPPORTED, "inverse-depth handle: covariance is not built in this mode");
  if (!h) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "null handle");
  release_stage_if_idle(h);
  if (!point_cov && !cam_cov) return 0;
  // Optimised intrinsics couple the cameras of a group: J'J of the views problem is an arrow per group, not block diagonal
  // (handled below by the arrow's Schur complement, linear in the number of cameras).
  // (point covariances on a handle with optimised intrinsics: not a case the reference can produce -- BundleAdjustTrack(s)
  // holds every camera constant and with them the intrinsics groups, bundle_adjuster.cc:204-212 -- refused)
  if (h->ni && point_cov) return set_error(THEIA_HIP_ERR_UNSUPPORTED, "point covariances with optimised intrinsics: BundleAdjustTrack(s) keeps all intrinsics constant (bundle_adjuster.cc:204-212)");
  bool any_var_point = false;
  for (int q = 0; q < h->np; ++q) any_var_point |= !h->pt_const[q];
  if (point_cov && h->ncv > 0)
    return set_error(THEIA_HIP_ERR_UNSUPPORTED, "point covariances need all cameras constant (the BundleAdjustTrack(s) problem)");
  if (cam_cov && any_var_point)
    return set_error(THEIA_HIP_ERR_UNSUPPORTED, "camera covariances need all points constant (the BundleAdjustView(s) problem)");
  if (cam_cov)
    for (int c = 0; c < h->nc; ++c)
      if (h->cam_red[c] >= 0 && h->cam_mask[c]) return set_error(THEIA_HIP_ERR_UNSUPPORTED, "covariance of a partially constant camera");
  const int pd = h->pd, NT = pd * (pd + 1) / 2;
  DevProblem Q = h->P;
  Q.scale_c = h->ones_c.p; Q.scale_p = h->ones_p.p; Q.scale_i = h->ones_i.p; Q.intr = h->intr[h->cur].p;
  Q.camrot_current = 0; h->camrot_valid = false;   // (the per-camera blocks are rebuilt with the unit scales, and are the solve's no longer)
  LmState st;
  std::memset(&st, 0, sizeof(st));
  st.radius = 1e300;   // no LM damping: D = clamp(diag) / radius vanishes against the diagonal
  HIP_TRY(hipMemcpyAsync(h->lm_state.p, &st, sizeof(st), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  const double* radius = &reinterpret_cast<const LmState*>(h->lm_state.p)->radius;
  HIP_TRY(hipMemsetAsync(h->reduce.p, 0, sizeof(double) * h->reduce.n, h->stream));
  if (h->Vinv.n) HIP_TRY(hipMemsetAsync(h->Vinv.p, 0, sizeof(double) * h->Vinv.n, h->stream));
  launch_linearize(Q, h->cam[h->cur].p, h->pts[h->cur].p, radius, h->rb, h->Vinv.p, h->gp.p, h->tile_part.p, h->stream);
  launch_long_linearize(Q, h->cam[h->cur].p, h->pts[h->cur].p, radius, h->rb, h->Vinv.p, h->gp.p, h->long_scratch.p, h->stream);
  if (point_cov) {
    std::vector<double> vi((size_t)NT * h->np);
    if (h->np) HIP_TRY(hipMemcpyAsync(vi.data(), h->Vinv.p, sizeof(double) * vi.size(), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (int q = 0; q < h->np; ++q)
      for (int a = 0; a < pd; ++a)
        for (int b = 0; b < pd; ++b)
          point_cov[(size_t)q * pd * pd + a * pd + b] = vi[(size_t)NT * q + (a >= b ? a * (a + 1) / 2 + b : b * (b + 1) / 2 + a)];
  }
  if (cam_cov) {
    const int n = h->n;
    std::vector<double> S((size_t)n * n);
    if (n) HIP_TRY(hipMemcpyAsync(S.data(), h->rb.S, sizeof(double) * S.size(), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    std::fill(cam_cov, cam_cov + 36 * (size_t)h->nc, 0.0);
    if (h->ni) {
      // Optimised intrinsics: with every point constant J'J is an ARROW per intrinsics group -- [G_g  B^T; B  D], D block diagonal
      // over the group's cameras (6 x 6 each), G_g the group's 10 x 10 block -- and groups do not couple.  The extrinsics block
      // of camera c of (J'J)^-1 is   D_c^-1 + Y_c (G_g - sum_c' B_c'^T D_c'^-1 B_c')^-1 Y_c^T,   Y_c = D_c^-1 B_c :
      // work linear in the number of cameras, no limit on the size of the reduced system (the dense host factorisation of the
      // earlier rounds stopped at 2048 columns).  Slots of parameters outside the optimised subset are empty rows: left out.
      const int ni = h->ni;
      std::vector<double> G((size_t)ni * 10);   // [group][10][10] lower triangle mirrored
      for (int i = 0; i < ni; ++i)
        for (int j = 0; j < 10; ++j) { const int gs = 10 * (i / 10), r = std::max(i, gs + j), q = std::min(i, gs + j); G[(size_t)i * 10 + j] = S[(size_t)r * n + q]; }
      std::vector<int> cg((size_t)std::max(1, h->nc));
      if (h->nc) HIP_TRY(hipMemcpy(cg.data(), h->cam_group.p, sizeof(int) * h->nc, hipMemcpyDeviceToHost));
      auto chol_inv = [](const double* A, int m, double* Ainv) -> bool {   // inverse of an SPD m x m matrix (m <= 10), row-major
        double L[100], Li[100];
        for (int i = 0; i < m; ++i)
          for (int j = 0; j <= i; ++j) {
            double v = A[i * m + j];
            for (int k = 0; k < j; ++k) v -= L[i * m + k] * L[j * m + k];
            if (i == j) { if (!(v > 0.0)) return false; L[i * m + i] = std::sqrt(v); }
            else L[i * m + j] = v / L[j * m + j];
          }
        for (int i = 0; i < m; ++i) {
          for (int j = 0; j < m; ++j) Li[i * m + j] = 0.0;
          Li[i * m + i] = 1.0 / L[i * m + i];
          for (int j = 0; j < i; ++j) {
            double v = 0.0;
            for (int k = j; k < i; ++k) v -= L[i * m + k] * Li[k * m + j];
            Li[i * m + j] = v / L[i * m + i];
          }
        }
        for (int a = 0; a < m; ++a)
          for (int b = 0; b < m; ++b) {
            double v = 0.0;
            for (int k = std::max(a, b); k < m; ++k) v += Li[k * m + a] * Li[k * m + b];
            Ainv[a * m + b] = v;
          }
        return true;
      };
      const int ngv = ni / 10;
      // per camera: D_c^-1 and Y_c = D_c^-1 B_c (6 x 10); per group: the Schur complement onto its intrinsics
      std::vector<double> Dinv((size_t)36 * h->nc, 0.0), Y((size_t)60 * h->nc, 0.0), SG(G);
      for (int c = 0; c < h->nc; ++c) {
        const int rc = h->cam_red[c];
        if (rc < 0) continue;
        const int o = ni + 6 * rc, gr = h->grp_red[cg[c]];
        double D[36];
        for (int a = 0; a < 6; ++a)
          for (int b = 0; b < 6; ++b) D[a * 6 + b] = S[(size_t)(o + std::max(a, b)) * n + o + std::min(a, b)];
        if (!chol_inv(D, 6, &Dinv[(size_t)36 * c])) return set_error(THEIA_HIP_ERR_INTERNAL, "camera %d: J'J is rank deficient (ceres::Covariance::Compute fails)", c);
        if (gr < 0) continue;
        double* Yc = &Y[(size_t)60 * c];
        for (int a = 0; a < 6; ++a)
          for (int k = 0; k < 10; ++k) {
            double v = 0.0;
            for (int b = 0; b < 6; ++b) v += Dinv[(size_t)36 * c + a * 6 + b] * S[(size_t)(o + b) * n + 10 * gr + k];
            Yc[a * 10 + k] = v;
          }
        for (int k = 0; k < 10; ++k)
          for (int l = 0; l < 10; ++l) {
            double v = 0.0;
            for (int a = 0; a < 6; ++a) v += S[(size_t)(o + a) * n + 10 * gr + k] * Yc[a * 10 + l];
            SG[(size_t)(10 * gr + k) * 10 + l] -= v;
          }
      }
      std::vector<double> SGinv((size_t)ni * 10, 0.0);   // [group][10][10], zero rows / columns at the empty slots
      for (int gr = 0; gr < ngv; ++gr) {
        int act[10], na = 0;
        for (int k = 0; k < 10; ++k) if (G[(size_t)(10 * gr + k) * 10 + k] != 0.0) act[na++] = k;
        if (!na) continue;
        double A[100], Ai[100];
        for (int a = 0; a < na; ++a)
          for (int b = 0; b < na; ++b) A[a * na + b] = SG[(size_t)(10 * gr + act[a]) * 10 + act[b]];
        if (!chol_inv(A, na, Ai)) return set_error(THEIA_HIP_ERR_INTERNAL, "J'J is rank deficient in intrinsics group slot %d (ceres::Covariance::Compute fails)", gr);
        for (int a = 0; a < na; ++a)
          for (int b = 0; b < na; ++b) SGinv[(size_t)(10 * gr + act[a]) * 10 + act[b]] = Ai[a * na + b];
      }
      for (int c = 0; c < h->nc; ++c) {
        const int rc = h->cam_red[c];
        if (rc < 0) continue;
        const int gr = h->grp_red[cg[c]];
        for (int a = 0; a < 6; ++a)
          for (int b = 0; b < 6; ++b) {
            double v = Dinv[(size_t)36 * c + a * 6 + b];
            if (gr >= 0) {
              const double* Yc = &Y[(size_t)60 * c];
              for (int k = 0; k < 10; ++k) {
                double t = 0.0;
                for (int l = 0; l < 10; ++l) t += SGinv[(size_t)(10 * gr + k) * 10 + l] * Yc[b * 10 + l];
                v += Yc[a * 10 + k] * t;
              }
            }
            cam_cov[(size_t)c * 36 + a * 6 + b] = v;
          }
      }
      release_stage_if_idle(h);
      return 0;
    }
    for (int c = 0; c < h->nc; ++c) {
      const int rc = h->cam_red[c];
      if (rc < 0) continue;
      double L[36], Li[36];
      bool ok = true;
      for (int i = 0; i < 6 && ok; ++i)
        for (int j = 0; j <= i; ++j) {
          double s = S[(size_t)(6 * rc + i) * n + 6 * rc + j];
          for (int k = 0; k < j; ++k) s -= L[i * 6 + k] * L[j * 6 + k];
          if (i == j) { if (!(s > 0.0)) { ok = false; break; } L[i * 6 + i] = std::sqrt(s); }
          else L[i * 6 + j] = s / L[j * 6 + j];
        }
      if (!ok) return set_error(THEIA_HIP_ERR_INTERNAL, "camera %d: J'J is rank deficient (ceres::Covariance::Compute fails)", c);
      for (int i = 0; i < 6; ++i) {   // Li = L^-1
        for (int j = 0; j < 6; ++j) Li[i * 6 + j] = 0.0;
        Li[i * 6 + i] = 1.0 / L[i * 6 + i];
        for (int j = 0; j < i; ++j) {
          double s = 0.0;
          for (int k = j; k < i; ++k) s -= L[i * 6 + k] * Li[k * 6 + j];
          Li[i * 6 + j] = s / L[i * 6 + i];
        }
      }
      for (int a = 0; a < 6; ++a)
        for (int b = 0; b < 6; ++b) {
          double s = 0.0;
          for (int k = std::max(a, b); k < 6; ++k) s += Li[k * 6 + a] * Li[k * 6 + b];
          cam_cov[(size_t)c * 36 + a * 6 + b] = s;
        }
    }
  }
  release_stage_if_idle(h);
  return 0;
}

int theia_hip_dense_spd_solve(int32_t n, const double* A, const double* b, double* x) {
  if (n < 0 || (n > 0 && (!A || !b || !x))) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "bad argument");
  if (n == 0) return 0;
  int rc = thip::ensure_device();
  if (rc) return rc;
  DevBuf<double> dA, dw, dflag;
  if ((rc = dA.alloc((size_t)n * n + n)) || (rc = dw.alloc(dense_cholesky_workspace(n))) || (rc = dflag.alloc(1))) return rc;
  HIP_TRY(hipMemcpy(dA.p, A, sizeof(double) * (size_t)n * n, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(dA.p + (size_t)n * n, b, sizeof(double) * n, hipMemcpyHostToDevice));
  HIP_TRY(hipMemset(dflag.p, 0, sizeof(double)));
  dense_cholesky_solve(n, dA.p, n, dA.p + (size_t)n * n, dw.p, dflag.p, nullptr);
  double flag = 0.0;
  if (getenv("THEIA_HIP_DEBUG_FACTOR")) {  // development aid: hand back the factor in place of A
    HIP_TRY(hipMemcpy(const_cast<double*>(A), dA.p, sizeof(double) * (size_t)n * n, hipMemcpyDeviceToHost));
  }
  HIP_TRY(hipMemcpy(x, dA.p + (size_t)n * n, sizeof(double) * n, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(&flag, dflag.p, sizeof(double), hipMemcpyDeviceToHost));
  if (flag != 0.0) return set_error(THEIA_HIP_ERR_INTERNAL, "matrix is not positive definite");
  return 0;
}

int theia_hip_ba_reduced_system(theia_ba_handle h, double radius, int32_t* n_out, double* S, double* rhs, int64_t capacity) {
  if (h && h->idh) return set_error(THEIA_HIP_ERR_UNSUPPORTED, "inverse-depth handle: the reduced-system dump is not built in this mode");
  if (!h || !n_out) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "null argument");
  int rc = compute_scale(h);
  if (rc) return rc;
  {
    LmState st;
    std::memset(&st, 0, sizeof(st));
    st.radius = radius;
    HIP_TRY(hipMemcpyAsync(h->lm_state.p, &st, sizeof(st), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));   // `st` is a stack object
  }
  rc = enqueue_linearize(h);
  if (rc) return rc;
  const int n = h->n;
  *n_out = n;
  if ((int64_t)n * n > capacity) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "capacity too small for %d x %d", n, n);
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (n) {
    HIP_TRY(hipMemcpy(S, h->rb.S, sizeof(double) * (size_t)n * n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(rhs, h->rb.rhs, sizeof(double) * n, hipMemcpyDeviceToHost));
  }
  for (int i = 0; i < n; ++i) for (int j = i + 1; j < n; ++j) S[(size_t)i * n + j] = S[(size_t)j * n + i];
  return 0;
}

}  // extern "C"
