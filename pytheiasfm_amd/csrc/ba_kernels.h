// ba_kernels.h -- launch interface of the BA device kernels (internal).
#pragma once
#include <algorithm>
#include <vector>
#include <hip/hip_runtime.h>
#include <cstdint>

namespace thip {

// One workgroup of the fused linearise + Schur kernel (ba_fused.hip): consecutive wave tiles whose tracks see at
// most kFusedMaxCams variable cameras.  Sub-chunks of 4 tiles are linearised into LDS; the pair products
// What_a What_b^T of every track are accumulated in registers by the lane that owns the target block (la, lb).
struct FusedRun {
  int tile0, ntiles;     // wave tiles of the run
  int cam_off, W;        // local camera table: frun_cams[cam_off + lc] = reduced camera index (ascending)
  int tgt_off, ntgt;     // target blocks: frun_tgt[tgt_off + k] = la | lb << 8 (lb <= la)
  int part_off;          // doubles offset of the run's partial sums: [ntgt][36] then [W][6][3]
  int gp;                // G | PS << 8.  G = consumer waves per track slice: 1, 2 or 4 (64 G >= max(ntgt, 6 W));
                         // PS = track slices per consumer wave (only with G == 1): floor(64 / PS) >= ntgt, 6 W <= 64
  int stage_off, nstage; // ba_fused.hip: frun_stage[stage_off + k] = camera index of the k-th per-camera block the run keeps in
                         // LDS: its W local cameras, then the CONSTANT cameras its tracks see (obs_lc = 0x80 | index among them)
};
inline int fused_tiles_per_subchunk(int) { return 4; }   // one wave per tile, 256-thread workgroups
constexpr int kFusedMaxCams = 22;        // -> at most 253 target blocks = one per thread
constexpr int kFusedMaxConst = 22;       // constant cameras a run may see (their blocks are staged in LDS next to the local cameras')
constexpr int kFusedMaxStage = kFusedMaxCams + kFusedMaxConst;
constexpr int kFusedTileTracks = 32;     // tracks per wave tile (128 per sub-chunk)
// fused assembly with intrinsics (ba_fused_intr.hip): compound camera blocks [extrinsics (6) | compact intrinsics rows (4)]
constexpr int kFusedIntrRows = 4, kFusedIntrWidth = 6 + kFusedIntrRows;
constexpr int kFusedMaxCamsIntr = 12;    // -> at most 78 target blocks x 3 lanes
constexpr int kFusedMaxConstIntr = 12;   // constant cameras a run of the compound-block plan may see (staged in LDS as above)
constexpr int kFusedMaxStageIntr = kFusedMaxCamsIntr + kFusedMaxConstIntr;

// Device-resident problem (SoA, observations sorted by point and packed into
// wave tiles of <= 64 observations that never split a point).
struct DevProblem {
  int nc, np, ncv, ntiles;
  int64_t nobs;
  int n;                       // reduced system size = 6 * ncv
  int pd;                      // point tangent dofs: 3 (SphereManifold<4>) or 4
  int loss_type;
  double loss_width;
  const double* intr;          // [ng][10]
  const int* group_model;      // [ng]
  const int* cam_group;        // [nc]
  const int* cam_red;          // [nc] reduced index or -1
  const uint8_t* cam_mask;     // [nc] bit q = extrinsics column q frozen
  const uint8_t* pt_const;     // [np]
  const double2* obs_uv;       // [nobs]
  const double2* obs_si;       // [nobs] or nullptr
  const int* obs_cam;          // [nobs]
  const int* obs_pt;           // [nobs]
  const int* tile_start;       // [ntiles]
  const int* tile_count;       // [ntiles]
  // tracks with > 64 observations (slow path, ba_kernels.hip "long tracks")
  int long_nobs, long_ntracks;
  const int* long_obs_index;   // [long_nobs] index into the sorted observation arrays
  const int* long_obs_slot;    // [long_nobs]
  const int* long_track_start; // [long_ntracks + 1]
  const int* long_track_pt;    // [long_ntracks]
  const double* scale_c;       // [nc][6] Jacobi scaling
  const double* scale_p;       // [np][pd]
  // intrinsics blocks (intrinsics_to_optimize != NONE): they come FIRST in the
  // reduced system, 10 slots per variable group; cameras follow at ni + 6 * rc
  int ni;                      // 10 * (#variable intrinsics groups), 0 = all intrinsics constant
  int ng_total;                // number of intrinsics groups
  const int* grp_k;            // [ng] number of parameters K of the group's model
  const int* grp_red;          // [ng] reduced group index or -1
  const unsigned* grp_free;    // [ng] bit q = parameter q is free
  const unsigned* red_free;    // [#variable groups] the same mask by reduced group index
  int intr_rows;               // intrinsics rows stored per record: 10, or 4 compact rows (k-th free parameter)
  const double* scale_i;       // [ng][10] Jacobi scaling of the intrinsics columns
  const double* intr_cand;     // [ng][10] candidate intrinsics (back-substitution / trial cost)
  const double* scale_red;     // [n] Jacobi scaling by reduced index (finalize)
  // gather-based Schur assembly (k_lin_obs + k_schur_diag + k_schur_blocks), ni == 0:
  // static lists built at create()
  double* rec;                 // per-observation records, camera-major: [#records][6*pd + 14] = {What | F | r} without
                               // intrinsics (What = W Li^T with V^-1 = Li^T Li; gp holds ghat = Li g of slot_pt),
                               // [#records][12*pd + 20 + 2*KI*pd + 3*KI] with intrinsics, KI = intr_rows (ba_kernels.hip)
  const uint8_t* obs_kind;     // [nobs] (sorted order) THEIA_OBS_* or null: depth-prior rows use the pseudo model
  double loss_width_depth;     // robust_loss_width_depth_prior
  const int* slot_pt;          // [#records] point of a record slot (records without intrinsics)
  const int* rec_slot;         // [nobs_main] record slot of a (sorted) observation, -1 = none
  const int* slot_obs;         // [#records] inverse: (sorted) observation of a record slot
  int n_diag_items, n_blk_items;
  const int* diag_items;       // [n_diag_items][4] {rc, first slot, end slot, atomic}
  const int* blk_items;        // [n_blk_items][5] {ri, rj, beg, end, atomic}
  const int2* blk_pairs;       // (slot of the obs of camera ri, slot of the obs of camera rj) of a common track
  const int* pt_sum_slot;      // [np] or null: pseudo-record slot of a track's summed intrinsics fields (k_lin_obs_intr), -1 = none
  const uint8_t* slot_in_sum;  // [#records] or null: the observation's track is summed (its own pair term is in the sum)
  const uint8_t* pt_sum_cnt;   // [np]: pseudo-records (summed groups) of a track, pt_sum_slot[p] .. + cnt - 1
  const int* sum_group;        // [#pseudo-records] reduced group of pseudo-record sum_base + j
  int sum_base;
  // fused linearise + Schur (ba_fused.hip), ni == 0: static plan built at create()
  int n_fruns;
  const FusedRun* fruns;
  const int* frun_order;       // [n_fruns] run indices, most expensive first: the order in which workgroups take runs
  int* frun_next;              // [2] work-queue heads of k_lin_schur / k_backsub_runs (zeroed by k_cam_prep before every launch)
  const int* frun_cams;
  const int* frun_stage;       // camera indices of the per-camera blocks a run stages in LDS (FusedRun::stage_off / nstage)
  const unsigned short* frun_tgt;
  const uint8_t* obs_lc;       // [nobs_main] local camera index inside the run; constant camera: 0x80 | index among the run's
                               // constant cameras (staged in LDS behind the W local ones: slot W + (lc & 0x7f)), for every
                               // block width (ba_fused.hip, ba_fused_intr.hip); tracks outside the runs keep 0xFF
  const uint8_t* obs_tl;       // [nobs_main] track index inside the sub-chunk
  const int* tile_trk_end;     // [ntiles] tracks of the sub-chunk up to and including this tile
  double* fpart;               // per-run partial sums
  double* camrot;              // [nc][kCamRot] per-camera blocks at the linearisation point (k_cam_prep), null = not in use
  double* camrot_cand;         // the same for the candidate cameras of the trial step (back-substitution)
  double* camdir;              // [nc][12] the trial step of each camera as {D (3 x 3), v (3)} (camera_step_direction; k_cam_update)
  int camrot_current;          // host-set per launch: camrot already holds the blocks of `cam` (launch_linearize_fused skips k_cam_prep)
  unsigned model_mask;         // bit m = some intrinsics group uses camera model m (picks the kernel instance)
  int fused_dbg;               // development switches (THEIA_HIP_FUSED_DBG): 1 = skip phase S, 2 = skip phase L arithmetic
  int n_sum_items;
  const int* sum_items;        // [n_sum_items][6] {ri, rj, tbeg, tend, dbeg, dend} into sum_src
  const int* sum_src;          // offsets into fpart
  // fused_bw > 0 (ba_fused_intr.hip): the camera-side blocks are fused_bw wide, a local camera of a run is a camera with a
  // variable extrinsics block OR a variable intrinsics group, sum_items / sum_src are the generic lists of k_sum_items
  // (n_sum_items first-level items followed by n_sum_items2 second-level ones; sources are int pairs)
  int fused_bw, n_sum_items2;
  unsigned fused_kmask;        // the free-parameter mask every variable intrinsics group shares, 0 = they differ
  // camera priors in use (compact list): 3 residuals each on one camera's extrinsics
  int n_priors;
  const int* prior_cam;        // [n_priors] camera index
  const int* prior_kind;       // [n_priors] THEIA_PRIOR_* bit
  const double* prior_vec;     // [n_priors][3]
  const double* prior_info;    // [n_priors][9] sqrt information, row-major
};

// Per-iteration reduced-system workspace: one contiguous buffer so that a
// single memset clears it and a single all-reduce sums it across ranks.
//   [ S (n*n) | rhs (n) | colsq (n) | gc (n) | scalars (16) ]
struct ReduceBuf {
  double* base;
  size_t count;     // doubles
  double* S; double* rhs; double* colsq; double* gc; double* scal;
};
enum {  // indices into ReduceBuf::scal: [0,8) are SUM-reduced, [8,16) MAX-reduced
  SC_COST = 0,      // sum 1/2 rho at x
  SC_INVALID = 1,   // > 0 if any functor returned false
  SC_NOTPD = 2,     // > 0 if a point block / the reduced matrix was not PD
  SC_GMAX = 8,      // max |gradient| (points part; cameras folded in by finalize)
  SC_COUNT = 16
};

// Inner iterations (ba_inner.hip): one sweep of block coordinate descent over the parameter buffers cam / pts / intr
// (cameras, then shared intrinsics, then points), each block by its own LM solve.
struct InnerArgs {
  DevProblem P;                 // topology, masks, loss (P.intr is not used: the intrinsics come from `intr`)
  const int* cam_obs_off;       // [nc + 1] per camera: range of cam_obs_idx
  const int* cam_obs_idx;       // (sorted) observation indices of the camera
  const int* grp_obs_off;       // [ng + 1], [..]: the same per intrinsics group (only with variable intrinsics)
  const int* grp_obs_idx;
  const int* trk_off;           // [ntracks + 1] observation ranges of the tracks in the sorted arrays
  int ntracks;
  int64_t nobs;                 // observations that enter the cost (sorted arrays, fixed residual blocks excluded)
  double* cam; double* pts; double* intr;   // in / out
  const int* gate;              // device flag: 0 = return at once
  int has_si = 0, has_kind = 0;  // set by the launchers: P.obs_si / P.obs_kind are real (otherwise valid stand-ins)
  int own_rank = 0, own_world = 1;   // sharded inner iterations: this launch sweeps the cameras / groups with index % own_world == own_rank
  // the intrinsics sweep with grp_wgs > 1 workgroups per group (co-resident: cooperative launch): per group and pass parity
  // grp_wgs partial sums of kInnerGroupSums doubles in grp_part, arrival counters in grp_bar ([ng] counters | abort flag | [ng] done flags | pad, zeroed before the launch)
  double* grp_part = nullptr; int* grp_bar = nullptr; int grp_wgs = 1; int grp_max_polls = 200000;
  int grp_parts = 1;            // parts a pass over a group's observations is dealt to (= the cooperative launch's workgroups per group; the one-workgroup launch walks them in order)
};
constexpr int kInnerGroupSums = 68;        // 55 (J'J, packed) + 10 (J'r) + cost + invalid count (+ 1 pad)
constexpr int kInnerGroupMaxWgs = 32;
inline int inner_group_wgs(int ng_total) { return std::max(1, std::min(kInnerGroupMaxWgs, 256 / std::max(1, ng_total))); }   // <= 256 co-resident workgroups (one per CU at this kernel's register count: 512 were refused)
void launch_inner_sweep(const InnerArgs& A, hipStream_t st, int stages = 7);   // stages: 1 cameras, 2 intrinsics groups, 4 points
// out[0] = |x0 - x|^2, out[1] = |x|^2 over the variable blocks
void launch_inner_norms(const InnerArgs& A, const double* cam0, const double* pts0, const double* intr0, double* out2, double* part,
                        hipStream_t st, int which = 0);   // part: scratch of 2 * kInnerCostBlocks doubles; which: 0 all blocks, 1 points, 2 cameras + intrinsics
// sharded inner iterations: a shard's candidate points into the (zeroed) global point buffer; the sweep's scalars
void launch_inner_scatter_points(int np, const double* pts, const int* global_index, double* gpts, hipStream_t st);
void launch_inner_combine(const double* reduced4, const double* cam2, double* out4, hipStream_t st);
// blocks of `stride` doubles: keep those with index % world == rank, zero the others (the ranks' buffers are then summed)
void launch_inner_keep_owned(double* x, int nblocks, int stride, int rank, int world, hipStream_t st);
// cost of every residual block at (cam, pts, intr): out[0] = sum rho / 2 (+ camera priors), out[1] > 0 if a functor failed;
// part: scratch of 2 * kInnerCostBlocks doubles
constexpr int kInnerCostBlocks = 512;
void launch_inner_cost(const InnerArgs& A, double* part, double* out2, hipStream_t st);
// both in two launches (the unsharded sweep): out4 = {step^2, |x|^2, cost, invalid}; part: 4 * kInnerCostBlocks doubles
void launch_inner_norms_cost(const InnerArgs& A, const double* cam0, const double* pts0, const double* intr0, double* out4, double* part,
                             hipStream_t st);

void launch_colnorm(const DevProblem& P, const double* cam, const double* pts, double* colsq_c,
                    double* colsq_p, double* colsq_i, hipStream_t st);
void launch_build_scale_red(const DevProblem& P, double* scale_red, hipStream_t st);
void launch_make_scale(int count, const double* colsq, double* scale, hipStream_t st);
void launch_linearize(const DevProblem& P, const double* cam, const double* pts, const double* radius /* device */,
                      const ReduceBuf& rb, double* Vinv, double* gp, double* tile_part,
                      hipStream_t st);
// fused path: k_cam_prep + k_lin_schur + k_schur_sum (S blocks are WRITTEN, the buffer must be clear where nothing lands)
void launch_linearize_fused(const DevProblem& P, const double* cam, const double* pts, const double* radius /* device */,
                            const ReduceBuf& rb, double* Vinv, double* tile_part, hipStream_t st);
void launch_linearize_fused_intr(const DevProblem& P, const double* cam, const double* pts, const double* radius /* device */,
                                 const ReduceBuf& rb, double* Vinv, double* tile_part, hipStream_t st);
// colsq_c[nc][6] / colsq_i[ng][10] += the squared column norms k_sum_items left by reduced index (Jacobi scaling)
void launch_scatter_colsq(const DevProblem& P, const double* colsq_red, double* colsq_c, double* colsq_i, hipStream_t st);
// per-camera blocks (rotation terms, masked scaling, intrinsics) of `cam` -> camrot (ba_fused.hip)
void launch_cam_prep(const DevProblem& P, const double* cam, const double* intr, double* camrot, hipStream_t st, const double* ycam = nullptr);
// K4 + K5 by persistent workgroups over the fused plan's runs (ba_fused.hip: k_backsub_runs); false = not applicable
bool launch_backsub_runs(const DevProblem& P, const double* pts, double* cand_pts, const double* Vinv, double* tile_part, hipStream_t st);
bool launch_backsub_runs_intr(const DevProblem& P, const double* pts, double* cand_pts, const double* Vinv, double* tile_part,
                              const double* y, hipStream_t st);   // the compound-block plan (free intrinsics); y: the solved step
void launch_reduce_tiles(int ntiles, const double* tile_part, int nfields, const int* field_to_scal,
                         const int* field_is_max, double* scal, hipStream_t st, double* red_part = nullptr);
// first stage of a two-stage reduction: kReduceBlocks workgroups fold contiguous slices of the per-tile partials into
// out[kReduceBlocks][nfields] (fixed order); the one-workgroup consumers then reduce kReduceBlocks rows instead of ntiles
constexpr int kReduceBlocks = 128;
void launch_reduce_tiles_stage1(int ntiles, const double* tile_part, int nfields, const int* field_is_max, double* out,
                                hipStream_t st);
// camera priors (ba_priors.h).  mode: PRIOR_COLNORM adds the squared column norms of their (unscaled) Jacobians
// to colsq_c[nc][6]; PRIOR_LINEARIZE adds J'J / J'r / g / column norms / cost to the reduced system at `cam`;
// PRIOR_TRIAL adds the model-cost change of the step y and the cost at `cand_cam` to scal_cost / scal_mcc;
// PRIOR_COST adds the cost at `cam` to *scal_cost; PRIOR_FIXED the cost of the priors of CONSTANT cameras.
enum { PRIOR_COLNORM = 0, PRIOR_LINEARIZE = 1, PRIOR_TRIAL = 2, PRIOR_COST = 3, PRIOR_FIXED = 4 };
void launch_cam_priors(const DevProblem& P, int mode, const double* cam, const double* cand_cam, const double* y,
                       const ReduceBuf* rb, double* colsq_c, double* scal_cost, double* scal_mcc, hipStream_t st);
// tile_part != null: the kernel first folds the per-tile partial sums of the linearisation into rb.scal
// (= launch_reduce_tiles cfg 0), saving that launch; only when no all-reduce sits between the two.
void launch_finalize_rcs(const DevProblem& P, const double* radius /* device */, const ReduceBuf& rb, hipStream_t st,
                         int ntiles = 0, const double* tile_part = nullptr, const int* f2s = nullptr,
                         const int* fmaxflag = nullptr, const uint8_t* tile_cls = nullptr, int want_cls = 0);
void launch_cam_update(const DevProblem& P, const double* cam, const double* y, double* cand_cam,
                       double* cand_intr, double* out_stepsq, double* out_xnormsq, hipStream_t st, double* zero16 = nullptr);
void launch_backsub(const DevProblem& P, const double* cam, const double* pts, const double* cand_cam,
                    double* cand_pts, const double* yc, const double* Vinv, double* tile_part,
                    double* scal, hipStream_t st);
void launch_evaluate(const DevProblem& P, const double* cam, const double* pts, double* residuals,
                     double* jac_cam, double* jac_pt, uint8_t* valid, double* tile_part,
                     hipStream_t st, double* jac_intr = nullptr);
void launch_cost_only(const DevProblem& P, const double* cam, const double* pts, double* tile_part,
                      double* scal, hipStream_t st);

void launch_long_colnorm(const DevProblem& P, const double* cam, const double* pts, double* colsq_c, double* colsq_p,
                         double* scratch, hipStream_t st, double* colsq_i = nullptr);
void launch_long_linearize(const DevProblem& P, const double* cam, const double* pts, const double* radius /* device */, const ReduceBuf& rb,
                           double* Vinv, double* gp, double* scratch, hipStream_t st);
void launch_long_backsub(const DevProblem& P, const double* cam, const double* pts, const double* cand_cam,
                         double* cand_pts, const double* yc, const double* Vinv, double* scratch, double* scalB,
                         hipStream_t st);

// dense SPD solve  A x = b  (lower triangle of row-major A, leading dim lda;
// A is overwritten by its Cholesky factor, b by x).  fail_flag (device) is
// incremented if a pivot is not positive.
size_t dense_cholesky_workspace(int n);  // doubles
void dense_cholesky_solve(int n, double* A, int lda, double* b, double* work, double* fail_flag, hipStream_t st);

// Tile-sparse, nested-dissection ordered, level-scheduled variant of the same
// solve (sparse_cholesky.hip).  adj = symmetric nt x nt (nt = ceil(n/64)) tile
// co-visibility matrix, or nullptr for "dense".  Same workspace as the dense
// solve; falls back to the dense schedule when the structure offers nothing.
struct CholPlan;
CholPlan* chol_plan_create(int n, const uint8_t* adj);
// a rank's plan of a sharded solve (sparse_cholesky.hip): its private tile columns first, the shared ones above them
CholPlan* chol_plan_create_sharded(int n, const uint8_t* adj, const uint8_t* tile_class);
void chol_plan_solve_phase(const CholPlan* plan, int phase, double* A, int lda, double* b, double* work, double* fail_flag, hipStream_t st);
int chol_plan_split_level(const CholPlan* plan);
const std::vector<int>& chol_plan_shared_tiles(const CholPlan* plan);
void chol_plan_destroy(CholPlan* plan);
// zero the tiles of A the plan's assembly / factorisation touch; false = dense plan (the caller clears everything)
// tail / tail_count: a vector of doubles to zero in the same launch (null / 0: none)
bool chol_plan_clear(const CholPlan* plan, double* A, int lda, hipStream_t st, double* tail = nullptr, size_t tail_count = 0);
int chol_plan_levels(const CholPlan* plan);
double chol_plan_flops(const CholPlan* plan);   // FP64 flops of one solve on the plan (n^3 / 3 for the dense schedule)
void chol_plan_solve(const CholPlan* plan, double* A, int lda, double* b, double* work, double* fail_flag, hipStream_t st);

#ifdef __HIPCC__
// Deterministic reduction of per-tile partials into a scalar block by ONE workgroup of 1024 threads:
// field f of tile t at part[t * nfields + f]; result -> scal[f2s[f]] (sum, or max if fmaxflag[f]).
// Every thread folds its tiles for all fields, a wave shuffle tree per field, the 16 wave results are combined
// in wave order.  Ends with a workgroup barrier: the scalars are visible to every thread afterwards.
__device__ __forceinline__ void reduce_tiles_body(int ntiles, const double* __restrict__ part, int nfields,
                                                  const int* __restrict__ f2s, const int* __restrict__ fmaxflag,
                                                  double* __restrict__ scal, double (*sm)[16], bool atomic_max = false) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double acc[8];
  bool ismax[8];
#pragma unroll
  for (int f = 0; f < 8; ++f) { acc[f] = 0.0; ismax[f] = f < nfields && fmaxflag[f] != 0; }
  // four tiles per thread and step, all loads (clamped, unconditional) in flight before the first use;
  // the accumulation order of a thread is unchanged (t, t + 1024, t + 2048, ...)
  for (int t0 = threadIdx.x; t0 < ntiles; t0 += 4096) {
    double v[4][8];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const double* row = part + (size_t)min(t0 + 1024 * u, ntiles - 1) * nfields;
#pragma unroll
      for (int f = 0; f < 8; ++f) v[u][f] = row[min(f, nfields - 1)];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (t0 + 1024 * u >= ntiles) break;
#pragma unroll
      for (int f = 0; f < 8; ++f)
        if (f < nfields) acc[f] = ismax[f] ? fmax(acc[f], v[u][f]) : acc[f] + v[u][f];
    }
  }
#pragma unroll
  for (int f = 0; f < 8; ++f) {
    if (f >= nfields) break;
    double v = acc[f];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_xor(v, off, 64); v = ismax[f] ? fmax(v, o) : v + o; }
    if (lane == 0) sm[f][wv] = v;
  }
  __syncthreads();
  if (threadIdx.x < (unsigned)nfields) {
    const int f = threadIdx.x;
    double v = sm[f][0];
    for (int w = 1; w < 16; ++w) v = ismax[f] ? fmax(v, sm[f][w]) : v + sm[f][w];
    // atomic_max: other workgroups of the same launch fold their own maxima into the same scalar (non-negative values: the
    // IEEE bit patterns order like the numbers; a maximum does not depend on the order of its operands)
    if (ismax[f] && atomic_max) { if (v == v) atomicMax(reinterpret_cast<unsigned long long*>(&scal[f2s[f]]), (unsigned long long)__double_as_longlong(v)); }
    else if (ismax[f]) scal[f2s[f]] = fmax(scal[f2s[f]], v);
    else scal[f2s[f]] += v;
  }
  __threadfence_block();
  __syncthreads();
}
#endif

}  // namespace thip
