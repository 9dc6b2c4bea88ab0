// ba_kernels.hip -- BA hot-path kernels for gfx950 (CDNA4), FP64.
//
// Layout / mapping (DESIGN.md "BA kernels"):
//  * observations are sorted by track and packed into WAVE TILES: one 64-lane
//    wavefront owns <= 64 consecutive observations that never split a track;
//    lane = observation.  All per-track sums (V_p = sum Jp^T Jp, g_p, the
//    back-substitution sum) are wave-level segmented reductions done with
//    cross-lane shuffles in a FIXED order (bitwise reproducible);
//  * the per-track Schur products  S_ij -= W_i V^-1 W_j^T  stage the W blocks of
//    the tile in LDS (64 x 6 x pd doubles per wave) and every lane walks its own
//    track's segment;
//  * Jacobians are never written to HBM: the linearize/Schur kernel and the
//    back-substitution kernel both recompute them from the 24 B/observation
//    SoA stream (FP64 flops are free relative to HBM/atomic traffic here).
//
// Reference arithmetic restated: src/theia/sfm/camera/reprojection_error.h:54-110
// (+ camera models), Ceres SchurEliminator semantics for the 3-group ordering of
// bundle_adjuster.cc:547-577, Ceres LM diagonal (levenberg_marquardt_strategy.cc).
#define THIP_LEAN_SQRT 1   // ba_device.h: fsqrt() without range scaling / class selects
#define THIP_LEAN_SINCOS 1 // ba_device.h: SphereManifold::Plus through polynomials for small steps
#include "ba_kernels.h"
#include "ba_priors.h"

#include "ba_lane.h"

namespace thip {

namespace {

// ---------------------------------------------------------------- colnorm
// Squared column norms of the unscaled Jacobian at the initial point: the
// Jacobi scaling 1/(1+sqrt(.)) is computed once from them
// (ceres trust_region_minimizer.cc, jacobi_scaling = true).
// ROT: the cameras as the per-camera blocks of k_cam_prep (P.camrot, written with the unit scales of this pass): no sin / cos and
// no camera -> group -> intrinsics chain per observation
template <int PD, bool INTR, bool ROT = false>
__global__ __launch_bounds__(kBlock) void k_colnorm(DevProblem P, const double* __restrict__ cam,
                                                    const double* __restrict__ pts,
                                                    double* __restrict__ colsq_c,
                                                    double* __restrict__ colsq_p,
                                                    double* __restrict__ colsq_i) {
  const int lane = threadIdx.x & 63;
  const int tile = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (tile >= P.ntiles) return;
  const int cnt = P.tile_count[tile];
  const int start = P.tile_start[tile];
  LaneLin<PD, INTR> L;
  if constexpr (ROT) lane_linearize<PD, true, INTR, true>(P, P.camrot, pts, start + lane, lane < cnt, lane, L);
  else lane_linearize<PD, true, INTR>(P, cam, pts, start + lane, lane < cnt, lane, L);
  const Segment sg = lane_segment(L.p, lane);
  // camera / intrinsics columns: k_colnorm_gather, or the fused kernel itself (compute_scale), no atomics
  const bool gather = P.slot_obs != nullptr || P.fused_bw > 0;
  if constexpr (INTR) {
    if (!gather && L.active && L.gr >= 0) {
#pragma unroll
      for (int q = 0; q < THEIA_MAX_INTRINSICS; ++q) {
        const double v = L.Jk[q] * L.Jk[q] + L.Jk[THEIA_MAX_INTRINSICS + q] * L.Jk[THEIA_MAX_INTRINSICS + q];
        if (v != 0.0) atomic_add(&colsq_i[(size_t)L.g * THEIA_MAX_INTRINSICS + q], v);
      }
    }
  }
  double in[PD], out[PD];
#pragma unroll
  for (int q = 0; q < PD; ++q) in[q] = L.Jt[q] * L.Jt[q] + L.Jt[PD + q] * L.Jt[PD + q];
  segment_allsum<PD>(sg, in, out);
  if (L.active && sg.head) {
#pragma unroll
    for (int q = 0; q < PD; ++q) colsq_p[(size_t)PD * L.p + q] = out[q];
  }
  if (!gather && L.active && L.rc >= 0) {
#pragma unroll
    for (int q = 0; q < 6; ++q) atomic_add(&colsq_c[6 * L.c + q], L.Jc[q] * L.Jc[q] + L.Jc[6 + q] * L.Jc[6 + q]);
  }
}

// Column norms of the camera (and intrinsics) blocks by the gather lists: one workgroup per contiguous
// record range of a camera / group (the diagonal items of the Schur assembly); the lanes re-linearise the
// range's observations and the workgroup sums 6 (10) squared column norms.  Replaces 6 global FP64 atomics
// per observation onto a few hundred addresses.
template <int PD, bool INTR, bool ROT = false>
__global__ __launch_bounds__(kBlock) void k_colnorm_gather(DevProblem P, const double* __restrict__ cam,
                                                           const double* __restrict__ pts,
                                                           double* __restrict__ colsq_c, double* __restrict__ colsq_i) {
  __shared__ double part[kWavesPerBlock][16];
  __shared__ int s_idx;
  int beg, end, atomic, group_item = 0;
  if constexpr (INTR) {
    const int* it = P.blk_items + 6 * blockIdx.x;
    if (it[0] != IT_CD && it[0] != IT_GV) return;
    group_item = it[0] == IT_GV;
    beg = it[3]; end = it[4]; atomic = it[5] & ITF_ATOMIC;
  } else {
    if ((int)blockIdx.x >= P.n_diag_items) return;
    const int* it = P.diag_items + 4 * blockIdx.x;
    beg = it[1]; end = it[2]; atomic = it[3];
  }
  constexpr int NV = INTR ? 10 : 6;
  double acc[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) acc[k] = 0.0;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int q = beg + threadIdx.x; q < end; q += kBlock) {
    LaneLin<PD, INTR> L;
    if constexpr (ROT) lane_linearize<PD, true, INTR, true>(P, P.camrot, pts, P.slot_obs[q], true, lane, L);
    else lane_linearize<PD, true, INTR>(P, cam, pts, P.slot_obs[q], true, lane, L);
    if (q == beg) s_idx = group_item ? L.g : L.c;
    if (group_item) {
      if constexpr (INTR) {
#pragma unroll
        for (int k = 0; k < 10; ++k) acc[k] += L.Jk[k] * L.Jk[k] + L.Jk[10 + k] * L.Jk[10 + k];
      }
    } else {
#pragma unroll
      for (int k = 0; k < 6; ++k) acc[k] += L.Jc[k] * L.Jc[k] + L.Jc[6 + k] * L.Jc[6 + k];
    }
  }
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const double v = wave_sum(acc[k]);
    if (lane == 0) part[wv][k] = v;
  }
  __syncthreads();
  const int nv = group_item ? 10 : 6;
  if ((int)threadIdx.x < nv) {
    const double v = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
    double* dst = group_item ? &colsq_i[(size_t)s_idx * THEIA_MAX_INTRINSICS + threadIdx.x] : &colsq_c[6 * (size_t)s_idx + threadIdx.x];
    if (atomic) atomic_add(dst, v); else *dst = v;
  }
}

__global__ void k_make_scale(int count, const double* __restrict__ colsq, double* __restrict__ scale) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) scale[i] = 1.0 / (1.0 + sqrt(colsq[i]));
}

// -------------------------------------------------- linearize + Schur (K1+K2)
// tile_part layout: [ntiles][4] = {cost, gmax_points, invalid, notpd}
// The first versions scattered the Schur blocks with FP64 atomics (global, then an
// LDS window per workgroup): ds_add_f64 costs ~200 cycles per wave instruction and
// the window left one workgroup per CU.  The problem topology is static, so the
// scatter is a gather:
//   k_lin_obs     : per wave tile: linearise, V_p / g_p by segmented sums,
//                   V_p^-1; writes per-point Vinv, g_p and a per-observation
//                   record {W = F^T E (6 x PD) | F (2 x 6) | r (2)}.  No LDS
//                   accumulators -> full occupancy.
//   k_schur_diag  : one wave per reduced camera walks the camera's observation
//                   list: U_c - sum T W^T, rhs, g_c, column norms.
//   k_schur_blocks: one wave per off-diagonal block (i, j) walks the host-built
//                   list of (obs of i, obs of j) pairs of common tracks:
//                   S_ij = - sum T_a W_b^T, T_a = W_a V^-1.
// Lanes accumulate in registers and a shuffle reduce-scatter leaves element e
// of the block in lane(e): no atomics (except for lists split into chunks),
// fixed summation order, each S entry written once.
// record of one observation: {What = F^T E Li^T (6 x PD) | F (2 x 6) | r (2)} = 32 doubles (two 128-B lines) for
// PD = 3, where V_p^-1 = Li^T Li (Li = inverse of the Cholesky factor of the damped point block).  Everything the
// Schur complement needs is a product of two such rows:  W_a V^-1 W_b^T = What_a What_b^T  and
// W V^-1 g_p = What ghat_p with ghat_p = Li g_p (stored per point in `gp`) -- no per-pair V^-1 gather, no T.
template <int PD> constexpr int rec_stride() { return 6 * PD + 14; }

template <int PD>
__global__ __launch_bounds__(kBlock) void k_lin_obs(DevProblem P, const double* __restrict__ cam,
                                                    const double* __restrict__ pts, const double* __restrict__ radius_p,
                                                    double* __restrict__ Vinv, double* __restrict__ gp,
                                                    double* __restrict__ tile_part) {
  constexpr int NT = PD * (PD + 1) / 2;
  constexpr int NW = 6 * PD;
  constexpr int RS = rec_stride<PD>();
  const double radius = *radius_p;   // device-resident: the LM step control runs on the GPU
  const int lane = threadIdx.x & 63;
  const int tile = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const bool tile_ok = tile < P.ntiles;
  const int cnt = tile_ok ? P.tile_count[tile] : 0;
  const int start = tile_ok ? P.tile_start[tile] : 0;
  LaneLin<PD> L;
  lane_linearize<PD, true>(P, cam, pts, start + lane, lane < cnt, lane, L);
  const int slot = (lane < cnt) ? P.rec_slot[start + lane] : -1;
  const Segment sg = lane_segment(L.p, lane);
  double in[NT + PD], tot[NT + PD];
#pragma unroll
  for (int a = 0; a < PD; ++a) {
#pragma unroll
    for (int b = 0; b <= a; ++b) in[lidx(a, b)] = L.Jt[a] * L.Jt[b] + L.Jt[PD + a] * L.Jt[PD + b];
    in[NT + a] = L.Jt[a] * L.r[0] + L.Jt[PD + a] * L.r[1];
  }
  segment_allsum<NT + PD>(sg, in, tot);
  // every lane of the track inverts the same V (lock-step anyway), no broadcast needed
  double V[NT], Vi[NT], g[PD];
#pragma unroll
  for (int k = 0; k < NT; ++k) V[k] = tot[k];
#pragma unroll
  for (int a = 0; a < PD; ++a) { g[a] = tot[NT + a]; V[lidx(a, a)] += fmin(fmax(V[lidx(a, a)], 1e-6), 1e32) / radius; }
  bool pd_ok = true;
  double Li[PD][PD];
#pragma unroll
  for (int a = 0; a < PD; ++a)
#pragma unroll
    for (int b = 0; b < PD; ++b) Li[a][b] = 0.0;
  if (L.active && !L.pconst) pd_ok = invert_spd<PD>(V, Vi, Li);
  if (!L.active || L.pconst || !pd_ok) {
#pragma unroll
    for (int k = 0; k < NT; ++k) Vi[k] = 0.0;
#pragma unroll
    for (int a = 0; a < PD; ++a)
#pragma unroll
      for (int b = 0; b < PD; ++b) Li[a][b] = 0.0;
  }
  double gmax = 0.0;
  if (L.active && sg.head && !L.pconst) {
#pragma unroll
    for (int k = 0; k < NT; ++k) Vinv[(size_t)NT * L.p + k] = Vi[k];
#pragma unroll
    for (int a = 0; a < PD; ++a) {
      double gh = 0.0;   // ghat = Li g (the Schur readers' side of W V^-1 g)
#pragma unroll
      for (int k = 0; k <= a; ++k) gh += Li[a][k] * g[k];
      gp[(size_t)PD * L.p + a] = gh;
      gmax = fmax(gmax, fabs(g[a] / P.scale_p[(size_t)PD * L.p + a]));
    }
  }
  if (slot >= 0) {
    double2* R = reinterpret_cast<double2*>(P.rec + (size_t)slot * RS);
    double w0[NW], w[NW];
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
      for (int b = 0; b < PD; ++b) w0[a * PD + b] = L.Jc[a] * L.Jt[b] + L.Jc[6 + a] * L.Jt[PD + b];
#pragma unroll
    for (int a = 0; a < 6; ++a)   // What = W Li^T
#pragma unroll
      for (int b = 0; b < PD; ++b) {
        double sacc = 0.0;
#pragma unroll
        for (int k = 0; k <= b; ++k) sacc += w0[a * PD + k] * Li[b][k];
        w[a * PD + b] = sacc;
      }
#pragma unroll
    for (int k = 0; k < NW / 2; ++k) R[k] = make_double2(w[2 * k], w[2 * k + 1]);
#pragma unroll
    for (int k = 0; k < 6; ++k) R[NW / 2 + k] = make_double2(L.Jc[2 * k], L.Jc[2 * k + 1]);
    R[NW / 2 + 6] = make_double2(L.r[0], L.r[1]);
  }
  const double cost = wave_sum(L.cost);
  gmax = wave_max(gmax);
  const double inval = wave_sum((L.active && !L.valid) ? 1.0 : 0.0);
  const double npd = wave_sum((L.active && !pd_ok && sg.head) ? 1.0 : 0.0);
  if (lane == 0 && tile_ok) {
    tile_part[4 * (size_t)tile + 0] = cost;
    tile_part[4 * (size_t)tile + 1] = gmax;
    tile_part[4 * (size_t)tile + 2] = inval;
    tile_part[4 * (size_t)tile + 3] = npd;
  }
}

// Sum v[0..N) over the 64 lanes and scatter the totals: after the call the
// lane with cnt > 0 holds the total of element `lo` in the return value.
// Each stage halves the values a lane carries (xor-shuffle exchange).
template <int N, int OFF>
THIP_DEV double wave_reduce_scatter(double (&v)[N], int lane, int& lo, int& cnt) {
  if constexpr (OFF == 0) {
    return v[0];
  } else {
    constexpr int H = (N + 1) / 2;
    const bool up = (lane & OFF) != 0;
    double w[H];
#pragma unroll
    for (int k = 0; k < H; ++k) {
      const double a = v[k];
      const double b = (k + H < N) ? v[(k + H < N) ? k + H : 0] : 0.0;
      const double send = up ? a : b;
      const double keep = up ? b : a;
      w[k] = keep + __shfl_xor(send, OFF, kWave);
    }
    lo += up ? H : 0;
    cnt = up ? cnt - H : min(cnt, H);
    return wave_reduce_scatter<H, OFF / 2>(w, lane, lo, cnt);
  }
}

// the four waves of a workgroup share one list; combine their totals in wave order
template <int N>
THIP_DEV double block_combine(double (*part)[40], double tot, int lo, int cnt, int wv, int tid) {
  if (cnt > 0) part[wv][lo] = tot;
  __syncthreads();
  return (tid < N) ? ((part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid])) : 0.0;
}

template <int PD, int K>
THIP_DEV void load_rec(const double* __restrict__ rec, int slot, int off2, double (&w)[K]) {
  const double2* R = reinterpret_cast<const double2*>(rec + (size_t)slot * rec_stride<PD>()) + off2;
#pragma unroll
  for (int k = 0; k < K / 2; ++k) { const double2 t = R[k]; w[2 * k] = t.x; w[2 * k + 1] = t.y; }
}

// one workgroup per (camera, chunk of its contiguous records)
template <int PD>
THIP_DEV void schur_diag_item(const DevProblem& P, int item, double (*part)[40], double* __restrict__ S,
                              double* __restrict__ rhs, double* __restrict__ colsq, double* __restrict__ gc,
                              const double* __restrict__ gp) {
  constexpr int NW = 6 * PD;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int* it = P.diag_items + 4 * item;
  const int rc = it[0], beg = it[1], end = it[2], atomic = it[3];
  double acc[39];
#pragma unroll
  for (int k = 0; k < 39; ++k) acc[k] = 0.0;
  for (int q = beg + tid; q < end; q += kBlock) {
    double W[NW], Jc[12], rt[8];
    const int p = P.slot_pt[q];
    load_rec<PD>(P.rec, q, 0, W);          // What
    load_rec<PD>(P.rec, q, NW / 2, Jc);
    {
      const double2 r2 = reinterpret_cast<const double2*>(P.rec + (size_t)q * rec_stride<PD>())[NW / 2 + 6];
      rt[0] = r2.x; rt[1] = r2.y;
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) {   // W V^-1 g_p = What ghat_p
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < PD; ++k) s += W[a * PD + k] * gp[(size_t)PD * p + k];
      rt[2 + a] = s;
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) {
#pragma unroll
      for (int b = 0; b <= a; ++b) {
        double s = Jc[a] * Jc[b] + Jc[6 + a] * Jc[6 + b];
#pragma unroll
        for (int k = 0; k < PD; ++k) s -= W[a * PD + k] * W[b * PD + k];
        acc[lidx(a, b)] += s;
      }
      const double jr = Jc[a] * rt[0] + Jc[6 + a] * rt[1];
      acc[21 + a] += jr - rt[2 + a];
      acc[27 + a] += jr;
      acc[33 + a] += Jc[a] * Jc[a] + Jc[6 + a] * Jc[6 + a];
    }
  }
  int lo = 0, cnt = 39;
  double tot = wave_reduce_scatter<39, 32>(acc, lane, lo, cnt);
  tot = block_combine<39>(part, tot, lo, cnt, wv, tid);
  if (tid >= 39) return;
  double* dst;
  if (tid < 21) {
    int a = 0;
    while ((a + 1) * (a + 2) / 2 <= tid) ++a;
    const int b = tid - a * (a + 1) / 2;
    dst = S + (size_t)(6 * rc + a) * P.n + 6 * rc + b;
  } else {
    dst = (tid < 27 ? rhs : (tid < 33 ? gc : colsq)) + 6 * rc + (tid - 21) % 6;
  }
  if (atomic) atomic_add(dst, tot); else *dst = tot;
}

// one workgroup per (block (ri, rj), chunk of its pair list)
template <int PD>
THIP_DEV void schur_block_item(const DevProblem& P, int item, double (*part)[40], double* __restrict__ S) {
  constexpr int NW = 6 * PD;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int* it = P.blk_items + 5 * item;
  const int ri = it[0], rj = it[1], beg = it[2], end = it[3], atomic = it[4];
  double acc[36];
#pragma unroll
  for (int k = 0; k < 36; ++k) acc[k] = 0.0;
  for (int q = beg + tid; q < end; q += kBlock) {
    const int2 ab = P.blk_pairs[q];
    double T[NW], Wb[NW];                  // What_a, What_b
    load_rec<PD>(P.rec, ab.x, 0, T);
    load_rec<PD>(P.rec, ab.y, 0, Wb);
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
      for (int b = 0; b < 6; ++b) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < PD; ++k) s += T[a * PD + k] * Wb[b * PD + k];
        acc[a * 6 + b] += s;
      }
  }
  int lo = 0, cnt = 36;
  double tot = wave_reduce_scatter<36, 32>(acc, lane, lo, cnt);
  tot = block_combine<36>(part, tot, lo, cnt, wv, tid);
  if (tid >= 36) return;
  const int a = tid / 6, b = tid % 6;
  if (ri == rj && b > a) return;   // two observations of one camera in a track: lower part of the diagonal block
  double* dst = S + (size_t)(6 * ri + a) * P.n + 6 * rj + b;
  if (atomic) atomic_add(dst, -tot); else *dst = -tot;
}

// K2: the camera (diagonal) items and the camera-pair (block) items in ONE launch --
// they are independent, and at C2 size every launch costs as much as it computes.
template <int PD>
__global__ __launch_bounds__(kBlock) void k_schur(DevProblem P, double* __restrict__ S, double* __restrict__ rhs,
                                                  double* __restrict__ colsq, double* __restrict__ gc,
                                                  const double* __restrict__ gp) {
  __shared__ double part[kWavesPerBlock][40];
  const int b = blockIdx.x;
  if (b < P.n_blk_items) schur_block_item<PD>(P, b, part, S);        // the long lists first
  else schur_diag_item<PD>(P, b - P.n_blk_items, part, S, rhs, colsq, gc, gp);
}

// ------------------------------------- gather-based Schur assembly with intrinsics (A10)
// The camera-side block of an observation is 16 wide: [intrinsics of its group (10) | extrinsics (6)].
// Per-observation record (INTR):
//   {W (6xPD) | T (6xPD) | F (2x6) | r (2) | T g (6) | WI (10xPD) | TI (10xPD) | Fk (2x10) | TI g (10)}
// with WI = Fk^T E, TI = WI V^-1.  Records are stored (group, camera)-major, so a camera's and a
// group's records are contiguous.  One workgroup per item; item = {type, row0, col0, beg, end, flags}:
//   pair items walk a list of (slot a, slot b) of a common track and accumulate  - TA_a WB_b^T :
//     CC  6x6   T_a  W_b^T   -> S[cam_a , cam_b ]      GG0/GG1  5x10  TI_a[rows] WI_b^T -> S[grp_a, grp_b]
//     CG  6x10  T_a  WI_b^T  -> S[cam_a , grp_b ]
//   diagonal items stream a contiguous slot range (per-observation terms):
//     CD   camera block, rhs, g, column norms (as without intrinsics)
//     CGD  F^T Fk - T WI^T            -> S[cam, grp of cam]
//     GD0/GD1  Fk^T Fk - TI WI^T (rows)-> S[grp, grp]   GV  Fk^T r - TI g | Fk^T r | column norms
// Reduced index: intrinsics slots first (10 per variable group), cameras at ni + 6 rc.

// KI = stored intrinsics rows per record: 10 (row = parameter index) or, when no variable group frees more than four
// parameters (the pipelines' FOCAL_LENGTH | RADIAL_DISTORTION frees three), 4 COMPACT rows: row k = the k-th free
// parameter of the observation's group (zero beyond the group's count).  The pair passes stream these rows, so the
// compact form cuts their traffic and FMAs by 2.5x.
template <int PD, int KI> constexpr int rec_stride_intr() { return 12 * PD + 20 + 2 * KI * PD + 3 * KI; }
template <int PD, int KI> struct RecI {   // offsets in double2 units
  static constexpr int W = 0, T = 3 * PD, F = 6 * PD, R = 6 * PD + 6, TG = 6 * PD + 7, WI = 6 * PD + 10,
                       TI = WI + KI * PD / 2, FK = TI + KI * PD / 2, TIG = FK + KI;
};
// parameter index (0..9) of stored row k of a group with free mask fm, or -1
template <int KI>
THIP_DEV int row_param(unsigned fm, int k) {
  if (KI == THEIA_MAX_INTRINSICS) return k;
  for (int j = 0; j < k; ++j) fm &= fm - 1u;
  return fm ? __ffs(fm) - 1 : -1;
}

template <int PD, int KI, int K>
THIP_DEV void load_reci(const double* __restrict__ rec, int slot, int off2, double (&w)[K]) {
  const double2* R = reinterpret_cast<const double2*>(rec + (size_t)slot * rec_stride_intr<PD, KI>()) + off2;
#pragma unroll
  for (int k = 0; k < K / 2; ++k) { const double2 t = R[k]; w[2 * k] = t.x; w[2 * k + 1] = t.y; }
}

// reduce NACC per-thread accumulators over the workgroup and write element e -> dst(e)
template <int NACC, typename FDST>
THIP_DEV void finish_item(double (&acc)[NACC], double (*part)[64], bool atomic, double sign, FDST dst_of) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  int lo = 0, cnt = NACC;
  const double tot = wave_reduce_scatter<NACC, 32>(acc, lane, lo, cnt);
  if (cnt > 0) part[wv][lo] = tot;
  __syncthreads();
  if (tid >= NACC) return;
  const double v = sign * ((part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]));
  double* dst = dst_of(tid);
  if (!dst) return;
  if (atomic) atomic_add(dst, v); else *dst = v;
}

// pair item: acc[RA x RB] += TA_a[ra0 + i][:] . WB_b[j][:]; GA / GB: the row / column side is an intrinsics block
// (stored row -> parameter index through the group's free mask), ra0 = first stored row of the item
template <int PD, int KI, int RA, int RB, bool GA, bool GB>
THIP_DEV void pair_item(const DevProblem& P, const int* it, int offA2, int offB2, int ra0, double (*part)[64], double* __restrict__ S) {
  const int row0 = it[1], col0 = it[2], beg = it[3], end = it[4], flags = it[5];
  double acc[RA * RB];
#pragma unroll
  for (int k = 0; k < RA * RB; ++k) acc[k] = 0.0;
  for (int q = beg + threadIdx.x; q < end; q += kBlock) {
    const int2 ab = P.blk_pairs[q];
    double TA[RA * PD], WB[RB * PD];
    load_reci<PD, KI>(P.rec, ab.x, offA2, TA);
    load_reci<PD, KI>(P.rec, ab.y, offB2, WB);
#pragma unroll
    for (int a = 0; a < RA; ++a)
#pragma unroll
      for (int b = 0; b < RB; ++b) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < PD; ++k) s += TA[a * PD + k] * WB[b * PD + k];
        acc[a * RB + b] += s;
      }
  }
  const int n = P.n;
  const bool lower = (flags & ITF_LOWER) != 0;
  const unsigned fma_ = GA ? P.red_free[row0 / THEIA_MAX_INTRINSICS] : 0u, fmb_ = GB ? P.red_free[col0 / THEIA_MAX_INTRINSICS] : 0u;
  finish_item<RA * RB>(acc, part, (flags & ITF_ATOMIC) != 0, -1.0, [&](int e) -> double* {
    int a = e / RB, b = e % RB;
    if (GA) { a = row_param<KI>(fma_, ra0 + a); if (a < 0) return nullptr; }
    if (GB) { b = row_param<KI>(fmb_, b); if (b < 0) return nullptr; }
    if (lower && col0 + b > row0 + a) return nullptr;
    return S + (size_t)(row0 + a) * n + col0 + b;
  });
}

// the same pass with the intrinsics part of the camera-side block (INTR records, see k_schur_intr)
template <int PD, int KI>
__global__ __launch_bounds__(kBlock) void k_lin_obs_intr(DevProblem P, const double* __restrict__ cam,
                                                    const double* __restrict__ pts, const double* __restrict__ radius_p,
                                                    double* __restrict__ Vinv, double* __restrict__ gp,
                                                    double* __restrict__ tile_part) {
  constexpr int NT = PD * (PD + 1) / 2;
  constexpr int NW = 6 * PD;
  constexpr int RS = rec_stride_intr<PD, KI>();
  const double radius = *radius_p;   // device-resident: the LM step control runs on the GPU
  const int lane = threadIdx.x & 63;
  const int tile = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const bool tile_ok = tile < P.ntiles;
  const int cnt = tile_ok ? P.tile_count[tile] : 0;
  const int start = tile_ok ? P.tile_start[tile] : 0;
  LaneLin<PD, true> L;
  lane_linearize<PD, true, true>(P, cam, pts, start + lane, lane < cnt, lane, L);
  const int slot = (lane < cnt) ? P.rec_slot[start + lane] : -1;
  const Segment sg = lane_segment(L.p, lane);
  double in[NT + PD], tot[NT + PD];
#pragma unroll
  for (int a = 0; a < PD; ++a) {
#pragma unroll
    for (int b = 0; b <= a; ++b) in[lidx(a, b)] = L.Jt[a] * L.Jt[b] + L.Jt[PD + a] * L.Jt[PD + b];
    in[NT + a] = L.Jt[a] * L.r[0] + L.Jt[PD + a] * L.r[1];
  }
  segment_allsum<NT + PD>(sg, in, tot);
  // every lane of the track inverts the same V (lock-step anyway), no broadcast needed
  double V[NT], Vi[NT], g[PD];
#pragma unroll
  for (int k = 0; k < NT; ++k) V[k] = tot[k];
#pragma unroll
  for (int a = 0; a < PD; ++a) { g[a] = tot[NT + a]; V[lidx(a, a)] += fmin(fmax(V[lidx(a, a)], 1e-6), 1e32) / radius; }
  bool pd_ok = true;
  if (L.active && !L.pconst) pd_ok = invert_spd<PD>(V, Vi);
  if (!L.active || L.pconst || !pd_ok) {
#pragma unroll
    for (int k = 0; k < NT; ++k) Vi[k] = 0.0;
  }
  double gmax = 0.0;
  if (L.active && sg.head && !L.pconst) {
#pragma unroll
    for (int k = 0; k < NT; ++k) Vinv[(size_t)NT * L.p + k] = Vi[k];
#pragma unroll
    for (int a = 0; a < PD; ++a) {
      gp[(size_t)PD * L.p + a] = g[a];
      gmax = fmax(gmax, fabs(g[a] / P.scale_p[(size_t)PD * L.p + a]));
    }
  }
  double wi[KI * PD];   // WI = Fk^T E of the lane's observation (zero without a record or a variable group)
#pragma unroll
  for (int k = 0; k < KI * PD; ++k) wi[k] = 0.0;
  if (slot >= 0) {
    double2* R = reinterpret_cast<double2*>(P.rec + (size_t)slot * RS);
    double w[NW], t[NW], tg[6];
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
      for (int b = 0; b < PD; ++b) w[a * PD + b] = L.Jc[a] * L.Jt[b] + L.Jc[6 + a] * L.Jt[PD + b];
#pragma unroll
    for (int a = 0; a < 6; ++a) {
#pragma unroll
      for (int b = 0; b < PD; ++b) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < PD; ++k) s += w[a * PD + k] * sym_get<PD>(Vi, k, b);
        t[a * PD + b] = s;
      }
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < PD; ++k) s += t[a * PD + k] * g[k];
      tg[a] = s;
    }
    using O = RecI<PD, KI>;
#pragma unroll
    for (int k = 0; k < NW / 2; ++k) R[O::W + k] = make_double2(w[2 * k], w[2 * k + 1]);
#pragma unroll
    for (int k = 0; k < NW / 2; ++k) R[O::T + k] = make_double2(t[2 * k], t[2 * k + 1]);
#pragma unroll
    for (int k = 0; k < 6; ++k) R[O::F + k] = make_double2(L.Jc[2 * k], L.Jc[2 * k + 1]);
    R[O::R] = make_double2(L.r[0], L.r[1]);
#pragma unroll
    for (int k = 0; k < 3; ++k) R[O::TG + k] = make_double2(tg[2 * k], tg[2 * k + 1]);
    // intrinsics part: WI = Fk^T E, TI = WI V^-1, TI g (KI stored rows)
    double jk[2 * KI];
    if (KI == THEIA_MAX_INTRINSICS) {
#pragma unroll
      for (int k = 0; k < 2 * KI; ++k) jk[k] = L.Jk[k];
    } else {
      unsigned fm = L.gr >= 0 ? P.red_free[L.gr] : 0u;
#pragma unroll
      for (int k = 0; k < KI; ++k) {
        double v0 = 0.0, v1 = 0.0;
        if (fm) {
          const int q = __ffs(fm) - 1;
          fm &= fm - 1u;
#pragma unroll
          for (int j = 0; j < THEIA_MAX_INTRINSICS; ++j) if (j == q) { v0 = L.Jk[j]; v1 = L.Jk[THEIA_MAX_INTRINSICS + j]; }
        }
        jk[k] = v0; jk[KI + k] = v1;
      }
    }
    double ti[KI * PD], tig[KI];
#pragma unroll
    for (int a = 0; a < KI; ++a) {
#pragma unroll
      for (int b = 0; b < PD; ++b) wi[a * PD + b] = jk[a] * L.Jt[b] + jk[KI + a] * L.Jt[PD + b];
    }
#pragma unroll
    for (int a = 0; a < KI; ++a) {
#pragma unroll
      for (int b = 0; b < PD; ++b) {
        double s2 = 0.0;
#pragma unroll
        for (int k = 0; k < PD; ++k) s2 += wi[a * PD + k] * sym_get<PD>(Vi, k, b);
        ti[a * PD + b] = s2;
      }
      double s3 = 0.0;
#pragma unroll
      for (int k = 0; k < PD; ++k) s3 += ti[a * PD + k] * g[k];
      tig[a] = s3;
    }
#pragma unroll
    for (int k = 0; k < KI * PD / 2; ++k) R[O::WI + k] = make_double2(wi[2 * k], wi[2 * k + 1]);
#pragma unroll
    for (int k = 0; k < KI * PD / 2; ++k) R[O::TI + k] = make_double2(ti[2 * k], ti[2 * k + 1]);
#pragma unroll
    for (int k = 0; k < KI; ++k) R[O::FK + k] = make_double2(jk[2 * k], jk[2 * k + 1]);
#pragma unroll
    for (int k = 0; k < KI / 2; ++k) R[O::TIG + k] = make_double2(tig[2 * k], tig[2 * k + 1]);
  }
  if (P.pt_sum_slot) {
    // Tracks that see few variable intrinsics groups: the sums of WI and TI = WI V^-1 over the track's observations of a
    // group go to a pseudo-record per group, and the camera x group / group x group pair lists hold (observation, sum)
    // and (sum, sum) pairs instead of every ordered pair of the track's observations (build_gather_lists_intr).
    const int base = (L.active && !L.pconst) ? P.pt_sum_slot[L.p] : -1;   // (every lane of the track: same values)
    const int ngt = base >= 0 ? (int)P.pt_sum_cnt[L.p] : 0;
    int nsum = 0;   // wave-uniform: most summed groups of a track of this tile (0: nothing to do)
    for (int k = 1; k <= 4; ++k) if (__ballot(ngt >= k) != 0ull) nsum = k;
    for (int k = 0; k < nsum; ++k) {
      const int gk = k < ngt ? P.sum_group[base - P.sum_base + k] : -2;
      double wsum[KI * PD];
#pragma unroll
      for (int j = 0; j < KI * PD; ++j) wsum[j] = (L.gr == gk) ? wi[j] : 0.0;
      segment_allsum_log<KI * PD>(sg, lane, wsum);
      if (sg.head && k < ngt) {
        using O = RecI<PD, KI>;
        double2* R = reinterpret_cast<double2*>(P.rec + (size_t)(base + k) * RS);
        double tsum[KI * PD];
#pragma unroll
        for (int a = 0; a < KI; ++a)
#pragma unroll
          for (int b = 0; b < PD; ++b) {
            double s2 = 0.0;
#pragma unroll
            for (int q = 0; q < PD; ++q) s2 += wsum[a * PD + q] * sym_get<PD>(Vi, q, b);
            tsum[a * PD + b] = s2;
          }
#pragma unroll
        for (int j = 0; j < KI * PD / 2; ++j) {
          R[O::WI + j] = make_double2(wsum[2 * j], wsum[2 * j + 1]);
          R[O::TI + j] = make_double2(tsum[2 * j], tsum[2 * j + 1]);
        }
      }
    }
  }
  const double cost = wave_sum(L.cost);
  gmax = wave_max(gmax);
  const double inval = wave_sum((L.active && !L.valid) ? 1.0 : 0.0);
  const double npd = wave_sum((L.active && !pd_ok && sg.head) ? 1.0 : 0.0);
  if (lane == 0 && tile_ok) {
    tile_part[4 * (size_t)tile + 0] = cost;
    tile_part[4 * (size_t)tile + 1] = gmax;
    tile_part[4 * (size_t)tile + 2] = inval;
    tile_part[4 * (size_t)tile + 3] = npd;
  }
}


// rows [ra0, ra0 + RA) of the group block  Fk^T Fk - TI WI^T  (row0 = first reduced index of the group)
template <int PD, int KI, int RA, int ra0>
THIP_DEV void gdiag_item(const DevProblem& P, int row0, int beg, int end, bool atomic, double (*part)[64],
                         double* __restrict__ S) {
  using O = RecI<PD, KI>;
  double acc[RA * KI];
#pragma unroll
  for (int k = 0; k < RA * KI; ++k) acc[k] = 0.0;
  for (int q = beg + threadIdx.x; q < end; q += kBlock) {
    double TI[RA * PD], WI[KI * PD], Jk[2 * KI];
    const bool noself = P.slot_in_sum && P.slot_in_sum[q];   // the observation's own TI WI^T term is in its track's sum
    load_reci<PD, KI>(P.rec, q, O::TI + (ra0 * PD) / 2, TI); load_reci<PD, KI>(P.rec, q, O::WI, WI); load_reci<PD, KI>(P.rec, q, O::FK, Jk);
#pragma unroll
    for (int a = 0; a < RA; ++a)
#pragma unroll
      for (int b = 0; b < KI; ++b) {
        double s = Jk[ra0 + a] * Jk[b] + Jk[KI + ra0 + a] * Jk[KI + b];
        if (!noself) {
#pragma unroll
          for (int k = 0; k < PD; ++k) s -= TI[a * PD + k] * WI[b * PD + k];
        }
        acc[a * KI + b] += s;
      }
  }
  const int n = P.n;
  const unsigned fm = P.red_free[row0 / THEIA_MAX_INTRINSICS];
  finish_item<RA * KI>(acc, part, atomic, 1.0, [&](int e) -> double* {
    const int a = row_param<KI>(fm, ra0 + e / KI), b = row_param<KI>(fm, e % KI);
    if (a < 0 || b < 0 || b > a) return nullptr;
    return S + (size_t)(row0 + a) * n + row0 + b;
  });
}

template <int PD, int KI>
__global__ __launch_bounds__(kBlock) void k_schur_intr(DevProblem P, double* __restrict__ S, double* __restrict__ rhs,
                                                       double* __restrict__ colsq, double* __restrict__ gc) {
  __shared__ double part[kWavesPerBlock][64];
  using O = RecI<PD, KI>;
  const int* it = P.blk_items + 6 * blockIdx.x;
  const int type = it[0], row0 = it[1], col0 = it[2], beg = it[3], end = it[4], flags = it[5];
  const bool atomic = (flags & ITF_ATOMIC) != 0;
  const int n = P.n;
  if (type == IT_CC) { pair_item<PD, KI, 6, 6, false, false>(P, it, O::T, O::W, 0, part, S); return; }
  if (type == IT_CG) { pair_item<PD, KI, 6, KI, false, true>(P, it, O::T, O::WI, 0, part, S); return; }
  // ten intrinsics rows are split 4 + 6 (register budget; 4 * PD doubles keep the 16-B alignment); four compact rows are one item
  if (type == IT_GG0) { pair_item<PD, KI, 4, KI, true, true>(P, it, O::TI, O::WI, 0, part, S); return; }
  if (KI == THEIA_MAX_INTRINSICS && type == IT_GG1) {
    pair_item<PD, KI, KI - 4 ? KI - 4 : 2, KI, true, true>(P, it, O::TI + 2 * PD, O::WI, 4, part, S);
    return;
  }
  if (type == IT_CD) {
    double acc[39];
#pragma unroll
    for (int k = 0; k < 39; ++k) acc[k] = 0.0;
    for (int q = beg + threadIdx.x; q < end; q += kBlock) {
      double W[6 * PD], T[6 * PD], Jc[12], rt[8];
      load_reci<PD, KI>(P.rec, q, O::W, W); load_reci<PD, KI>(P.rec, q, O::T, T);
      load_reci<PD, KI>(P.rec, q, O::F, Jc); load_reci<PD, KI>(P.rec, q, O::R, rt);
#pragma unroll
      for (int a = 0; a < 6; ++a) {
#pragma unroll
        for (int b = 0; b <= a; ++b) {
          double s = Jc[a] * Jc[b] + Jc[6 + a] * Jc[6 + b];
#pragma unroll
          for (int k = 0; k < PD; ++k) s -= T[a * PD + k] * W[b * PD + k];
          acc[lidx(a, b)] += s;
        }
        const double jr = Jc[a] * rt[0] + Jc[6 + a] * rt[1];
        acc[21 + a] += jr - rt[2 + a];
        acc[27 + a] += jr;
        acc[33 + a] += Jc[a] * Jc[a] + Jc[6 + a] * Jc[6 + a];
      }
    }
    finish_item<39>(acc, part, atomic, 1.0, [&](int e) -> double* {
      if (e < 21) {
        int a = 0;
        while ((a + 1) * (a + 2) / 2 <= e) ++a;
        return S + (size_t)(row0 + a) * n + row0 + (e - a * (a + 1) / 2);
      }
      return (e < 27 ? rhs : (e < 33 ? gc : colsq)) + row0 + (e - 21) % 6;
    });
    return;
  }
  if (type == IT_CGD) {
    double acc[6 * KI];
#pragma unroll
    for (int k = 0; k < 6 * KI; ++k) acc[k] = 0.0;
    for (int q = beg + threadIdx.x; q < end; q += kBlock) {
      double T[6 * PD], WI[KI * PD], Jc[12], Jk[2 * KI];
      const bool noself = P.slot_in_sum && P.slot_in_sum[q];   // (the (observation, track sum) pair carries T_a WI_a^T)
      load_reci<PD, KI>(P.rec, q, O::T, T); load_reci<PD, KI>(P.rec, q, O::WI, WI);
      load_reci<PD, KI>(P.rec, q, O::F, Jc); load_reci<PD, KI>(P.rec, q, O::FK, Jk);
#pragma unroll
      for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = 0; b < KI; ++b) {
          double s = Jc[a] * Jk[b] + Jc[6 + a] * Jk[KI + b];
          if (!noself) {
#pragma unroll
            for (int k = 0; k < PD; ++k) s -= T[a * PD + k] * WI[b * PD + k];
          }
          acc[a * KI + b] += s;
        }
    }
    const unsigned fm = P.red_free[col0 / THEIA_MAX_INTRINSICS];
    finish_item<6 * KI>(acc, part, atomic, 1.0, [&](int e) -> double* {
      const int b = row_param<KI>(fm, e % KI);
      return b < 0 ? nullptr : S + (size_t)(row0 + e / KI) * n + col0 + b;
    });
    return;
  }
  if (type == IT_GD0) { gdiag_item<PD, KI, 4, 0>(P, row0, beg, end, atomic, part, S); return; }
  if (KI == THEIA_MAX_INTRINSICS && type == IT_GD1) { gdiag_item<PD, KI, KI - 4 ? KI - 4 : 2, KI - 4 ? 4 : 0>(P, row0, beg, end, atomic, part, S); return; }
  if (type == IT_GV) {
    double acc[3 * KI];
#pragma unroll
    for (int k = 0; k < 3 * KI; ++k) acc[k] = 0.0;
    for (int q = beg + threadIdx.x; q < end; q += kBlock) {
      double Jk[2 * KI], rt[2], tig[KI];
      load_reci<PD, KI>(P.rec, q, O::FK, Jk); load_reci<PD, KI>(P.rec, q, O::R, rt); load_reci<PD, KI>(P.rec, q, O::TIG, tig);
#pragma unroll
      for (int a = 0; a < KI; ++a) {
        const double jr = Jk[a] * rt[0] + Jk[KI + a] * rt[1];
        acc[a] += jr - tig[a];
        acc[KI + a] += jr;
        acc[2 * KI + a] += Jk[a] * Jk[a] + Jk[KI + a] * Jk[KI + a];
      }
    }
    const unsigned fm = P.red_free[row0 / THEIA_MAX_INTRINSICS];
    finish_item<3 * KI>(acc, part, atomic, 1.0, [&](int e) -> double* {
      const int b = row_param<KI>(fm, e % KI);
      return b < 0 ? nullptr : (e < KI ? rhs : (e < 2 * KI ? gc : colsq)) + row0 + b;
    });
    return;
  }
}

// Deterministic reduction of per-tile partials into the scalar block.
// field f of tile t at tile_part[t*nfields + f]; result -> scal[field_to_scal[f]]
// (sum, or max if field_is_max[f]).
__global__ __launch_bounds__(1024) void k_reduce_tiles(int ntiles, const double* __restrict__ part,
                                                       int nfields, const int* __restrict__ f2s,
                                                       const int* __restrict__ fmaxflag,
                                                       double* __restrict__ scal) {
  __shared__ double sm[8][16];
  reduce_tiles_body(ntiles, part, nfields, f2s, fmaxflag, scal, sm);
}

__global__ __launch_bounds__(256) void k_reduce_tiles_stage1(int ntiles, const double* __restrict__ part, int nfields,
                                                             const int* __restrict__ fmaxflag, double* __restrict__ out) {
  __shared__ double sm[8][4];
  const int nb = gridDim.x, b = blockIdx.x;
  const int per = (ntiles + nb - 1) / nb;
  const int t0 = b * per, t1 = min(ntiles, t0 + per);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double acc[8];
  bool ismax[8];
#pragma unroll
  for (int f = 0; f < 8; ++f) { acc[f] = 0.0; ismax[f] = f < nfields && fmaxflag[f] != 0; }
  for (int t = t0 + (int)threadIdx.x; t < t1; t += 256) {
    const double* row = part + (size_t)t * nfields;
#pragma unroll
    for (int f = 0; f < 8; ++f)
      if (f < nfields) { const double v = row[f]; acc[f] = ismax[f] ? fmax(acc[f], v) : acc[f] + v; }
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
  if ((int)threadIdx.x < nfields) {
    const int f = threadIdx.x;
    double v = sm[f][0];
    for (int w = 1; w < 4; ++w) v = ismax[f] ? fmax(v, sm[f][w]) : v + sm[f][w];
    out[(size_t)b * nfields + f] = v;
  }
}

// Add the LM diagonal to the camera-side blocks (intrinsics + extrinsics) and
// fold their gradient into the gradient max-norm: S_dd += clamp(colsq_d) / radius.
// ceil(n / 1024) workgroups of 1024 threads, one diagonal entry per thread (the entries are n + 1 doubles apart: one cache
// line each -- a single workgroup over n = 6000 was bound by what one CU keeps in flight, 22 us); with tile_part workgroup 0
// starts with the tile reduction of the linearisation.  The maxima meet in scal[SC_GMAX] through an atomic maximum.
__global__ __launch_bounds__(1024) void k_finalize_rcs(DevProblem P, const double* __restrict__ radius_p, double* __restrict__ S,
                                                       const double* __restrict__ colsq, const double* __restrict__ gc,
                                                       double* __restrict__ scal, int ntiles, const double* __restrict__ tile_part,
                                                       const int* __restrict__ f2s, const int* __restrict__ fmaxflag,
                                                       const uint8_t* __restrict__ tile_cls, int want_cls) {
  __shared__ double sm[16];
  __shared__ double smr[8][16];
  if (tile_part && blockIdx.x == 0) reduce_tiles_body(ntiles, tile_part, 4, f2s, fmaxflag, scal, smr, true);
  const double radius = *radius_p;
  double gmax = 0.0;
  const int d = blockIdx.x * 1024 + (int)threadIdx.x;
  // tile_cls (sharded solve with a distributed K3): only the columns of the 64-wide tiles of class want_cls
  if (d < P.n && (!tile_cls || tile_cls[d >> 6] == want_cls)) {
    const double sv = S[(size_t)d * P.n + d], cv = colsq[d], gv = gc[d], sr = P.scale_red[d];
    S[(size_t)d * P.n + d] = sv + fmin(fmax(cv, 1e-6), 1e32) / radius;
    gmax = fabs(gv / sr);
  }
  gmax = wave_max(gmax);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = gmax;
  __syncthreads();
  if (threadIdx.x == 0) {
    double v = sm[0];
    for (int w = 1; w < 16; ++w) v = fmax(v, sm[w]);
    if (v == v) atomicMax(reinterpret_cast<unsigned long long*>(&scal[SC_GMAX]), (unsigned long long)__double_as_longlong(v));   // (fmax drops a NaN too)
  }
}

// Jacobi scaling by reduced index (frozen columns: 1).  One workgroup.
__global__ void k_build_scale_red(DevProblem P, double* __restrict__ scale_red) {
  for (int d = threadIdx.x; d < P.n; d += blockDim.x) scale_red[d] = 1.0;
  __syncthreads();
  if (P.ni)
    for (int i = threadIdx.x; i < THEIA_MAX_INTRINSICS * P.ng_total; i += blockDim.x) {
      const int g = i / THEIA_MAX_INTRINSICS, q = i % THEIA_MAX_INTRINSICS;
      const int gr = P.grp_red[g];
      if (gr >= 0 && ((P.grp_free[g] >> q) & 1u)) scale_red[10 * gr + q] = P.scale_i[i];
    }
  for (int i = threadIdx.x; i < 6 * P.nc; i += blockDim.x) {
    const int c = i / 6, q = i % 6;
    const int rc = P.cam_red[c];
    if (rc >= 0 && !((P.cam_mask[c] >> q) & 1u)) scale_red[P.ni + 6 * rc + q] = P.scale_c[i];
  }
}

// candidate cameras and intrinsics: x + (-y) * scale (intrinsics projected onto
// their bounds, bundle_adjuster.cc:406-427); their |step|^2 and |x+|^2.
// one extrinsics component of the candidate camera: the state's, or state + (-y) * scale on a free column
THIP_DEV double cand_ext_component(const DevProblem& P, int c, int rc, int q, double x, const double* __restrict__ yc) {
  if (rc >= 0 && !((P.cam_mask[c] >> q) & 1u)) return __builtin_fma(-yc[6 * rc + q], P.scale_c[6 * c + q], x);
  return x;
}

// PREP (fused path without intrinsics): the workgroups behind the first one write the candidate's per-camera blocks
// (P.camrot_cand) and the cameras' steps as {D, v} (P.camdir) -- what a k_cam_prep launch on the candidate did next on the
// stream, from the SAME candidate components (cand_ext_component), so the two launches' work runs side by side: one launch
// and its dependent boundary less per iteration.  (Folding that work into the single reducing workgroup was slower: 26 us
// against 13 + 7.)
// the candidate intrinsics of a group: x + (-y) * scale on its free parameters, projected onto their bounds
// (bundle_adjuster.cc:406-427); gr < 0 (constant group): the state's
THIP_DEV void cand_intrinsics(const DevProblem& P, int g, int gr, const double* __restrict__ y, double (&kk)[THEIA_MAX_INTRINSICS]) {
  for (int q = 0; q < THEIA_MAX_INTRINSICS; ++q) {
    kk[q] = P.intr[(size_t)g * THEIA_MAX_INTRINSICS + q];
    if (gr >= 0 && ((P.grp_free[g] >> q) & 1u)) kk[q] = __builtin_fma(-y[10 * gr + q], P.scale_i[(size_t)g * THEIA_MAX_INTRINSICS + q], kk[q]);
  }
  if (gr >= 0) {
    const int model = P.group_model[g];
    if (kk[0] < 1.0) kk[0] = 1.0;
    if (model == THEIA_CAM_DOUBLE_SPHERE) { kk[5] = fmin(1.0, fmax(-1.0, kk[5])); kk[6] = fmin(1.0, fmax(0.0, kk[6])); }
    if (model == THEIA_CAM_EXTENDED_UNIFIED) { kk[5] = fmin(1.0, fmax(0.0, kk[5])); kk[6] = fmax(0.1, kk[6]); }
  }
}

// PREP: 0 = the candidate parameters only; 1 = fused path without intrinsics (blocks + {D, v} steps, above); 2 = fused path with
// free intrinsics: the candidate's blocks carry the candidate intrinsics of the camera's group (cand_intrinsics, the same
// arithmetic as the reducing workgroup's), no {D, v} (the back-substitution multiplies the solved step itself there)
template <int PREP>
__global__ __launch_bounds__(1024) void k_cam_update(DevProblem P, const double* __restrict__ cam, const double* __restrict__ y,
                             double* __restrict__ cand, double* __restrict__ cand_intr,
                             double* __restrict__ out_stepsq, double* __restrict__ out_xnormsq,
                             double* __restrict__ zero16) {
  const double* yc = y + P.ni;
  if constexpr (PREP != 0) {
    if (blockIdx.x > 0) {
      if (threadIdx.x >= 256) return;   // 256 cameras per workgroup, as k_cam_prep: the blocks spread over the CUs
      const int c = (blockIdx.x - 1) * 256 + threadIdx.x;
      if (c == 0 && P.frun_next) { P.frun_next[0] = 0; P.frun_next[1] = 0; }   // (as k_cam_prep: heads of the run queues)
      if (c >= P.nc) return;
      const int rc = P.cam_red[c];
      double xp[6], dl[6];
      for (int q = 0; q < 6; ++q) {
        const double x = cam[6 * c + q];
        xp[q] = cand_ext_component(P, c, rc, q, x, yc);
        dl[q] = (rc >= 0 && !((P.cam_mask[c] >> q) & 1u)) ? (-yc[6 * rc + q]) * P.scale_c[6 * c + q] : 0.0;
      }
      if constexpr (PREP == 2) {
        const int g = P.cam_group[c];
        double kk[THEIA_MAX_INTRINSICS];
        cand_intrinsics(P, g, P.grp_red ? P.grp_red[g] : -1, y, kk);
        cam_prep_one_k(P, c, xp, kk, P.camrot_cand);
        return;
      }
      cam_prep_one(P, c, xp, P.intr, P.camrot_cand);
      double ext[6], out[12];
      RotTerms rt;
      camrot_load(P.camrot + (size_t)kCamRot * c, ext, rt);
      camera_step_direction(ext + 3, rt, dl, dl + 3, out);
      for (int k = 0; k < 12; ++k) P.camdir[(size_t)12 * c + k] = out[k];
      return;
    }
  }
  __shared__ double s1[1024], s2[1024];
  double st = 0.0, xn = 0.0;
  for (int c = threadIdx.x; c < P.nc; c += blockDim.x) {
    const int rc = P.cam_red[c];
    for (int q = 0; q < 6; ++q) {
      const double x = cam[6 * c + q];
      const double xp = cand_ext_component(P, c, rc, q, x, yc);
      if (rc >= 0) {
        st += (x - xp) * (x - xp);
        xn += xp * xp;
      }
      cand[6 * c + q] = xp;
    }
  }
  if (cand_intr) {
    for (int g = threadIdx.x; g < P.ng_total; g += blockDim.x) {
      const int gr = P.grp_red ? P.grp_red[g] : -1;
      double kk[THEIA_MAX_INTRINSICS];
      cand_intrinsics(P, g, gr, y, kk);
      for (int q = 0; q < THEIA_MAX_INTRINSICS; ++q) {
        const double x = P.intr[(size_t)g * THEIA_MAX_INTRINSICS + q];
        cand_intr[(size_t)g * THEIA_MAX_INTRINSICS + q] = kk[q];
        if (gr >= 0 && q < P.grp_k[g]) { st += (x - kk[q]) * (x - kk[q]); xn += kk[q] * kk[q]; }
      }
    }
  }
  s1[threadIdx.x] = st; s2[threadIdx.x] = xn;
  __syncthreads();
  for (int s = blockDim.x / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) { s1[threadIdx.x] += s1[threadIdx.x + s]; s2[threadIdx.x] += s2[threadIdx.x + s]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    // first kernel of the trial-step group: it also clears the group's scalar block (saves a memset node)
    if (zero16) for (int k = 0; k < 16; ++k) zero16[k] = 0.0;
    *out_stepsq = s1[0]; *out_xnormsq = s2[0];
  }
}

// ------------------------------- back-substitution + candidate + trial cost
// tile_part: [ntiles][5] = {cand_cost, mcc, stepsq, xnormsq, invalid}
// ROT: the per-camera blocks of k_cam_prep (P.camrot for the state, P.camrot_cand for the candidate cameras) replace
// the per-observation sincos and the chained camera -> group -> intrinsics gathers.
// KMASK != 0 (INTR): every variable group frees exactly these intrinsics; the Jacobian columns of the others are never formed
template <int PD, bool INTR, bool ROT = false, unsigned KMASK = 0u>
// (148 VGPRs = three waves per SIMD; capped at 128 for four, 24 of them spill and the kernel takes 55 us longer)
__global__ __launch_bounds__(kBlock) void k_backsub(DevProblem P, const double* __restrict__ cam,
                                                    const double* __restrict__ pts,
                                                    const double* __restrict__ cand_cam,
                                                    double* __restrict__ cand_pts,
                                                    const double* __restrict__ yc,
                                                    const double* __restrict__ Vinv,
                                                    double* __restrict__ tile_part) {
  constexpr int NT = PD * (PD + 1) / 2;
  const int lane = threadIdx.x & 63;
  const int tile = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const bool tile_ok = tile < P.ntiles;
  const int cnt = tile_ok ? P.tile_count[tile] : 0;
  const int start = tile_ok ? P.tile_start[tile] : 0;
  LaneLin<PD, INTR> L;
  constexpr bool DIRC = ROT && !INTR;   // the camera block only ever multiplies y_c here: directional derivative instead
  if constexpr (DIRC) lane_linearize<PD, true, INTR, true, kModelsAll, true>(P, P.camrot, pts, start + lane, lane < cnt, lane, L);
  else if constexpr (ROT) lane_linearize<PD, true, INTR, true>(P, P.camrot, pts, start + lane, lane < cnt, lane, L);
  else lane_linearize<PD, true, INTR>(P, cam, pts, start + lane, lane < cnt, lane, L);
  const Segment sg = lane_segment_all(L.p, lane);
  // m_c = F y_c   (yc points at the camera part; the intrinsics part sits ni before it)
  double mc[2] = {0.0, 0.0};
  if constexpr (DIRC) {
    mc[0] = L.mc[0]; mc[1] = L.mc[1];   // zero for constant cameras (their {D, v} is zero) and inactive lanes
  } else if (L.active && L.rc >= 0) {
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const double yv = yc[6 * L.rc + q];
      mc[0] += L.Jc[q] * yv;
      mc[1] += L.Jc[6 + q] * yv;
    }
  }
  if constexpr (INTR) {
    if (L.active && L.gr >= 0) {
      const double* yi = yc - P.ni + 10 * L.gr;
#pragma unroll
      for (int q = 0; q < THEIA_MAX_INTRINSICS; ++q) {
        if (KMASK != 0u && !((KMASK >> q) & 1u)) continue;   // (a frozen parameter's step is zero)
        mc[0] += L.Jk[q] * yi[q];
        mc[1] += L.Jk[THEIA_MAX_INTRINSICS + q] * yi[q];
      }
    }
  }
  // t = E^T (r - F y_c), summed over the track
  double in[PD], tsum[PD];
#pragma unroll
  for (int q = 0; q < PD; ++q) in[q] = L.Jt[q] * (L.r[0] - mc[0]) + L.Jt[PD + q] * (L.r[1] - mc[1]);
  if constexpr (ROT) {
#pragma unroll
    for (int q = 0; q < PD; ++q) tsum[q] = in[q];
    segment_allsum_log<PD>(sg, lane, tsum);
  } else {
    segment_allsum<PD>(sg, in, tsum);
  }
  double Vi[NT];
#pragma unroll
  for (int k = 0; k < NT; ++k) Vi[k] = (L.active && !L.pconst) ? Vinv[(size_t)NT * L.p + k] : 0.0;
  double yp[PD];
#pragma unroll
  for (int a = 0; a < PD; ++a) {
    double s = 0.0;
#pragma unroll
    for (int b = 0; b < PD; ++b) s += sym_get<PD>(Vi, a, b) * tsum[b];
    yp[a] = s;
  }
  // step = -y ; model residual m = Js * step ; mcc = -m . (r + m/2)
  double mcc = 0.0;
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    double m = -mc[a];
#pragma unroll
    for (int q = 0; q < PD; ++q) m -= L.Jt[a * PD + q] * yp[q];
    mcc -= m * (L.r[a] + m / 2.0);
  }
  if (!L.active) mcc = 0.0;
  // candidate point (every lane of the track computes the same value)
  double Xp[4] = {L.X[0], L.X[1], L.X[2], L.X[3]};
  double stepsq = 0.0, xnormsq = 0.0;
  if (L.active && !L.pconst) {
    double d[PD];
#pragma unroll
    for (int q = 0; q < PD; ++q) d[q] = -yp[q] * P.scale_p[(size_t)PD * L.p + q];
    if (PD == 3) {
      const double d3[3] = {d[0], d[1], d[2]};
      sphere_plus(L.X, d3, Xp);
    } else {
#pragma unroll
      for (int q = 0; q < PD; ++q) Xp[q] = L.X[q] + d[q];
    }
    if (sg.head) {
      reinterpret_cast<double4*>(cand_pts)[L.p] = make_double4(Xp[0], Xp[1], Xp[2], Xp[3]);
#pragma unroll
      for (int q = 0; q < 4; ++q) { stepsq += (L.X[q] - Xp[q]) * (L.X[q] - Xp[q]); xnormsq += Xp[q] * Xp[q]; }
    }
  }
  // trial cost at the candidate
  double ccost = 0.0;
  bool cvalid = true;
  if (L.active) {
    double ext[6];
    const double2 uv = P.obs_uv[start + lane];
    double six = 1.0, siy = 1.0;
    if (P.obs_si) { const double2 s = P.obs_si[start + lane]; six = s.x; siy = s.y; }
    ObsLin ol;
    const bool depth_row = P.obs_kind && P.obs_kind[start + lane];
    if constexpr (ROT) {
      const double* cr = P.camrot_cand + (size_t)kCamRot * L.c;
      RotTerms rt;
      double kc[12];
      camrot_load(cr, ext, rt);
      load_d2<12>(cr + kCamRotIntr, kc);   // intrinsics (10) | model | reduced index
      observe_rot<false>(depth_row ? THIP_MODEL_DEPTH_ROW : (int)kc[kCamRotModel - kCamRotIntr], ext, rt, kc, Xp, uv.x, uv.y, six, siy, ol);
    } else {
#pragma unroll
      for (int i = 0; i < 6; ++i) ext[i] = cand_cam[6 * L.c + i];
      const int g = P.cam_group[L.c];
      const double* kc = (INTR ? P.intr_cand : P.intr) + (size_t)g * THEIA_MAX_INTRINSICS;
      observe<false>(depth_row ? THIP_MODEL_DEPTH_ROW : P.group_model[g], ext, kc, Xp, uv.x, uv.y, six, siy, ol);
    }
    cvalid = ol.valid;
    const double s2 = ol.r[0] * ol.r[0] + ol.r[1] * ol.r[1];
    double rho1;
    ccost = 0.5 * (P.loss_type == THEIA_LOSS_TRIVIAL ? s2 : loss_eval(P.loss_type, depth_row ? P.loss_width_depth : P.loss_width, s2, &rho1));
  }
  ccost = wave_sum_all(ccost);   // no lane has left the kernel: the DPP reductions apply
  mcc = wave_sum_all(mcc);
  stepsq = wave_sum_all(stepsq);
  xnormsq = wave_sum_all(xnormsq);
  const double inval = wave_count(!cvalid);
  if (lane == 0 && tile_ok) {
    double* tp = tile_part + 5 * (size_t)tile;
    tp[0] = ccost; tp[1] = mcc; tp[2] = stepsq; tp[3] = xnormsq; tp[4] = inval;
  }
}

// ------------------------------------------------ introspection / cost only
// residuals/Jacobians in SORTED observation order (host un-permutes).
// tile_part: [ntiles][2] = {cost, invalid}
template <int PD, bool WANT_JAC, bool INTR = false>
__global__ __launch_bounds__(kBlock) void k_evaluate(DevProblem P, const double* __restrict__ cam,
                                                     const double* __restrict__ pts,
                                                     double* __restrict__ residuals,
                                                     double* __restrict__ jac_cam,
                                                     double* __restrict__ jac_pt,
                                                     uint8_t* __restrict__ valid,
                                                     double* __restrict__ tile_part,
                                                     double* __restrict__ jac_intr = nullptr) {
  const int lane = threadIdx.x & 63;
  const int tile = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (tile >= P.ntiles) return;
  const int cnt = P.tile_count[tile];
  const int start = P.tile_start[tile];
  LaneLin<PD, INTR> L;
  lane_linearize<PD, WANT_JAC, INTR>(P, cam, pts, start + lane, lane < cnt, lane, L);
  if (L.active) {
    const size_t o = (size_t)start + lane;
    if constexpr (INTR) {
      if (jac_intr) for (int i = 0; i < 2 * THEIA_MAX_INTRINSICS; ++i) jac_intr[2 * THEIA_MAX_INTRINSICS * o + i] = L.Jk[i];
    }
    if (residuals) { residuals[2 * o] = L.r[0]; residuals[2 * o + 1] = L.r[1]; }
    if (WANT_JAC) {
      if (jac_cam) for (int i = 0; i < 12; ++i) jac_cam[12 * o + i] = L.Jc[i];
      if (jac_pt) for (int i = 0; i < 2 * PD; ++i) jac_pt[2 * PD * o + i] = L.Jt[i];
    }
    if (valid) valid[o] = L.valid ? 1 : 0;
  }
  const double cost = wave_sum(L.cost);
  const double inval = wave_sum((L.active && !L.valid) ? 1.0 : 0.0);
  if (lane == 0) { tile_part[2 * (size_t)tile] = cost; tile_part[2 * (size_t)tile + 1] = inval; }
}

// ------------------------------------------------------------- long tracks
// Tracks with more than 64 observations do not fit a wave tile.  They are rare
// (a few percent of real reconstructions) and take this slower path: one thread
// per observation, per-track sums through global FP64 atomics into a small
// scratch [track][NT + PD], pair products by re-linearising the partner.
// Results are the same quantities the tiled kernels produce.
struct LongView {
  int nobs;                    // observations of long tracks
  int ntracks;
  const int* obs_index;        // [nobs] index into the sorted observation arrays
  const int* obs_slot;         // [nobs] long-track slot of the observation
  const int* track_start;      // [ntracks + 1] offsets into obs_index
  const int* track_pt;         // [ntracks] point id
};

THIP_DEV void atomic_max_nonneg(double* p, double v) {
  atomicMax(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v));
}

// The camera-side block of one observation as reduced columns: [free intrinsics of its group (10) | extrinsics (6)]
// with intrinsics, the six extrinsics columns without.  col < 0 = constant column.
template <int PD, bool INTR>
struct CamSide {
  static constexpr int N = INTR ? 16 : 6;
  double F0[N], F1[N];
  int col[N];
};
template <int PD, bool INTR>
THIP_DEV void cam_side(const DevProblem& P, const LaneLin<PD, INTR>& L, CamSide<PD, INTR>& C) {
  constexpr int O = INTR ? THEIA_MAX_INTRINSICS : 0;
  if constexpr (INTR) {
    const unsigned fm = L.gr >= 0 ? P.grp_free[L.g] : 0u;
#pragma unroll
    for (int q = 0; q < THEIA_MAX_INTRINSICS; ++q) {
      C.F0[q] = L.Jk[q]; C.F1[q] = L.Jk[THEIA_MAX_INTRINSICS + q];
      C.col[q] = ((fm >> q) & 1u) ? 10 * L.gr + q : -1;
    }
  }
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    C.F0[O + q] = L.Jc[q]; C.F1[O + q] = L.Jc[6 + q];
    C.col[O + q] = L.rc >= 0 ? P.ni + 6 * L.rc + q : -1;
  }
}

// pass A: per-observation linearisation -> per-track V/g (scratch), camera terms.
//   MODE 0: column norms only (Jacobi scaling pass); MODE 1: linearize.
template <int PD, int MODE, bool INTR>
__global__ void k_long_accum(DevProblem P, LongView Lv, const double* __restrict__ cam, const double* __restrict__ pts,
                             double* __restrict__ scratch, double* __restrict__ S, double* __restrict__ rhs,
                             double* __restrict__ colsq, double* __restrict__ gc, double* __restrict__ scal,
                             double* __restrict__ colsq_c0, double* __restrict__ colsq_i0) {
  constexpr int NT = PD * (PD + 1) / 2;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= Lv.nobs) return;
  LaneLin<PD, INTR> L;
  lane_linearize<PD, true, INTR>(P, cam, pts, Lv.obs_index[t], true, 0, L);
  double* sc = scratch + (size_t)Lv.obs_slot[t] * (NT + PD);
#pragma unroll
  for (int a = 0; a < PD; ++a) {
#pragma unroll
    for (int b = 0; b <= a; ++b) atomic_add(&sc[lidx(a, b)], L.Jt[a] * L.Jt[b] + L.Jt[PD + a] * L.Jt[PD + b]);
    if (MODE == 1) atomic_add(&sc[NT + a], L.Jt[a] * L.r[0] + L.Jt[PD + a] * L.r[1]);
  }
  if (MODE == 0) {
    if (L.rc >= 0)
      for (int q = 0; q < 6; ++q) atomic_add(&colsq_c0[6 * L.c + q], L.Jc[q] * L.Jc[q] + L.Jc[6 + q] * L.Jc[6 + q]);
    if constexpr (INTR) {
      if (L.gr >= 0)
        for (int q = 0; q < THEIA_MAX_INTRINSICS; ++q) {
          const double v = L.Jk[q] * L.Jk[q] + L.Jk[THEIA_MAX_INTRINSICS + q] * L.Jk[THEIA_MAX_INTRINSICS + q];
          if (v != 0.0) atomic_add(&colsq_i0[(size_t)L.g * THEIA_MAX_INTRINSICS + q], v);
        }
    }
    return;
  }
  atomic_add(&scal[SC_COST], L.cost);
  if (!L.valid) atomic_add(&scal[SC_INVALID], 1.0);
  CamSide<PD, INTR> C;
  cam_side<PD, INTR>(P, L, C);
  const int n = P.n;
  for (int a = 0; a < C.N; ++a) {
    const int ca = C.col[a];
    if (ca < 0) continue;
    const double jr = C.F0[a] * L.r[0] + C.F1[a] * L.r[1];
    atomic_add(&rhs[ca], jr);
    atomic_add(&gc[ca], jr);
    atomic_add(&colsq[ca], C.F0[a] * C.F0[a] + C.F1[a] * C.F1[a]);
    for (int b = 0; b < C.N; ++b) {
      const int cb = C.col[b];
      if (cb < 0 || cb > ca) continue;   // lower triangle of S
      atomic_add(&S[(size_t)ca * n + cb], C.F0[a] * C.F0[b] + C.F1[a] * C.F1[b]);
    }
  }
}

// pass B: per long track: colsq_p (MODE 0) or damped inverse, g_p, gradient max (MODE 1)
template <int PD, int MODE>
__global__ void k_long_track(DevProblem P, LongView Lv, const double* __restrict__ radius_p, const double* __restrict__ scratch,
                             double* __restrict__ colsq_p, double* __restrict__ Vinv, double* __restrict__ gp,
                             double* __restrict__ scal) {
  constexpr int NT = PD * (PD + 1) / 2;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= Lv.ntracks) return;
  const int p = Lv.track_pt[t];
  const double* sc = scratch + (size_t)t * (NT + PD);
  if (MODE == 0) {
    for (int a = 0; a < PD; ++a) colsq_p[(size_t)PD * p + a] = sc[lidx(a, a)];
    return;
  }
  if (P.pt_const[p]) return;
  const double radius = *radius_p;
  double V[NT], Vi[NT];
  for (int k = 0; k < NT; ++k) V[k] = sc[k];
  for (int a = 0; a < PD; ++a) V[lidx(a, a)] += fmin(fmax(V[lidx(a, a)], 1e-6), 1e32) / radius;
  const bool ok = invert_spd<PD>(V, Vi);
  if (!ok) { atomic_add(&scal[SC_NOTPD], 1.0); for (int k = 0; k < NT; ++k) Vi[k] = 0.0; }
  for (int k = 0; k < NT; ++k) Vinv[(size_t)NT * p + k] = Vi[k];
  double gmax = 0.0;
  for (int a = 0; a < PD; ++a) {
    gp[(size_t)PD * p + a] = sc[NT + a];
    gmax = fmax(gmax, fabs(sc[NT + a] / P.scale_p[(size_t)PD * p + a]));
  }
  atomic_max_nonneg(&scal[SC_GMAX], gmax);
}

// pass C: Schur products of long tracks: thread = observation i, walks the
// track's observations j (re-linearised), S_ij -= W_i V^-1 W_j^T, rhs -= W_i V^-1 g.
template <int PD, bool INTR>
__global__ void k_long_schur(DevProblem P, LongView Lv, const double* __restrict__ cam, const double* __restrict__ pts,
                             const double* __restrict__ Vinv, const double* __restrict__ gp, double* __restrict__ S,
                             double* __restrict__ rhs) {
  constexpr int NT = PD * (PD + 1) / 2;
  constexpr int NC = CamSide<PD, INTR>::N;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= Lv.nobs) return;
  LaneLin<PD, INTR> L;
  lane_linearize<PD, true, INTR>(P, cam, pts, Lv.obs_index[t], true, 0, L);
  if (L.pconst) return;
  CamSide<PD, INTR> C;
  cam_side<PD, INTR>(P, L, C);
  bool any = false;
  for (int a = 0; a < NC; ++a) any = any || C.col[a] >= 0;
  if (!any) return;
  double Vi[NT], T[NC * PD];
  for (int k = 0; k < NT; ++k) Vi[k] = Vinv[(size_t)NT * L.p + k];
  for (int a = 0; a < NC; ++a) {
    double W[PD];
    for (int b = 0; b < PD; ++b) W[b] = C.F0[a] * L.Jt[b] + C.F1[a] * L.Jt[PD + b];
    double wy = 0.0;
    for (int b = 0; b < PD; ++b) {
      double s = 0.0;
      for (int k = 0; k < PD; ++k) s += W[k] * sym_get<PD>(Vi, k, b);
      T[a * PD + b] = s;
      wy += s * gp[(size_t)PD * L.p + b];   // (W Vinv) g = W (Vinv g)
    }
    if (C.col[a] >= 0) atomic_add(&rhs[C.col[a]], -wy);
  }
  const int n = P.n, slot = Lv.obs_slot[t];
  for (int u = Lv.track_start[slot]; u < Lv.track_start[slot + 1]; ++u) {
    LaneLin<PD, INTR> M;
    lane_linearize<PD, true, INTR>(P, cam, pts, Lv.obs_index[u], true, 0, M);
    CamSide<PD, INTR> D;
    cam_side<PD, INTR>(P, M, D);
    for (int a = 0; a < NC; ++a) {
      const int ca = C.col[a];
      if (ca < 0) continue;
      for (int b = 0; b < NC; ++b) {
        const int cb = D.col[b];
        if (cb < 0 || cb > ca) continue;   // every ordered pair (t, u) lands once in the lower triangle
        double s = 0.0;
        for (int k = 0; k < PD; ++k) s += T[a * PD + k] * (D.F0[b] * M.Jt[k] + D.F1[b] * M.Jt[PD + k]);
        atomic_add(&S[(size_t)ca * n + cb], -s);
      }
    }
  }
}

// F y over the camera-side block of an observation (y = solution of the reduced system, all columns)
template <int PD, bool INTR>
THIP_DEV void cam_side_apply(const CamSide<PD, INTR>& C, const double* __restrict__ y, double (&m)[2]) {
  m[0] = m[1] = 0.0;
  for (int q = 0; q < C.N; ++q)
    if (C.col[q] >= 0) { m[0] += C.F0[q] * y[C.col[q]]; m[1] += C.F1[q] * y[C.col[q]]; }
}

// back-substitution, pass 1: t_p = sum E^T (r - F y_c) into the scratch
template <int PD, bool INTR>
__global__ void k_long_back1(DevProblem P, LongView Lv, const double* __restrict__ cam, const double* __restrict__ pts,
                             const double* __restrict__ y, double* __restrict__ scratch) {
  constexpr int NT = PD * (PD + 1) / 2;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= Lv.nobs) return;
  LaneLin<PD, INTR> L;
  lane_linearize<PD, true, INTR>(P, cam, pts, Lv.obs_index[t], true, 0, L);
  CamSide<PD, INTR> C;
  cam_side<PD, INTR>(P, L, C);
  double mc[2];
  cam_side_apply<PD, INTR>(C, y, mc);
  double* sc = scratch + (size_t)Lv.obs_slot[t] * (NT + PD);
  for (int q = 0; q < PD; ++q) atomic_add(&sc[q], L.Jt[q] * (L.r[0] - mc[0]) + L.Jt[PD + q] * (L.r[1] - mc[1]));
}

// back-substitution, pass 2: y_p, model cost change, candidate point + trial cost
template <int PD, bool INTR>
__global__ void k_long_back2(DevProblem P, LongView Lv, const double* __restrict__ cam, const double* __restrict__ pts,
                             const double* __restrict__ cand_cam, double* __restrict__ cand_pts,
                             const double* __restrict__ y, const double* __restrict__ Vinv,
                             const double* __restrict__ scratch, double* __restrict__ scalB) {
  constexpr int NT = PD * (PD + 1) / 2;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= Lv.nobs) return;
  const int o = Lv.obs_index[t];
  LaneLin<PD, INTR> L;
  lane_linearize<PD, true, INTR>(P, cam, pts, o, true, 0, L);
  CamSide<PD, INTR> C;
  cam_side<PD, INTR>(P, L, C);
  const int slot = Lv.obs_slot[t];
  const double* sc = scratch + (size_t)slot * (NT + PD);
  double yp[PD];
  for (int a = 0; a < PD; ++a) {
    double s = 0.0;
    if (!L.pconst) for (int b = 0; b < PD; ++b) {
      const double v = (a >= b) ? Vinv[(size_t)NT * L.p + lidx(a, b)] : Vinv[(size_t)NT * L.p + lidx(b, a)];
      s += v * sc[b];
    }
    yp[a] = s;
  }
  double mcv[2];
  cam_side_apply<PD, INTR>(C, y, mcv);
  double mcc = 0.0;
  for (int a = 0; a < 2; ++a) {
    double m = -mcv[a];
    for (int q = 0; q < PD; ++q) m -= L.Jt[a * PD + q] * yp[q];
    mcc -= m * (L.r[a] + m / 2.0);
  }
  double Xp[4] = {L.X[0], L.X[1], L.X[2], L.X[3]};
  if (!L.pconst) {
    double d[PD];
    for (int q = 0; q < PD; ++q) d[q] = -yp[q] * P.scale_p[(size_t)PD * L.p + q];
    if (PD == 3) { const double d3[3] = {d[0], d[1], d[2]}; sphere_plus(L.X, d3, Xp); }
    else for (int q = 0; q < PD; ++q) Xp[q] = L.X[q] + d[q];
    if (t == Lv.track_start[slot]) {  // first observation of the track writes the point
      reinterpret_cast<double4*>(cand_pts)[L.p] = make_double4(Xp[0], Xp[1], Xp[2], Xp[3]);
      double st = 0.0, xn = 0.0;
      for (int q = 0; q < 4; ++q) { st += (L.X[q] - Xp[q]) * (L.X[q] - Xp[q]); xn += Xp[q] * Xp[q]; }
      atomic_add(&scalB[2], st); atomic_add(&scalB[3], xn);
    }
  }
  double ext[6];
  for (int i = 0; i < 6; ++i) ext[i] = cand_cam[6 * L.c + i];
  const int g = P.cam_group[L.c];
  const double2 uv = P.obs_uv[o];
  double six = 1.0, siy = 1.0;
  if (P.obs_si) { const double2 s = P.obs_si[o]; six = s.x; siy = s.y; }
  ObsLin ol;
  const bool depth_row = P.obs_kind && P.obs_kind[o];
  const double* kc = (INTR ? P.intr_cand : P.intr) + (size_t)g * THEIA_MAX_INTRINSICS;
  observe<false>(depth_row ? THIP_MODEL_DEPTH_ROW : P.group_model[g], ext, kc, Xp, uv.x, uv.y, six, siy, ol);
  double rho1;
  const double cc = 0.5 * loss_eval(P.loss_type, depth_row ? P.loss_width_depth : P.loss_width, ol.r[0] * ol.r[0] + ol.r[1] * ol.r[1], &rho1);
  atomic_add(&scalB[0], cc); atomic_add(&scalB[1], mcc);
  if (!ol.valid) atomic_add(&scalB[4], 1.0);
}

inline int tile_blocks(int ntiles) { return (ntiles + kWavesPerBlock - 1) / kWavesPerBlock; }

}  // namespace

// ------------------------------------------------------------------ launchers
void launch_colnorm(const DevProblem& P, const double* cam, const double* pts, double* colsq_c,
                    double* colsq_p, double* colsq_i, hipStream_t st) {
  if (P.ntiles == 0) return;
  const int nb = tile_blocks(P.ntiles);
  // fused plan without intrinsics: the cameras through pre-rotated blocks (the handle keeps P.camrot for k_lin_schur)
  const bool rot = !P.ni && P.camrot && P.n_fruns > 0 && !getenv("THEIA_HIP_COLNORM_DIRECT");
  if (P.ni) {
    if (P.pd == 3) k_colnorm<3, true><<<nb, kBlock, 0, st>>>(P, cam, pts, colsq_c, colsq_p, colsq_i);
    else k_colnorm<4, true><<<nb, kBlock, 0, st>>>(P, cam, pts, colsq_c, colsq_p, colsq_i);
  } else if (rot) {
    launch_cam_prep(P, cam, P.intr, P.camrot, st);   // (P carries the unit scales of this pass: the caller rebuilds the blocks afterwards)
    if (P.pd == 3) k_colnorm<3, false, true><<<nb, kBlock, 0, st>>>(P, cam, pts, colsq_c, colsq_p, colsq_i);
    else k_colnorm<4, false, true><<<nb, kBlock, 0, st>>>(P, cam, pts, colsq_c, colsq_p, colsq_i);
  } else {
    if (P.pd == 3) k_colnorm<3, false><<<nb, kBlock, 0, st>>>(P, cam, pts, colsq_c, colsq_p, colsq_i);
    else k_colnorm<4, false><<<nb, kBlock, 0, st>>>(P, cam, pts, colsq_c, colsq_p, colsq_i);
  }
  if (!P.slot_obs) return;
  if (P.ni && P.n_blk_items) {
    if (P.pd == 3) k_colnorm_gather<3, true><<<P.n_blk_items, kBlock, 0, st>>>(P, cam, pts, colsq_c, colsq_i);
    else k_colnorm_gather<4, true><<<P.n_blk_items, kBlock, 0, st>>>(P, cam, pts, colsq_c, colsq_i);
  } else if (!P.ni && P.n_diag_items && rot) {
    if (P.pd == 3) k_colnorm_gather<3, false, true><<<P.n_diag_items, kBlock, 0, st>>>(P, cam, pts, colsq_c, colsq_i);
    else k_colnorm_gather<4, false, true><<<P.n_diag_items, kBlock, 0, st>>>(P, cam, pts, colsq_c, colsq_i);
  } else if (!P.ni && P.n_diag_items) {
    if (P.pd == 3) k_colnorm_gather<3, false><<<P.n_diag_items, kBlock, 0, st>>>(P, cam, pts, colsq_c, colsq_i);
    else k_colnorm_gather<4, false><<<P.n_diag_items, kBlock, 0, st>>>(P, cam, pts, colsq_c, colsq_i);
  }
}

void launch_build_scale_red(const DevProblem& P, double* scale_red, hipStream_t st) {
  if (P.n == 0) return;
  k_build_scale_red<<<1, 1024, 0, st>>>(P, scale_red);
}

void launch_make_scale(int count, const double* colsq, double* scale, hipStream_t st) {
  if (count == 0) return;
  k_make_scale<<<(count + 255) / 256, 256, 0, st>>>(count, colsq, scale);
}

void launch_linearize(const DevProblem& P, const double* cam, const double* pts, const double* radius,
                      const ReduceBuf& rb, double* Vinv, double* gp, double* tile_part, hipStream_t st) {
  if (P.ntiles == 0) return;
  if (P.n_fruns > 0 && P.ni == 0) {   // fused linearise + Schur (ba_fused.hip)
    launch_linearize_fused(P, cam, pts, radius, rb, Vinv, tile_part, st);
    return;
  }
  if (P.n_fruns > 0 && P.fused_bw > 0) {   // the same with compound [extrinsics | intrinsics] blocks (ba_fused_intr.hip)
    launch_linearize_fused_intr(P, cam, pts, radius, rb, Vinv, tile_part, st);
    return;
  }
  if (P.rec && P.ni > 0) {   // intrinsics optimised: 16-wide camera-side blocks on the gather lists
    const int g = tile_blocks(P.ntiles);
#define THIP_INTR(PD_, KI_)                                                                              \
    do {                                                                                                   \
      k_lin_obs_intr<PD_, KI_><<<g, kBlock, 0, st>>>(P, cam, pts, radius, Vinv, gp, tile_part);            \
      if (P.n_blk_items) k_schur_intr<PD_, KI_><<<P.n_blk_items, kBlock, 0, st>>>(P, rb.S, rb.rhs, rb.colsq, rb.gc); \
    } while (0)
    if (P.pd == 3) { if (P.intr_rows == 4) THIP_INTR(3, 4); else THIP_INTR(3, 10); }
    else { if (P.intr_rows == 4) THIP_INTR(4, 4); else THIP_INTR(4, 10); }
#undef THIP_INTR
    return;
  }
  if (P.rec && P.ni == 0) {
    const int g = tile_blocks(P.ntiles);
    if (P.pd == 3) k_lin_obs<3><<<g, kBlock, 0, st>>>(P, cam, pts, radius, Vinv, gp, tile_part);
    else k_lin_obs<4><<<g, kBlock, 0, st>>>(P, cam, pts, radius, Vinv, gp, tile_part);
    if (P.n_diag_items + P.n_blk_items) {
      if (P.pd == 3) k_schur<3><<<P.n_diag_items + P.n_blk_items, kBlock, 0, st>>>(P, rb.S, rb.rhs, rb.colsq, rb.gc, gp);
      else k_schur<4><<<P.n_diag_items + P.n_blk_items, kBlock, 0, st>>>(P, rb.S, rb.rhs, rb.colsq, rb.gc, gp);
    }
    return;
  }
}

// one thread per prior; all updates are FP64 atomics (a handful of priors per camera at most)
__global__ void k_cam_priors(DevProblem P, int mode, const double* __restrict__ cam, const double* __restrict__ cand,
                             const double* __restrict__ y, double* __restrict__ S, double* __restrict__ rhs,
                             double* __restrict__ colsq, double* __restrict__ gc, double* __restrict__ colsq_c,
                             double* __restrict__ scal_cost, double* __restrict__ scal_mcc) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n_priors) return;
  const int c = P.prior_cam[i], kind = P.prior_kind[i];
  const int rc = P.cam_red[c];
  const double* vec = P.prior_vec + 3 * (size_t)i;
  const double* info = P.prior_info + 9 * (size_t)i;
  double r[3], J[18];
  if (mode == PRIOR_FIXED || mode == PRIOR_COST) {
    if ((mode == PRIOR_FIXED) != (rc < 0)) return;
    camera_prior(kind, cam + 6 * (size_t)c, vec, info, false, r, J);
    atomic_add(scal_cost, 0.5 * ((r[0] * r[0] + r[1] * r[1]) + r[2] * r[2]));
    return;
  }
  if (rc < 0) return;
  camera_prior(kind, cam + 6 * (size_t)c, vec, info, true, r, J);
  const unsigned mask = P.cam_mask[c];
  for (int q = 0; q < 6; ++q) {
    const double sc = ((mask >> q) & 1u) ? 0.0 : P.scale_c[6 * c + q];
    J[q] *= sc; J[6 + q] *= sc; J[12 + q] *= sc;
  }
  if (mode == PRIOR_COLNORM) {
    for (int q = 0; q < 6; ++q) atomic_add(&colsq_c[6 * c + q], (J[q] * J[q] + J[6 + q] * J[6 + q]) + J[12 + q] * J[12 + q]);
    return;
  }
  const int base = P.ni + 6 * rc;
  if (mode == PRIOR_LINEARIZE) {
    const int n = P.n;
    for (int a = 0; a < 6; ++a) {
      for (int b = 0; b <= a; ++b)
        atomic_add(&S[(size_t)(base + a) * n + base + b], (J[a] * J[b] + J[6 + a] * J[6 + b]) + J[12 + a] * J[12 + b]);
      const double jr = (J[a] * r[0] + J[6 + a] * r[1]) + J[12 + a] * r[2];
      atomic_add(&rhs[base + a], jr);
      atomic_add(&gc[base + a], jr);
      atomic_add(&colsq[base + a], (J[a] * J[a] + J[6 + a] * J[6 + a]) + J[12 + a] * J[12 + a]);
    }
    atomic_add(scal_cost, 0.5 * ((r[0] * r[0] + r[1] * r[1]) + r[2] * r[2]));
    return;
  }
  // PRIOR_TRIAL: model residual change m = -J y, mcc = -m (r + m / 2); cost at the candidate
  double mcc = 0.0;
  for (int a = 0; a < 3; ++a) {
    double m = 0.0;
    for (int q = 0; q < 6; ++q) m -= J[6 * a + q] * y[base + q];
    mcc -= m * (r[a] + m / 2.0);
  }
  double rc3[3];
  camera_prior(kind, cand + 6 * (size_t)c, vec, info, false, rc3, J);
  atomic_add(scal_mcc, mcc);
  atomic_add(scal_cost, 0.5 * ((rc3[0] * rc3[0] + rc3[1] * rc3[1]) + rc3[2] * rc3[2]));
}

void launch_cam_priors(const DevProblem& P, int mode, const double* cam, const double* cand_cam, const double* y,
                       const ReduceBuf* rb, double* colsq_c, double* scal_cost, double* scal_mcc, hipStream_t st) {
  if (P.n_priors == 0) return;
  k_cam_priors<<<(P.n_priors + 63) / 64, 64, 0, st>>>(P, mode, cam, cand_cam, y, rb ? rb->S : nullptr, rb ? rb->rhs : nullptr,
                                                      rb ? rb->colsq : nullptr, rb ? rb->gc : nullptr, colsq_c, scal_cost, scal_mcc);
}

void launch_reduce_tiles(int ntiles, const double* tile_part, int nfields, const int* field_to_scal,
                         const int* field_is_max, double* scal, hipStream_t st, double* red_part) {
  if (red_part && ntiles > 4 * kReduceBlocks) {   // two fixed-order stages: one workgroup over 50k tile rows costs 50 us
    k_reduce_tiles_stage1<<<kReduceBlocks, 256, 0, st>>>(ntiles, tile_part, nfields, field_is_max, red_part);
    k_reduce_tiles<<<1, 1024, 0, st>>>(kReduceBlocks, red_part, nfields, field_to_scal, field_is_max, scal);
    return;
  }
  k_reduce_tiles<<<1, 1024, 0, st>>>(ntiles, tile_part, nfields, field_to_scal, field_is_max, scal);
}

void launch_reduce_tiles_stage1(int ntiles, const double* tile_part, int nfields, const int* field_is_max, double* out,
                                hipStream_t st) {
  k_reduce_tiles_stage1<<<kReduceBlocks, 256, 0, st>>>(ntiles, tile_part, nfields, field_is_max, out);
}

void launch_finalize_rcs(const DevProblem& P, const double* radius, const ReduceBuf& rb, hipStream_t st, int ntiles,
                         const double* tile_part, const int* f2s, const int* fmaxflag, const uint8_t* tile_cls, int want_cls) {
  k_finalize_rcs<<<std::max(1, (P.n + 1023) / 1024), 1024, 0, st>>>(P, radius, rb.S, rb.colsq, rb.gc, rb.scal, ntiles, tile_part, f2s, fmaxflag,
                                                                     tile_cls, want_cls);
}

// 1 / 2 when launch_cam_update writes the candidate's per-camera blocks itself (launch_backsub then skips its k_cam_prep)
static int cam_update_preps(const DevProblem& P) {
  if (P.n_fruns == 0 || !P.camrot || !P.camrot_cand || P.ntiles == 0 || getenv("THEIA_HIP_CAM_PREP_SEPARATE")) return 0;
  if (!P.ni) return P.camdir ? 1 : 0;
  return P.fused_bw > 0 ? 2 : 0;
}

void launch_cam_update(const DevProblem& P, const double* cam, const double* y, double* cand_cam,
                       double* cand_intr, double* out_stepsq, double* out_xnormsq, hipStream_t st, double* zero16) {
  const int prep = cam_update_preps(P);
  if (prep == 1) k_cam_update<1><<<1 + (P.nc + 255) / 256, 1024, 0, st>>>(P, cam, y, cand_cam, cand_intr, out_stepsq, out_xnormsq, zero16);
  else if (prep == 2) k_cam_update<2><<<1 + (P.nc + 255) / 256, 1024, 0, st>>>(P, cam, y, cand_cam, cand_intr, out_stepsq, out_xnormsq, zero16);
  else k_cam_update<0><<<1, 1024, 0, st>>>(P, cam, y, cand_cam, cand_intr, out_stepsq, out_xnormsq, zero16);
}

void launch_backsub(const DevProblem& P, const double* cam, const double* pts, const double* cand_cam,
                    double* cand_pts, const double* yc, const double* Vinv, double* tile_part,
                    double* scal, hipStream_t st) {
  (void)scal;
  if (P.ntiles == 0) return;
  const double* ycc = yc + P.ni;  // camera part of the solution
  if (!P.ni && P.n_fruns > 0 && P.camrot && P.camrot_cand) {   // fused path: the state's blocks are in P.camrot already
    // the candidate's blocks + the cameras' steps as {D, v} (P.camdir): written by launch_cam_update's extra workgroups
    if (!cam_update_preps(P)) launch_cam_prep(P, cand_cam, P.intr, P.camrot_cand, st, ycc);
    if (launch_backsub_runs(P, pts, cand_pts, Vinv, tile_part, st)) return;   // round 5: per-run camera blocks in LDS, prefetched stream
    if (P.pd == 3) k_backsub<3, false, true><<<tile_blocks(P.ntiles), kBlock, 0, st>>>(P, cam, pts, cand_cam, cand_pts, ycc, Vinv, tile_part);
    else k_backsub<4, false, true><<<tile_blocks(P.ntiles), kBlock, 0, st>>>(P, cam, pts, cand_cam, cand_pts, ycc, Vinv, tile_part);
    return;
  }
  if (P.ni && P.fused_bw > 0 && P.n_fruns > 0 && P.camrot && P.camrot_cand) {
    // fused path with intrinsics: the state's blocks are in P.camrot; the candidate cameras with the candidate intrinsics
    if (cam_update_preps(P) != 2) launch_cam_prep(P, cand_cam, P.intr_cand, P.camrot_cand, st);   // (else: launch_cam_update's extra workgroups)
    if (launch_backsub_runs_intr(P, pts, cand_pts, Vinv, tile_part, yc, st)) return;   // round 5: over the runs, camera blocks in LDS
    constexpr unsigned kFR = (1u << 0) | (3u << 5);   // FOCAL_LENGTH | RADIAL_DISTORTION (ba_fused_intr.hip: kMaskFocalRadial)
    if (P.fused_kmask == kFR) {
      if (P.pd == 3) k_backsub<3, true, true, kFR><<<tile_blocks(P.ntiles), kBlock, 0, st>>>(P, cam, pts, cand_cam, cand_pts, ycc, Vinv, tile_part);
      else k_backsub<4, true, true, kFR><<<tile_blocks(P.ntiles), kBlock, 0, st>>>(P, cam, pts, cand_cam, cand_pts, ycc, Vinv, tile_part);
    } else {
      if (P.pd == 3) k_backsub<3, true, true><<<tile_blocks(P.ntiles), kBlock, 0, st>>>(P, cam, pts, cand_cam, cand_pts, ycc, Vinv, tile_part);
      else k_backsub<4, true, true><<<tile_blocks(P.ntiles), kBlock, 0, st>>>(P, cam, pts, cand_cam, cand_pts, ycc, Vinv, tile_part);
    }
    return;
  }
  if (P.ni) {
    if (P.pd == 3) k_backsub<3, true><<<tile_blocks(P.ntiles), kBlock, 0, st>>>(P, cam, pts, cand_cam, cand_pts, ycc, Vinv, tile_part);
    else k_backsub<4, true><<<tile_blocks(P.ntiles), kBlock, 0, st>>>(P, cam, pts, cand_cam, cand_pts, ycc, Vinv, tile_part);
  } else {
    if (P.pd == 3) k_backsub<3, false><<<tile_blocks(P.ntiles), kBlock, 0, st>>>(P, cam, pts, cand_cam, cand_pts, ycc, Vinv, tile_part);
    else k_backsub<4, false><<<tile_blocks(P.ntiles), kBlock, 0, st>>>(P, cam, pts, cand_cam, cand_pts, ycc, Vinv, tile_part);
  }
}

void launch_evaluate(const DevProblem& P, const double* cam, const double* pts, double* residuals,
                     double* jac_cam, double* jac_pt, uint8_t* valid, double* tile_part, hipStream_t st, double* jac_intr) {
  if (P.ntiles == 0) return;
  if (P.ni) {
    if (P.pd == 3) k_evaluate<3, true, true><<<tile_blocks(P.ntiles), kBlock, 0, st>>>(P, cam, pts, residuals, jac_cam, jac_pt, valid, tile_part, jac_intr);
    else k_evaluate<4, true, true><<<tile_blocks(P.ntiles), kBlock, 0, st>>>(P, cam, pts, residuals, jac_cam, jac_pt, valid, tile_part, jac_intr);
    return;
  }
  if (P.pd == 3)
    k_evaluate<3, true><<<tile_blocks(P.ntiles), kBlock, 0, st>>>(P, cam, pts, residuals, jac_cam, jac_pt, valid, tile_part);
  else
    k_evaluate<4, true><<<tile_blocks(P.ntiles), kBlock, 0, st>>>(P, cam, pts, residuals, jac_cam, jac_pt, valid, tile_part);
}

void launch_cost_only(const DevProblem& P, const double* cam, const double* pts, double* tile_part,
                      double* scal, hipStream_t st) {
  (void)scal;
  if (P.ntiles == 0) return;
  if (P.pd == 3)
    k_evaluate<3, false><<<tile_blocks(P.ntiles), kBlock, 0, st>>>(P, cam, pts, nullptr, nullptr, nullptr, nullptr, tile_part);
  else
    k_evaluate<4, false><<<tile_blocks(P.ntiles), kBlock, 0, st>>>(P, cam, pts, nullptr, nullptr, nullptr, nullptr, tile_part);
}

}  // namespace thip

namespace thip {

// ------------------------------------------------------------ long-track path
namespace {
LongView long_view(const DevProblem& P) {
  LongView v;
  v.nobs = P.long_nobs; v.ntracks = P.long_ntracks; v.obs_index = P.long_obs_index; v.obs_slot = P.long_obs_slot;
  v.track_start = P.long_track_start; v.track_pt = P.long_track_pt;
  return v;
}
template <int PD> constexpr int long_stride() { return PD * (PD + 1) / 2 + PD; }
}  // namespace

void launch_long_colnorm(const DevProblem& P, const double* cam, const double* pts, double* colsq_c, double* colsq_p,
                         double* scratch, hipStream_t st, double* colsq_i) {
  if (P.long_nobs == 0) return;
  const LongView v = long_view(P);
  const int gb = (v.nobs + 127) / 128, tb = (v.ntracks + 127) / 128;
#define THIP_LONG(PD_, INTR_)                                                                                          \
  do {                                                                                                                 \
    (void)hipMemsetAsync(scratch, 0, sizeof(double) * long_stride<PD_>() * v.ntracks, st);                             \
    k_long_accum<PD_, 0, INTR_><<<gb, 128, 0, st>>>(P, v, cam, pts, scratch, nullptr, nullptr, nullptr, nullptr, nullptr, colsq_c, colsq_i); \
    k_long_track<PD_, 0><<<tb, 128, 0, st>>>(P, v, nullptr, scratch, colsq_p, nullptr, nullptr, nullptr);              \
  } while (0)
  if (P.pd == 3) { if (P.ni) THIP_LONG(3, true); else THIP_LONG(3, false); }
  else { if (P.ni) THIP_LONG(4, true); else THIP_LONG(4, false); }
#undef THIP_LONG
}

void launch_long_linearize(const DevProblem& P, const double* cam, const double* pts, const double* radius, const ReduceBuf& rb,
                           double* Vinv, double* gp, double* scratch, hipStream_t st) {
  if (P.long_nobs == 0) return;
  const LongView v = long_view(P);
  const int gb = (v.nobs + 127) / 128, tb = (v.ntracks + 127) / 128;
#define THIP_LONG(PD_, INTR_)                                                                                          \
  do {                                                                                                                 \
    (void)hipMemsetAsync(scratch, 0, sizeof(double) * long_stride<PD_>() * v.ntracks, st);                             \
    k_long_accum<PD_, 1, INTR_><<<gb, 128, 0, st>>>(P, v, cam, pts, scratch, rb.S, rb.rhs, rb.colsq, rb.gc, rb.scal, nullptr, nullptr); \
    k_long_track<PD_, 1><<<tb, 128, 0, st>>>(P, v, radius, scratch, nullptr, Vinv, gp, rb.scal);                       \
    k_long_schur<PD_, INTR_><<<gb, 128, 0, st>>>(P, v, cam, pts, Vinv, gp, rb.S, rb.rhs);                              \
  } while (0)
  if (P.pd == 3) { if (P.ni) THIP_LONG(3, true); else THIP_LONG(3, false); }
  else { if (P.ni) THIP_LONG(4, true); else THIP_LONG(4, false); }
#undef THIP_LONG
}

// yc = the whole solution vector of the reduced system (intrinsics columns first)
void launch_long_backsub(const DevProblem& P, const double* cam, const double* pts, const double* cand_cam,
                         double* cand_pts, const double* yc, const double* Vinv, double* scratch, double* scalB,
                         hipStream_t st) {
  if (P.long_nobs == 0) return;
  const LongView v = long_view(P);
  const int gb = (v.nobs + 127) / 128;
#define THIP_LONG(PD_, INTR_)                                                                                          \
  do {                                                                                                                 \
    (void)hipMemsetAsync(scratch, 0, sizeof(double) * long_stride<PD_>() * v.ntracks, st);                             \
    k_long_back1<PD_, INTR_><<<gb, 128, 0, st>>>(P, v, cam, pts, yc, scratch);                                         \
    k_long_back2<PD_, INTR_><<<gb, 128, 0, st>>>(P, v, cam, pts, cand_cam, cand_pts, yc, Vinv, scratch, scalB);        \
  } while (0)
  if (P.pd == 3) { if (P.ni) THIP_LONG(3, true); else THIP_LONG(3, false); }
  else { if (P.ni) THIP_LONG(4, true); else THIP_LONG(4, false); }
#undef THIP_LONG
}

}  // namespace thip
