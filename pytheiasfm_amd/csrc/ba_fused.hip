// ba_fused.hip -- K1 + K2 of the BA hot path in ONE pass over the observations (gfx950, FP64).
//
// What the reference computes here: the Jacobian blocks of every residual
// (src/theia/sfm/camera/reprojection_error.h:54-110 through ceres AutoDiff) and the Schur complement that
// eliminates the point blocks (ceres SchurEliminator for the ordering of bundle_adjuster.cc:547-577):
//     S = sum_c F^T F  -  sum_tracks sum_{a,b in track} (F_a^T E_a) V^-1 (E_b^T F_b),   rhs = F^T r - W V^-1 g.
//
// The first generation (k_lin_obs + k_schur, ba_kernels.hip) wrote a 256-B record per observation to HBM and
// re-read every record once per partner observation of its track: 525 MB moved for 14 MB of algorithmic
// traffic at 200 views / 50k tracks.  Here the records never leave the CU:
//
//   * a workgroup owns a RUN of consecutive wave tiles (tracks are ordered by their first variable camera, so a
//     run sees a short window of cameras: <= 22, its "local cameras");
//   * the per-camera blocks (k_cam_prep) of the run's local cameras and of the constant cameras its tracks see are
//     staged in LDS once per run;
//   * per sub-chunk of 4 tiles (<= 256 observations, <= 128 tracks), phase L: lane = observation linearises it
//     (closed-form Jacobians; its stream was loaded during the previous sub-chunk's phase S), the track sums
//     V_p / g_p go through the lanes' own LDS record slots, V_p^-1 = Li^T Li, and the lane leaves
//     F (2 x 6), Ehat = E Li^T (2 x PD), r, r - Ehat ghat in LDS, plus its slot in the track's row of a
//     (track, local camera) -> slot table;
//   * phase S: lane = TARGET BLOCK (la, lb) of the run's local camera pairs.  Every wave walks the tracks of its
//     slice; for each track the lane looks up the two slots and accumulates F_a^T (Ehat_a Ehat_b^T - [a = b] I) F_b
//     into 36 REGISTERS.
//     All lanes of a wave read the <= L records of one track: consecutive LDS slots, conflict free, mostly
//     broadcasts.  The per-observation terms (F^T F, F^T r - What ghat, gradient, column norms) go to lanes
//     (local camera, row) the same way;
//   * at the end of the run the slices are combined through LDS in a fixed order and the run's blocks go to a
//     partial-sum buffer; k_schur_sum adds the partial blocks of each S block in run order and WRITES it.
// No atomics, no pair lists, fixed summation order (bitwise reproducible).
//
// HBM traffic per LM iteration: the 24 B / observation stream + one byte pair of plan indices, the parameters,
// V^-1 out, and the partial blocks (ntgt * 288 B + W * 432 B per run, written and read once).
#define THIP_LEAN_SQRT 1   // ba_device.h: fsqrt() without range scaling / class selects
#define THIP_LEAN_SINCOS 1 // ba_device.h: SphereManifold::Plus with sin(t) / t and cos(t) as polynomials in t^2 for small steps
#include "ba_fused_lin.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>

namespace thip {
namespace {

constexpr int kRowBytes = 24;                         // slot-table row: kFusedMaxCams rounded up to 8
static_assert(kFusedMaxCams <= kRowBytes, "slot-table row too short");

// ycam (candidate launch only): the camera part of the solved step y; the step of every camera is then also left as
// {D, v} in P.camdir (camera_step_direction, from the STATE's rotation terms in P.camrot) for k_backsub.
__global__ __launch_bounds__(256) void k_cam_prep(DevProblem P, const double* __restrict__ cam, const double* __restrict__ intr,
                                                  double* __restrict__ camrot, const double* __restrict__ ycam) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0 && P.frun_next) { P.frun_next[0] = 0; P.frun_next[1] = 0; }   // heads of the run queues of k_lin_schur / k_backsub_runs (the next kernels on the stream)
  if (c >= P.nc) return;
  cam_prep_one(P, c, cam + 6 * (size_t)c, intr, camrot);
  if (ycam && P.camdir) {
    const int rc = P.cam_red[c];
    const unsigned mask = P.cam_mask[c];
    double dl[6], ext[6], out[12];
    for (int q = 0; q < 6; ++q)   // exactly the step k_cam_update added to the free columns
      dl[q] = (rc >= 0 && !((mask >> q) & 1u)) ? (-ycam[6 * rc + q]) * P.scale_c[6 * c + q] : 0.0;
    RotTerms rt;
    camrot_load(P.camrot + (size_t)kCamRot * c, ext, rt);
    camera_step_direction(ext + 3, rt, dl, dl + 3, out);
    for (int k = 0; k < 12; ++k) P.camdir[(size_t)12 * c + k] = out[k];
  }
}

// Track-local OR of one int per lane (log-step, as segment_allsum_log); every lane of the track gets the result.
THIP_DEV unsigned segment_or(const Segment& s, int lane, unsigned v) {
  const int pos = lane - s.start;
  for (int d = 1; d < s.maxlen; d <<= 1) {
    const unsigned o = (unsigned)__shfl_down((int)v, d, kWave);
    if (pos + d < s.len) v |= o;
  }
  return (unsigned)__shfl((int)v, s.start, kWave);
}

// LDS record of one observation: {F (2 x 6) | Ehat = E Li^T (2 x PD) | r (2)}, padded to an odd number of 16-B pieces
// (lanes reading consecutive records then fall on distinct bank groups).  What = F^T Ehat is never formed: the block
// product is  What_a What_b^T = F_a^T (Ehat_a Ehat_b^T) F_b  -- a 2 x 2 core between the two camera Jacobians,
// the same 108 FMAs (PD = 3) from 20 stored doubles instead of 32.
#ifndef THIP_FUSED_WAVES
#define THIP_FUSED_WAVES 2   // workgroups per CU the register allocation aims at (63 KB of LDS per workgroup: two fit)
#endif
template <int PD> constexpr int rec_doubles() { return PD == 3 ? 22 : 26; }   // 12 + 2 PD + 2 + 2 used

// Phase S of one sub-chunk: lane = target block (la, lb) of the track slice it serves; per-observation terms by the
// lanes (local camera, row).  acc[] / dacc[] only ever see constant indices (they are registers of the caller).
template <int PD, int RD>
THIP_DEV void fused_phase_s(const double* __restrict__ s_rec, const uint8_t* __restrict__ s_tslot, const unsigned* __restrict__ s_tmask,
                            int ntr, int tstride, int t0, bool slice_ok, unsigned tbits, int la, int lb, double diag_core,
                            int dP, int dbase, unsigned dbit, int dlc, int da, double (&acc)[36], double (&dacc)[3]) {
#pragma unroll 1
  for (int base = 0; base < ntr; base += tstride) {   // wave-uniform trip count
    {
      const int t = base + t0;
      const unsigned mask = (slice_ok && t < ntr) ? s_tmask[t] : 0u;
      if ((mask & tbits) == tbits) {
        const unsigned ca = s_tslot[t * kRowBytes + la], cb = s_tslot[t * kRowBytes + lb];
        double Fa[12], Ea[2 * PD], Fb[12], Eb[2 * PD];
        const double2* pa = reinterpret_cast<const double2*>(s_rec + ca * RD);
        const double2* pb = reinterpret_cast<const double2*>(s_rec + cb * RD);
#pragma unroll
        for (int q = 0; q < 6; ++q) { const double2 u = pa[q]; Fa[2 * q] = u.x; Fa[2 * q + 1] = u.y; }
#pragma unroll
        for (int q = 0; q < PD; ++q) { const double2 u = pa[6 + q]; Ea[2 * q] = u.x; Ea[2 * q + 1] = u.y; }
#pragma unroll
        for (int q = 0; q < 6; ++q) { const double2 u = pb[q]; Fb[2 * q] = u.x; Fb[2 * q + 1] = u.y; }
#pragma unroll
        for (int q = 0; q < PD; ++q) { const double2 u = pb[6 + q]; Eb[2 * q] = u.x; Eb[2 * q + 1] = u.y; }
        double M[2][2];   // Ehat_a Ehat_b^T
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            double sm = 0.0;
#pragma unroll
            for (int q = 0; q < PD; ++q) sm += Ea[i * PD + q] * Eb[j * PD + q];
            M[i][j] = sm;
          }
        M[0][0] -= diag_core; M[1][1] -= diag_core;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
          const double t0v = Fa[a] * M[0][0] + Fa[6 + a] * M[1][0];   // (F_a^T M)[a][0..1]
          const double t1v = Fa[a] * M[0][1] + Fa[6 + a] * M[1][1];
#pragma unroll
          for (int b2 = 0; b2 < 6; ++b2)   // two chained FMAs: `acc += x y + z w` compiles to mul + fma + add (no reassociation)
            acc[a * 6 + b2] = __builtin_fma(t1v, Fb[6 + b2], __builtin_fma(t0v, Fb[b2], acc[a * 6 + b2]));
        }
      }
    }
    // per-observation terms: lane = (local camera, row), over the dP tracks this wave serves in the step
#pragma unroll 1
    for (int g2 = 0; g2 < dP; ++g2) {
      const int t = base + dbase + g2;
      const unsigned mask = (t < ntr) ? s_tmask[t] : 0u;
      if (mask & dbit) {
        const unsigned sd = s_tslot[t * kRowBytes + dlc];
        const double2* px = reinterpret_cast<const double2*>(s_rec + sd * RD);
        const double2 rr = px[6 + PD], rv = px[6 + PD + 1];   // r,  r - Ehat ghat
        const double fa0 = s_rec[sd * RD + da], fa1 = s_rec[sd * RD + 6 + da];   // column da of F (no dynamic register indexing)
        dacc[0] = __builtin_fma(fa1, rv.y, __builtin_fma(fa0, rv.x, dacc[0]));
        dacc[1] = __builtin_fma(fa1, rr.y, __builtin_fma(fa0, rr.x, dacc[1]));
        dacc[2] = __builtin_fma(fa1, fa1, __builtin_fma(fa0, fa0, dacc[2]));
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// Round 5: the same pass with (i) the per-camera blocks of a run staged in LDS once per run -- its W local cameras and
// the constant cameras its tracks see (FusedRun::stage_off / nstage) -- instead of a 320-B gather from HBM per
// observation, (ii) the observation stream of the NEXT sub-chunk (uv, point index, plan bytes; then the point, its
// Jacobi scaling and constant flag) loaded while the current one is linearised / multiplied, so that phase L starts on
// registers, (iii) the problem read through the kernel arguments (scalar loads), no LDS copy of it.
// Development instrumentation (THEIA_HIP_FUSED_STAMPS=1 selects an instrumented instance of the kernel): s_memtime deltas
// summed over all waves, per section of the kernel.
__device__ unsigned long long g_fused_stamps[16];

// Phase L of one wave tile on prefetched registers: the linearisation (lin5, ba_fused_lin.h), the track sums, V^-1, the record
// and the slot table.
template <int PD, unsigned MODELS, int LOSSK, bool STAMPS>
THIP_DEV void fused_phase_l5(const DevProblem& P, const LanePre<PD>& c, const double* __restrict__ s_cam, int W, int tile, bool tile_ok,
                             int wv, int lane, double inv_radius, double* __restrict__ Vinv, double* __restrict__ tile_part,
                             double* __restrict__ s_rec, uint8_t* __restrict__ s_tslot, unsigned* __restrict__ s_tmask,
                             unsigned long long (&stamps)[12]) {
  Stamp<STAMPS> sl;
  sl.start();
  constexpr int NT = PD * (PD + 1) / 2;
  constexpr int RD = rec_doubles<PD>();
  const bool active = c.active;
  const bool is_tgt = !(c.lc & 0x80u);
  const unsigned cslot = is_tgt ? c.lc : (unsigned)W + (c.lc & 0x7fu);
  const double* cb = s_cam + cslot * kCamLds;
  double r[2], cost, Jc[12], Jt[2 * PD];
  bool valid;
  lin5<PD, MODELS, LOSSK, false>(P, c, cb, r, cost, valid, Jc, Jt, nullptr);
  if constexpr (STAMPS) { asm volatile("" :: "v"(Jt[0]), "v"(Jc[11]), "v"(r[1])); }
  sl.lap(stamps, 2);
  const int pseg = active ? c.p : -1 - lane;   // inactive lanes are their own segment
  const Segment sg = lane_segment_all(pseg, lane);
  double tot[NT + PD];
#pragma unroll
  for (int a = 0; a < PD; ++a) {
#pragma unroll
    for (int b = 0; b <= a; ++b) tot[lidx(a, b)] = Jt[a] * Jt[b] + Jt[PD + a] * Jt[PD + b];
    tot[NT + a] = Jt[a] * r[0] + Jt[PD + a] * r[1];
  }
  // Track sums: every lane leaves its NT + PD products (and its local-camera bit) in its OWN record slot -- free until the
  // record is written below -- and every lane of a track then adds the slots of the track's lanes in lane order: <= maxlen
  // steps of five wide LDS reads (all lanes of a track read one address: broadcasts) against the 92 ds_bpermute of the
  // log-step scans, which the LDS serves at ~16 cycles each (the stamps showed them costing as much as the whole
  // linearisation).  Same bits in every lane of the track (same operands, same order).
  unsigned tmask = 0u;
  {
    constexpr int NS = NT + PD;                       // 9 (PD = 3) or 14 doubles, then one word for the camera bit
    double* mine = s_rec + (wv * 64 + lane) * RD;
#pragma unroll
    for (int k = 0; k + 1 < NS; k += 2) *reinterpret_cast<double2*>(mine + k) = make_double2(tot[k], tot[k + 1]);
    if constexpr (NS & 1) mine[NS - 1] = tot[NS - 1];
    reinterpret_cast<unsigned*>(mine + NS)[0] = (active && is_tgt) ? (1u << c.lc) : 0u;
#pragma unroll
    for (int k = 0; k < NS; ++k) tot[k] = 0.0;
    for (int j = 0; j < sg.maxlen; ++j) {             // wave-uniform trip count
      const bool take = j < sg.len;
      const double* oth = s_rec + (wv * 64 + min(sg.start + j, 63)) * RD;
      double v[NS];
#pragma unroll
      for (int k = 0; k + 1 < NS; k += 2) { const double2 u = *reinterpret_cast<const double2*>(oth + k); v[k] = u.x; v[k + 1] = u.y; }
      if constexpr (NS & 1) v[NS - 1] = oth[NS - 1];
      const unsigned ob = reinterpret_cast<const unsigned*>(oth + NS)[0];
      if (take) {
#pragma unroll
        for (int k = 0; k < NS; ++k) tot[k] += v[k];
        tmask |= ob;
      }
    }
  }
  if constexpr (STAMPS) { asm volatile("" :: "v"(tot[0]), "v"(tot[NT + PD - 1]), "v"(tmask)); }
  sl.lap(stamps, 3);
  double V[NT], Vi[NT], g[PD];
#pragma unroll
  for (int k = 0; k < NT; ++k) V[k] = tot[k];
#pragma unroll
  for (int a = 0; a < PD; ++a) { g[a] = tot[NT + a]; V[lidx(a, a)] += fmin(fmax(V[lidx(a, a)], 1e-6), 1e32) * inv_radius; }
  bool pd_ok = true;
  double Li[PD][PD];
#pragma unroll
  for (int a = 0; a < PD; ++a)
#pragma unroll
    for (int b = 0; b < PD; ++b) Li[a][b] = 0.0;
  const bool pvar = active && !c.pconst;
  if (pvar) pd_ok = invert_spd<PD>(V, Vi, Li);
  if (!pvar || !pd_ok) {
#pragma unroll
    for (int k = 0; k < NT; ++k) Vi[k] = 0.0;
#pragma unroll
    for (int a = 0; a < PD; ++a)
#pragma unroll
      for (int b = 0; b < PD; ++b) Li[a][b] = 0.0;
  }
  double gmax = 0.0;
  double gh[PD];   // ghat = Li g, in every lane of the track (the record carries r - Ehat ghat)
#pragma unroll
  for (int a = 0; a < PD; ++a) {
    double sm = 0.0;
#pragma unroll
    for (int k = 0; k <= a; ++k) sm += Li[a][k] * g[k];
    gh[a] = sm;
  }
  if constexpr (STAMPS) { asm volatile("" :: "v"(gh[0]), "v"(gh[PD - 1]), "v"(Vi[0])); }
  sl.lap(stamps, 4);
  if (active && sg.head) {
    s_tmask[c.tl] = tmask;
    if (!c.pconst) {
#pragma unroll
      for (int k = 0; k < NT; ++k) Vinv[(size_t)NT * c.p + k] = Vi[k];
#pragma unroll
      for (int a = 0; a < PD; ++a) gmax = fmax(gmax, fabs(g[a] / c.sp[a]));
    }
  }
  if (active && is_tgt) {
    const int slot = wv * 64 + lane;
    double eh[2 * PD];   // Ehat = E Li^T
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int b = 0; b < PD; ++b) {
        double sm = 0.0;
#pragma unroll
        for (int k = 0; k <= b; ++k) sm += Jt[i * PD + k] * Li[b][k];
        eh[i * PD + b] = sm;
      }
    double2* Rr = reinterpret_cast<double2*>(s_rec + slot * RD);
#pragma unroll
    for (int k = 0; k < 6; ++k) Rr[k] = make_double2(Jc[2 * k], Jc[2 * k + 1]);
#pragma unroll
    for (int k = 0; k < PD; ++k) Rr[6 + k] = make_double2(eh[2 * k], eh[2 * k + 1]);
    Rr[6 + PD] = make_double2(r[0], r[1]);
    double v1[2];   // r - Ehat ghat: the camera's rhs row is F^T (r - Ehat ghat)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      double sm = 0.0;
#pragma unroll
      for (int b = 0; b < PD; ++b) sm += eh[i * PD + b] * gh[b];
      v1[i] = r[i] - sm;
    }
    Rr[6 + PD + 1] = make_double2(v1[0], v1[1]);
    s_tslot[c.tl * kRowBytes + c.lc] = (uint8_t)slot;
  }
  sl.lap(stamps, 5);
  const double costw = wave_sum_all(cost);
  gmax = wave_max_all(gmax);
  const double inval = wave_count(active && !valid);
  const double npd = wave_count(active && !pd_ok && sg.head);
  if (lane == 0 && tile_ok) {
    tile_part[4 * (size_t)tile + 0] = costw;
    tile_part[4 * (size_t)tile + 1] = gmax;
    tile_part[4 * (size_t)tile + 2] = inval;
    tile_part[4 * (size_t)tile + 3] = npd;
  }
  sl.lap(stamps, 6);
}

// tile_part layout as k_lin_obs: [ntiles][4] = {cost, gmax_points, invalid, notpd}
template <int PD, int TPS, unsigned MODELS, int LOSSK, bool STAMPS = false>
__global__ __launch_bounds__(64 * TPS, THIP_FUSED_WAVES) void k_lin_schur(DevProblem P, const double* __restrict__ pts,
                                                           const double* __restrict__ radius_p,
                                                           double* __restrict__ Vinv, double* __restrict__ tile_part) {
  constexpr int RD = rec_doubles<PD>();
  constexpr int SUB = TPS * kWave;                    // observations per sub-chunk
  constexpr int SUBT = TPS * kFusedTileTracks;        // tracks per sub-chunk
  constexpr int NWV = TPS;                            // waves of the workgroup
  __shared__ __attribute__((aligned(16))) double s_rec[SUB * RD];
  __shared__ __attribute__((aligned(16))) double s_cam[kFusedMaxStage * kCamLds];
  __shared__ uint8_t s_tslot[SUBT * kRowBytes];   // (track, local camera) -> record slot, valid where the mask bit is set
  __shared__ unsigned s_tmask[SUBT];              // local cameras of a track
  static_assert(SUB * 18 <= SUB * RD, "slice-combination scratch does not fit the record buffer");
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const double radius = *radius_p;
  const double inv_radius = 1.0 / radius;
  __shared__ int s_next;
  unsigned long long stamps[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) stamps[k] = 0;
  Stamp<STAMPS> sk;
  sk.start();
  // runs are taken from a queue, longest first (ba_solver.hip), ONE RUN AHEAD: the pop's round trip -- and, behind it in the
  // in-order memory counter, the drain of the previous run's partial-block stores -- is off the path between two runs
  int pending = 0;
  if (tid == 0) pending = atomicAdd(P.frun_next, 1);
  for (;;) {
  __syncthreads();   // the previous run's slice combination is done with the LDS scratch (and s_next, s_cam)
  if (tid == 0) s_next = pending;
  __syncthreads();
  const int rix = __builtin_amdgcn_readfirstlane(s_next);   // (scalar: the run's fields and the tile geometry are s_loads)
  if (rix >= P.n_fruns) break;
  if (tid == 0) pending = atomicAdd(P.frun_next, 1);
  const FusedRun run = P.fruns[P.frun_order[rix]];
  const int nsc = (run.ntiles + TPS - 1) / TPS;
  // ---- the run's per-camera blocks (k_cam_prep) -> LDS, 16 B per thread and step
  for (int j = tid; j < run.nstage * (kCamRot / 2); j += SUB) {
    const int k = j / (kCamRot / 2), piece = j - k * (kCamRot / 2);
    const int cidx = P.frun_stage[run.stage_off + k];
    reinterpret_cast<double2*>(s_cam + k * kCamLds)[piece] = reinterpret_cast<const double2*>(P.camrot + (size_t)kCamRot * cidx)[piece];
  }
  LanePre<PD> cur;
  int tile = 0; bool tile_ok = false;
  pre_level1<PD, TPS>(P, run, 0, wv, lane, tile, tile_ok, cur);
  pre_level2<PD>(P, pts, cur);
  __syncthreads();   // s_cam complete

  // ---- phase-S role: which target block / per-camera row this lane owns, which tracks it walks
  const int G = run.gp & 0xff, PS = run.gp >> 8; // waves per slice (1, 2, 4) / slices per wave (>= 1, only with G == 1)
  int tix, t0, tstride;
  bool slice_ok = true;
  const int B = 64 / PS;
  if (G == 1) { const int g = lane / B; tix = lane - g * B; slice_ok = g < PS; t0 = wv * PS + g; tstride = NWV * PS; }
  else { tix = (wv % G) * 64 + lane; t0 = wv / G; tstride = NWV / G; }
  const bool has_tgt = slice_ok && tix < run.ntgt;
  int la = 0, lb = 0;
  if (has_tgt) { const unsigned us = P.frun_tgt[run.tgt_off + tix]; la = us & 0xffu; lb = us >> 8; }
  const int dix = (G == 1) ? lane : tix;         // per-camera rows: lanes of the wave (G == 1) / of the wave group
  const bool has_d = dix < 6 * run.W;
  const int dlc = has_d ? dix / 6 : 0, da = dix % 6;
  const int dP = (G == 1) ? PS : 1, dbase = (G == 1) ? wv * PS : wv / G;
  const unsigned tbits = has_tgt ? ((1u << la) | (1u << lb)) : 0xffffffffu;   // no target: never a subset (bit 31 unused)
  const unsigned dbit = has_d ? (1u << dlc) : 0x80000000u;
  // a DIAGONAL target (la == lb) accumulates  What What^T - F^T F  (its 2 x 2 core is  Ehat Ehat^T - I): the camera's
  // F^T F rides the pair products, the per-observation lanes keep three sums (rhs, gradient, column norm) only
  const double diag_core = (has_tgt && la == lb) ? 1.0 : 0.0;
  double acc[36], dacc[3];
#pragma unroll
  for (int k = 0; k < 36; ++k) acc[k] = 0.0;
#pragma unroll
  for (int k = 0; k < 3; ++k) dacc[k] = 0.0;

  sk.lap(stamps, 0);
  for (int sc = 0; sc < nsc; ++sc) {
    // ------------------------------------------------------------------ phase L: lane = observation
    LanePre<PD> nxt;
    int ntile = 0; bool ntile_ok = false;
    const int scn = min(sc + 1, nsc - 1);   // (the last sub-chunk reloads itself: unconditional loads, nothing is used)
    pre_level1<PD, TPS>(P, run, scn, wv, lane, ntile, ntile_ok, nxt);
    if (P.fused_dbg & 16) __builtin_amdgcn_s_setprio(1);   // development: issue priority to the latency-bound phase
    fused_phase_l5<PD, MODELS, LOSSK, STAMPS>(P, cur, s_cam, run.W, tile, tile_ok, wv, lane, inv_radius, Vinv, tile_part, s_rec, s_tslot, s_tmask, stamps);
    sk.lap(stamps, 1);
    pre_level2<PD>(P, pts, nxt);
    if (P.fused_dbg & 16) __builtin_amdgcn_s_setprio(0);
    // phase S is pure issue (FMAs on registers and LDS words), phase L is a chain of latencies: with the co-resident workgroup's
    // wave in the other phase, giving S the issue slots first and letting L fill the gaps measured 0.3725 against 0.3958 ms
    if (!(P.fused_dbg & 32)) __builtin_amdgcn_s_setprio(1);
    __syncthreads();
    sk.lap(stamps, 7);
    // ------------------------------------------------------------------ phase S: lane = target block
    if (!(P.fused_dbg & 1)) {
      const int last_tile = min(run.tile0 + TPS * sc + TPS - 1, run.tile0 + run.ntiles - 1);
      const int ntr = P.tile_trk_end[last_tile];
      fused_phase_s<PD, RD>(s_rec, s_tslot, s_tmask, ntr, tstride, t0, slice_ok, tbits, la, lb, diag_core, dP, dbase, dbit, dlc, da, acc, dacc);
    }
    sk.lap(stamps, 8);
    if (!(P.fused_dbg & 32)) __builtin_amdgcn_s_setprio(0);
    __syncthreads();
    sk.lap(stamps, 9);
    if constexpr (STAMPS) stamps[11] += 1;
    cur = nxt; tile = ntile; tile_ok = ntile_ok;
  }
  // ---------------------------------------------------------------- combine the track slices, fixed order
  double* scratch = s_rec;   // SUB x 18 doubles
  double* out = P.fpart + run.part_off;
  const int nrep = (G == 1) ? NWV * PS : NWV / G;
#pragma unroll
  for (int h = 0; h < 2; ++h) {   // fully unrolled: acc[] must only ever see constant indices (else it lives in scratch)
    if (h) __syncthreads();
#pragma unroll
    for (int q = 0; q < 18; ++q) scratch[tid * 18 + q] = acc[18 * h + q];
    __syncthreads();
    // every ENTRY (target, q) of the half blocks is summed over the target's replicas, in slice order (the order, and so the
    // bits, of the one-lane-per-target loop this replaces: that left ntgt of the 256 lanes adding nrep - 1 rows of 18 each),
    // and leaves for the partial-sum buffer at once: consecutive threads on consecutive doubles of a 144-B half block
    for (int e = tid; e < run.ntgt * 18; e += SUB) {
      const int k = e / 18, q = e - 18 * k;
      double v = scratch[k * 18 + q];   // (replica 0 of target k is thread k)
#pragma unroll 1
      for (int r = 1; r < nrep; ++r) {
        const int oth = (G == 1) ? ((r / PS) * 64 + (r % PS) * B + k) : (k + r * G * 64);
        v += scratch[oth * 18 + q];
      }
      out[(size_t)k * 36 + 18 * h + q] = v;
    }
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 3; ++q) scratch[tid * 3 + q] = dacc[q];
  __syncthreads();
  if (has_d && tid == dix) {
    const int nrd = (G == 1) ? NWV : NWV / G;
    double v[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) v[q] = scratch[tid * 3 + q];
#pragma unroll 1
    for (int r = 1; r < nrd; ++r) {
      const int oth = (G == 1) ? (r * 64 + dix) : (dix + r * G * 64);
#pragma unroll
      for (int q = 0; q < 3; ++q) v[q] += scratch[oth * 3 + q];
    }
    double* od = out + (size_t)run.ntgt * 36 + (size_t)dix * 3;   // [local camera][row][rhs, gradient, column norm]
#pragma unroll
    for (int q = 0; q < 3; ++q) od[q] = v[q];
  }
  sk.lap(stamps, 10);
  }   // runs of this workgroup
  if constexpr (STAMPS) {
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < 12; ++k) atomicAdd(&g_fused_stamps[k], stamps[k]);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// K4 + K5 on the same skeleton (round 5): back-substitution of the point steps, candidate points, model-cost change and the
// trial cost, by persistent workgroups over the RUNS of the fused plan -- the per-camera blocks of a run (state: k_cam_prep;
// trial step {D, v}: P.camdir; candidate: P.camrot_cand) staged in LDS once per run, the observation stream of the next
// sub-chunk (and the points, their scaling and V^-1) prefetched, the problem through scalar loads, track sums through LDS
// slots.  The arithmetic is k_backsub<PD, false, true>'s (ba_kernels.hip): F y_c as the directional derivative
// s Jq (D p - w v), y_p = V^-1 E^T (r - F y_c), SphereManifold::Plus, residual at the candidate.  No phase S, no barrier
// inside a run: the waves of a workgroup only share the staged cameras.
// tile_part: [ntiles][5] = {cand_cost, mcc, stepsq, xnormsq, invalid}
constexpr int kCamLdsB = 70;   // doubles per staged camera: camrot (40) | camdir (12) | candidate ext + R (16) | pad (2); 35 16-B pieces

// INTR (the compound-block plan, free intrinsics; KMASK as k_lin_schur_i): the camera block of an observation multiplies the
// solved step itself -- F y_c + Fk y_k, from lin5's 2 x 6 and 2 x 10 blocks -- and the candidate carries its own intrinsics.
// Staged per camera then: camrot (40) | y_c (6) | y_k scaled, free parameters only (10) | candidate ext + R (16) | candidate
// intrinsics (10) = 82 doubles, 41 16-B pieces.  ysol: the solved step, intrinsics slots first (null without INTR).
constexpr int kCamLdsBI = 82;
template <int PD, int TPS, unsigned MODELS, int LOSSK, int WPS, bool INTR = false, unsigned KMASK = 0u>
__global__ __launch_bounds__(64 * TPS, WPS) void k_backsub_runs(DevProblem P, const double* __restrict__ pts, double* __restrict__ cand_pts,
                                                             const double* __restrict__ Vinv, double* __restrict__ tile_part,
                                                             const double* __restrict__ ysol = nullptr) {
  constexpr int NT = PD * (PD + 1) / 2;
  constexpr int SUB = TPS * kWave;
  constexpr int PITCH = INTR ? kCamLdsBI : kCamLdsB;
  constexpr int OFF_CAND = INTR ? 56 : 52;          // candidate ext + R inside a staged block
  constexpr int OFF_CINTR = INTR ? 72 : kCamRotIntr;   // the intrinsics the candidate is projected with
  __shared__ __attribute__((aligned(16))) double s_cam[(INTR ? kFusedMaxStageIntr : kFusedMaxStage) * PITCH];
  __shared__ __attribute__((aligned(16))) double s_slot[TPS][64 * 4];
  __shared__ int s_next;
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  int pending = 0;
  if (tid == 0) pending = atomicAdd(P.frun_next + 1, 1);
  for (;;) {
  __syncthreads();   // the previous run is done with s_cam (and s_next)
  if (tid == 0) s_next = pending;
  __syncthreads();
  const int rix = __builtin_amdgcn_readfirstlane(s_next);
  if (rix >= P.n_fruns) break;
  if (tid == 0) pending = atomicAdd(P.frun_next + 1, 1);
  const FusedRun run = P.fruns[P.frun_order[rix]];
  const int nsc = (run.ntiles + TPS - 1) / TPS;
  if constexpr (INTR) {
    for (int j = tid; j < run.nstage * (kCamLdsBI / 2); j += SUB) {
      const int k = j / (kCamLdsBI / 2), piece = j - k * (kCamLdsBI / 2);
      const int cidx = P.frun_stage[run.stage_off + k];
      double2 v = make_double2(0.0, 0.0);
      if (piece < 20) v = reinterpret_cast<const double2*>(P.camrot + (size_t)kCamRot * cidx)[piece];
      else if (piece < 28) {   // the solved step of the camera (6) and of its group's free intrinsics (10, Jacobi-scaled)
        const int rc = P.cam_red[cidx], g = P.cam_group[cidx], gr = P.grp_red[g];
        const unsigned fm = gr >= 0 ? (KMASK != 0u ? KMASK : P.grp_free[g]) : 0u;
        double e2[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int e = 2 * (piece - 20) + h;
          double val = 0.0;
          if (e < 6) { if (rc >= 0) val = ysol[P.ni + 6 * rc + e]; }
          else if ((fm >> (e - 6)) & 1u) val = ysol[10 * gr + (e - 6)] * P.scale_i[(size_t)g * THEIA_MAX_INTRINSICS + (e - 6)];
          e2[h] = val;
        }
        v = make_double2(e2[0], e2[1]);
      }
      else if (piece < 36) v = reinterpret_cast<const double2*>(P.camrot_cand + (size_t)kCamRot * cidx)[piece - 28];
      else v = reinterpret_cast<const double2*>(P.camrot_cand + (size_t)kCamRot * cidx + kCamRotIntr)[piece - 36];
      reinterpret_cast<double2*>(s_cam + k * kCamLdsBI)[piece] = v;
    }
  } else
  for (int j = tid; j < run.nstage * (kCamLdsB / 2); j += SUB) {
    const int k = j / (kCamLdsB / 2), piece = j - k * (kCamLdsB / 2);
    const int cidx = P.frun_stage[run.stage_off + k];
    double2 v = make_double2(0.0, 0.0);
    if (piece < 20) v = reinterpret_cast<const double2*>(P.camrot + (size_t)kCamRot * cidx)[piece];
    else if (piece < 26) v = reinterpret_cast<const double2*>(P.camdir + (size_t)12 * cidx)[piece - 20];
    else if (piece < 34) v = reinterpret_cast<const double2*>(P.camrot_cand + (size_t)kCamRot * cidx)[piece - 26];
    reinterpret_cast<double2*>(s_cam + k * kCamLdsB)[piece] = v;
  }
  LanePre<PD> cur;
  double viq[NT];
  int tile = 0; bool tile_ok = false;
  pre_level1<PD, TPS>(P, run, 0, wv, lane, tile, tile_ok, cur);
  pre_level2<PD>(P, pts, cur);
#pragma unroll
  for (int k = 0; k < NT; ++k) viq[k] = Vinv[(size_t)NT * cur.p + k];
  __syncthreads();   // s_cam complete
  for (int sc = 0; sc < nsc; ++sc) {
    LanePre<PD> nxt;
    int ntile = 0; bool ntile_ok = false;
    const int scn = min(sc + 1, nsc - 1);
    pre_level1<PD, TPS>(P, run, scn, wv, lane, ntile, ntile_ok, nxt);
    // ---------------------------------------------------------------- this lane's observation
    const LanePre<PD>& c = cur;
    const bool active = c.active;
    const unsigned cslot = (c.lc & 0x80u) ? (unsigned)run.W + (c.lc & 0x7fu) : c.lc;
    const double* cb = s_cam + cslot * PITCH;
    const double X[4] = {c.X.x, c.X.y, c.X.z, c.X.w};
    const int model = c.depth ? THIP_MODEL_DEPTH_ROW : (int)cb[kCamRotModel];
    double r[2], mc[2], Jt[2 * PD];
    bool behind = false;
    if constexpr (INTR) {
      // the compound camera block times the solved step: F y_c + Fk y_k (lin5 leaves r, F, Fk zero where nothing is evaluated,
      // the point block zero for a constant point)
      double Jc[12], Jk[2 * THEIA_MAX_INTRINSICS], cst;
      bool vld;
      lin5<PD, MODELS, LOSSK, true>(P, c, cb, r, cst, vld, Jc, Jt, Jk);
      const double* ys = cb + 40;
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        double sm = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) sm += Jc[6 * a + k] * ys[k];
#pragma unroll
        for (int k = 0; k < THEIA_MAX_INTRINSICS; ++k) {
          if (KMASK != 0u && !((KMASK >> k) & 1u)) continue;   // (a frozen parameter's step is zero)
          sm += Jk[THEIA_MAX_INTRINSICS * a + k] * ys[6 + k];
        }
        mc[a] = sm;
      }
    } else {
      const double2 c01 = *reinterpret_cast<const double2*>(cb), c2w = *reinterpret_cast<const double2*>(cb + 2);
      const double C[3] = {c01.x, c01.y, c2w.x};
      const double p[3] = {X[0] - X[3] * C[0], X[1] - X[3] * C[1], X[2] - X[3] * C[2]};
      behind = (p[0] * p[0] + p[1] * p[1] + p[2] * p[2]) < 1e-8;
      double R[9];
      {
        const double2 r0 = *reinterpret_cast<const double2*>(cb + 6), r1 = *reinterpret_cast<const double2*>(cb + 8), r2 = *reinterpret_cast<const double2*>(cb + 10),
                      r3 = *reinterpret_cast<const double2*>(cb + 12), r4 = *reinterpret_cast<const double2*>(cb + 14);
        R[0] = r0.x; R[1] = r0.y; R[2] = r1.x; R[3] = r1.y; R[4] = r2.x; R[5] = r2.y; R[6] = r3.x; R[7] = r3.y; R[8] = r4.x;
      }
      const double q[3] = {R[0] * p[0] + R[1] * p[1] + R[2] * p[2], R[3] * p[0] + R[4] * p[1] + R[5] * p[2], R[6] * p[0] + R[7] * p[1] + R[8] * p[2]};
      double uvp[2], Jq[6];
      project<true, false, MODELS>(model, cb + kCamRotIntr, q, uvp, Jq);
      r[0] = c.si.x * (uvp[0] - c.uv.x); r[1] = c.si.y * (uvp[1] - c.uv.y);
      double sr = 1.0;
      if constexpr (LOSSK != 0) {
        double rho1;
        (void)loss_eval_k<LOSSK>(P.loss_type, c.depth ? P.loss_width_depth : P.loss_width, r[0] * r[0] + r[1] * r[1], &rho1);
        sr = fsqrt(rho1);
        r[0] *= sr; r[1] *= sr;
      }
      {
        // the camera's step as {D, v}: F y_c = -sr s Jq (D p - w v)   (camera_step_direction, ba_device.h; zero for constant cameras)
        double u[3];
  #pragma unroll
        for (int i = 0; i < 3; ++i) u[i] = (cb[40 + 3 * i] * p[0] + cb[40 + 3 * i + 1] * p[1] + cb[40 + 3 * i + 2] * p[2]) - X[3] * cb[49 + i];
        double v[4] = {X[0], X[1], X[2], 1.0}, beta = 0.0, nx = 1.0;
        if constexpr (PD == 3) {
          const double sigma = X[0] * X[0] + X[1] * X[1] + X[2] * X[2];
          nx = fsqrt(X[3] * X[3] + sigma);
          if (sigma <= DBL_EPSILON) { if (X[3] < 0.0) beta = 2.0; }
          else {
            const double vp = (X[3] <= 0.0) ? X[3] - nx : -sigma / (X[3] + nx);
            beta = 2.0 * vp * vp / (sigma + vp * vp);
            const double ivp = 1.0 / vp;
            v[0] *= ivp; v[1] *= ivp; v[2] *= ivp;
          }
        }
        const double sia[2] = {c.si.x * sr, c.si.y * sr};
  #pragma unroll
        for (int a = 0; a < 2; ++a) {
          const double* jq = Jq + 3 * a;
          mc[a] = -sia[a] * (jq[0] * u[0] + jq[1] * u[1] + jq[2] * u[2]);
          const double A0 = jq[0] * R[0] + jq[1] * R[3] + jq[2] * R[6];
          const double A1 = jq[0] * R[1] + jq[1] * R[4] + jq[2] * R[7];
          const double A2 = jq[0] * R[2] + jq[1] * R[5] + jq[2] * R[8];
          const double j4[4] = {A0, A1, A2, -(A0 * C[0] + A1 * C[1] + A2 * C[2])};
          if constexpr (PD == 3) {
            const double jv = j4[0] * v[0] + j4[1] * v[1] + j4[2] * v[2] + j4[3] * v[3];
  #pragma unroll
            for (int k = 0; k < 3; ++k) Jt[3 * a + k] = (sia[a] * c.sp[k]) * (nx * (j4[k] - beta * v[k] * jv));
          } else {
  #pragma unroll
            for (int k = 0; k < PD; ++k) Jt[PD * a + k] = (sia[a] * c.sp[k]) * j4[k];
          }
        }
      }
    }
    if (behind || !active) { r[0] = 0.0; r[1] = 0.0; mc[0] = 0.0; mc[1] = 0.0; }
    if (behind || !active || c.pconst) {
#pragma unroll
      for (int i = 0; i < 2 * PD; ++i) Jt[i] = 0.0;
    }
    // t = E^T (r - F y_c), summed over the track through the wave's LDS slots
    const int pseg = active ? c.p : -1 - lane;
    const Segment sg = lane_segment_all(pseg, lane);
    double tsum[PD];
    {
      double* mine = s_slot[wv] + lane * 4;
      double in[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int k = 0; k < PD; ++k) in[k] = Jt[k] * (r[0] - mc[0]) + Jt[PD + k] * (r[1] - mc[1]);
      *reinterpret_cast<double2*>(mine) = make_double2(in[0], in[1]);
      *reinterpret_cast<double2*>(mine + 2) = make_double2(in[2], in[3]);
#pragma unroll
      for (int k = 0; k < PD; ++k) tsum[k] = 0.0;
      for (int j = 0; j < sg.maxlen; ++j) {
        const double* oth = s_slot[wv] + min(sg.start + j, 63) * 4;
        const double2 a = *reinterpret_cast<const double2*>(oth), b = *reinterpret_cast<const double2*>(oth + 2);
        const double v4[4] = {a.x, a.y, b.x, b.y};
        if (j < sg.len) {
#pragma unroll
          for (int k = 0; k < PD; ++k) tsum[k] += v4[k];
        }
      }
    }
    const bool pvar = active && !c.pconst;
    double yp[PD];
#pragma unroll
    for (int a = 0; a < PD; ++a) {
      double sm = 0.0;
#pragma unroll
      for (int b = 0; b < PD; ++b) sm += (pvar ? viq[a >= b ? lidx(a, b) : lidx(b, a)] : 0.0) * tsum[b];
      yp[a] = sm;
    }
    // step = -y ; model residual m = Js * step ; mcc = -m . (r + m/2)
    double mcc = 0.0;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      double m = -mc[a];
#pragma unroll
      for (int k = 0; k < PD; ++k) m -= Jt[a * PD + k] * yp[k];
      mcc -= m * (r[a] + m / 2.0);
    }
    if (!active) mcc = 0.0;
    double Xp[4] = {X[0], X[1], X[2], X[3]};
    double stepsq = 0.0, xnormsq = 0.0;
    if (pvar) {
      double d[PD];
#pragma unroll
      for (int k = 0; k < PD; ++k) d[k] = -yp[k] * c.sp[k];
      if constexpr (PD == 3) { const double d3[3] = {d[0], d[1], d[2]}; sphere_plus(X, d3, Xp); }
      else {
#pragma unroll
        for (int k = 0; k < PD; ++k) Xp[k] = X[k] + d[k];
      }
      if (sg.head) {
        reinterpret_cast<double4*>(cand_pts)[c.p] = make_double4(Xp[0], Xp[1], Xp[2], Xp[3]);
#pragma unroll
        for (int k = 0; k < 4; ++k) { stepsq += (X[k] - Xp[k]) * (X[k] - Xp[k]); xnormsq += Xp[k] * Xp[k]; }
      }
    }
    // trial cost at the candidate (camera block: the candidate's position / rotation, the same intrinsics)
    double ccost = 0.0;
    bool cvalid = true;
    {
      const double* cc = cb + OFF_CAND;
      const double2 d01 = *reinterpret_cast<const double2*>(cc), d2w = *reinterpret_cast<const double2*>(cc + 2);
      const double Cc[3] = {d01.x, d01.y, d2w.x};
      const double pc[3] = {Xp[0] - Xp[3] * Cc[0], Xp[1] - Xp[3] * Cc[1], Xp[2] - Xp[3] * Cc[2]};
      const bool cbehind = (pc[0] * pc[0] + pc[1] * pc[1] + pc[2] * pc[2]) < 1e-8;
      double Rc[9];
      {
        const double2 r0 = *reinterpret_cast<const double2*>(cc + 6), r1 = *reinterpret_cast<const double2*>(cc + 8), r2 = *reinterpret_cast<const double2*>(cc + 10),
                      r3 = *reinterpret_cast<const double2*>(cc + 12), r4 = *reinterpret_cast<const double2*>(cc + 14);
        Rc[0] = r0.x; Rc[1] = r0.y; Rc[2] = r1.x; Rc[3] = r1.y; Rc[4] = r2.x; Rc[5] = r2.y; Rc[6] = r3.x; Rc[7] = r3.y; Rc[8] = r4.x;
      }
      const double qc[3] = {Rc[0] * pc[0] + Rc[1] * pc[1] + Rc[2] * pc[2], Rc[3] * pc[0] + Rc[4] * pc[1] + Rc[5] * pc[2], Rc[6] * pc[0] + Rc[7] * pc[1] + Rc[8] * pc[2]};
      double uvc[2], Jqc[6];
      cvalid = project<false, false, MODELS>(model, cb + OFF_CINTR, qc, uvc, Jqc);
      const double rc0 = c.si.x * (uvc[0] - c.uv.x), rc1 = c.si.y * (uvc[1] - c.uv.y);
      double s2 = rc0 * rc0 + rc1 * rc1;
      if (cbehind) { cvalid = false; s2 = 0.0; }
      double rho = s2;
      if constexpr (LOSSK != 0) { double rho1; rho = loss_eval_k<LOSSK>(P.loss_type, c.depth ? P.loss_width_depth : P.loss_width, s2, &rho1); }
      ccost = 0.5 * rho;
      if (!active) { ccost = 0.0; cvalid = true; }
    }
    ccost = wave_sum_all(ccost);
    mcc = wave_sum_all(mcc);
    stepsq = wave_sum_all(stepsq);
    xnormsq = wave_sum_all(xnormsq);
    const double inval = wave_count(!cvalid);
    if (lane == 0 && tile_ok) {
      double* tp = tile_part + 5 * (size_t)tile;
      tp[0] = ccost; tp[1] = mcc; tp[2] = stepsq; tp[3] = xnormsq; tp[4] = inval;
    }
    // ---------------------------------------------------------------- the next sub-chunk's points
    pre_level2<PD>(P, pts, nxt);
#pragma unroll
    for (int k = 0; k < NT; ++k) viq[k] = Vinv[(size_t)NT * nxt.p + k];
    cur = nxt; tile = ntile; tile_ok = ntile_ok;
  }
  }   // runs of this workgroup
}

// One wave per S block (ri, rj): the partial blocks of the runs that touch it are added in run order and the block
// is WRITTEN (never accumulated): S_ij = - sum What_a What_b^T; for a camera (ri == rj) also the per-observation
// sums  F^T F - ..,  rhs = F^T r - What ghat,  gradient, column norms.
__global__ __launch_bounds__(256) void k_schur_sum(int nitems, const int* __restrict__ items, const int* __restrict__ src,
                                                   const double* __restrict__ part, double* __restrict__ S, int n,
                                                   double* __restrict__ rhs, double* __restrict__ gc,
                                                   double* __restrict__ colsq, int* __restrict__ qhead) {
  const int it = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (blockIdx.x == 0 && threadIdx.x == 0 && qhead) *qhead = 0;   // k_lin_schur (the previous kernel on the stream) is done with its run queue: ready for the next launch
  if (it >= nitems) return;
  const int* d = items + 6 * it;
  const int ri = d[0], rj = d[1], tbeg = d[2], tend = d[3], dbeg = d[4], dend = d[5];
  // the sources of a block are added in run order (fixed), but their loads are independent: eight in flight
  auto sum_sources = [&](int qb, int qe, int off) {
    double v = 0.0;
    int q = qb;
    for (; q + 8 <= qe; q += 8) {
      double t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = part[(size_t)src[q + u] + off];
#pragma unroll
      for (int u = 0; u < 8; ++u) v += t[u];
    }
    double t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = (q + u < qe) ? part[(size_t)src[min(q + u, qe - 1)] + off] : 0.0;
#pragma unroll
    for (int u = 0; u < 8; ++u) if (q + u < qe) v += t[u];
    return v;
  };
  if (ri != rj) {
    if (lane >= 36) return;
    const double v = (tend > tbeg) ? sum_sources(tbeg, tend, lane) : 0.0;
    S[(size_t)(6 * ri + lane / 6) * n + 6 * rj + lane % 6] = -v;
    return;
  }
  if (lane >= 54) return;
  const int a = lane / 9, j = lane % 9;
  if (j < 6) {   // the diagonal target's sum is  What What^T - F^T F
    const double w = (tend > tbeg) ? sum_sources(tbeg, tend, a * 6 + j) : 0.0;
    if (j <= a) S[(size_t)(6 * ri + a) * n + 6 * ri + j] = -w;
    return;
  }
  const double v = (dend > dbeg) ? sum_sources(dbeg, dend, a * 3 + (j - 6)) : 0.0;
  if (j == 6) {
    rhs[6 * ri + a] = v;
  } else if (j == 7) {
    gc[6 * ri + a] = v;
  } else {
    colsq[6 * ri + a] = v;
  }
}

}  // namespace

void launch_cam_prep(const DevProblem& P, const double* cam, const double* intr, double* camrot, hipStream_t st, const double* ycam) {
  if (P.nc > 0) k_cam_prep<<<(P.nc + 255) / 256, 256, 0, st>>>(P, cam, intr, camrot, ycam);
}

void launch_linearize_fused(const DevProblem& P, const double* cam, const double* pts, const double* radius,
                            const ReduceBuf& rb, double* Vinv, double* tile_part, hipStream_t st) {
  if (P.n_fruns == 0) return;
  if (!P.camrot_current || P.n_sum_items == 0) launch_cam_prep(P, cam, P.intr, P.camrot, st);   // (otherwise k_schur_sum left the run queue's head at zero)
  static const int wgs = [] { const char* e = getenv("THEIA_HIP_FUSED_WGS"); return e ? std::max(1, atoi(e)) : 512; }();   // 2 per CU
  const int grid = std::min(P.n_fruns, wgs);
  const bool trig = (P.model_mask & ~kModelsNoTrig) != 0;   // FOV / fisheye groups present
  {
    const int lk = loss_class(P.loss_type);
    static const bool stamps = getenv("THEIA_HIP_FUSED_STAMPS") != nullptr;   // development: instrumented instance + report
    if (stamps && P.pd == 3 && !trig && lk == 0) {
      static int launches = 0;
      k_lin_schur<3, 4, kModelsNoTrig, 0, true><<<grid, 256, 0, st>>>(P, pts, radius, Vinv, tile_part);
      if (++launches % 16 == 0) {
        unsigned long long hst[16], zero[16] = {0};
        (void)hipStreamSynchronize(st);
        (void)hipMemcpyFromSymbol(hst, HIP_SYMBOL(g_fused_stamps), sizeof(hst));
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_fused_stamps), zero, sizeof(zero));
        static const char* nm[12] = {"run setup", "phase L", " L math", " L scan", " L inverse", " L stores", " L reductions", "prefetch2+barrier", "phase S", "barrier after S", "combine+write", "sub-chunks"};
        double tot = 0.0;
        for (int k : {0, 1, 7, 8, 9, 10}) tot += (double)hst[k];
        fprintf(stderr, "theia_hip fused stamps (16 launches, wave-summed s_memtime ticks; share of the sections' total):\n");
        for (int k = 0; k < 12; ++k) fprintf(stderr, "  %-20s %14llu  %5.1f%%  per wave sub-chunk %8.0f\n", nm[k], hst[k], 100.0 * (double)hst[k] / tot, (double)hst[k] / std::max(1.0, (double)hst[11]));
      }
      if (P.n_sum_items)
        k_schur_sum<<<(P.n_sum_items + 3) / 4, 256, 0, st>>>(P.n_sum_items, P.sum_items, P.sum_src, P.fpart, rb.S, P.n, rb.rhs, rb.gc, rb.colsq, P.frun_next);
      return;
    }
#define THIP_LS(PD_, M_) do { \
      if (lk == 0) k_lin_schur<PD_, 4, M_, 0><<<grid, 256, 0, st>>>(P, pts, radius, Vinv, tile_part); \
      else if (lk == 1) k_lin_schur<PD_, 4, M_, 1><<<grid, 256, 0, st>>>(P, pts, radius, Vinv, tile_part); \
      else k_lin_schur<PD_, 4, M_, 2><<<grid, 256, 0, st>>>(P, pts, radius, Vinv, tile_part); } while (0)
    if (P.pd == 3) { if (trig) THIP_LS(3, kModelsAll); else THIP_LS(3, kModelsNoTrig); }
    else { if (trig) THIP_LS(4, kModelsAll); else THIP_LS(4, kModelsNoTrig); }
#undef THIP_LS
  }
  if (P.n_sum_items)
    k_schur_sum<<<(P.n_sum_items + 3) / 4, 256, 0, st>>>(P.n_sum_items, P.sum_items, P.sum_src, P.fpart, rb.S, P.n, rb.rhs,
                                                         rb.gc, rb.colsq, P.frun_next);
}

// K4 + K5 over the runs of the fused plan (k_backsub_runs); false: not applicable (the caller takes k_backsub)
bool launch_backsub_runs(const DevProblem& P, const double* pts, double* cand_pts, const double* Vinv, double* tile_part, hipStream_t st) {
  static const bool off = getenv("THEIA_HIP_BACKSUB_TILES") != nullptr;   // development: the round-4 kernel (one wave per tile, gathers)
  if (off || P.ni || P.n_fruns == 0 || !P.camrot || !P.camrot_cand || !P.camdir || P.fused_bw) return false;
  static const int wgs_env = [] { const char* e = getenv("THEIA_HIP_BACKSUB_WGS"); return e ? std::max(1, atoi(e)) : 0; }();
  const bool trig = (P.model_mask & ~kModelsNoTrig) != 0;
  const int lk = loss_class(P.loss_type);
  static const int wps = [] { const char* e = getenv("THEIA_HIP_BACKSUB_WAVES"); return e ? atoi(e) : 3; }();   // waves per SIMD the register allocation aims at (trivial loss: 3 with 56 B of spills measured 0.901 ms per iteration against 0.925 at 2 without)
  const int grid = std::min(P.n_fruns, wgs_env ? wgs_env : 256 * (wps == 3 && lk == 0 ? 3 : 2));
#define THIP_BS(PD_, M_) do { \
    if (wps == 3 && lk == 0) k_backsub_runs<PD_, 4, M_, 0, 3><<<grid, 256, 0, st>>>(P, pts, cand_pts, Vinv, tile_part); \
    else if (lk == 0) k_backsub_runs<PD_, 4, M_, 0, 2><<<grid, 256, 0, st>>>(P, pts, cand_pts, Vinv, tile_part); \
    else if (lk == 1) k_backsub_runs<PD_, 4, M_, 1, 2><<<grid, 256, 0, st>>>(P, pts, cand_pts, Vinv, tile_part); \
    else k_backsub_runs<PD_, 4, M_, 2, 2><<<grid, 256, 0, st>>>(P, pts, cand_pts, Vinv, tile_part); } while (0)
  if (P.pd == 3) { if (trig) THIP_BS(3, kModelsAll); else THIP_BS(3, kModelsNoTrig); }
  else { if (trig) THIP_BS(4, kModelsAll); else THIP_BS(4, kModelsNoTrig); }
#undef THIP_BS
  return true;
}

// the same over the runs of the compound-block plan (free intrinsics): y = the solved step, intrinsics slots first
bool launch_backsub_runs_intr(const DevProblem& P, const double* pts, double* cand_pts, const double* Vinv, double* tile_part,
                              const double* y, hipStream_t st) {
  static const bool off = getenv("THEIA_HIP_BACKSUB_TILES") != nullptr;
  if (off || !P.ni || !P.fused_bw || P.n_fruns == 0 || !P.camrot || !P.camrot_cand || !P.frun_stage) return false;
  constexpr unsigned kFR = (1u << 0) | (3u << 5);   // FOCAL_LENGTH | RADIAL_DISTORTION (ba_fused_intr.hip: kMaskFocalRadial)
  const bool trig = (P.model_mask & ~kModelsNoTrig) != 0;
  const bool trivial = P.loss_type == THEIA_LOSS_TRIVIAL;
  const int grid = std::min(P.n_fruns, 512);
#define THIP_BSI2(PD_, M_, LK_) do { \
    if (P.fused_kmask == kFR) k_backsub_runs<PD_, 4, M_, LK_, 2, true, kFR><<<grid, 256, 0, st>>>(P, pts, cand_pts, Vinv, tile_part, y); \
    else k_backsub_runs<PD_, 4, M_, LK_, 2, true, 0u><<<grid, 256, 0, st>>>(P, pts, cand_pts, Vinv, tile_part, y); } while (0)
#define THIP_BSI(PD_, M_) do { if (trivial) THIP_BSI2(PD_, M_, 0); else THIP_BSI2(PD_, M_, 2); } while (0)
  if (P.pd == 3) { if (trig) THIP_BSI(3, kModelsAll); else THIP_BSI(3, kModelsNoTrig); }
  else { if (trig) THIP_BSI(4, kModelsAll); else THIP_BSI(4, kModelsNoTrig); }
#undef THIP_BSI
#undef THIP_BSI2
  return true;
}

}  // namespace thip
