// ba_fused_intr.hip -- K1 + K2 in one pass over the observations when camera intrinsics are optimised
// (intrinsics_to_optimize != NONE: the pipelines' default FOCAL_LENGTH | RADIAL_DISTORTION,
// src/theia/sfm/reconstruction_estimator_options.h:281-283; shared blocks per bundle_adjuster.cc:382-475).
//
// The camera-side Jacobian of an observation is the COMPOUND block  Jc = [F (2 x 6) | Fk (2 x KI)]  of its camera's
// extrinsics and the (up to four, compact) free intrinsics of its camera's group.  The kernel treats every local
// camera of a run as if it owned its intrinsics: per pair of cameras (a, b) that share a track it accumulates the
// (6 + KI) x (6 + KI) block
//       Jc_a^T (Ehat_a Ehat_b^T - [a == b] I) Jc_b        (Ehat = E Li^T, V^-1 = Li^T Li, as in ba_fused.hip)
// in registers, three (four) lanes per block with three rows each, the records {Jc | Ehat | r | r - Ehat ghat} staying in LDS.
// Sharing is applied afterwards, where it is linear: the intrinsics rows / columns of all cameras of a group are SUMMED
// into the group's rows / columns of S by k_sum_items (S_shared = P^T S_private P with P the 0/1 map from per-camera
// intrinsics to the group's).  No group sums per track, no pair lists, no records in HBM (the first generation,
// k_lin_obs_intr + k_schur_intr, moved 14 GB per iteration at 1000 views / 500k tracks for 0.14 GB of algorithmic
// traffic); the price is (6 + KI)^2 / 36 = 2.25x (2.8x) the block products of the camera-only kernel.
//
// Partial sums of a run: [ntgt][BW x BW] then [W][BW][3] (rhs, gradient, squared column norm).  k_sum_items adds, per
// block of S, the listed pieces of those partial blocks in list order (fixed, bitwise reproducible) and WRITES the block;
// long lists (the group x group blocks of a problem with one shared camera get a piece from every block of every run)
// go through a second level.
#define THIP_LEAN_SQRT 1
#include "ba_fused_lin.h"

#include <algorithm>
#include <cstdlib>

namespace thip {
namespace {

// Lane split of a compound block of width BW = 6 + KI: lane `sub` of a target owns RPL consecutive rows of it, the last lanes
// one row less where BW is not a multiple of RPL -- KI = 3: three lanes of three rows {0, 3, 6}; KI = 4: four lanes {0, 3, 6, 8};
// KI = 7 (every intrinsic of the pinhole model free): four lanes of 4 + 3 + 3 + 3 rows {0, 4, 7, 10}; KI = 10 (the ten of the
// radial-tangential model; nine of the fisheye model in ten rows): four lanes of four rows -- so that the 55 target
// blocks of ten cameras still fit the 256 threads of a workgroup (the register budget is no constraint there: the records
// of that width leave room for one workgroup per CU).
template <int KI> constexpr int rpl() { return KI <= 4 ? 3 : 4; }   // rows of the block per lane (accumulators: RPL x BW)
template <int KI> constexpr int lanes_per_target() { return (6 + KI + rpl<KI>() - 1) / rpl<KI>(); }
template <int KI> constexpr int full_lanes() { return lanes_per_target<KI>() - (rpl<KI>() * lanes_per_target<KI>() - (6 + KI)); }   // lanes with RPL rows
template <int KI> THIP_DEV int sub_row0(int sub) { return sub < full_lanes<KI>() ? rpl<KI>() * sub : rpl<KI>() * full_lanes<KI>() + (rpl<KI>() - 1) * (sub - full_lanes<KI>()); }
template <int KI> THIP_DEV int sub_rows(int sub) { return sub < full_lanes<KI>() ? rpl<KI>() : rpl<KI>() - 1; }
static_assert(lanes_per_target<3>() == 3 && lanes_per_target<4>() == 4 && full_lanes<4>() == 2 && lanes_per_target<7>() == 4 && full_lanes<7>() == 1, "lane split");
template <int KI> constexpr int bwp() { return (6 + KI + 1) & ~1; }   // row stride of Jc inside a record (BW rounded up to even: 16-B aligned rows)
constexpr int kRowBytesI = 16;             // slot-table row: the most local cameras of a run, rounded up
static_assert(kFusedMaxCamsIntr <= kRowBytesI, "slot-table row too short");
// LDS record of one observation: {Jc row 0 (BWP) | Jc row 1 (BWP) | Ehat (2 x PD, interleaved as in ba_fused.hip) | r (2) |
// r - Ehat ghat (2)}, at least the RPL x BW doubles a lane parks there when the track slices are combined, an odd number of
// 16-B pieces: 30 / 34 doubles (PD = 3 / 4) with three or four intrinsics rows, 54 with seven, 66 with ten
template <int PD, int KI> constexpr int reci_doubles() {
  int rd = 2 * bwp<KI>() + 2 * PD + 4;
  if (rd < rpl<KI>() * (6 + KI)) rd = rpl<KI>() * (6 + KI);
  rd = (rd + 1) & ~1;
  if ((rd / 2) % 2 == 0) rd += 2;
  return rd;
}
static_assert(reci_doubles<3, 3>() == 30 && reci_doubles<4, 3>() == 34 && reci_doubles<3, 4>() == 30 && reci_doubles<4, 4>() == 34, "record layout of the three / four row instances");

THIP_DEV unsigned segment_or_i(const Segment& s, int lane, unsigned v) {
  const int pos = lane - s.start;
  for (int d = 1; d < s.maxlen; d <<= 1) {
    const unsigned o = (unsigned)__shfl_down((int)v, d, kWave);
    if (pos + d < s.len) v |= o;
  }
  return (unsigned)__shfl((int)v, s.start, kWave);
}

// k-th set bit of a compile-time mask (-1: fewer bits)
constexpr int nth_bit(unsigned m, int k) {
  for (int q = 0; q < 32; ++q) if ((m >> q) & 1u) { if (k == 0) return q; --k; }
  return -1;
}

// KMASK != 0: every variable group frees exactly the parameters KMASK (known at create()): the compact rows are picked
// at compile time and the Jacobian columns of the other intrinsics are never computed.  KMASK == 0: per-lane masks.
// Round 5: the linearisation is lin5 (ba_fused_lin.h) on the prefetched observation `c` and the camera's block in LDS (s_cam:
// the run's local cameras, then the constant cameras its tracks see), as in k_lin_schur; the generic lane_linearize with its
// 320-B gather per observation was two thirds of this kernel (407 of 612 us at C4 with the pair products switched off).
template <int PD, int TPS, unsigned MODELS, unsigned KMASK, int LOSSK, int KI>
__device__ __forceinline__ void fusedi_phase_l(const DevProblem& P, const LanePre<PD>& c, const double* __restrict__ s_cam, int W, int tile,
                                               bool tile_ok, int wv, int lane, double inv_radius,
                                               double* __restrict__ Vinv, double* __restrict__ tile_part,
                                               double* __restrict__ s_rec, uint8_t* __restrict__ s_tslot,
                                               unsigned* __restrict__ s_tmask) {
  constexpr int NT = PD * (PD + 1) / 2;
  constexpr int RD = reci_doubles<PD, KI>();
  constexpr int kBWP = bwp<KI>();
  constexpr int KR = kBWP - 6;             // compact intrinsics rows of a record (KI of them in use, the rest zero)
  const bool active = c.active;
  const bool is_tgt = !(c.lc & 0x80u);
  const unsigned cslot = is_tgt ? c.lc : (unsigned)W + (c.lc & 0x7fu);
  const double* cb = s_cam + cslot * kCamLds;
  struct { double r[2], Jc[12], Jt[2 * PD], Jk[2 * THEIA_MAX_INTRINSICS]; double cost; bool valid, pconst; int p; } L;
  lin5<PD, MODELS, LOSSK, true>(P, c, cb, L.r, L.cost, L.valid, L.Jc, L.Jt, L.Jk);
  L.pconst = c.pconst; L.p = active ? c.p : -1 - lane;
  // Jacobi scale and free mask of the camera's group (constant group: zero rows); an inactive lane's rows are zero
  const int grp = (int)cb[kCamRotGroup];
  const int gr = active ? P.grp_red[grp] : -1;
  unsigned fm = gr >= 0 ? P.grp_free[grp] : 0u;
  // compact intrinsics rows: row k = the k-th free parameter of the camera's group
  double jk[2 * KR];
  if constexpr (KMASK != 0u) {
    static_assert(KMASK == 0u || KR == 4, "a compile-time mask picks at most four rows");
    constexpr int Q[4] = {nth_bit(KMASK, 0), nth_bit(KMASK, 1), nth_bit(KMASK, 2), nth_bit(KMASK, 3)};
#pragma unroll
    for (int k = 0; k < KR; ++k) {
      double sc = 0.0;
      if (Q[k] >= 0 && gr >= 0) sc = P.scale_i[(size_t)grp * THEIA_MAX_INTRINSICS + (Q[k] >= 0 ? Q[k] : 0)];
      jk[k] = Q[k] >= 0 ? L.Jk[Q[k] >= 0 ? Q[k] : 0] * sc : 0.0;
      jk[KR + k] = Q[k] >= 0 ? L.Jk[THEIA_MAX_INTRINSICS + (Q[k] >= 0 ? Q[k] : 0)] * sc : 0.0;
    }
  } else {
#pragma unroll
    for (int k = 0; k < KR; ++k) {
      double v0 = 0.0, v1 = 0.0;
      if (fm) {
        const int q = __ffs(fm) - 1;
        fm &= fm - 1u;
        const double sc = P.scale_i[(size_t)grp * THEIA_MAX_INTRINSICS + q];
#pragma unroll
        for (int j = 0; j < THEIA_MAX_INTRINSICS; ++j) if (j == q) { v0 = L.Jk[j] * sc; v1 = L.Jk[THEIA_MAX_INTRINSICS + j] * sc; }
      }
      jk[k] = v0; jk[KR + k] = v1;
    }
  }
  const int tl = active ? (int)c.tl : 0;
  const unsigned lc = (active && is_tgt) ? c.lc : 0xffu;
  // the camera-side rows of the record leave the registers at once (20 doubles that nothing below needs again)
  if (active && lc != 0xffu) {
    double2* R = reinterpret_cast<double2*>(s_rec + (wv * 64 + lane) * RD);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int q = 0; q < 3; ++q) R[(kBWP / 2) * i + q] = make_double2(L.Jc[6 * i + 2 * q], L.Jc[6 * i + 2 * q + 1]);
#pragma unroll
      for (int q = 0; q < KR / 2; ++q) R[(kBWP / 2) * i + 3 + q] = make_double2(jk[KR * i + 2 * q], jk[KR * i + 2 * q + 1]);
    }
  }
  const Segment sg = lane_segment_all(L.p, lane);
  double tot[NT + PD];
#pragma unroll
  for (int a = 0; a < PD; ++a) {
#pragma unroll
    for (int b = 0; b <= a; ++b) tot[lidx(a, b)] = L.Jt[a] * L.Jt[b] + L.Jt[PD + a] * L.Jt[PD + b];
    tot[NT + a] = L.Jt[a] * L.r[0] + L.Jt[PD + a] * L.r[1];
  }
  unsigned tmask;
  if constexpr (NT + PD + 1 <= RD - 2 * kBWP) {
    // the track sums through the lanes' own record slots (round 5, as ba_fused.hip): the part of the record behind the camera
    // rows is free until Ehat is written below; every lane of a track adds the slots of the track's lanes in lane order
    // (broadcast reads) instead of 92 ds_bpermute of the log-step scans
    constexpr int NS = NT + PD;
    tmask = 0u;
    double* mine = s_rec + (wv * 64 + lane) * RD + 2 * kBWP;
#pragma unroll
    for (int k = 0; k + 1 < NS; k += 2) *reinterpret_cast<double2*>(mine + k) = make_double2(tot[k], tot[k + 1]);
    if constexpr (NS & 1) mine[NS - 1] = tot[NS - 1];
    reinterpret_cast<unsigned*>(mine + NS)[0] = (active && lc != 0xffu) ? (1u << lc) : 0u;
#pragma unroll
    for (int k = 0; k < NS; ++k) tot[k] = 0.0;
    for (int j = 0; j < sg.maxlen; ++j) {   // wave-uniform trip count
      const double* oth = s_rec + (wv * 64 + min(sg.start + j, 63)) * RD + 2 * kBWP;
      double v[NS];
#pragma unroll
      for (int k = 0; k + 1 < NS; k += 2) { const double2 u = *reinterpret_cast<const double2*>(oth + k); v[k] = u.x; v[k + 1] = u.y; }
      if constexpr (NS & 1) v[NS - 1] = oth[NS - 1];
      const unsigned ob = reinterpret_cast<const unsigned*>(oth + NS)[0];
      if (j < sg.len) {
#pragma unroll
        for (int k = 0; k < NS; ++k) tot[k] += v[k];
        tmask |= ob;
      }
    }
  } else {
    segment_allsum_log<NT + PD>(sg, lane, tot);
    tmask = segment_or_i(sg, lane, (active && lc != 0xffu) ? (1u << lc) : 0u);
  }
  double V[NT], Vi[NT], g[PD];
#pragma unroll
  for (int q = 0; q < NT; ++q) V[q] = tot[q];
#pragma unroll
  for (int a = 0; a < PD; ++a) { g[a] = tot[NT + a]; V[lidx(a, a)] += fmin(fmax(V[lidx(a, a)], 1e-6), 1e32) * inv_radius; }
  bool pd_ok = true;
  double Li[PD][PD];
#pragma unroll
  for (int a = 0; a < PD; ++a)
#pragma unroll
    for (int b = 0; b < PD; ++b) Li[a][b] = 0.0;
  if (active && !L.pconst) pd_ok = invert_spd<PD>(V, Vi, Li);
  if (!active || L.pconst || !pd_ok) {
#pragma unroll
    for (int q = 0; q < NT; ++q) Vi[q] = 0.0;
#pragma unroll
    for (int a = 0; a < PD; ++a)
#pragma unroll
      for (int b = 0; b < PD; ++b) Li[a][b] = 0.0;
  }
  double gmax = 0.0;
  double gh[PD];
#pragma unroll
  for (int a = 0; a < PD; ++a) {
    double s = 0.0;
#pragma unroll
    for (int q = 0; q <= a; ++q) s += Li[a][q] * g[q];
    gh[a] = s;
  }
  if (active && sg.head) {
    s_tmask[tl] = tmask;
    if (!L.pconst) {
#pragma unroll
      for (int q = 0; q < NT; ++q) Vinv[(size_t)NT * L.p + q] = Vi[q];
#pragma unroll
      for (int a = 0; a < PD; ++a) gmax = fmax(gmax, fabs(g[a] / c.sp[a]));
    }
  }
  if (active && lc != 0xffu) {
    const int slot = wv * 64 + lane;
    double eh[2 * PD];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int b = 0; b < PD; ++b) {
        double s = 0.0;
#pragma unroll
        for (int q = 0; q <= b; ++q) s += L.Jt[i * PD + q] * Li[b][q];
        eh[i * PD + b] = s;
      }
    double2* R = reinterpret_cast<double2*>(s_rec + slot * RD);
#pragma unroll
    for (int q = 0; q < PD; ++q) R[kBWP + q] = make_double2(eh[2 * q], eh[2 * q + 1]);
    R[kBWP + PD] = make_double2(L.r[0], L.r[1]);
    double v1[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      double sm = 0.0;
#pragma unroll
      for (int b = 0; b < PD; ++b) sm += eh[i * PD + b] * gh[b];
      v1[i] = L.r[i] - sm;
    }
    R[kBWP + PD + 1] = make_double2(v1[0], v1[1]);
    s_tslot[tl * kRowBytesI + lc] = (uint8_t)slot;
  }
  const double cost = wave_sum_all(L.cost);
  gmax = wave_max_all(gmax);
  const double inval = wave_count(active && !L.valid);
  const double npd = wave_count(active && !pd_ok && sg.head);
  if (lane == 0 && tile_ok) {
    tile_part[4 * (size_t)tile + 0] = cost;
    tile_part[4 * (size_t)tile + 1] = gmax;
    tile_part[4 * (size_t)tile + 2] = inval;
    tile_part[4 * (size_t)tile + 3] = npd;
  }
}

// KI = 3 or 4: compact intrinsics rows in use (the widest free mask of the problem's groups); a run's partial blocks are
// BW x BW = (6 + KI)^2 doubles per target and BW x 3 per camera (ba_solver.hip: build_sum_items_intr reads them so).
template <int PD, int TPS, unsigned MODELS, int KI, unsigned KMASK, int LOSSK>
__global__ __launch_bounds__(64 * TPS, 2) void k_lin_schur_i(DevProblem P, const double* __restrict__ pts,
                                                             const double* __restrict__ radius_p,
                                                             double* __restrict__ Vinv, double* __restrict__ tile_part) {
  constexpr int RD = reci_doubles<PD, KI>();
  constexpr int BW = 6 + KI;
  constexpr int kBWP = bwp<KI>();
  constexpr int NS = lanes_per_target<KI>();
  constexpr int OE = 2 * kBWP;                         // offset of Ehat inside a record
  constexpr int SUB = TPS * kWave;
  constexpr int SUBT = TPS * kFusedTileTracks;
  constexpr int NWV = TPS;
  constexpr int kRPL = rpl<KI>();
  constexpr int NA = kRPL * BW;                        // accumulators of a lane: its rows of the block
  __shared__ __attribute__((aligned(16))) double s_rec[SUB * RD];
  __shared__ __attribute__((aligned(16))) double s_cam[kFusedMaxStageIntr * kCamLds];   // the run's camera blocks (k_cam_prep): local, then constant
  __shared__ uint8_t s_tslot[SUBT * kRowBytesI];
  __shared__ unsigned s_tmask[SUBT];
  static_assert(NA <= RD, "slice-combination scratch does not fit the record buffer");
  __shared__ int s_next;
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const double inv_radius = 1.0 / *radius_p;
  const bool colnorm_only = (P.fused_dbg & 8) != 0;   // compute_scale: only the per-(camera, row) sums are read afterwards
  int pending = 0;   // the run queue is popped one run ahead (the pop's round trip is off the path between two runs)
  if (tid == 0) pending = atomicAdd(P.frun_next, 1);
  for (;;) {
    __syncthreads();
    if (tid == 0) s_next = pending;
    __syncthreads();
    const int rix = __builtin_amdgcn_readfirstlane(s_next);   // (scalar: the run's fields and the tile geometry are s_loads)
    if (rix >= P.n_fruns) break;
    if (tid == 0) pending = atomicAdd(P.frun_next, 1);
    const FusedRun run = P.fruns[P.frun_order[rix]];
    const int nsc = (run.ntiles + TPS - 1) / TPS;
    // the run's per-camera blocks -> LDS, 16 B per thread and step; the first sub-chunk's observation stream meanwhile
    for (int j = tid; j < run.nstage * (kCamRot / 2); j += SUB) {
      const int k = j / (kCamRot / 2), piece = j - k * (kCamRot / 2);
      const int cidx = P.frun_stage[run.stage_off + k];
      reinterpret_cast<double2*>(s_cam + k * kCamLds)[piece] = reinterpret_cast<const double2*>(P.camrot + (size_t)kCamRot * cidx)[piece];
    }
    LanePre<PD> cur;
    int tile = 0; bool tile_ok = false;
    pre_level1<PD, TPS>(P, run, 0, wv, lane, tile, tile_ok, cur);
    pre_level2<PD>(P, pts, cur);
    __syncthreads();

    const int G = run.gp & 0xff, PS = run.gp >> 8;
    const int B = 64 / PS;
    double acc[NA], dacc[3];
#pragma unroll
    for (int k = 0; k < NA; ++k) acc[k] = 0.0;
#pragma unroll
    for (int k = 0; k < 3; ++k) dacc[k] = 0.0;

    for (int sc = 0; sc < nsc; ++sc) {
      LanePre<PD> nxt;
      int ntile = 0; bool ntile_ok = false;
      const int scn = min(sc + 1, nsc - 1);   // (the last sub-chunk reloads itself: unconditional loads, nothing is used)
      pre_level1<PD, TPS>(P, run, scn, wv, lane, ntile, ntile_ok, nxt);
      fusedi_phase_l<PD, TPS, MODELS, KMASK, LOSSK, KI>(P, cur, s_cam, run.W, tile, tile_ok, wv, lane, inv_radius, Vinv, tile_part, s_rec, s_tslot, s_tmask);
      pre_level2<PD>(P, pts, nxt);
      __builtin_amdgcn_s_setprio(1);   // phase S is pure issue, phase L a chain of latencies: S first, L fills the gaps (ba_fused.hip)
      __syncthreads();
      {
        // ---- phase-S role (recomputed per sub-chunk: a dozen integers that need not stay in registers across phase L).  lix = lane index inside the track slice this lane serves: target block lix / NS, rows
      // sub_row0(lix % NS) .. of it; for the per-observation terms lane (local camera, row) = dix
      int lix, t0, tstride;
      bool slice_ok = true;
      if (G == 1) { const int g = lane / B; lix = lane - g * B; slice_ok = g < PS; t0 = wv * PS + g; tstride = NWV * PS; }
      else { lix = (wv % G) * 64 + lane; t0 = wv / G; tstride = NWV / G; }
      const int tix = lix / NS, sub = lix - tix * NS;
      const int row0 = sub_row0<KI>(sub);
      const bool has_tgt = slice_ok && tix < run.ntgt;
      int la = 0, lb = 0;
      if (has_tgt) { const unsigned us = P.frun_tgt[run.tgt_off + tix]; la = us & 0xffu; lb = us >> 8; }
      const int dix = (G == 1) ? lane : lix;
      const bool has_d = dix < BW * run.W;
      const int dlc = has_d ? dix / BW : 0, da = dix % BW;
      const int dP = (G == 1) ? PS : 1, dbase = (G == 1) ? wv * PS : wv / G;
      const unsigned tbits = has_tgt ? ((1u << la) | (1u << lb)) : 0xffffffffu;
      const unsigned dbit = has_d ? (1u << dlc) : 0x80000000u;
      const double diag_core = (has_tgt && la == lb) ? 1.0 : 0.0;

        const int last_tile = min(run.tile0 + TPS * sc + TPS - 1, run.tile0 + run.ntiles - 1);
        const int ntr = P.tile_trk_end[last_tile];
#pragma unroll 1
        for (int base = 0; base < ntr; base += tstride) {
          if (!colnorm_only) {
            const int t = base + t0;
            const unsigned mask = (slice_ok && t < ntr) ? s_tmask[t] : 0u;
            if ((mask & tbits) == tbits) {
              const unsigned ca = s_tslot[t * kRowBytesI + la], cb = s_tslot[t * kRowBytesI + lb];
              const double* ra = s_rec + ca * RD;
              const double* rb = s_rec + cb * RD;
              double Ea[2 * PD], Eb[2 * PD];
              const double2* pa = reinterpret_cast<const double2*>(ra + OE);
              const double2* pb = reinterpret_cast<const double2*>(rb);
#pragma unroll
              for (int q = 0; q < PD; ++q) { const double2 u = pa[q]; Ea[2 * q] = u.x; Ea[2 * q + 1] = u.y; }
#pragma unroll
              for (int q = 0; q < PD; ++q) { const double2 u = pb[kBWP + q]; Eb[2 * q] = u.x; Eb[2 * q + 1] = u.y; }
              double M[2][2];
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
              // rows row0 .. row0 + 2 of Jc_a through the 2 x 2 core (a lane with two rows reads one column of the next
              // lane's share: an accumulator row that is never stored)
              double t0v[kRPL], t1v[kRPL];
#pragma unroll
              for (int a = 0; a < kRPL; ++a) {
                const double f0 = ra[row0 + a], f1 = ra[kBWP + row0 + a];
                t0v[a] = f0 * M[0][0] + f1 * M[1][0];
                t1v[a] = f0 * M[0][1] + f1 * M[1][1];
              }
              // Jc_b in pieces of five columns (the whole block would hold 20 more registers across the products)
#pragma unroll
              for (int hb = 0; hb < (BW + 4) / 5; ++hb) {
                double Fb0[5], Fb1[5];
#pragma unroll
                for (int q = 0; q < 5; ++q) { Fb0[q] = (5 * hb + q < kBWP) ? rb[5 * hb + q] : 0.0; Fb1[q] = (5 * hb + q < kBWP) ? rb[kBWP + 5 * hb + q] : 0.0; }
#pragma unroll
                for (int a = 0; a < kRPL; ++a)
#pragma unroll
                  for (int q = 0; q < 5; ++q)
                    if (5 * hb + q < BW)   // two chained FMAs (`acc += x y + z w` compiles to mul + fma + add)
                      acc[a * BW + 5 * hb + q] = __builtin_fma(t1v[a], Fb1[q], __builtin_fma(t0v[a], Fb0[q], acc[a * BW + 5 * hb + q]));
              }
            }
          }
#pragma unroll 1
          for (int g2 = 0; g2 < dP; ++g2) {
            const int t = base + dbase + g2;
            const unsigned mask = (t < ntr) ? s_tmask[t] : 0u;
            if (mask & dbit) {
              const unsigned sd = s_tslot[t * kRowBytesI + dlc];
              const double2* px = reinterpret_cast<const double2*>(s_rec + sd * RD);
              const double2 rr = px[kBWP + PD], rv = px[kBWP + PD + 1];
              const double fa0 = s_rec[sd * RD + da], fa1 = s_rec[sd * RD + kBWP + da];
              dacc[0] = __builtin_fma(fa1, rv.y, __builtin_fma(fa0, rv.x, dacc[0]));
              dacc[1] = __builtin_fma(fa1, rr.y, __builtin_fma(fa0, rr.x, dacc[1]));
              dacc[2] = __builtin_fma(fa1, fa1, __builtin_fma(fa0, fa0, dacc[2]));
            }
          }
        }
      }
      __builtin_amdgcn_s_setprio(0);
      __syncthreads();
      cur = nxt; tile = ntile; tile_ok = ntile_ok;
    }
    // ---- combine the track slices in a fixed order; the first replica of a lane role writes the rows it owns
    // (the lane's role again:  lix = lane index inside the track slice this lane serves: target block lix / NS, rows
    // sub_row0(lix % NS) .. of it; for the per-observation terms lane (local camera, row) = dix
    int lix, t0, tstride;
    bool slice_ok = true;
    if (G == 1) { const int g = lane / B; lix = lane - g * B; slice_ok = g < PS; t0 = wv * PS + g; tstride = NWV * PS; }
    else { lix = (wv % G) * 64 + lane; t0 = wv / G; tstride = NWV / G; }
    const int tix = lix / NS, sub = lix - tix * NS;
    const int row0 = sub_row0<KI>(sub);
    const bool has_tgt = slice_ok && tix < run.ntgt;
    int la = 0, lb = 0;
    if (has_tgt) { const unsigned us = P.frun_tgt[run.tgt_off + tix]; la = us & 0xffu; lb = us >> 8; }
    const int dix = (G == 1) ? lane : lix;
    const bool has_d = dix < BW * run.W;
    const int dlc = has_d ? dix / BW : 0, da = dix % BW;
    const int dP = (G == 1) ? PS : 1, dbase = (G == 1) ? wv * PS : wv / G;
    const unsigned tbits = has_tgt ? ((1u << la) | (1u << lb)) : 0xffffffffu;
    const unsigned dbit = has_d ? (1u << dlc) : 0x80000000u;
    const double diag_core = (has_tgt && la == lb) ? 1.0 : 0.0;
    double* scratch = s_rec;   // NA doubles per thread (NA <= RD)
    double* out = P.fpart + run.part_off;
    const int nrep = (G == 1) ? NWV * PS : NWV / G;
#pragma unroll
    for (int q = 0; q < NA; ++q) scratch[tid * NA + q] = acc[q];
    __syncthreads();
    if (has_tgt && tid == lix) {
      const int nrows = sub_rows<KI>(sub);
#pragma unroll 1
      for (int a = 0; a < kRPL; ++a) {
        if (a >= nrows) break;
        double v[BW];
#pragma unroll
        for (int q = 0; q < BW; ++q) v[q] = scratch[tid * NA + a * BW + q];
#pragma unroll 1
        for (int r = 1; r < nrep; ++r) {
          const int oth = (G == 1) ? ((r / PS) * 64 + (r % PS) * B + lix) : (lix + r * G * 64);
#pragma unroll
          for (int q = 0; q < BW; ++q) v[q] += scratch[oth * NA + a * BW + q];
        }
#pragma unroll
        for (int q = 0; q < BW; ++q) out[(size_t)tix * (BW * BW) + (row0 + a) * BW + q] = v[q];
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
      double* od = out + (size_t)run.ntgt * (BW * BW) + ((size_t)dlc * BW + da) * 3;
#pragma unroll
      for (int q = 0; q < 3; ++q) od[q] = v[q];
    }
  }
}

// One wave per item.  item = {row0, col0, nr | nc << 4 | kind << 8 | rows_are_group << 12 | cols_are_group << 13, beg, end,
// dst}; source q = {offset into part, r0 | c0 << 4 | transposed << 8 | row stride << 16}: element (i, j) of the item takes
// part[off + (r0 + i) stride + c0 + j]  (transposed: part[off + (r0 + j) stride + c0 + i]).  The sources of an item are dealt
// to 64 / (nr nc) lane groups, every group adds its share in list order with eight loads in flight, the groups are combined
// in order.  kinds: 0 = block of S, written negated (S = - sum: the partial blocks hold  What What^T - Jc^T Jc);
// 1 = the same, lower triangle only (a diagonal block); 2 = the three vectors (j = 0 rhs, 1 gradient, 2 squared column
// norms), written as they are; 3 = intermediate sum of a long list, to part[dst + e] (read by a second-level item).
// A group-side row / column index k is compact: it lands on parameter row_param(free mask, k) of the group's ten slots.
constexpr int SK_BLOCK = 0, SK_LOWER = 1, SK_VEC = 2, SK_CHUNK = 3;
THIP_DEV int compact_param(unsigned fm, int k) {
  for (int j = 0; j < k; ++j) fm &= fm - 1u;
  return fm ? __ffs(fm) - 1 : -1;
}
__global__ __launch_bounds__(256) void k_sum_items(int nitems, const int* __restrict__ items, const int2* __restrict__ src,
                                                   double* __restrict__ part, double* __restrict__ S, int n,
                                                   double* __restrict__ rhs, double* __restrict__ gc,
                                                   double* __restrict__ colsq, const unsigned* __restrict__ red_free) {
  const int it = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (it >= nitems) return;
  const int* d = items + 6 * it;
  const int row0 = d[0], col0 = d[1], code = d[2], beg = d[3], end = d[4], dst = d[5];
  const int nr = code & 15, nc = (code >> 4) & 15, kind = (code >> 8) & 15;
  const bool rgrp = (code >> 12) & 1, cgrp = (code >> 13) & 1;
  const int ne = nr * nc, nsl = 64 / ne;
  const int slot = lane / ne, e = lane - slot * ne;
  const int i = e / nc, j = e - i * nc;
  const bool live = slot < nsl;
  double v = 0.0;
  if (live) {
    auto addr = [&](int q) -> size_t {
      const int2 s = src[q];
      const int r0 = s.y & 15, c0 = (s.y >> 4) & 15, tr = (s.y >> 8) & 1, stride = (s.y >> 16) & 0xffff;
      return (size_t)s.x + (size_t)(tr ? (r0 + j) * stride + c0 + i : (r0 + i) * stride + c0 + j);
    };
    int q = beg + slot;
    for (; q + 7 * nsl < end; q += 8 * nsl) {
      double t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = part[addr(q + u * nsl)];
#pragma unroll
      for (int u = 0; u < 8; ++u) v += t[u];
    }
    for (; q < end; q += nsl) v += part[addr(q)];
  }
  // the lane groups in order (group s holds sources beg + s, beg + s + nsl, ...)
  double tot = v;
  for (int s = 1; s < nsl; ++s) {
    const double o = __shfl(v, min(63, e + s * ne), kWave);
    tot += o;
  }
  if (slot != 0 || !live) return;
  if (kind == SK_CHUNK) { part[(size_t)dst + e] = tot; return; }
  int ri = row0 + i, cj = col0 + j;
  // (a group-side item may start at compact row / column `off` of its group: row0 = 10 g + off -- the 10 x 10 group blocks of the
  // 16-row plan are two items of five compact rows)
  if (rgrp) { const int g10 = row0 / THEIA_MAX_INTRINSICS * THEIA_MAX_INTRINSICS; const int q = compact_param(red_free[row0 / THEIA_MAX_INTRINSICS], i + (row0 - g10)); if (q < 0) return; ri = g10 + q; }
  if (kind == SK_VEC) {
    if (j == 0) rhs[ri] = tot; else if (j == 1) gc[ri] = tot; else colsq[ri] = tot;
    return;
  }
  if (cgrp) { const int g10 = col0 / THEIA_MAX_INTRINSICS * THEIA_MAX_INTRINSICS; const int q = compact_param(red_free[col0 / THEIA_MAX_INTRINSICS], j + (col0 - g10)); if (q < 0) return; cj = g10 + q; }
  if (kind == SK_LOWER && cj > ri) return;
  S[(size_t)ri * n + cj] = -tot;
}

// rb.colsq by reduced index -> += the per-camera / per-group arrays of the Jacobi scaling (compute_scale)
__global__ void k_scatter_colsq(DevProblem P, const double* __restrict__ colsq_red, double* __restrict__ colsq_c,
                                double* __restrict__ colsq_i) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 6 * P.nc) {
    const int c = i / 6, q = i % 6, rc = P.cam_red[c];
    if (rc >= 0 && !((P.cam_mask[c] >> q) & 1u)) colsq_c[i] += colsq_red[P.ni + 6 * rc + q];
  }
  if (i < THEIA_MAX_INTRINSICS * P.ng_total) {
    const int g = i / THEIA_MAX_INTRINSICS, q = i % THEIA_MAX_INTRINSICS, gr = P.grp_red[g];
    if (gr >= 0 && ((P.grp_free[g] >> q) & 1u)) colsq_i[i] += colsq_red[THEIA_MAX_INTRINSICS * gr + q];
  }
}

}  // namespace

void launch_linearize_fused_intr(const DevProblem& P, const double* cam, const double* pts, const double* radius,
                                 const ReduceBuf& rb, double* Vinv, double* tile_part, hipStream_t st) {
  if (P.n_fruns == 0) return;
  launch_cam_prep(P, cam, P.intr, P.camrot, st);
  static const int wgs = [] { const char* e = getenv("THEIA_HIP_FUSED_WGS"); return e ? std::max(1, atoi(e)) : 512; }();
  const int grid = std::min(P.n_fruns, wgs);
  const bool trig = (P.model_mask & ~kModelsNoTrig) != 0;
// FOCAL_LENGTH | RADIAL_DISTORTION on the perspective / double-sphere / unified models (parameters 0, 5, 6): the pipelines' default
constexpr unsigned kMaskFocalRadial = (1u << 0) | (3u << 5);
#define THIP_LSI2(PD_, M_, LK_)                                                                              \
  do {                                                                                                         \
    if (P.fused_bw == 9 && P.fused_kmask == kMaskFocalRadial)                                                     \
      k_lin_schur_i<PD_, 4, M_, 3, kMaskFocalRadial, LK_><<<grid, 256, 0, st>>>(P, pts, radius, Vinv, tile_part);    \
    else if (P.fused_bw == 9) k_lin_schur_i<PD_, 4, M_, 3, 0u, LK_><<<grid, 256, 0, st>>>(P, pts, radius, Vinv, tile_part); \
    else if (P.fused_bw == 10) k_lin_schur_i<PD_, 4, M_, 4, 0u, LK_><<<grid, 256, 0, st>>>(P, pts, radius, Vinv, tile_part); \
    else if (P.fused_bw == 13) k_lin_schur_i<PD_, 4, M_, 7, 0u, LK_><<<grid, 256, 0, st>>>(P, pts, radius, Vinv, tile_part);   /* up to seven rows */ \
    else k_lin_schur_i<PD_, 4, M_, 10, 0u, LK_><<<grid, 256, 0, st>>>(P, pts, radius, Vinv, tile_part);   /* 16: up to ten rows */ \
  } while (0)
  // (two loss instances: the trivial loss without corrector code, everything else with the full corrector)
#define THIP_LSI(PD_, M_) do { if (P.loss_type == THEIA_LOSS_TRIVIAL) THIP_LSI2(PD_, M_, 0); else THIP_LSI2(PD_, M_, 2); } while (0)
  if (P.pd == 3) { if (trig) THIP_LSI(3, kModelsAll); else THIP_LSI(3, kModelsNoTrig); }
  else { if (trig) THIP_LSI(4, kModelsAll); else THIP_LSI(4, kModelsNoTrig); }
#undef THIP_LSI2
#undef THIP_LSI
  const int2* src = reinterpret_cast<const int2*>(P.sum_src);
  if (P.n_sum_items)
    k_sum_items<<<(P.n_sum_items + 3) / 4, 256, 0, st>>>(P.n_sum_items, P.sum_items, src, P.fpart, rb.S, P.n, rb.rhs, rb.gc,
                                                         rb.colsq, P.red_free);
  if (P.n_sum_items2)
    k_sum_items<<<(P.n_sum_items2 + 3) / 4, 256, 0, st>>>(P.n_sum_items2, P.sum_items + 6 * (size_t)P.n_sum_items, src, P.fpart,
                                                          rb.S, P.n, rb.rhs, rb.gc, rb.colsq, P.red_free);
}

void launch_scatter_colsq(const DevProblem& P, const double* colsq_red, double* colsq_c, double* colsq_i, hipStream_t st) {
  const int cnt = std::max(6 * P.nc, THEIA_MAX_INTRINSICS * P.ng_total);
  if (cnt > 0) k_scatter_colsq<<<(cnt + 255) / 256, 256, 0, st>>>(P, colsq_red, colsq_c, colsq_i);
}

}  // namespace thip
