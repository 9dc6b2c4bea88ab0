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
//   * per sub-chunk of 4 tiles (<= 256 observations, <= 128 tracks), phase L: lane = observation linearises it
//     (closed-form Jacobians, per-camera rotation terms precomputed by k_cam_prep), the track sums V_p / g_p
//     are log-step shuffle scans inside the wave, V_p^-1 = Li^T Li, and the lane leaves
//     What = F^T E Li^T (6 x PD), F (2 x 6), r in LDS, plus its slot in the track's row of a
//     (track, local camera) -> slot table;
//   * phase S: lane = TARGET BLOCK (la, lb) of the run's local camera pairs.  Every wave walks the tracks of its
//     slice; for each track the lane looks up the two slots and accumulates What_a What_b^T into 36 REGISTERS.
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
#include "ba_lane.h"

#include <algorithm>
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
  if (c == 0 && P.frun_next) *P.frun_next = 0;   // head of k_lin_schur's run queue (the next kernel on the stream)
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
#define THIP_FUSED_WAVES 2   // workgroups per CU the register allocation aims at (LDS allows 3)
#endif
#ifdef THIP_PHASE_L_CALL
#define THIP_PHASE_L_ATTR __attribute__((noinline))
#else
#define THIP_PHASE_L_ATTR __forceinline__
#endif
template <int PD> constexpr int rec_doubles() { return PD == 3 ? 22 : 26; }   // 12 + 2 PD + 2 + 2 used

// Phase L of one sub-chunk (lane = observation): linearise, reduce V_p / g_p over the track, invert, leave the record
// {F | Ehat | r} and the track's slot-table row in LDS.  NOT inlined on purpose: the caller keeps 45 FP64 accumulators
// alive across this phase; as a call they are saved once around it (callee-saved registers) instead of pushing the
// register allocation of the pair-product loop into scratch.
template <int PD, int TPS, unsigned MODELS>
__device__ THIP_PHASE_L_ATTR void fused_phase_l(const DevProblem* __restrict__ Pp, const FusedRun* __restrict__ runp,
                                                        const double* __restrict__ pts, double inv_radius, int sc,
                                                        double* __restrict__ Vinv, double* __restrict__ tile_part,
                                                        double* __restrict__ s_rec,
                                                        uint8_t* __restrict__ s_tslot, unsigned* __restrict__ s_tmask) {
  constexpr int NT = PD * (PD + 1) / 2;
  constexpr int RD = rec_doubles<PD>();
  const DevProblem& P = *Pp;
  const FusedRun& run = *runp;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
      const int tile = run.tile0 + TPS * sc + wv;
      const bool tile_ok = tile < run.tile0 + run.ntiles;
      const int cnt = tile_ok ? P.tile_count[tile] : 0;
      const int start = tile_ok ? P.tile_start[tile] : 0;
      const bool active = lane < cnt && !(P.fused_dbg & 2);
      LaneLin<PD> L;
      lane_linearize<PD, true, false, true, MODELS>(P, P.camrot, pts, start + lane, active, lane, L);
      const Segment sg = lane_segment_all(L.p, lane);
      double tot[NT + PD];
#pragma unroll
      for (int a = 0; a < PD; ++a) {
#pragma unroll
        for (int b = 0; b <= a; ++b) tot[lidx(a, b)] = L.Jt[a] * L.Jt[b] + L.Jt[PD + a] * L.Jt[PD + b];
        tot[NT + a] = L.Jt[a] * L.r[0] + L.Jt[PD + a] * L.r[1];
      }
      segment_allsum_log<NT + PD>(sg, lane, tot);
      const int o = start + lane;
      const int tl = active ? P.obs_tl[o] : 0;
      const unsigned lc = active ? P.obs_lc[o] : 0xffu;
      const unsigned tmask = segment_or(sg, lane, (active && lc != 0xffu) ? (1u << lc) : 0u);
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
      double gh[PD];   // ghat = Li g, in every lane of the track (the record carries r - Ehat ghat)
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
          for (int a = 0; a < PD; ++a) gmax = fmax(gmax, fabs(g[a] / P.scale_p[(size_t)PD * L.p + a]));
        }
      }
      if (active && lc != 0xffu) {
        const int slot = wv * 64 + lane;
        double eh[2 * PD];   // Ehat = E Li^T
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
        for (int q = 0; q < 6; ++q) R[q] = make_double2(L.Jc[2 * q], L.Jc[2 * q + 1]);
#pragma unroll
        for (int q = 0; q < PD; ++q) R[6 + q] = make_double2(eh[2 * q], eh[2 * q + 1]);
        R[6 + PD] = make_double2(L.r[0], L.r[1]);
        double v1[2];   // r - Ehat ghat: the camera's rhs row is F^T (r - Ehat ghat)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          double sm = 0.0;
#pragma unroll
          for (int b = 0; b < PD; ++b) sm += eh[i * PD + b] * gh[b];
          v1[i] = L.r[i] - sm;
        }
        R[6 + PD + 1] = make_double2(v1[0], v1[1]);
        s_tslot[tl * kRowBytes + lc] = (uint8_t)slot;
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

// tile_part layout as k_lin_obs: [ntiles][4] = {cost, gmax_points, invalid, notpd}
template <int PD, int TPS, unsigned MODELS>
__global__ __launch_bounds__(64 * TPS, THIP_FUSED_WAVES) void k_lin_schur(DevProblem P, const double* __restrict__ pts,
                                                           const double* __restrict__ radius_p,
                                                           double* __restrict__ Vinv, double* __restrict__ tile_part) {
  constexpr int NT = PD * (PD + 1) / 2;
  constexpr int RD = rec_doubles<PD>();
  constexpr int OE = 12, OR = 12 + 2 * PD;           // offsets of Ehat and r inside a record
  constexpr int SUB = TPS * kWave;                    // observations per sub-chunk
  constexpr int SUBT = TPS * kFusedTileTracks;        // tracks per sub-chunk
  constexpr int NWV = TPS;                            // waves of the workgroup
  __shared__ __attribute__((aligned(16))) double s_rec[SUB * RD];
  __shared__ uint8_t s_tslot[SUBT * kRowBytes];   // (track, local camera) -> record slot, valid where the mask bit is set
  __shared__ unsigned s_tmask[SUBT];              // local cameras of a track
  static_assert(SUB * 18 <= SUB * RD, "slice-combination scratch does not fit the record buffer");

  __shared__ DevProblem s_P;     // phase L is a real call: it reads the problem through these LDS copies
  __shared__ FusedRun s_run;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) s_P = P;
  const double radius = *radius_p;
  const double inv_radius = 1.0 / radius;
  // persistent workgroups: a few hundred of them take runs from a queue (per-workgroup set-up -- and what the
  // register allocator parks in scratch at entry -- is paid once per workgroup, not once per run)
  __shared__ int s_next;
  for (;;) {
  __syncthreads();   // the previous run's slice combination is done with the LDS scratch (and s_run, s_next)
  if (tid == 0) s_next = atomicAdd(P.frun_next, 1);   // runs are taken from a queue, longest first (ba_solver.hip)
  __syncthreads();
  if (s_next >= P.n_fruns) break;
  const FusedRun run = P.fruns[P.frun_order[s_next]];
  if (tid == 0) s_run = run;
  const int nsc = (run.ntiles + TPS - 1) / TPS;
  __syncthreads();

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

  for (int sc = 0; sc < nsc; ++sc) {
    // ------------------------------------------------------------------ phase L: lane = observation
    // the problem is read through its LDS copy: with the kernel-argument struct (SGPR bases, global instead of flat loads)
    // the timing is the same and 11 more VGPRs spill (WRITE_SIZE +50 MB per launch at 1000 views / 500k tracks)
    fused_phase_l<PD, TPS, MODELS>(&s_P, &s_run, pts, inv_radius, sc, Vinv, tile_part, s_rec, s_tslot, s_tmask);
    __syncthreads();
    // ------------------------------------------------------------------ phase S: lane = target block
    if (!(P.fused_dbg & 1)) {
      const int last_tile = min(run.tile0 + TPS * sc + TPS - 1, run.tile0 + run.ntiles - 1);
      const int ntr = P.tile_trk_end[last_tile];
#pragma unroll 1
      for (int base = 0; base < ntr; base += tstride) {   // wave-uniform trip count
        // pair products: lane = target block (la, lb) of the track slice it serves
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
            // column da of F: two LDS reads (no dynamic register indexing)
            const double fa0 = s_rec[sd * RD + da], fa1 = s_rec[sd * RD + 6 + da];
            dacc[0] = __builtin_fma(fa1, rv.y, __builtin_fma(fa0, rv.x, dacc[0]));
            dacc[1] = __builtin_fma(fa1, rr.y, __builtin_fma(fa0, rr.x, dacc[1]));
            dacc[2] = __builtin_fma(fa1, fa1, __builtin_fma(fa0, fa0, dacc[2]));
          }
        }
      }
    }
    __syncthreads();
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
    if (has_tgt && tid == tix) {   // the first replica of the target adds the others, in slice order
      double v[18];
#pragma unroll
      for (int q = 0; q < 18; ++q) v[q] = scratch[tid * 18 + q];
#pragma unroll 1
      for (int r = 1; r < nrep; ++r) {
        const int oth = (G == 1) ? ((r / PS) * 64 + (r % PS) * B + tix) : (tix + r * G * 64);
#pragma unroll
        for (int q = 0; q < 18; ++q) v[q] += scratch[oth * 18 + q];
      }
#pragma unroll
      for (int q = 0; q < 18; ++q) out[(size_t)tix * 36 + 18 * h + q] = v[q];
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
  }   // runs of this workgroup
}

// One wave per S block (ri, rj): the partial blocks of the runs that touch it are added in run order and the block
// is WRITTEN (never accumulated): S_ij = - sum What_a What_b^T; for a camera (ri == rj) also the per-observation
// sums  F^T F - ..,  rhs = F^T r - What ghat,  gradient, column norms.
__global__ __launch_bounds__(256) void k_schur_sum(int nitems, const int* __restrict__ items, const int* __restrict__ src,
                                                   const double* __restrict__ part, double* __restrict__ S, int n,
                                                   double* __restrict__ rhs, double* __restrict__ gc,
                                                   double* __restrict__ colsq) {
  const int it = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
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
  launch_cam_prep(P, cam, P.intr, P.camrot, st);
  static const int wgs = [] { const char* e = getenv("THEIA_HIP_FUSED_WGS"); return e ? std::max(1, atoi(e)) : 512; }();   // 2 per CU
  const int grid = std::min(P.n_fruns, wgs);
  const bool trig = (P.model_mask & ~kModelsNoTrig) != 0;   // FOV / fisheye groups present
  if (P.pd == 3) {
    if (trig) k_lin_schur<3, 4, kModelsAll><<<grid, 256, 0, st>>>(P, pts, radius, Vinv, tile_part);
    else k_lin_schur<3, 4, kModelsNoTrig><<<grid, 256, 0, st>>>(P, pts, radius, Vinv, tile_part);
  } else {
    if (trig) k_lin_schur<4, 4, kModelsAll><<<grid, 256, 0, st>>>(P, pts, radius, Vinv, tile_part);
    else k_lin_schur<4, 4, kModelsNoTrig><<<grid, 256, 0, st>>>(P, pts, radius, Vinv, tile_part);
  }
  if (P.n_sum_items)
    k_schur_sum<<<(P.n_sum_items + 3) / 4, 256, 0, st>>>(P.n_sum_items, P.sum_items, P.sum_src, P.fpart, rb.S, P.n, rb.rhs,
                                                         rb.gc, rb.colsq);
}

}  // namespace thip
