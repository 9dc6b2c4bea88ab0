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
#include "ba_lane.h"

namespace thip {
namespace {

constexpr int kSub = 4 * kWave;                       // observations per sub-chunk
constexpr int kSubTracks = 4 * kFusedTileTracks;      // 128
constexpr int kRowBytes = 24;                         // slot-table row: kFusedMaxCams rounded up to 8
constexpr int kAux = 14;                              // F (12) | r (2)
static_assert(kFusedMaxCams <= kRowBytes, "slot-table row too short");

__global__ __launch_bounds__(256) void k_cam_prep(int nc, const double* __restrict__ cam, double* __restrict__ camrot) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < nc) camrot_store(cam + 6 * (size_t)c, camrot + (size_t)kCamRot * c);
}

// tile_part layout as k_lin_obs: [ntiles][4] = {cost, gmax_points, invalid, notpd}
template <int PD>
__global__ __launch_bounds__(256, 2) void k_lin_schur(DevProblem P, const double* __restrict__ pts,
                                                      const double* __restrict__ radius_p, double* __restrict__ Vinv,
                                                      double* __restrict__ tile_part) {
  constexpr int NT = PD * (PD + 1) / 2;
  constexpr int NW = 6 * PD;
  __shared__ __attribute__((aligned(16))) double s_what[kSub * NW];
  __shared__ __attribute__((aligned(16))) double s_aux[kSub * kAux];
  __shared__ double s_ghat[kSubTracks * PD];
  __shared__ uint8_t s_tslot[kSubTracks * kRowBytes];   // (track, local camera) -> record slot, valid where the mask bit is set
  __shared__ unsigned s_tmask[2][kSubTracks];            // local cameras of a track (double buffered: cleared one sub-chunk ahead)

  const FusedRun run = P.fruns[blockIdx.x];
  const double radius = *radius_p;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  // phase-S role of this lane
  const int G = run.G, S = 4 / G;
  const int slice = wv / G, tix = (wv % G) * 64 + lane;
  const bool has_tgt = tix < run.ntgt;
  int la = 0, lb = 0;
  if (has_tgt) { const unsigned us = P.frun_tgt[run.tgt_off + tix]; la = us & 0xffu; lb = us >> 8; }
  const bool has_d = tix < 6 * run.W;
  const int dlc = has_d ? tix / 6 : 0, da = tix % 6;
  double acc[36], dacc[9];
#pragma unroll
  for (int k = 0; k < 36; ++k) acc[k] = 0.0;
#pragma unroll
  for (int k = 0; k < 9; ++k) dacc[k] = 0.0;

  if (tid < kSubTracks) s_tmask[0][tid] = 0u;
  __syncthreads();

  const int nsc = (run.ntiles + 3) >> 2;
  for (int sc = 0; sc < nsc; ++sc) {
    const int buf = sc & 1;
    // ------------------------------------------------------------------ phase L: lane = observation
    {
      const int tile = run.tile0 + 4 * sc + wv;
      const bool tile_ok = tile < run.tile0 + run.ntiles;
      const int cnt = tile_ok ? P.tile_count[tile] : 0;
      const int start = tile_ok ? P.tile_start[tile] : 0;
      const bool active = lane < cnt;
      LaneLin<PD> L;
      lane_linearize<PD, true, false, true>(P, P.camrot, pts, start + lane, active, lane, L);
      const Segment sg = lane_segment(L.p, lane);
      double tot[NT + PD];
#pragma unroll
      for (int a = 0; a < PD; ++a) {
#pragma unroll
        for (int b = 0; b <= a; ++b) tot[lidx(a, b)] = L.Jt[a] * L.Jt[b] + L.Jt[PD + a] * L.Jt[PD + b];
        tot[NT + a] = L.Jt[a] * L.r[0] + L.Jt[PD + a] * L.r[1];
      }
      segment_allsum_log<NT + PD>(sg, lane, tot);
      double V[NT], Vi[NT], g[PD];
#pragma unroll
      for (int q = 0; q < NT; ++q) V[q] = tot[q];
#pragma unroll
      for (int a = 0; a < PD; ++a) { g[a] = tot[NT + a]; V[lidx(a, a)] += fmin(fmax(V[lidx(a, a)], 1e-6), 1e32) / radius; }
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
      const int o = start + lane;
      const int tl = active ? P.obs_tl[o] : 0;
      const unsigned lc = active ? P.obs_lc[o] : 0xffu;
      if (active && sg.head) {
        double gh[PD];
#pragma unroll
        for (int a = 0; a < PD; ++a) {   // ghat = Li g
          double s = 0.0;
#pragma unroll
          for (int q = 0; q <= a; ++q) s += Li[a][q] * g[q];
          gh[a] = s;
        }
#pragma unroll
        for (int a = 0; a < PD; ++a) s_ghat[tl * PD + a] = gh[a];
        if (!L.pconst) {
#pragma unroll
          for (int q = 0; q < NT; ++q) Vinv[(size_t)NT * L.p + q] = Vi[q];
#pragma unroll
          for (int a = 0; a < PD; ++a) gmax = fmax(gmax, fabs(g[a] / P.scale_p[(size_t)PD * L.p + a]));
        }
      }
      if (active && lc != 0xffu) {
        const int slot = wv * 64 + lane;
        double w0[NW], w[NW];
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
          for (int b = 0; b < PD; ++b) w0[a * PD + b] = L.Jc[a] * L.Jt[b] + L.Jc[6 + a] * L.Jt[PD + b];
#pragma unroll
        for (int a = 0; a < 6; ++a)   // What = W Li^T
#pragma unroll
          for (int b = 0; b < PD; ++b) {
            double s = 0.0;
#pragma unroll
            for (int q = 0; q <= b; ++q) s += w0[a * PD + q] * Li[b][q];
            w[a * PD + b] = s;
          }
        double2* R = reinterpret_cast<double2*>(s_what + slot * NW);
#pragma unroll
        for (int q = 0; q < NW / 2; ++q) R[q] = make_double2(w[2 * q], w[2 * q + 1]);
        double2* A2 = reinterpret_cast<double2*>(s_aux + slot * kAux);
#pragma unroll
        for (int q = 0; q < 6; ++q) A2[q] = make_double2(L.Jc[2 * q], L.Jc[2 * q + 1]);
        A2[6] = make_double2(L.r[0], L.r[1]);
        s_tslot[tl * kRowBytes + lc] = (uint8_t)slot;
        atomicOr(&s_tmask[buf][tl], 1u << lc);
      }
      const double cost = wave_sum(L.cost);
      gmax = wave_max(gmax);
      const double inval = wave_sum((active && !L.valid) ? 1.0 : 0.0);
      const double npd = wave_sum((active && !pd_ok && sg.head) ? 1.0 : 0.0);
      if (lane == 0 && tile_ok) {
        tile_part[4 * (size_t)tile + 0] = cost;
        tile_part[4 * (size_t)tile + 1] = gmax;
        tile_part[4 * (size_t)tile + 2] = inval;
        tile_part[4 * (size_t)tile + 3] = npd;
      }
    }
    __syncthreads();
    // ------------------------------------------------------------------ phase S: lane = target block
    {
      // the other slot table is free now: clear it for the next sub-chunk
      if (tid < kSubTracks) s_tmask[buf ^ 1][tid] = 0u;
      const int last_tile = min(run.tile0 + 4 * sc + 3, run.tile0 + run.ntiles - 1);
      const int ntr = P.tile_trk_end[last_tile];
      const unsigned tbits = has_tgt ? ((1u << la) | (1u << lb)) : 0xffffffffu;   // no target: never a subset (bit 31 unused)
      const unsigned dbit = has_d ? (1u << dlc) : 0x80000000u;
      for (int t = slice; t < ntr; t += S) {
        const uint8_t* row = s_tslot + t * kRowBytes;
        const unsigned mask = s_tmask[buf][t];
        const unsigned sa = row[la], sb = row[lb], sd = row[dlc];
        if ((mask & tbits) == tbits) {
          double A[NW], B[NW];
          const double2* pa = reinterpret_cast<const double2*>(s_what + sa * NW);
          const double2* pb = reinterpret_cast<const double2*>(s_what + sb * NW);
#pragma unroll
          for (int q = 0; q < NW / 2; ++q) { const double2 u = pa[q]; A[2 * q] = u.x; A[2 * q + 1] = u.y; }
#pragma unroll
          for (int q = 0; q < NW / 2; ++q) { const double2 u = pb[q]; B[2 * q] = u.x; B[2 * q + 1] = u.y; }
#pragma unroll
          for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b = 0; b < 6; ++b) {
              double s = acc[a * 6 + b];
#pragma unroll
              for (int q = 0; q < PD; ++q) s += A[a * PD + q] * B[b * PD + q];
              acc[a * 6 + b] = s;
            }
        }
        if (mask & dbit) {
          double F[12], r2[2];
          const double2* px = reinterpret_cast<const double2*>(s_aux + sd * kAux);
#pragma unroll
          for (int q = 0; q < 6; ++q) { const double2 u = px[q]; F[2 * q] = u.x; F[2 * q + 1] = u.y; }
          { const double2 u = px[6]; r2[0] = u.x; r2[1] = u.y; }
          double fa0 = 0.0, fa1 = 0.0;   // F[da], F[6 + da] without dynamic register indexing
#pragma unroll
          for (int q = 0; q < 6; ++q) { if (q == da) { fa0 = F[q]; fa1 = F[6 + q]; } }
#pragma unroll
          for (int q = 0; q < 6; ++q) dacc[q] += fa0 * F[q] + fa1 * F[6 + q];
          const double jr = fa0 * r2[0] + fa1 * r2[1];
          double wg = 0.0;
#pragma unroll
          for (int q = 0; q < PD; ++q) wg += s_what[sd * NW + da * PD + q] * s_ghat[t * PD + q];
          dacc[6] += jr - wg;
          dacc[7] += jr;
          dacc[8] += fa0 * fa0 + fa1 * fa1;
        }
      }
    }
    __syncthreads();
  }
  // ---------------------------------------------------------------- combine the track slices, fixed order
  double* scratch = s_what;   // 256 x 18 doubles
  double* out = P.fpart + run.part_off;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (h) __syncthreads();
#pragma unroll
    for (int q = 0; q < 18; ++q) scratch[tid * 18 + q] = acc[18 * h + q];
    __syncthreads();
    if (slice == 0 && has_tgt) {
#pragma unroll
      for (int q = 0; q < 18; ++q) {
        double v = scratch[tid * 18 + q];
        for (int s = 1; s < S; ++s) v += scratch[(tid + s * G * 64) * 18 + q];
        out[(size_t)tix * 36 + 18 * h + q] = v;
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 9; ++q) scratch[tid * 9 + q] = dacc[q];
  __syncthreads();
  if (slice == 0 && has_d) {
    double* od = out + (size_t)run.ntgt * 36 + (size_t)tix * 9;
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      double v = scratch[tid * 9 + q];
      for (int s = 1; s < S; ++s) v += scratch[(tid + s * G * 64) * 9 + q];
      od[q] = v;
    }
  }
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
  if (ri != rj) {
    if (lane >= 36) return;
    double v = 0.0;
    for (int q = tbeg; q < tend; ++q) v += part[(size_t)src[q] + lane];
    S[(size_t)(6 * ri + lane / 6) * n + 6 * rj + lane % 6] = -v;
    return;
  }
  if (lane >= 54) return;
  const int a = lane / 9, j = lane % 9;
  double v = 0.0;
  for (int q = dbeg; q < dend; ++q) v += part[(size_t)src[q] + lane];
  if (j < 6) {
    double w = 0.0;
    for (int q = tbeg; q < tend; ++q) w += part[(size_t)src[q] + a * 6 + j];
    if (j <= a) S[(size_t)(6 * ri + a) * n + 6 * ri + j] = v - w;
  } else if (j == 6) {
    rhs[6 * ri + a] = v;
  } else if (j == 7) {
    gc[6 * ri + a] = v;
  } else {
    colsq[6 * ri + a] = v;
  }
}

}  // namespace

void launch_linearize_fused(const DevProblem& P, const double* cam, const double* pts, const double* radius,
                            const ReduceBuf& rb, double* Vinv, double* tile_part, hipStream_t st) {
  if (P.n_fruns == 0) return;
  k_cam_prep<<<(P.nc + 255) / 256, 256, 0, st>>>(P.nc, cam, P.camrot);
  if (P.pd == 3) k_lin_schur<3><<<P.n_fruns, 256, 0, st>>>(P, pts, radius, Vinv, tile_part);
  else k_lin_schur<4><<<P.n_fruns, 256, 0, st>>>(P, pts, radius, Vinv, tile_part);
  if (P.n_sum_items)
    k_schur_sum<<<(P.n_sum_items + 3) / 4, 256, 0, st>>>(P.n_sum_items, P.sum_items, P.sum_src, P.fpart, rb.S, P.n, rb.rhs,
                                                         rb.gc, rb.colsq);
}

}  // namespace thip
