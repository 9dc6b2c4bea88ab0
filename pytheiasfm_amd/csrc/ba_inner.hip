// Inner iterations of the bundle adjustment (Ceres 2.2 TrustRegionMinimizer::DoInnerIterationsIfNeeded +
// CoordinateDescentMinimizer, enabled by default: bundle_adjustment.h:144, wired at bundle_adjuster.cc:75,329-333).
//
// After every trust-region candidate Ceres runs one sweep of block coordinate descent over the reversed elimination
// ordering the reference hands it (`inner_iteration_ordering->Reverse()`): group 0 = camera extrinsics, 1 = shared
// intrinsics, 2 = points.  The blocks of a group are independent (no residual touches two of them), each is minimised
// ALONE -- every other block held at its current value, later groups see the earlier groups' results -- by a fresh
// Levenberg-Marquardt solve with the default Minimizer::Options (coordinate_descent_minimizer.cc:229-262: LM, DENSE_QR,
// at most 50 iterations, tolerances 1e-6 / 1e-10 / 1e-8, initial radius 1e4, Jacobi scaling) over the residual blocks
// that depend on it, loss functions included.
//
// On the device the three groups are three launches over the candidate buffers:
//   k_inner_views   one WAVE per variable camera: lane = observation of the camera's list (built at create()), 6 x 6
//                   normal equations by wave reduction (the camera's prior rows included);
//   k_inner_groups  one WORKGROUP per variable intrinsics group: 10 x 10 over every observation of the group's cameras;
//   k_inner_tracks  one THREAD per variable point: PD x PD (SphereManifold<4> tangent or plain XYZW).
// Every kernel reads a device flag first (inner iterations switch themselves off when their relative progress drops
// below inner_iteration_tolerance = 1e-3, trust_region_minimizer.cc) and returns at once when it is clear.
// The normal equations are solved by Cholesky instead of Ceres' QR of [J; D]: the same step up to round-off.
//
// One block's LM solve follows trust_region_minimizer.cc with LevenbergMarquardtStrategy (the rules of ba_solver.hip's
// lm_control_body): Jacobi scaling 1 / (1 + sqrt(H_qq)) from the first linearisation; D = clamp(H_qq, 1e-6, 1e32) / radius;
// a step is valid iff the solve succeeded and its model-cost change is positive (else radius /= decrease_factor, five in a
// row end the solve); parameter / function tolerance on the candidate; rho > 1e-3 accepts with
// radius /= max(1/3, 1 - (2 rho - 1)^3); "if the optimization is a failure ... it won't change the parameters".
#define THIP_LEAN_SINCOS 1 // ba_device.h: SphereManifold::Plus through polynomials for small steps (the point solves)
#include "ba_kernels.h"
#include "ba_device.h"
#include "ba_priors.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>

namespace thip {
namespace {

constexpr int kInnerMaxIterations = 50;              // Solver::Options defaults behind Minimizer::Options()
constexpr double kInnerFunctionTolerance = 1e-6, kInnerGradientTolerance = 1e-10, kInnerParameterTolerance = 1e-8;
constexpr double kInnerMaxRadius = 1e16;

__device__ __forceinline__ int itri(int a, int b) { return a * (a + 1) / 2 + b; }

// ------------------------------------------------------------------ team solves (cameras, shared intrinsics)
// Round 5.  A block whose residuals are summed by a TEAM of threads (a workgroup per camera; up to 32 workgroups per intrinsics
// group) runs block_lm's rules as a state machine in LDS: the team evaluates what the state asks for -- the normal equations at
// x, or the cost at the candidate -- and leaves the totals in LDS; ONE thread advances the state (its H / L / step arrays are
// LDS arrays: dynamic indexing without scratch; rounds 3-4 kept them per thread, 1120 B of scratch per lane in k_inner_groups)
// and writes what to evaluate next.  Every workgroup of a team sees the same totals, so the same transitions.
// As in the point solves the Jacobi scaling is applied to the SUMS (H_ab s_a s_b, g_a s_a), which saves Ceres' second
// evaluation at the starting point.
enum { TL_LIN = 0, TL_COST = 1, TL_DONE = 2 };
template <int N>
struct TeamLm {
  static constexpr int NT = N * (N + 1) / 2;
  double x[N], xc[N], scale[N], H[NT], g[N], L[NT], y[N], z[N];
  double x_cost, gmax, radius, decrease_factor, x_norm, mcc, stepsq, xnormsq;
  double base_normsq;   // |.|^2 of the entries of the parameter block that are not among the N (held entries of a compacted block)
  int iter, invalid_steps, step_successful, first, phase, frozen;   // frozen: bit q = parameter q is held (its column is zero)
};
template <int N>
__device__ __forceinline__ void team_lm_init(TeamLm<N>& S, const double* x0, unsigned frozen, double base_normsq = 0.0) {
  double xn = base_normsq;
  S.base_normsq = base_normsq;
  for (int q = 0; q < N; ++q) { S.x[q] = x0[q]; S.xc[q] = x0[q]; S.scale[q] = 1.0; xn += x0[q] * x0[q]; }
  S.x_norm = sqrt(xn);
  S.radius = 1e4; S.decrease_factor = 2.0; S.iter = 0; S.invalid_steps = 0; S.step_successful = 1; S.first = 1;
  S.phase = TL_LIN; S.frozen = (int)frozen; S.x_cost = 0.0; S.gmax = 0.0;
}
// (H + D) y = g on the LDS arrays; false if not positive definite
template <int N>
__device__ __forceinline__ bool team_lm_solve(TeamLm<N>& S) {
  for (int i = 0; i < N; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = S.H[itri(i, j)] + (i == j ? fmin(fmax(S.H[itri(i, i)], 1e-6), 1e32) / S.radius : 0.0);
      for (int k = 0; k < j; ++k) s -= S.L[itri(i, k)] * S.L[itri(j, k)];
      if (i == j) { if (!(s > 0.0)) return false; S.L[itri(i, i)] = sqrt(s); }
      else S.L[itri(i, j)] = s / S.L[itri(j, j)];
    }
  for (int i = 0; i < N; ++i) {
    double s = S.g[i];
    for (int k = 0; k < i; ++k) s -= S.L[itri(i, k)] * S.z[k];
    S.z[i] = s / S.L[itri(i, i)];
  }
  for (int i = N - 1; i >= 0; --i) {
    double s = S.z[i];
    for (int k = i + 1; k < N; ++k) s -= S.L[itri(k, i)] * S.y[k];
    S.y[i] = s / S.L[itri(i, i)];
  }
  return true;
}
// tot: {J'J (packed lower, UNSCALED), J'r, cost, invalid} after TL_LIN, {cost, invalid} (at [NT + N], [NT + N + 1]) after TL_COST
template <int N>
__device__ __forceinline__ void team_lm_advance(TeamLm<N>& S, const double* tot) {
  constexpr int NT = N * (N + 1) / 2;
  if (S.phase == TL_LIN) {
    if (S.first) {
      S.first = 0;
      if (tot[NT + N + 1] > 0.0 || !isfinite(tot[NT + N])) { S.phase = TL_DONE; return; }   // failure: the parameters stay as they are
      for (int q = 0; q < N; ++q) S.scale[q] = 1.0 / (1.0 + sqrt(tot[itri(q, q)]));
    }
    for (int a = 0; a < N; ++a) {
      for (int b = 0; b <= a; ++b) S.H[itri(a, b)] = tot[itri(a, b)] * S.scale[a] * S.scale[b];
      S.g[a] = tot[NT + a] * S.scale[a];
    }
    S.x_cost = tot[NT + N];
    S.gmax = 0.0;
    for (int q = 0; q < N; ++q) S.gmax = fmax(S.gmax, fabs(S.g[q] / S.scale[q]));
  } else {   // the candidate's cost
    S.invalid_steps = 0;
    double cand_cost = tot[NT + N];
    if (tot[NT + N + 1] > 0.0 || !isfinite(cand_cost)) cand_cost = DBL_MAX;
    if (sqrt(S.stepsq) <= kInnerParameterTolerance * (S.x_norm + kInnerParameterTolerance)) { S.phase = TL_DONE; return; }
    const double cost_change = S.x_cost - cand_cost;
    if (fabs(cost_change) <= kInnerFunctionTolerance * S.x_cost) { S.phase = TL_DONE; return; }
    const double rho = cost_change / S.mcc;
    if (rho > 1e-3) {
      for (int q = 0; q < N; ++q) S.x[q] = S.xc[q];
      S.x_norm = sqrt(S.xnormsq);
      const double t = 2.0 * rho - 1.0;
      S.radius = fmin(kInnerMaxRadius, S.radius / fmax(1.0 / 3.0, 1.0 - t * t * t));
      S.decrease_factor = 2.0; S.step_successful = 1;
      S.phase = TL_LIN;   // re-linearise at the new point, then try a step from there
      return;
    }
    S.radius /= S.decrease_factor; S.decrease_factor *= 2.0; S.step_successful = 0;
  }
  // try steps until one needs its cost (or the solve ends)
  while (true) {
    if (S.iter >= kInnerMaxIterations || (S.step_successful && S.gmax <= kInnerGradientTolerance) || S.radius <= 1e-32) { S.phase = TL_DONE; return; }
    ++S.iter;
    const bool pd = team_lm_solve<N>(S);
    double yg = 0.0, yHy = 0.0;
    if (pd)
      for (int a = 0; a < N; ++a) {
        yg += S.y[a] * S.g[a];
        double row = 0.0;
        for (int b = 0; b < N; ++b) row += S.H[a >= b ? itri(a, b) : itri(b, a)] * S.y[b];
        yHy += S.y[a] * row;
      }
    S.mcc = yg - 0.5 * yHy;
    double stepsq = 0.0, xnormsq = S.base_normsq;
    for (int q = 0; q < N; ++q) {
      const double xq = S.x[q];
      const double cq = (pd && !((S.frozen >> q) & 1)) ? xq + (-S.y[q] * S.scale[q]) : xq;
      S.xc[q] = cq;
      stepsq += (xq - cq) * (xq - cq); xnormsq += cq * cq;
    }
    S.stepsq = stepsq; S.xnormsq = xnormsq;
    if (pd && isfinite(S.mcc) && isfinite(stepsq) && S.mcc > 0.0) { S.phase = TL_COST; return; }
    if (++S.invalid_steps >= 5) { S.phase = TL_DONE; return; }
    S.radius /= S.decrease_factor; S.decrease_factor *= 2.0; S.step_successful = 0;
  }
}

__device__ __forceinline__ double wave_sum64(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
// workgroup total of NS per-thread sums -> tot[0 .. NS) in LDS (wave trees, then the four waves in order); red: [4][NS] of LDS
template <int NS>
__device__ __forceinline__ void team_reduce(const double (&acc)[NS], double* __restrict__ red, double* __restrict__ tot) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  __syncthreads();   // (red / tot of the previous pass have been read)
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    const double v = wave_sum64(acc[k]);
    if (lane == 0) red[wv * NS + k] = v;
  }
  __syncthreads();
  if (tid < NS) tot[tid] = ((red[tid] + red[NS + tid]) + red[2 * NS + tid]) + red[3 * NS + tid];
  __syncthreads();
}

struct ObsRef { double2 uv; double six, siy; int cam, pt; bool depth_row; };
__device__ __forceinline__ ObsRef load_obs(const InnerArgs& A, int o) {
  ObsRef r;
  r.uv = A.P.obs_uv[o];
  // the optional arrays always point at mapped memory here (launch_inner_* substitutes obs_uv / obs_cam when they are
  // absent): the compiler was seen hoisting these loads above the null test
  const double2 si = A.P.obs_si[o];
  r.six = A.has_si ? si.x : 1.0; r.siy = A.has_si ? si.y : 1.0;
  r.cam = A.P.obs_cam[o]; r.pt = A.P.obs_pt[o];
  r.depth_row = A.has_kind && A.P.obs_kind[o];
  return r;
}
__device__ __forceinline__ double obs_loss(const InnerArgs& A, const ObsRef& r, double s, double* rho1) {
  return loss_eval(A.P.loss_type, r.depth_row ? A.P.loss_width_depth : A.P.loss_width, s, rho1);
}
// MODELS / LOSSK: the camera models and the loss class of a kernel instance (as ba_fused.hip: the tan / atan / log constants
// stay out of the instances that do not need them).  LOSSK: 0 trivial, 1 Huber / SoftLOne / Tukey / Truncated, 2 every loss.
template <int LOSSK>
__device__ __forceinline__ double pt_loss(int type, double a, double s, double* rho1) {
  if constexpr (LOSSK == 0) { *rho1 = 1.0; return s; }
  else if constexpr (LOSSK == 1) {
    switch (type) {
      case THEIA_LOSS_HUBER: case THEIA_LOSS_SOFTLONE: case THEIA_LOSS_TUKEY: case THEIA_LOSS_TRUNCATED: return loss_eval(type, a, s, rho1);
      default: *rho1 = 1.0; return s;
    }
  } else return loss_eval(type, a, s, rho1);
}
template <int LOSSK>
__device__ __forceinline__ double obs_loss_k(const InnerArgs& A, const ObsRef& r, double s, double* rho1) {
  return pt_loss<LOSSK>(A.P.loss_type, r.depth_row ? A.P.loss_width_depth : A.P.loss_width, s, rho1);
}
inline int inner_loss_class(int lt) { return lt == THEIA_LOSS_TRIVIAL ? 0 : ((lt == THEIA_LOSS_CAUCHY || lt == THEIA_LOSS_ARCTAN) ? 2 : 1); }

// ------------------------------------------------------------------ cameras
// One WORKGROUP per variable camera (round 4: one wave): thread = observation of the camera's list (built at create()), the
// 6 x 6 normal equations by wave trees and the four waves in order (the camera's prior rows included).
// PRIORS: the instance carries the camera-prior rows (their sin / cos / atan2 code and constants: 352 registers against 2xx)
template <unsigned MODELS, int LOSSK, bool PRIORS>
__global__ __launch_bounds__(256) void k_inner_views(InnerArgs A) {
  if (!*A.gate) return;
  constexpr int N = 6, NT = 21, NS = NT + N + 2;
  __shared__ TeamLm<N> S;
  __shared__ double red[4 * NS], tot[NS];
  const int tid = threadIdx.x;
  const int c = blockIdx.x;
  if (c >= A.P.nc || A.P.cam_red[c] < 0) return;
  if (A.own_world > 1 && c % A.own_world != A.own_rank) return;   // another rank sweeps this camera
  const unsigned mask = A.P.cam_mask[c];
  if ((mask & 0x3fu) == 0x3fu) return;
  const int beg = A.cam_obs_off[c], end = A.cam_obs_off[c + 1];
  const int grp = A.P.cam_group[c];
  const int model = A.P.group_model[grp];
  double intr[THEIA_MAX_INTRINSICS];
#pragma unroll
  for (int q = 0; q < THEIA_MAX_INTRINSICS; ++q) intr[q] = A.intr[(size_t)grp * THEIA_MAX_INTRINSICS + q];
  if (tid == 0) team_lm_init<N>(S, A.cam + 6 * (size_t)c, mask & 0x3fu);
  __syncthreads();
  while (true) {
    const int phase = S.phase;   // (uniform: written by thread 0 before the barrier that ended the previous pass)
    if (phase == TL_DONE) break;
    const bool want_jac = phase == TL_LIN;
    double ext[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) ext[q] = want_jac ? S.x[q] : S.xc[q];
    double acc[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) acc[k] = 0.0;
    RotTerms rt;   // one camera, one rotation: the terms once per evaluation, not once per observation
    rotation_terms(ext + 3, rt);
    for (int i = beg + tid; i < end; i += 256) {
      const ObsRef ob = load_obs(A, A.cam_obs_idx[i]);
      const double4 Xv = reinterpret_cast<const double4*>(A.pts)[ob.pt];
      const double X[4] = {Xv.x, Xv.y, Xv.z, Xv.w};
      ObsLin ol;
      const int m = ob.depth_row ? THIP_MODEL_DEPTH_ROW : model;
      if (want_jac) observe_rot<true, false, ObsLin, MODELS>(m, ext, rt, intr, X, ob.uv.x, ob.uv.y, ob.six, ob.siy, ol);
      else observe_rot<false, false, ObsLin, MODELS>(m, ext, rt, intr, X, ob.uv.x, ob.uv.y, ob.six, ob.siy, ol);
      if (!ol.valid) acc[NT + N + 1] += 1.0;
      double rho1;
      acc[NT + N] += 0.5 * obs_loss_k<LOSSK>(A, ob, ol.r[0] * ol.r[0] + ol.r[1] * ol.r[1], &rho1);
      if (!want_jac) continue;
      const double sr = LOSSK == 0 ? 1.0 : sqrt(rho1);
      const double r0 = sr * ol.r[0], r1 = sr * ol.r[1];
      double J[12];
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        const double sc = ((mask >> q) & 1u) ? 0.0 : sr;
        J[q] = ol.Jc[q] * sc; J[6 + q] = ol.Jc[6 + q] * sc;
      }
#pragma unroll
      for (int a = 0; a < 6; ++a) {
#pragma unroll
        for (int b = 0; b <= a; ++b) acc[a * (a + 1) / 2 + b] += J[a] * J[b] + J[6 + a] * J[6 + b];
        acc[NT + a] += J[a] * r0 + J[6 + a] * r1;
      }
    }
    // the camera's prior rows (3 residuals each, no loss): bundle_adjuster.cc:291-313
    if constexpr (PRIORS)
    for (int i = tid; i < A.P.n_priors; i += 256) {
      if (A.P.prior_cam[i] != c) continue;
      double r[3], Jp[18];
      camera_prior(A.P.prior_kind[i], ext, A.P.prior_vec + 3 * (size_t)i, A.P.prior_info + 9 * (size_t)i, want_jac, r, Jp);
      acc[NT + N] += 0.5 * ((r[0] * r[0] + r[1] * r[1]) + r[2] * r[2]);
      if (!want_jac) continue;
#pragma unroll
      for (int row = 0; row < 3; ++row) {
        double J[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) J[q] = ((mask >> q) & 1u) ? 0.0 : Jp[6 * row + q];
#pragma unroll
        for (int a = 0; a < 6; ++a) {
#pragma unroll
          for (int b = 0; b <= a; ++b) acc[a * (a + 1) / 2 + b] += J[a] * J[b];
          acc[NT + a] += J[a] * r[row];
        }
      }
    }
    team_reduce<NS>(acc, red, tot);
    if (tid == 0) team_lm_advance<N>(S, tot);
    __syncthreads();
  }
  if (tid < 6) A.cam[6 * (size_t)c + tid] = S.x[tid];
}

// ------------------------------------------------------------------ shared intrinsics
// grp_wgs workgroups per group (round 4; one workgroup walked all of a group's observations per LM pass: 7.7 ms per sweep
// at C4 with eight groups of 375 000 observations).  Every workgroup of a group runs the SAME state machine on the SAME sums: a
// pass over the observations is dealt to grp_parts parts, part s to workgroup s, each leaves its partial sums in global memory,
// they meet at a per-group arrival counter, and every one of them adds the partials in part order -- identical bits
// everywhere, so the control flow stays in step without any further exchange.  The workgroups of a group must be resident
// together: the launch is cooperative (launch_inner_sweep).  The one-workgroup launch that follows (the whole sweep where
// the cooperative launch is refused, the groups a timed-out launch left undone otherwise) walks the SAME grp_parts parts one
// after the other and adds them in the same order: the same bits as the cooperative launch (the two orders
// differed in the low bits, and a timeout decided which one a run got).
// Round 5: the cameras come as the per-camera blocks k_inner_cam_blocks leaves after the camera sweep (rotation terms once
// per camera, not a sincos per observation and pass); accumulators on registers (every index a constant).
// KC: rows of the solve -- 10 (every parameter has its row) or 4 COMPACT rows (row k = the k-th free parameter) when no
// group frees more than four (DevProblem::intr_rows; the pipelines' FOCAL_LENGTH | RADIAL_DISTORTION frees three): 16 sums per
// observation instead of 67.
template <unsigned MODELS, int LOSSK, int KC>
__global__ __launch_bounds__(256) void k_inner_groups(InnerArgs A) {
  if (!*A.gate) return;
  constexpr int K = THEIA_MAX_INTRINSICS, NT = KC * (KC + 1) / 2, NS = NT + KC + 2;   // 55 + 10 + cost + invalid = 67 (KC = 10)
  static_assert(NS + 1 <= kInnerGroupSums, "partial-sum stride");
  __shared__ TeamLm<KC> S;
  __shared__ double red[4 * NS], tot[NS + 1], ptot[NS], s_kk[K];
  __shared__ int s_gone;
  const int NB = A.grp_wgs > 1 ? A.grp_wgs : 1;          // workgroups of a group in THIS launch
  const int NP = A.grp_parts > 1 ? A.grp_parts : 1;      // parts a pass is dealt to (NB == NP, or NB == 1: one workgroup walks them)
  const int grp = blockIdx.x / NB, sub = blockIdx.x - grp * NB;
  if (grp >= A.P.ng_total || A.P.grp_red[grp] < 0) return;
  if (A.own_world > 1 && grp % A.own_world != A.own_rank) return;
  const unsigned free_mask = A.P.grp_free[grp];
  if (!free_mask) return;
  if (NB == 1 && A.grp_bar && A.grp_bar[A.P.ng_total + 1 + grp]) return;   // the follow-up launch: this group is done
  const int tid = threadIdx.x;
  const int model = A.P.group_model[grp];
  const int beg = A.grp_obs_off[grp], end = A.grp_obs_off[grp + 1];
  // row k of the solve <-> parameter idx[k] (-1: no such row; KC = 10: the identity)
  int idx[KC];
  {
    int nf = 0;
#pragma unroll
    for (int k = 0; k < KC; ++k) idx[k] = -1;
    if constexpr (KC == K) {
#pragma unroll
      for (int q = 0; q < K; ++q) idx[q] = q;
    } else {
#pragma unroll
      for (int q = 0; q < K; ++q)
        if ((free_mask >> q) & 1u) {
#pragma unroll
          for (int k = 0; k < KC; ++k) if (k == nf) idx[k] = q;
          ++nf;
        }
    }
  }
  int pass = 0;   // barriers passed so far: the same number in every workgroup of the group
  if (tid == 0) {
    const double* x0 = A.intr + (size_t)grp * K;
    double xs[KC], base = 0.0;
    unsigned frozen = 0u;
#pragma unroll
    for (int k = 0; k < KC; ++k) { xs[k] = idx[k] >= 0 ? x0[idx[k]] : 0.0; if (idx[k] < 0 || !((free_mask >> idx[k]) & 1u)) frozen |= 1u << k; }
    if (KC != K) for (int q = 0; q < K; ++q) if (!((free_mask >> q) & 1u)) base += x0[q] * x0[q];
    team_lm_init<KC>(S, xs, frozen, base);
  }
  __syncthreads();
  while (true) {
    const int phase = S.phase;
    if (phase == TL_DONE) break;
    const bool want_jac = phase == TL_LIN;
    if (tid < K) {   // the full parameter vector of this pass: the group's intrinsics with the rows of the solve in place
      double v = A.intr[(size_t)grp * K + tid];
#pragma unroll
      for (int k = 0; k < KC; ++k) if (idx[k] == tid) v = want_jac ? S.x[k] : S.xc[k];
      s_kk[tid] = v;
    }
    __syncthreads();
    double kk[K];
#pragma unroll
    for (int q = 0; q < K; ++q) kk[q] = s_kk[q];
    for (int part = (NB > 1 ? sub : 0); part < (NB > 1 ? sub + 1 : NP); ++part) {
      double acc[NS];
#pragma unroll
      for (int k = 0; k < NS; ++k) acc[k] = 0.0;
      for (int i = beg + part * 256 + tid; i < end; i += 256 * NP) {
        const ObsRef ob = load_obs(A, A.grp_obs_idx[i]);
        if (ob.depth_row) continue;   // a depth-prior row does not depend on the intrinsics: constant in this block's problem
        const double4 Xv = reinterpret_cast<const double4*>(A.pts)[ob.pt];
        const double X[4] = {Xv.x, Xv.y, Xv.z, Xv.w};
        double ext[6];
        RotTerms rt;
        camrot_load(A.P.camrot_cand + (size_t)kCamRot * ob.cam, ext, rt);
        ObsLinK ol;
        if (want_jac) observe_rot<false, true, ObsLinK, MODELS>(model, ext, rt, kk, X, ob.uv.x, ob.uv.y, ob.six, ob.siy, ol);   // (residual and Jk: no camera / point blocks here)
        else observe_rot<false, false, ObsLinK, MODELS>(model, ext, rt, kk, X, ob.uv.x, ob.uv.y, ob.six, ob.siy, ol);
        if (!ol.valid) acc[NT + KC + 1] += 1.0;
        double rho1;
        acc[NT + KC] += 0.5 * obs_loss_k<LOSSK>(A, ob, ol.r[0] * ol.r[0] + ol.r[1] * ol.r[1], &rho1);
        if (!want_jac) continue;
        const double sr = LOSSK == 0 ? 1.0 : sqrt(rho1);
        const double r0 = sr * ol.r[0], r1 = sr * ol.r[1];
        double J[2 * KC];
#pragma unroll
        for (int k = 0; k < KC; ++k) {
          double j0 = 0.0, j1 = 0.0;
          if constexpr (KC == K) { if ((free_mask >> k) & 1u) { j0 = ol.Jk[k]; j1 = ol.Jk[K + k]; } }
          else {
#pragma unroll
            for (int q = 0; q < K; ++q) if (idx[k] == q) { j0 = ol.Jk[q]; j1 = ol.Jk[K + q]; }   // (uniform selects: every index a constant)
          }
          J[k] = j0 * sr; J[KC + k] = j1 * sr;
        }
#pragma unroll
        for (int a = 0; a < KC; ++a) {
#pragma unroll
          for (int b = 0; b <= a; ++b) acc[a * (a + 1) / 2 + b] += J[a] * J[b] + J[KC + a] * J[KC + b];
          acc[NT + a] += J[a] * r0 + J[KC + a] * r1;
        }
      }
      team_reduce<NS>(acc, red, ptot);
      if (NB == 1) {   // one workgroup: the parts in order, as the cooperative launch adds its workgroups' partials
        if (tid < NS) tot[tid] = part == 0 ? ptot[tid] : tot[tid] + ptot[tid];
        __syncthreads();
      }
    }
    if (NB > 1) {
      double* part = A.grp_part + ((size_t)grp * 2 + (pass & 1)) * NB * kInnerGroupSums;
      if (tid < NS) part[(size_t)sub * kInnerGroupSums + tid] = ptot[tid];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __syncthreads();
      if (tid == 0) {
        __hip_atomic_fetch_add(A.grp_bar + grp, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        const int want = NB * (pass + 1);
        int* abort_flag = A.grp_bar + A.P.ng_total;
        int polls = 0, gone = 0;
        while (__hip_atomic_load(A.grp_bar + grp, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {
          // a partner that never arrives (its workgroup is not resident: two such launches of different processes can split
          // the CUs between them) must not hang the device: after ~0.1 s every workgroup leaves, nothing is written, and the
          // one-workgroup-per-group launch that always follows redoes the groups that are not marked done
          if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) || ++polls > A.grp_max_polls) {
            __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            gone = 1;
            break;
          }
          __builtin_amdgcn_s_sleep(8);
        }
        s_gone = gone;
      }
      __syncthreads();
      if (s_gone) return;   // (nothing written: the follow-up launch redoes this group)
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      if (tid < NS) {
        double v = 0.0;
        for (int w = 0; w < NB; ++w) { const double u = __builtin_nontemporal_load(part + (size_t)w * kInnerGroupSums + tid); v = w == 0 ? u : v + u; }
        tot[tid] = v;
      }
      ++pass;
      __syncthreads();
    }
    if (tid == 0) team_lm_advance<KC>(S, tot);
    __syncthreads();
  }
  if (sub == 0) {
    if (tid < KC) {
      int q = -1;
#pragma unroll
      for (int k = 0; k < KC; ++k) if (k == tid) q = idx[k];   // (idx[] only ever sees constant indices: registers)
      if (q >= 0) A.intr[(size_t)grp * K + q] = S.x[tid];
    }
    if (tid == 0 && A.grp_bar) A.grp_bar[A.P.ng_total + 1 + grp] = 1;
  }
}

// ------------------------------------------------------------------ points
// Round 5: lane = OBSERVATION.  (Rounds 3-4 ran one thread per point with the generic block_lm<>: its H / g / step arrays
// and the out-of-line accumulate() lived in scratch -- 1128 B per lane, 685 MB written per launch at C4 -- and a wave
// took as long as its longest track times its slowest solve.)  The tracks of a wave tile (the tiles of the main
// kernels: <= 64 observations, never splitting a track) are solved TOGETHER: every lane keeps its own observation's
// camera terms in registers and a copy of its track's LM state; a linearisation is one evaluation per lane and one
// sum over the track's lanes (through LDS slots, as ba_fused.hip), the 3 x 3 / 4 x 4 solve is replicated in the
// track's lanes on identical bits, and the wave iterates until its last track has stopped.  Tracks outside the tiles
// (> 64 observations, a camera seen twice, ...) get one wave each with the lanes striding over the observations.
// Everything is unrolled on registers: no scratch.
//
// The solver rules are block_lm's (trust_region_minimizer.cc) with one saving: Ceres evaluates the Jacobian twice at the
// start (once for the Jacobi scaling, once scaled); here the second is the first rescaled, H_ab s_a s_b and g_a s_a.
template <int PD> constexpr int pt_nsum() { return PD * (PD + 1) / 2 + PD + 2; }   // J'J (packed) | J'r | cost | invalid
template <int PD> constexpr int pt_nsum_padded() { return (pt_nsum<PD>() + 1) & ~1; }

struct PtCam {   // what one observation needs of its camera (k_inner_cam_blocks: rotation terms, intrinsics, model)
  double C[3], R[9], intr[THEIA_MAX_INTRINSICS];
  int model;
};
__device__ __forceinline__ void pt_load_cam(const double* __restrict__ cr, bool depth_row, PtCam& c) {
  double v[16];
  load_d2<16>(cr, v);
#pragma unroll
  for (int i = 0; i < 3; ++i) c.C[i] = v[i];
#pragma unroll
  for (int i = 0; i < 9; ++i) c.R[i] = v[6 + i];
  double kc[12];
  load_d2<12>(cr + kCamRotIntr, kc);
#pragma unroll
  for (int i = 0; i < THEIA_MAX_INTRINSICS; ++i) c.intr[i] = kc[i];
  c.model = depth_row ? THIP_MODEL_DEPTH_ROW : (int)kc[kCamRotModel - kCamRotIntr];
}
// One observation at the point X: out[] += {J'J (packed lower), J'r, cost, invalid} of its loss-corrected, tangent-space,
// UNSCALED point block (WANT_JAC) or {cost, invalid} only.  reprojection_error.h:54-110 as observe_rot(), point side only.
template <int PD, bool WANT_JAC, unsigned MODELS, int LOSSK>
__device__ __forceinline__ void pt_eval(const InnerArgs& A, const PtCam& c, double2 uv, double six, double siy, bool depth_row,
                                        const double (&X)[4], double* __restrict__ out) {
  constexpr int NT = PD * (PD + 1) / 2;
  const double p[3] = {X[0] - X[3] * c.C[0], X[1] - X[3] * c.C[1], X[2] - X[3] * c.C[2]};
  const double sq = p[0] * p[0] + p[1] * p[1] + p[2] * p[2];
  if (sq < 1e-8) {   // the functor returns false: no residual, no Jacobian (cost of a zero residual: 0)
    out[WANT_JAC ? NT + PD + 1 : 1] += 1.0;
    return;
  }
  const double q[3] = {c.R[0] * p[0] + c.R[1] * p[1] + c.R[2] * p[2], c.R[3] * p[0] + c.R[4] * p[1] + c.R[5] * p[2],
                       c.R[6] * p[0] + c.R[7] * p[1] + c.R[8] * p[2]};
  double uvp[2], Jq[6];
  const bool ok = project<WANT_JAC, false, MODELS>(c.model, c.intr, q, uvp, Jq);
  double r[2] = {six * (uvp[0] - uv.x), siy * (uvp[1] - uv.y)};
  double rho1;
  const double rho = pt_loss<LOSSK>(A.P.loss_type, depth_row ? A.P.loss_width_depth : A.P.loss_width, r[0] * r[0] + r[1] * r[1], &rho1);
  if (!WANT_JAC) { out[0] += 0.5 * rho; if (!ok) out[1] += 1.0; return; }
  out[NT + PD] += 0.5 * rho;
  if (!ok) out[NT + PD + 1] += 1.0;
  const double sr = LOSSK == 0 ? 1.0 : sqrt(rho1);
  r[0] *= sr; r[1] *= sr;
  double J[2 * PD];
  double v[4] = {X[0], X[1], X[2], 1.0}, beta = 0.0, nx = 1.0;
  if constexpr (PD == 3) { householder4(X, v, beta); nx = sqrt(X[0] * X[0] + X[1] * X[1] + X[2] * X[2] + X[3] * X[3]); }
  const double s[2] = {six * sr, siy * sr};
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const double* jq = Jq + 3 * a;
    const double A0 = jq[0] * c.R[0] + jq[1] * c.R[3] + jq[2] * c.R[6];
    const double A1 = jq[0] * c.R[1] + jq[1] * c.R[4] + jq[2] * c.R[7];
    const double A2 = jq[0] * c.R[2] + jq[1] * c.R[5] + jq[2] * c.R[8];
    const double j4[4] = {s[a] * A0, s[a] * A1, s[a] * A2, -s[a] * (A0 * c.C[0] + A1 * c.C[1] + A2 * c.C[2])};
    if constexpr (PD == 3) {
      const double jv = j4[0] * v[0] + j4[1] * v[1] + j4[2] * v[2] + j4[3] * v[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) J[3 * a + k] = nx * (j4[k] - beta * v[k] * jv);
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) J[4 * a + k] = j4[k];
    }
  }
  int k = 0;
#pragma unroll
  for (int a = 0; a < PD; ++a) {
#pragma unroll
    for (int b = 0; b <= a; ++b) out[k++] += J[a] * J[b] + J[PD + a] * J[PD + b];
  }
#pragma unroll
  for (int a = 0; a < PD; ++a) out[NT + a] += J[a] * r[0] + J[PD + a] * r[1];
}

// (H + D) y = g for the packed lower H, D = diag d: Cholesky, fully unrolled; false if not positive definite
template <int N>
__device__ __forceinline__ bool pt_chol_solve(const double (&H)[N * (N + 1) / 2], const double (&d)[N], const double (&g)[N], double (&y)[N]) {
  double L[N * (N + 1) / 2];
  bool ok = true;
#pragma unroll
  for (int i = 0; i < N; ++i) {
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      double s = H[i * (i + 1) / 2 + j] + (i == j ? d[i] : 0.0);
#pragma unroll
      for (int k = 0; k < j; ++k) s -= L[i * (i + 1) / 2 + k] * L[j * (j + 1) / 2 + k];
      if (i == j) { if (!(s > 0.0)) ok = false; L[i * (i + 1) / 2 + i] = sqrt(s); }
      else L[i * (i + 1) / 2 + j] = s / L[j * (j + 1) / 2 + j];
    }
  }
  double z[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    double s = g[i];
#pragma unroll
    for (int k = 0; k < i; ++k) s -= L[i * (i + 1) / 2 + k] * z[k];
    z[i] = s / L[i * (i + 1) / 2 + i];
  }
#pragma unroll
  for (int i = N - 1; i >= 0; --i) {
    double s = z[i];
#pragma unroll
    for (int k = i + 1; k < N; ++k) s -= L[k * (k + 1) / 2 + i] * y[k];
    y[i] = s / L[i * (i + 1) / 2 + i];
  }
  return ok;
}

// The per-track LM, replicated in the lanes that hold the track (a tile: the lanes of its segment; a long track: the whole
// wave).  sumJ(X, out[NS]) / sumR(X, out[2]) return the TRACK's totals, identical bits in all of its lanes; `live` = this
// lane's track takes part at all.  The wave leaves when its last track has stopped; x holds the result.
template <int PD, class SumJ, class SumR>
__device__ __forceinline__ void pt_block_lm(double (&x)[4], bool live, SumJ&& sumJ, SumR&& sumR) {
  constexpr int NT = PD * (PD + 1) / 2, NS = pt_nsum<PD>();
  double tot[NS];
  bool done = !live, first = true, relin = true;
  double scale[PD], H[NT], g[PD];
#pragma unroll
  for (int q = 0; q < PD; ++q) scale[q] = 1.0;
  double x_cost = 0.0, gmax = 0.0;
  double radius = 1e4, decrease_factor = 2.0, x_norm = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3]);
  bool step_successful = true;
  int iter = 0, invalid_steps = 0;
  while (true) {
    if (__any(relin)) {   // (ONE copy of the linearisation in the code: the first pass and every accepted step come through here)
      sumJ(x, tot);
      if (first) {
        if (tot[NT + PD + 1] > 0.0 || !isfinite(tot[NT + PD])) done = true;   // failure: the parameters stay as they are
#pragma unroll
        for (int q = 0; q < PD; ++q) scale[q] = 1.0 / (1.0 + sqrt(tot[q * (q + 1) / 2 + q]));
        first = false;
      }
      if (relin) {
#pragma unroll
        for (int a = 0; a < PD; ++a) {
#pragma unroll
          for (int b = 0; b <= a; ++b) H[a * (a + 1) / 2 + b] = tot[a * (a + 1) / 2 + b] * scale[a] * scale[b];
          g[a] = tot[NT + a] * scale[a];
        }
        x_cost = tot[NT + PD];
        gmax = 0.0;
#pragma unroll
        for (int a = 0; a < PD; ++a) gmax = fmax(gmax, fabs(g[a] / scale[a]));
        relin = false;
      }
    }
    if (!done && (iter >= kInnerMaxIterations || (step_successful && gmax <= kInnerGradientTolerance) || radius <= 1e-32)) done = true;
    if (!__any(!done)) break;
    ++iter;
    double d[PD], y[PD];
#pragma unroll
    for (int q = 0; q < PD; ++q) d[q] = fmin(fmax(H[q * (q + 1) / 2 + q], 1e-6), 1e32) / radius;
    const bool pd = pt_chol_solve<PD>(H, d, g, y);
    double yg = 0.0, yHy = 0.0;
#pragma unroll
    for (int a = 0; a < PD; ++a) {
      yg += y[a] * g[a];
      double row = 0.0;
#pragma unroll
      for (int b = 0; b < PD; ++b) row += H[a >= b ? a * (a + 1) / 2 + b : b * (b + 1) / 2 + a] * y[b];
      yHy += y[a] * row;
    }
    const double mcc = yg - 0.5 * yHy;
    double xc[4], stepsq = 0.0, xnormsq = 0.0;
    {
      double step[PD];
#pragma unroll
      for (int q = 0; q < PD; ++q) step[q] = -y[q] * scale[q];
      if constexpr (PD == 3) { const double d3[3] = {step[0], step[1], step[2]}; sphere_plus(x, d3, xc); }
      else {
#pragma unroll
        for (int q = 0; q < 4; ++q) xc[q] = x[q] + step[q];
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) { stepsq += (x[q] - xc[q]) * (x[q] - xc[q]); xnormsq += xc[q] * xc[q]; }
    const bool bad = !(pd && isfinite(mcc) && isfinite(stepsq) && mcc > 0.0);
    if (!done && bad) {
      if (++invalid_steps >= 5) done = true;
      else { radius /= decrease_factor; decrease_factor *= 2.0; step_successful = false; }
    }
    const bool want_cost = !done && !bad;
    if (!want_cost) {   // (a finite stand-in for the lanes whose candidate is not used: no NaN walks through the evaluation)
#pragma unroll
      for (int q = 0; q < 4; ++q) xc[q] = x[q];
    }
    double ct[2];
    sumR(xc, ct);
    if (want_cost) {
      invalid_steps = 0;
      double cand_cost = ct[0];
      if (ct[1] > 0.0 || !isfinite(cand_cost)) cand_cost = DBL_MAX;
      const double cost_change = x_cost - cand_cost;
      if (sqrt(stepsq) <= kInnerParameterTolerance * (x_norm + kInnerParameterTolerance)) done = true;
      else if (fabs(cost_change) <= kInnerFunctionTolerance * x_cost) done = true;
      else {
        const double rho = cost_change / mcc;
        if (rho > 1e-3) {
#pragma unroll
          for (int q = 0; q < 4; ++q) x[q] = xc[q];
          x_norm = sqrt(xnormsq);
          const double t = 2.0 * rho - 1.0;
          radius = fmin(kInnerMaxRadius, radius / fmax(1.0 / 3.0, 1.0 - t * t * t));
          decrease_factor = 2.0; step_successful = true; relin = true;
        } else {
          radius /= decrease_factor; decrease_factor *= 2.0; step_successful = false;
        }
      }
    }
  }
}

// Track totals of N doubles per lane through the lanes' LDS slots (NP = N rounded up to even, 16-B pieces): every lane of a
// track adds the slots of the track's lanes in lane order -- the same operands in the same order in all of them.
template <int N, int NP>
__device__ __forceinline__ void pt_segment_sum(double* __restrict__ wslots, int lane, int seg_start, int seg_len, int seg_maxlen, double (&v)[N]) {
  double* mine = wslots + lane * NP;
#pragma unroll
  for (int k = 0; k < NP; k += 2) *reinterpret_cast<double2*>(mine + k) = make_double2(v[k], k + 1 < N ? v[k + 1] : 0.0);
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] = 0.0;
  for (int j = 0; j < seg_maxlen; ++j) {   // wave-uniform trip count
    const double* oth = wslots + min(seg_start + j, 63) * NP;
    double u[NP];
#pragma unroll
    for (int k = 0; k < NP; k += 2) { const double2 t = *reinterpret_cast<const double2*>(oth + k); u[k] = t.x; u[k + 1] = t.y; }
    if (j < seg_len) {
#pragma unroll
      for (int k = 0; k < N; ++k) v[k] += u[k];
    }
  }
}

__device__ __forceinline__ double pt_wave_sum(double v) {   // every lane gets the same bits
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Tiles of the main kernels: one wave per tile, lane = observation.
template <int PD, unsigned MODELS, int LOSSK>
__global__ __launch_bounds__(256, (MODELS == kModelsNoTrig && LOSSK < 2) ? 2 : 1) void k_inner_tracks(InnerArgs A) {   // (tan / atan / log constants: one wave per SIMD and 512 registers rather than spills)
  if (!*A.gate) return;
  constexpr int NS = pt_nsum<PD>(), NSP = pt_nsum_padded<PD>();
  __shared__ __attribute__((aligned(16))) double s_slots[4][64 * NSP];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int tile = blockIdx.x * 4 + wv;
  if (tile >= A.P.ntiles) return;
  const int cnt = A.P.tile_count[tile], start = A.P.tile_start[tile];
  const bool active = lane < cnt;
  const int o = start + min(lane, max(cnt, 1) - 1);
  const ObsRef ob = load_obs(A, o);
  const bool pconst = A.P.pt_const[ob.pt] != 0;
  PtCam cam;
  pt_load_cam(A.P.camrot_cand + (size_t)kCamRot * ob.cam, ob.depth_row, cam);
  double x[4];
  { const double4 Xv = reinterpret_cast<const double4*>(A.pts)[ob.pt]; x[0] = Xv.x; x[1] = Xv.y; x[2] = Xv.z; x[3] = Xv.w; }
  // the track's lanes: [seg_start, seg_start + seg_len); inactive lanes are tracks of their own
  const int pkey = active ? ob.pt : -1 - lane;
  const int prev = __shfl_up(pkey, 1, 64);
  const bool head = lane == 0 || pkey != prev;
  const unsigned long long Hm = __ballot(head);
  const unsigned long long low = (lane == 63) ? ~0ull : ((1ull << (lane + 1)) - 1ull);
  const int seg_start = 63 - __clzll((long long)(Hm & low));
  const unsigned long long Hn = Hm & ~low;
  const int seg_len = (Hn ? (__ffsll((long long)Hn) - 1) : 64) - seg_start;
  int seg_maxlen = seg_len;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) seg_maxlen = max(seg_maxlen, __shfl_xor(seg_maxlen, off, 64));
  double* wslots = s_slots[wv];
  auto sumJ = [&](const double (&X)[4], double (&out)[NS]) {
#pragma unroll
    for (int k = 0; k < NS; ++k) out[k] = 0.0;
    if (active) pt_eval<PD, true, MODELS, LOSSK>(A, cam, ob.uv, ob.six, ob.siy, ob.depth_row, X, out);
    pt_segment_sum<NS, NSP>(wslots, lane, seg_start, seg_len, seg_maxlen, out);
  };
  auto sumR = [&](const double (&X)[4], double (&out)[2]) {
    out[0] = 0.0; out[1] = 0.0;
    if (active) pt_eval<PD, false, MODELS, LOSSK>(A, cam, ob.uv, ob.six, ob.siy, ob.depth_row, X, out);
    pt_segment_sum<2, 2>(wslots, lane, seg_start, seg_len, seg_maxlen, out);
  };
  pt_block_lm<PD>(x, active && !pconst, sumJ, sumR);
  if (active && head && !pconst) reinterpret_cast<double4*>(A.pts)[ob.pt] = make_double4(x[0], x[1], x[2], x[3]);
}

// Tracks outside the tiles (DevProblem::long_*): one wave per track, the lanes stride over its observations.
template <int PD>
__global__ __launch_bounds__(256) void k_inner_long_tracks(InnerArgs A) {
  if (!*A.gate) return;
  constexpr int NS = pt_nsum<PD>();
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= A.P.long_ntracks) return;
  const int p = A.P.long_track_pt[t];
  if (A.P.pt_const[p]) return;
  const int l0 = A.P.long_track_start[t], l1 = A.P.long_track_start[t + 1];
  double x[4];
  { const double4 Xv = reinterpret_cast<const double4*>(A.pts)[p]; x[0] = Xv.x; x[1] = Xv.y; x[2] = Xv.z; x[3] = Xv.w; }
  auto sumJ = [&](const double (&X)[4], double (&out)[NS]) {
#pragma unroll
    for (int k = 0; k < NS; ++k) out[k] = 0.0;
    for (int i = l0 + lane; i < l1; i += 64) {
      const ObsRef ob = load_obs(A, A.P.long_obs_index[i]);
      PtCam cam;
      pt_load_cam(A.P.camrot_cand + (size_t)kCamRot * ob.cam, ob.depth_row, cam);
      pt_eval<PD, true, kModelsAll, 2>(A, cam, ob.uv, ob.six, ob.siy, ob.depth_row, X, out);
    }
#pragma unroll
    for (int k = 0; k < NS; ++k) out[k] = pt_wave_sum(out[k]);
  };
  auto sumR = [&](const double (&X)[4], double (&out)[2]) {
    out[0] = 0.0; out[1] = 0.0;
    for (int i = l0 + lane; i < l1; i += 64) {
      const ObsRef ob = load_obs(A, A.P.long_obs_index[i]);
      PtCam cam;
      pt_load_cam(A.P.camrot_cand + (size_t)kCamRot * ob.cam, ob.depth_row, cam);
      pt_eval<PD, false, kModelsAll, 2>(A, cam, ob.uv, ob.six, ob.siy, ob.depth_row, X, out);
    }
    out[0] = pt_wave_sum(out[0]); out[1] = pt_wave_sum(out[1]);
  };
  pt_block_lm<PD>(x, true, sumJ, sumR);
  if (lane == 0) reinterpret_cast<double4*>(A.pts)[p] = make_double4(x[0], x[1], x[2], x[3]);
}

// the cameras at the inner-iteration point (after the camera and intrinsics sweeps) as k_cam_prep-style blocks
__global__ __launch_bounds__(256) void k_inner_cam_blocks(InnerArgs A) {
  if (!*A.gate) return;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= A.P.nc) return;
  double* o = A.P.camrot_cand + (size_t)kCamRot * c;
  camrot_store(A.cam + 6 * (size_t)c, o);
  const int g = A.P.cam_group[c];
  for (int q = 0; q < 6; ++q) o[kCamRotScale + q] = 0.0;   // (not read by the track solves)
  for (int q = 0; q < THEIA_MAX_INTRINSICS; ++q) o[kCamRotIntr + q] = A.intr[(size_t)g * THEIA_MAX_INTRINSICS + q];
  o[kCamRotModel] = (double)A.P.group_model[g];
  o[kCamRotRed] = (double)A.P.cam_red[c];
  o[kCamRotGroup] = (double)g; o[39] = 0.0;
}

// |x - x_inner|^2 and |x_inner|^2 over the variable blocks (ParameterToleranceReached measures the step to the point
// the inner iterations ended at): out[0] = step^2, out[1] = |x_inner|^2.  Two fixed-order stages: kInnerCostBlocks
// workgroups over contiguous slices of the blocks (one workgroup over 500k points took 0.44 ms), then one workgroup
// over their partial sums.
__device__ __forceinline__ void inner_norms_body(const InnerArgs& A, const double* __restrict__ cam0, const double* __restrict__ pts0,
                                                 const double* __restrict__ intr0, double* __restrict__ part, int which, int nb, int blk,
                                                 double* s1, double* s2) {
  // which: 0 = every variable block, 1 = the points only (a track shard's share), 2 = cameras + intrinsics only
  double a = 0.0, b = 0.0;
  const int tid = threadIdx.x;
  if (which != 2) {
    const int per = (A.P.np + nb - 1) / nb, p0 = blk * per, p1 = min(A.P.np, p0 + per);
    for (int p = p0 + tid; p < p1; p += 256) {
      if (A.P.pt_const[p]) continue;
      for (int q = 0; q < 4; ++q) { const double u = A.pts[4 * (size_t)p + q], v = pts0[4 * (size_t)p + q]; a += (u - v) * (u - v); b += u * u; }
    }
  }
  if (which != 1) {
    const int per = (A.P.nc + nb - 1) / nb, c0 = blk * per, c1 = min(A.P.nc, c0 + per);
    for (int c = c0 + tid; c < c1; c += 256) {
      if (A.P.cam_red[c] < 0) continue;
      for (int q = 0; q < 6; ++q) { const double u = A.cam[6 * (size_t)c + q], v = cam0[6 * (size_t)c + q]; a += (u - v) * (u - v); b += u * u; }
    }
  }
  if (which != 1 && A.P.ni && blk == 0)
    for (int g = tid; g < A.P.ng_total; g += 256) {
      if (A.P.grp_red[g] < 0) continue;
      for (int q = 0; q < A.P.grp_k[g]; ++q) {
        const double u = A.intr[(size_t)g * THEIA_MAX_INTRINSICS + q], v = intr0[(size_t)g * THEIA_MAX_INTRINSICS + q];
        a += (u - v) * (u - v); b += u * u;
      }
    }
  s1[tid] = a; s2[tid] = b;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) { s1[tid] += s1[tid + s]; s2[tid] += s2[tid + s]; }
    __syncthreads();
  }
  if (tid == 0) { part[2 * blk] = s1[0]; part[2 * blk + 1] = s2[0]; }
}
__global__ __launch_bounds__(256) void k_inner_norms(InnerArgs A, const double* __restrict__ cam0, const double* __restrict__ pts0,
                                                     const double* __restrict__ intr0, double* __restrict__ part, int which) {
  __shared__ double s1[256], s2[256];
  inner_norms_body(A, cam0, pts0, intr0, part, which, gridDim.x, blockIdx.x, s1, s2);
}
__device__ __forceinline__ void inner_norms_reduce_body(int nparts, const double* __restrict__ part, double* __restrict__ out, double* s1, double* s2) {
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 256) { a += part[2 * i]; b += part[2 * i + 1]; }
  s1[threadIdx.x] = a; s2[threadIdx.x] = b;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) { s1[threadIdx.x] += s1[threadIdx.x + s]; s2[threadIdx.x] += s2[threadIdx.x + s]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out[0] = s1[0]; out[1] = s2[0]; }
}

__global__ __launch_bounds__(256) void k_inner_norms_reduce(int nparts, const double* __restrict__ part, double* __restrict__ out) {
  __shared__ double s1[256], s2[256];
  inner_norms_reduce_body(nparts, part, out, s1, s2);
}

// cost at the inner-iteration point: every observation row (tiles or not), then the camera priors; two fixed-order stages
__device__ __forceinline__ void inner_cost_body(const InnerArgs& A, double* __restrict__ part, int nb, int blk, double* s1, double* s2) {
  if (!*A.gate) return;
  double cst = 0.0, inv = 0.0;
  for (int64_t o = (int64_t)blk * 256 + threadIdx.x; o < A.nobs; o += (int64_t)nb * 256) {
    const ObsRef ob = load_obs(A, (int)o);
    const int grp = A.P.cam_group[ob.cam];
    const double4 Xv = reinterpret_cast<const double4*>(A.pts)[ob.pt];
    const double X[4] = {Xv.x, Xv.y, Xv.z, Xv.w};
    ObsLin ol;
    observe<false, false>(ob.depth_row ? THIP_MODEL_DEPTH_ROW : A.P.group_model[grp], A.cam + 6 * (size_t)ob.cam,
                          A.intr + (size_t)grp * THEIA_MAX_INTRINSICS, X, ob.uv.x, ob.uv.y, ob.six, ob.siy, ol);
    if (!ol.valid) inv += 1.0;
    double rho1;
    cst += 0.5 * obs_loss(A, ob, ol.r[0] * ol.r[0] + ol.r[1] * ol.r[1], &rho1);
  }
  s1[threadIdx.x] = cst; s2[threadIdx.x] = inv;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) { s1[threadIdx.x] += s1[threadIdx.x + s]; s2[threadIdx.x] += s2[threadIdx.x + s]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { part[2 * blk] = s1[0]; part[2 * blk + 1] = s2[0]; }
}
__global__ __launch_bounds__(256) void k_inner_cost(InnerArgs A, double* __restrict__ part) {
  __shared__ double s1[256], s2[256];
  inner_cost_body(A, part, gridDim.x, blockIdx.x, s1, s2);
}
__device__ __forceinline__ void inner_cost_reduce_body(const InnerArgs& A, const double* __restrict__ part, int nblocks, double* __restrict__ out,
                                                       double* s1, double* s2) {
  if (!*A.gate) return;
  double cst = 0.0, inv = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += 256) { cst += part[2 * b]; inv += part[2 * b + 1]; }
  for (int i = threadIdx.x; i < A.P.n_priors; i += 256) {
    if (A.P.cam_red[A.P.prior_cam[i]] < 0) continue;   // priors of constant cameras sit in the fixed cost
    double r[3], Jp[18];
    camera_prior(A.P.prior_kind[i], A.cam + 6 * (size_t)A.P.prior_cam[i], A.P.prior_vec + 3 * (size_t)i, A.P.prior_info + 9 * (size_t)i, false, r, Jp);
    cst += 0.5 * ((r[0] * r[0] + r[1] * r[1]) + r[2] * r[2]);
  }
  s1[threadIdx.x] = cst; s2[threadIdx.x] = inv;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) { s1[threadIdx.x] += s1[threadIdx.x + s]; s2[threadIdx.x] += s2[threadIdx.x + s]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out[0] = s1[0]; out[1] = s2[0]; }
}
__global__ __launch_bounds__(256) void k_inner_cost_reduce(InnerArgs A, const double* __restrict__ part, int nblocks, double* __restrict__ out) {
  __shared__ double s1[256], s2[256];
  inner_cost_reduce_body(A, part, nblocks, out, s1, s2);
}
// the unsharded sweep's closing scalars in two launches instead of four: the step norms (workgroups [0, nbn)) and the cost
// (workgroups [nbn, nbn + nbc)) side by side, then their two second stages by two workgroups.  Same partitions, same orders.
__global__ __launch_bounds__(256) void k_inner_norms_cost(InnerArgs A, const double* __restrict__ cam0, const double* __restrict__ pts0,
                                                          const double* __restrict__ intr0, double* __restrict__ part_n, int nbn,
                                                          double* __restrict__ part_c, int nbc) {
  __shared__ double s1[256], s2[256];
  if ((int)blockIdx.x < nbn) inner_norms_body(A, cam0, pts0, intr0, part_n, 0, nbn, blockIdx.x, s1, s2);
  else inner_cost_body(A, part_c, nbc, blockIdx.x - nbn, s1, s2);
}
__global__ __launch_bounds__(256) void k_inner_norms_cost_reduce(InnerArgs A, const double* __restrict__ part_n, int nbn, double* __restrict__ out_n,
                                                                 const double* __restrict__ part_c, int nbc, double* __restrict__ out_c) {
  __shared__ double s1[256], s2[256];
  if (blockIdx.x == 0) inner_norms_reduce_body(nbn, part_n, out_n, s1, s2);
  else inner_cost_reduce_body(A, part_c, nbc, out_c, s1, s2);
}

// sharded solves: this shard's candidate points at their global indices (the rest of the buffer is zero: the ranks' buffers
// are SUMMED into the full point set), and the sweep's scalars put together after the all-reduce
__global__ void k_inner_scatter_points(int np, const double* __restrict__ pts, const int* __restrict__ global_index, double* __restrict__ gpts) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= np) return;
  const double4 v = reinterpret_cast<const double4*>(pts)[p];
  reinterpret_cast<double4*>(gpts)[global_index[p]] = v;
}
__global__ void k_inner_keep_owned(double* __restrict__ x, int nblocks, int stride, int rank, int world) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nblocks * stride) return;
  if ((i / stride) % world != rank) x[i] = 0.0;
}
__global__ void k_inner_combine(const double* __restrict__ reduced4, const double* __restrict__ cam2, double* __restrict__ out4) {
  out4[0] = reduced4[0] + cam2[0]; out4[1] = reduced4[1] + cam2[1]; out4[2] = reduced4[2]; out4[3] = reduced4[3];
}

}  // namespace

// optional observation arrays -> always-valid pointers + flags (see load_obs)
static InnerArgs normalised(const InnerArgs& in) {
  InnerArgs A = in;
  A.has_si = in.P.obs_si != nullptr; A.has_kind = in.P.obs_kind != nullptr;
  if (!A.has_si) A.P.obs_si = in.P.obs_uv;
  if (!A.has_kind) A.P.obs_kind = reinterpret_cast<const uint8_t*>(in.P.obs_cam);
  return A;
}

void launch_inner_cost(const InnerArgs& A0, double* part, double* out2, hipStream_t st) {
  const InnerArgs A = normalised(A0);
  const int nb = (int)std::max<int64_t>(1, std::min<int64_t>(kInnerCostBlocks, (A.nobs + 255) / 256));
  k_inner_cost<<<nb, 256, 0, st>>>(A, part);
  k_inner_cost_reduce<<<1, 256, 0, st>>>(A, part, nb, out2);
}

void launch_inner_sweep(const InnerArgs& A0, hipStream_t st, int stages) {   // stages: 1 cameras, 2 intrinsics groups, 4 points
  const InnerArgs A = normalised(A0);
  static const int skip = [] { const char* e = getenv("THEIA_HIP_INNER_SKIP"); return e ? atoi(e) : 0; }();   // development switch
  const bool lean = (A.P.model_mask & ~kModelsNoTrig) == 0 && inner_loss_class(A.P.loss_type) < 2;   // no FOV / fisheye group, no log / atan loss
  const int lk1 = inner_loss_class(A.P.loss_type) == 0 ? 0 : 1;
  if (A.P.nc > 0 && (stages & 1) && !(skip & 1)) {
    if (A.P.n_priors > 0) {
      if (lean && lk1 == 0) k_inner_views<kModelsNoTrig, 0, true><<<A.P.nc, 256, 0, st>>>(A);
      else if (lean) k_inner_views<kModelsNoTrig, 1, true><<<A.P.nc, 256, 0, st>>>(A);
      else k_inner_views<kModelsAll, 2, true><<<A.P.nc, 256, 0, st>>>(A);
    } else {
      if (lean && lk1 == 0) k_inner_views<kModelsNoTrig, 0, false><<<A.P.nc, 256, 0, st>>>(A);
      else if (lean) k_inner_views<kModelsNoTrig, 1, false><<<A.P.nc, 256, 0, st>>>(A);
      else k_inner_views<kModelsAll, 2, false><<<A.P.nc, 256, 0, st>>>(A);
    }
  }
  if (A.P.ni > 0 && A.P.ng_total > 0 && (stages & 2) && !(skip & 2)) {
    // the cameras as per-camera blocks at the swept extrinsics (rotation terms once per camera; the blocks' intrinsics are
    // not read here: the candidate intrinsics come from the state machine)
    k_inner_cam_blocks<<<(A.P.nc + 255) / 256, 256, 0, st>>>(A);
    // several workgroups per group need all of them resident at once: a cooperative launch guarantees that (or fails);
    // not inside a stream capture, and one workgroup per group whenever it is refused
    bool done = false;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(st, &cap);
    static const bool single = getenv("THEIA_HIP_INNER_GROUPS_SINGLE") != nullptr;
    const int parts = (A.grp_wgs > 1 && A.grp_part && A.grp_bar) ? A.grp_wgs : 1;   // the summation order of BOTH launches
    const bool compact = A.P.intr_rows == 4;   // no group frees more than four parameters (create())
    if (parts > 1 && cap == hipStreamCaptureStatusNone && !single) {
      if (hipMemsetAsync(A.grp_bar, 0, sizeof(int) * (2 * (size_t)A.P.ng_total + 2), st) == hipSuccess) {
        InnerArgs Ac = A;
        Ac.grp_parts = parts;
        // (development / test switch: THEIA_HIP_INNER_GROUPS_MAX_POLLS=0 makes the first waiting workgroup give up at once,
        // which exercises the give-up-and-redo path of the follow-up launch)
        static const int max_polls = [] { const char* e = getenv("THEIA_HIP_INNER_GROUPS_MAX_POLLS"); return e ? atoi(e) : 200000; }();
        Ac.grp_max_polls = max_polls;
        void* args[] = {&Ac};
        const void* fn = compact ? (lean ? (lk1 == 0 ? (const void*)k_inner_groups<kModelsNoTrig, 0, 4> : (const void*)k_inner_groups<kModelsNoTrig, 1, 4>) : (const void*)k_inner_groups<kModelsAll, 2, 4>)
                                 : (lean ? (lk1 == 0 ? (const void*)k_inner_groups<kModelsNoTrig, 0, 10> : (const void*)k_inner_groups<kModelsNoTrig, 1, 10>) : (const void*)k_inner_groups<kModelsAll, 2, 10>);
        const hipError_t e = hipLaunchCooperativeKernel(fn, dim3((unsigned)(A.P.ng_total * A.grp_wgs)), dim3(256), args, 0, st);
        if (e == hipSuccess) done = true; else (void)hipGetLastError();
      }
    }
    // one workgroup per group: the whole sweep when the cooperative launch was not possible, otherwise the safety net that
    // returns at once for every group the cooperative launch marked done (all of them, unless it had to give up).  It walks
    // the same `parts` parts in the same order: the same bits either way.
    {
      InnerArgs A1 = A; A1.grp_wgs = 1; A1.grp_parts = parts; if (!done) A1.grp_bar = nullptr;
#define THIP_IG(KC_) do { \
        if (lean && lk1 == 0) k_inner_groups<kModelsNoTrig, 0, KC_><<<A.P.ng_total, 256, 0, st>>>(A1); \
        else if (lean) k_inner_groups<kModelsNoTrig, 1, KC_><<<A.P.ng_total, 256, 0, st>>>(A1); \
        else k_inner_groups<kModelsAll, 2, KC_><<<A.P.ng_total, 256, 0, st>>>(A1); } while (0)
      if (compact) THIP_IG(4); else THIP_IG(10);
#undef THIP_IG
    }
  }
  if (A.ntracks > 0 && (stages & 4) && !(skip & 4)) {
    // the per-camera blocks at the inner-iteration point (rotation terms, intrinsics, model: one gather per observation)
    k_inner_cam_blocks<<<(A.P.nc + 255) / 256, 256, 0, st>>>(A);
    if (A.P.ntiles > 0) {
      const unsigned nb = (unsigned)(A.P.ntiles + 3) / 4;
      const bool trig = (A.P.model_mask & ~kModelsNoTrig) != 0;   // FOV / fisheye groups present
      const int lk = inner_loss_class(A.P.loss_type);
#define THIP_IT(PD_, M_) do { \
        if (lk == 0) k_inner_tracks<PD_, M_, 0><<<nb, 256, 0, st>>>(A); \
        else if (lk == 1) k_inner_tracks<PD_, M_, 1><<<nb, 256, 0, st>>>(A); \
        else k_inner_tracks<PD_, M_, 2><<<nb, 256, 0, st>>>(A); } while (0)
      if (A.P.pd == 3) { if (trig) THIP_IT(3, kModelsAll); else THIP_IT(3, kModelsNoTrig); }
      else { if (trig) THIP_IT(4, kModelsAll); else THIP_IT(4, kModelsNoTrig); }
#undef THIP_IT
    }
    if (A.P.long_ntracks > 0) {
      if (A.P.pd == 3) k_inner_long_tracks<3><<<(A.P.long_ntracks + 3) / 4, 256, 0, st>>>(A); else k_inner_long_tracks<4><<<(A.P.long_ntracks + 3) / 4, 256, 0, st>>>(A);
    }
  }
}
// part: 4 * kInnerCostBlocks doubles (the norms' partial sums, then the cost's)
void launch_inner_norms_cost(const InnerArgs& A0, const double* cam0, const double* pts0, const double* intr0, double* out4, double* part,
                             hipStream_t st) {
  const InnerArgs A = normalised(A0);
  const int nbc = (int)std::max<int64_t>(1, std::min<int64_t>(kInnerCostBlocks, (A.nobs + 255) / 256));
  double* part_c = part + 2 * (size_t)kInnerCostBlocks;
  k_inner_norms_cost<<<kInnerCostBlocks + nbc, 256, 0, st>>>(A, cam0, pts0, intr0, part, kInnerCostBlocks, part_c, nbc);
  k_inner_norms_cost_reduce<<<2, 256, 0, st>>>(A, part, kInnerCostBlocks, out4, part_c, nbc, out4 + 2);
}
void launch_inner_norms(const InnerArgs& A, const double* cam0, const double* pts0, const double* intr0, double* out2, double* part,
                        hipStream_t st, int which) {
  k_inner_norms<<<kInnerCostBlocks, 256, 0, st>>>(A, cam0, pts0, intr0, part, which);
  k_inner_norms_reduce<<<1, 256, 0, st>>>(kInnerCostBlocks, part, out2);
}

void launch_inner_scatter_points(int np, const double* pts, const int* global_index, double* gpts, hipStream_t st) {
  if (np > 0) k_inner_scatter_points<<<(np + 255) / 256, 256, 0, st>>>(np, pts, global_index, gpts);
}
void launch_inner_keep_owned(double* x, int nblocks, int stride, int rank, int world, hipStream_t st) {
  const int n = nblocks * stride;
  if (n > 0) k_inner_keep_owned<<<(n + 255) / 256, 256, 0, st>>>(x, nblocks, stride, rank, world);
}
void launch_inner_combine(const double* reduced4, const double* cam2, double* out4, hipStream_t st) {
  k_inner_combine<<<1, 1, 0, st>>>(reduced4, cam2, out4);
}

}  // namespace thip
