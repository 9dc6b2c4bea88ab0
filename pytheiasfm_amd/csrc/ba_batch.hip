// ba_batch.hip -- batches of INDEPENDENT small bundle adjustments, one whole
// Levenberg-Marquardt solve per wavefront, no host round trips.
//
// Replaces N calls of BundleAdjustView (bundle_adjustment.cc:220-237: one view's
// 6 extrinsics against constant tracks -- camera localisation in the incremental
// pipeline, and the LO-RANSAC refinement of the absolute-pose estimator,
// estimate_calibrated_absolute_pose.cc:120-153) by one launch.  The pipelines
// call these from a thread pool, thousands of times per reconstruction
// (SURVEY 8f rank 1): on a GPU they are one batch.
//
// Same trust-region rules as ba_solver.hip (Ceres 2.2 TrustRegionMinimizer +
// LevenbergMarquardtStrategy, Jacobi scaling, loss corrector), restated for a
// 6 x 6 system whose model-cost change is  y'g - y'Hy/2.  Lane = observation
// (stride 64); the 6 x 6 normal equations are wave-reduced in a fixed order.
#include "ba_device.h"
#include "theia_hip_internal.h"

#include <chrono>
#include <cmath>
#include <cstring>
#include <vector>

#define HIP_TRY(expr)                                                                             \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess) return thip::set_error(THEIA_HIP_ERR_INTERNAL, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

namespace thip {
namespace {

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__device__ __forceinline__ double wsum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

struct ViewBatch {
  int num;
  const int64_t* offsets;
  const int* counts;      // optional: problem p = [offsets[p], offsets[p] + counts[p]) (device-built batches)
  const double2* uv;
  const double2* si;      // or nullptr
  const double4* X;
  double* cam;            // [num][6] in/out
  const double* intr;     // [num][10]
  const int* model;
  const uint8_t* mask;    // [num] frozen extrinsics columns (bit q)
  int loss_type;
  double loss_width;
  int max_iterations;
  double function_tolerance, gradient_tolerance, parameter_tolerance, max_radius;
};

struct ViewOut {
  int success, term, iters, nsucc;
  double initial_cost, final_cost;
};

// residual-only cost of the camera `ext` over the wave's observation range
__device__ double view_cost(const ViewBatch& B, int p, const double* ext, int lane, double* invalid) {
  const int64_t beg = B.offsets[p], end = B.counts ? beg + B.counts[p] : B.offsets[p + 1];
  const int model = B.model[p];
  const double* intr = B.intr + (size_t)p * THEIA_MAX_INTRINSICS;
  double cost = 0.0, inv = 0.0;
  for (int64_t o = beg + lane; o < end; o += 64) {
    const double2 uv = B.uv[o];
    double six = 1.0, siy = 1.0;
    if (B.si) { const double2 s = B.si[o]; six = s.x; siy = s.y; }
    const double4 Xv = B.X[o];
    const double X[4] = {Xv.x, Xv.y, Xv.z, Xv.w};
    ObsLin ol;
    observe<false, false>(model, ext, intr, X, uv.x, uv.y, six, siy, ol);
    if (!ol.valid) inv += 1.0;
    double rho1;
    cost += 0.5 * loss_eval(B.loss_type, B.loss_width, ol.r[0] * ol.r[0] + ol.r[1] * ol.r[1], &rho1);
  }
  *invalid = wsum(inv);
  return wsum(cost);
}

// linearise at `ext`: H (packed lower 21) = J'J, g = J'r, cost; J scaled by `scale`
__device__ void view_linearize(const ViewBatch& B, int p, const double* ext, const double* scale, unsigned mask,
                               int lane, double* H, double* g, double* cost, double* invalid) {
  const int64_t beg = B.offsets[p], end = B.counts ? beg + B.counts[p] : B.offsets[p + 1];
  const int model = B.model[p];
  const double* intr = B.intr + (size_t)p * THEIA_MAX_INTRINSICS;
  double acc[28];
#pragma unroll
  for (int k = 0; k < 28; ++k) acc[k] = 0.0;
  double inv = 0.0;
  for (int64_t o = beg + lane; o < end; o += 64) {
    const double2 uv = B.uv[o];
    double six = 1.0, siy = 1.0;
    if (B.si) { const double2 s = B.si[o]; six = s.x; siy = s.y; }
    const double4 Xv = B.X[o];
    const double X[4] = {Xv.x, Xv.y, Xv.z, Xv.w};
    ObsLin ol;
    observe<true, false>(model, ext, intr, X, uv.x, uv.y, six, siy, ol);
    if (!ol.valid) inv += 1.0;
    double rho1;
    const double rho = loss_eval(B.loss_type, B.loss_width, ol.r[0] * ol.r[0] + ol.r[1] * ol.r[1], &rho1);
    const double sr = sqrt(rho1);
    const double r0 = sr * ol.r[0], r1 = sr * ol.r[1];
    double J[12];
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const double sc = ((mask >> q) & 1u) ? 0.0 : sr * scale[q];
      J[q] = ol.Jc[q] * sc; J[6 + q] = ol.Jc[6 + q] * sc;
    }
    int k = 0;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
#pragma unroll
      for (int b = 0; b <= a; ++b) acc[k++] += J[a] * J[b] + J[6 + a] * J[6 + b];
      acc[21 + a] += J[a] * r0 + J[6 + a] * r1;
    }
    acc[27] += 0.5 * rho;
  }
#pragma unroll
  for (int k = 0; k < 28; ++k) acc[k] = wsum(acc[k]);
#pragma unroll
  for (int k = 0; k < 21; ++k) H[k] = acc[k];
#pragma unroll
  for (int k = 0; k < 6; ++k) g[k] = acc[21 + k];
  *cost = acc[27];
  *invalid = wsum(inv);
}

__device__ __forceinline__ int tri(int a, int b) { return a * (a + 1) / 2 + b; }

// (H + diag(d)) y = g by Cholesky; false if not positive definite
__device__ bool solve6(const double* H, const double* d, const double* g, double* y) {
  double L[21];
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = H[tri(i, j)] + (i == j ? d[i] : 0.0);
      for (int k = 0; k < j; ++k) s -= L[tri(i, k)] * L[tri(j, k)];
      if (i == j) { if (!(s > 0.0)) return false; L[tri(i, i)] = sqrt(s); }
      else L[tri(i, j)] = s / L[tri(j, j)];
    }
  double z[6];
  for (int i = 0; i < 6; ++i) {
    double s = g[i];
    for (int k = 0; k < i; ++k) s -= L[tri(i, k)] * z[k];
    z[i] = s / L[tri(i, i)];
  }
  for (int i = 5; i >= 0; --i) {
    double s = z[i];
    for (int k = i + 1; k < 6; ++k) s -= L[tri(k, i)] * y[k];
    y[i] = s / L[tri(i, i)];
  }
  return true;
}

__global__ __launch_bounds__(256) void k_view_lm(ViewBatch B, ViewOut* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= B.num) return;
  const unsigned mask = B.mask ? B.mask[p] : 0u;
  double x[6];
#pragma unroll
  for (int q = 0; q < 6; ++q) x[q] = B.cam[(size_t)p * 6 + q];
  ViewOut R;
  R.success = 0; R.term = THEIA_TERM_NO_CONVERGENCE; R.iters = 0; R.nsucc = 0; R.initial_cost = 0.0; R.final_cost = 0.0;
  double scale[6] = {1.0, 1.0, 1.0, 1.0, 1.0, 1.0};
  double H[21], g[6], x_cost, invalid;
  // Jacobi scaling from the column norms at the initial point (once per solve)
  view_linearize(B, p, x, scale, mask, lane, H, g, &x_cost, &invalid);
#pragma unroll
  for (int q = 0; q < 6; ++q) scale[q] = 1.0 / (1.0 + sqrt(H[tri(q, q)]));
  const bool all_const = (mask & 0x3fu) == 0x3fu;
  double radius = 1e4, decrease_factor = 2.0;
  bool step_successful = true, need_linearize = true;
  int iter = 0, invalid_steps = 0, term = THEIA_TERM_NO_CONVERGENCE;
  double x_norm = 0.0, minimum_cost = 0.0, gmax = 0.0;
#pragma unroll
  for (int q = 0; q < 6; ++q) x_norm += x[q] * x[q];
  x_norm = sqrt(x_norm);
  bool first = true;
  while (true) {
    if (need_linearize) {
      view_linearize(B, p, x, scale, mask, lane, H, g, &x_cost, &invalid);
      gmax = 0.0;
#pragma unroll
      for (int q = 0; q < 6; ++q) gmax = fmax(gmax, fabs(g[q] / scale[q]));
      need_linearize = false;
    }
    if (first) {
      first = false;
      R.initial_cost = x_cost;
      minimum_cost = x_cost;
      if (invalid > 0.0 || !isfinite(x_cost)) { term = THEIA_TERM_FAILURE; R.final_cost = x_cost; break; }
      if (all_const) { term = THEIA_TERM_CONVERGENCE; break; }
    }
    if (iter >= B.max_iterations) { term = THEIA_TERM_NO_CONVERGENCE; break; }
    if (step_successful && gmax <= B.gradient_tolerance) { term = THEIA_TERM_CONVERGENCE; break; }
    if (radius <= 1e-32) { term = THEIA_TERM_CONVERGENCE; break; }
    ++iter;
    double d[6], y[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) d[q] = fmin(fmax(H[tri(q, q)], 1e-6), 1e32) / radius;
    const bool pd = solve6(H, d, g, y);
    // model cost change of the step -y:  y'g - y'Hy/2  (H without the LM diagonal)
    double yg = 0.0, yHy = 0.0;
    for (int a = 0; a < 6; ++a) {
      yg += y[a] * g[a];
      double row = 0.0;
      for (int b = 0; b < 6; ++b) row += H[a >= b ? tri(a, b) : tri(b, a)] * y[b];
      yHy += y[a] * row;
    }
    const double mcc = yg - 0.5 * yHy;
    double cand[6], stepsq = 0.0, xnormsq = 0.0;
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      cand[q] = ((mask >> q) & 1u) ? x[q] : x[q] - y[q] * scale[q];
      stepsq += (x[q] - cand[q]) * (x[q] - cand[q]);
      xnormsq += cand[q] * cand[q];
    }
    const bool step_valid = pd && isfinite(mcc) && isfinite(stepsq) && mcc > 0.0;
    if (!step_valid) {
      if (++invalid_steps >= 5) { term = THEIA_TERM_FAILURE; break; }
      radius /= decrease_factor; decrease_factor *= 2.0; step_successful = false;
      continue;
    }
    invalid_steps = 0;
    double cinv;
    double cand_cost = view_cost(B, p, cand, lane, &cinv);
    if (cinv > 0.0 || !isfinite(cand_cost)) cand_cost = DBL_MAX;
    const double step_norm = sqrt(stepsq);
    if (step_norm <= B.parameter_tolerance * (x_norm + B.parameter_tolerance)) { term = THEIA_TERM_CONVERGENCE; break; }
    const double cost_change = x_cost - cand_cost;
    if (fabs(cost_change) <= B.function_tolerance * x_cost) { term = THEIA_TERM_CONVERGENCE; break; }
    const double rho = cost_change / mcc;
    if (rho > 1e-3) {
#pragma unroll
      for (int q = 0; q < 6; ++q) x[q] = cand[q];
      x_norm = sqrt(xnormsq);
      const double t = 2.0 * rho - 1.0;
      radius = radius / fmax(1.0 / 3.0, 1.0 - t * t * t);
      radius = fmin(B.max_radius, radius);
      decrease_factor = 2.0; step_successful = true; need_linearize = true;
      R.nsucc++;
      if (cand_cost < minimum_cost) minimum_cost = cand_cost;
    } else {
      radius /= decrease_factor; decrease_factor *= 2.0; step_successful = false;
    }
  }
  R.iters = iter; R.term = term; R.success = term != THEIA_TERM_FAILURE;
  if (term != THEIA_TERM_FAILURE) R.final_cost = minimum_cost;
  if (lane == 0) {
    out[p] = R;
#pragma unroll
    for (int q = 0; q < 6; ++q) B.cam[(size_t)p * 6 + q] = x[q];
  }
}

// ------------------------------------------------------------------ tracks
// N independent BundleAdjustTrack problems (bundle_adjustment.cc:262-285,389-418 with one track each;
// estimate_track.cc:289 refines every triangulated track this way from a thread pool): the point is
// the only variable block, all cameras are constant.  One THREAD per track runs the whole LM.
struct TrackBatch {
  int num;                       // points
  const int64_t* offsets;        // [num+1] into the observation arrays (sorted by point)
  const double2* uv;
  const double2* si;             // or nullptr
  const int* obs_cam;
  const double* cam;             // [nc][6]
  const double* intr;            // [ng][10]
  const int* group_model;
  const int* cam_group;
  const uint8_t* pt_const;       // or nullptr
  double* pts;                   // [num][4] in/out
  int loss_type;
  double loss_width;
  int max_iterations;
  double function_tolerance, gradient_tolerance, parameter_tolerance, max_radius;
};

template <int PD>
__device__ void track_linearize(const TrackBatch& B, int p, const double* X, const double* scale, double* H, double* g,
                                double* cost, bool* invalid, bool want_jac) {
  constexpr int NT = PD * (PD + 1) / 2;
  for (int k = 0; k < NT; ++k) H[k] = 0.0;
  for (int k = 0; k < PD; ++k) g[k] = 0.0;
  double c = 0.0;
  bool inv = false;
  for (int64_t o = B.offsets[p]; o < B.offsets[p + 1]; ++o) {
    const int cidx = B.obs_cam[o];
    const int grp = B.cam_group[cidx];
    const double2 uv = B.uv[o];
    double six = 1.0, siy = 1.0;
    if (B.si) { const double2 s = B.si[o]; six = s.x; siy = s.y; }
    ObsLin ol;
    if (want_jac) observe<true, false>(B.group_model[grp], B.cam + 6 * (size_t)cidx, B.intr + (size_t)grp * THEIA_MAX_INTRINSICS, X, uv.x, uv.y, six, siy, ol);
    else observe<false, false>(B.group_model[grp], B.cam + 6 * (size_t)cidx, B.intr + (size_t)grp * THEIA_MAX_INTRINSICS, X, uv.x, uv.y, six, siy, ol);
    if (!ol.valid) inv = true;
    double rho1;
    const double rho = loss_eval(B.loss_type, B.loss_width, ol.r[0] * ol.r[0] + ol.r[1] * ol.r[1], &rho1);
    c += 0.5 * rho;
    if (!want_jac) continue;
    const double sr = sqrt(rho1);
    const double r0 = sr * ol.r[0], r1 = sr * ol.r[1];
    double J[2 * PD];
    if (PD == 3) {
      double Jt[6];
      to_tangent(X, ol.Jx, Jt);
      for (int q = 0; q < 3; ++q) { J[q] = Jt[q] * sr * scale[q]; J[PD + q] = Jt[3 + q] * sr * scale[q]; }
    } else {
      for (int q = 0; q < PD; ++q) { J[q] = ol.Jx[q] * sr * scale[q]; J[PD + q] = ol.Jx[4 + q] * sr * scale[q]; }
    }
    int k = 0;
    for (int a = 0; a < PD; ++a) {
      for (int b = 0; b <= a; ++b) H[k++] += J[a] * J[b] + J[PD + a] * J[PD + b];
      g[a] += J[a] * r0 + J[PD + a] * r1;
    }
  }
  *cost = c; *invalid = inv;
}

template <int PD>
__device__ bool solve_small(const double* H, const double* d, const double* g, double* y) {
  constexpr int NT = PD * (PD + 1) / 2;
  double L[NT];
  for (int i = 0; i < PD; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = H[tri(i, j)] + (i == j ? d[i] : 0.0);
      for (int k = 0; k < j; ++k) s -= L[tri(i, k)] * L[tri(j, k)];
      if (i == j) { if (!(s > 0.0)) return false; L[tri(i, i)] = sqrt(s); }
      else L[tri(i, j)] = s / L[tri(j, j)];
    }
  double z[PD];
  for (int i = 0; i < PD; ++i) {
    double s = g[i];
    for (int k = 0; k < i; ++k) s -= L[tri(i, k)] * z[k];
    z[i] = s / L[tri(i, i)];
  }
  for (int i = PD - 1; i >= 0; --i) {
    double s = z[i];
    for (int k = i + 1; k < PD; ++k) s -= L[tri(k, i)] * y[k];
    y[i] = s / L[tri(i, i)];
  }
  return true;
}

template <int PD>
__global__ __launch_bounds__(64) void k_track_lm(TrackBatch B, ViewOut* __restrict__ out) {
  constexpr int NT = PD * (PD + 1) / 2;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= B.num) return;
  ViewOut R;
  R.success = 1; R.term = THEIA_TERM_CONVERGENCE; R.iters = 0; R.nsucc = 0; R.initial_cost = 0.0; R.final_cost = 0.0;
  double X[4];
  for (int q = 0; q < 4; ++q) X[q] = B.pts[4 * (size_t)p + q];
  double scale[PD], H[NT], g[PD], x_cost;
  for (int q = 0; q < PD; ++q) scale[q] = 1.0;
  bool invalid;
  const bool is_const = (B.pt_const && B.pt_const[p]) || B.offsets[p + 1] == B.offsets[p];
  track_linearize<PD>(B, p, X, scale, H, g, &x_cost, &invalid, true);
  if (is_const) {   // nothing to optimise: report the cost of its residual blocks
    R.initial_cost = R.final_cost = x_cost;
    out[p] = R;
    return;
  }
  for (int q = 0; q < PD; ++q) scale[q] = 1.0 / (1.0 + sqrt(H[tri(q, q)]));
  double radius = 1e4, decrease_factor = 2.0;
  bool step_successful = true, need_linearize = true, first = true;
  int iter = 0, invalid_steps = 0, term = THEIA_TERM_NO_CONVERGENCE;
  double x_norm = sqrt(X[0] * X[0] + X[1] * X[1] + X[2] * X[2] + X[3] * X[3]);
  double minimum_cost = 0.0, gmax = 0.0;
  while (true) {
    if (need_linearize) {
      track_linearize<PD>(B, p, X, scale, H, g, &x_cost, &invalid, true);
      gmax = 0.0;
      for (int q = 0; q < PD; ++q) gmax = fmax(gmax, fabs(g[q] / scale[q]));
      need_linearize = false;
    }
    if (first) {
      first = false;
      R.initial_cost = x_cost; minimum_cost = x_cost;
      if (invalid || !isfinite(x_cost)) { term = THEIA_TERM_FAILURE; R.final_cost = x_cost; break; }
    }
    if (iter >= B.max_iterations) { term = THEIA_TERM_NO_CONVERGENCE; break; }
    if (step_successful && gmax <= B.gradient_tolerance) { term = THEIA_TERM_CONVERGENCE; break; }
    if (radius <= 1e-32) { term = THEIA_TERM_CONVERGENCE; break; }
    ++iter;
    double d[PD], y[PD];
    for (int q = 0; q < PD; ++q) d[q] = fmin(fmax(H[tri(q, q)], 1e-6), 1e32) / radius;
    const bool pd = solve_small<PD>(H, d, g, y);
    double yg = 0.0, yHy = 0.0;
    for (int a = 0; a < PD; ++a) {
      yg += y[a] * g[a];
      double row = 0.0;
      for (int b = 0; b < PD; ++b) row += H[a >= b ? tri(a, b) : tri(b, a)] * y[b];
      yHy += y[a] * row;
    }
    const double mcc = yg - 0.5 * yHy;
    double Xp[4] = {X[0], X[1], X[2], X[3]};
    if (PD == 3) {
      const double d3[3] = {-y[0] * scale[0], -y[1] * scale[1], -y[2] * scale[2]};
      sphere_plus(X, d3, Xp);
    } else {
      for (int q = 0; q < PD; ++q) Xp[q] = X[q] - y[q] * scale[q];
    }
    double stepsq = 0.0, xnormsq = 0.0;
    for (int q = 0; q < 4; ++q) { stepsq += (X[q] - Xp[q]) * (X[q] - Xp[q]); xnormsq += Xp[q] * Xp[q]; }
    const bool step_valid = pd && isfinite(mcc) && isfinite(stepsq) && mcc > 0.0;
    if (!step_valid) {
      if (++invalid_steps >= 5) { term = THEIA_TERM_FAILURE; break; }
      radius /= decrease_factor; decrease_factor *= 2.0; step_successful = false;
      continue;
    }
    invalid_steps = 0;
    double cand_cost, Hd[NT], gd[PD];
    bool cinv;
    track_linearize<PD>(B, p, Xp, scale, Hd, gd, &cand_cost, &cinv, false);
    if (cinv || !isfinite(cand_cost)) cand_cost = DBL_MAX;
    const double step_norm = sqrt(stepsq);
    if (step_norm <= B.parameter_tolerance * (x_norm + B.parameter_tolerance)) { term = THEIA_TERM_CONVERGENCE; break; }
    const double cost_change = x_cost - cand_cost;
    if (fabs(cost_change) <= B.function_tolerance * x_cost) { term = THEIA_TERM_CONVERGENCE; break; }
    const double rho = cost_change / mcc;
    if (rho > 1e-3) {
      for (int q = 0; q < 4; ++q) X[q] = Xp[q];
      x_norm = sqrt(xnormsq);
      const double t = 2.0 * rho - 1.0;
      radius = fmin(B.max_radius, radius / fmax(1.0 / 3.0, 1.0 - t * t * t));
      decrease_factor = 2.0; step_successful = true; need_linearize = true;
      R.nsucc++;
      if (cand_cost < minimum_cost) minimum_cost = cand_cost;
    } else {
      radius /= decrease_factor; decrease_factor *= 2.0; step_successful = false;
    }
  }
  R.iters = iter; R.term = term; R.success = term != THEIA_TERM_FAILURE;
  if (term != THEIA_TERM_FAILURE) R.final_cost = minimum_cost;
  out[p] = R;
  for (int q = 0; q < 4; ++q) B.pts[4 * (size_t)p + q] = X[q];
}

// ---- per-track reprojection statistics: the sweep of SetOutlierTracksToUnestimated
// (set_outlier_tracks_to_unestimated.cc:64-139: mean squared reprojection error over the track's views, any view
// with negative depth, and SufficientTriangulationAngle, triangulation.cc:236-250).  One thread per track.
__global__ __launch_bounds__(64) void k_track_stats(TrackBatch B, double* __restrict__ mean_sq_err, int* __restrict__ behind,
                                                    double* __restrict__ min_ray_cos) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= B.num) return;
  double X[4];
  for (int q = 0; q < 4; ++q) X[q] = B.pts[4 * (size_t)p + q];
  const double Xh[3] = {X[0] / X[3], X[1] / X[3], X[2] / X[3]};   // track->Point().hnormalized()
  double sum = 0.0, mincos = 2.0;
  int nb = 0;
  const int64_t beg = B.offsets[p], end = B.offsets[p + 1];
  for (int64_t o = beg; o < end; ++o) {
    const int c = B.obs_cam[o];
    const double* ext = B.cam + 6 * (size_t)c;
    const int grp = B.cam_group[c];
    // depth = (R (X - w C))_z / w  (Camera::ProjectPoint, camera.cc:206-216)
    RotTerms rt;
    rotation_terms(ext + 3, rt);
    const double px = X[0] - X[3] * ext[0], py = X[1] - X[3] * ext[1], pz = X[2] - X[3] * ext[2];
    double qz;
    if (rt.small) qz = pz + (ext[3] * py - ext[4] * px);                     // p + w x p (first order)
    else qz = rt.R[6] * px + rt.R[7] * py + rt.R[8] * pz;
    if (qz / X[3] < 0.0) nb++;
    const double2 uv = B.uv[o];
    ObsLin ol;
    observe<false, false>(B.group_model[grp], ext, B.intr + (size_t)grp * THEIA_MAX_INTRINSICS, X, uv.x, uv.y, 1.0, 1.0, ol);
    sum += ol.r[0] * ol.r[0] + ol.r[1] * ol.r[1];
    // ray_i . ray_j over the earlier views
    double ri[3] = {Xh[0] - ext[0], Xh[1] - ext[1], Xh[2] - ext[2]};
    const double ni = sqrt(ri[0] * ri[0] + ri[1] * ri[1] + ri[2] * ri[2]);
    ri[0] /= ni; ri[1] /= ni; ri[2] /= ni;
    for (int64_t o2 = beg; o2 < o; ++o2) {
      const double* e2 = B.cam + 6 * (size_t)B.obs_cam[o2];
      double rj[3] = {Xh[0] - e2[0], Xh[1] - e2[1], Xh[2] - e2[2]};
      const double nj = sqrt(rj[0] * rj[0] + rj[1] * rj[1] + rj[2] * rj[2]);
      const double d = (ri[0] * (rj[0] / nj) + ri[1] * (rj[1] / nj)) + ri[2] * (rj[2] / nj);
      mincos = fmin(mincos, d);
    }
  }
  mean_sq_err[p] = sum / (double)(end - beg);
  behind[p] = nb;
  min_ray_cos[p] = mincos;
}

// ---- N-view triangulation of the other two TriangulationMethodType values (estimate_track.cc:239-257) ----
// Camera::GetProjectionMatrix (camera.cc:195-200): K [R | -R c], K = [f s cx; 0 f a cy; 0 0 1] from the model's own
// focal length / aspect ratio / skew / principal point slots (FOV and division-undistortion have no skew slot).
__device__ inline void track_projection_matrix(int model, const double* k, const double* ext, double* P) {
  RotTerms rt;
  rotation_terms(ext + 3, rt);
  const bool noskew = model == THEIA_CAM_FOV || model == THEIA_CAM_DIVISION_UNDISTORTION;
  const double f = k[0], fa = k[0] * k[1], sk = noskew ? 0.0 : k[2];
  const double cx = noskew ? k[2] : k[3], cy = noskew ? k[3] : k[4];
  double Rt[12];
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) Rt[4 * r + c] = rt.R[3 * r + c];
    Rt[4 * r + 3] = -((rt.R[3 * r] * ext[0] + rt.R[3 * r + 1] * ext[1]) + rt.R[3 * r + 2] * ext[2]);
  }
  for (int c = 0; c < 4; ++c) {
    P[c] = (f * Rt[c] + sk * Rt[4 + c]) + cx * Rt[8 + c];
    P[4 + c] = fa * Rt[4 + c] + cy * Rt[8 + c];
    P[8 + c] = Rt[8 + c];
  }
}

// eigenvector of the smallest eigenvalue of a symmetric 4 x 4 matrix (cyclic Jacobi; A row-major, destroyed)
__device__ inline void smallest_eigenvector4(double* A, double* x) {
  double V[16];
  for (int i = 0; i < 16; ++i) V[i] = (i % 5 == 0) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 40; ++sweep) {
    double off = 0.0, diag = 0.0;
    for (int i = 0; i < 4; ++i) { diag += A[5 * i] * A[5 * i]; for (int j = i + 1; j < 4; ++j) off += A[4 * i + j] * A[4 * i + j]; }
    if (!(off > 1e-34 * diag)) break;
    for (int pI = 0; pI < 3; ++pI)
      for (int q = pI + 1; q < 4; ++q) {
        const double apq = A[4 * pI + q];
        if (apq == 0.0) continue;
        const double theta = (A[5 * q] - A[5 * pI]) / (2.0 * apq);
        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
        for (int k = 0; k < 4; ++k) {   // A <- A J
          const double akp = A[4 * k + pI], akq = A[4 * k + q];
          A[4 * k + pI] = c * akp - sn * akq; A[4 * k + q] = sn * akp + c * akq;
        }
        for (int k = 0; k < 4; ++k) {   // A <- J^T A
          const double apk = A[4 * pI + k], aqk = A[4 * q + k];
          A[4 * pI + k] = c * apk - sn * aqk; A[4 * q + k] = sn * apk + c * aqk;
        }
        for (int k = 0; k < 4; ++k) {
          const double vkp = V[4 * k + pI], vkq = V[4 * k + q];
          V[4 * k + pI] = c * vkp - sn * vkq; V[4 * k + q] = sn * vkp + c * vkq;
        }
      }
  }
  int best = 0;
  for (int i = 1; i < 4; ++i) if (A[5 * i] < A[5 * best]) best = i;
  for (int k = 0; k < 4; ++k) x[k] = V[4 * k + best];
}

// TriangulateNView (triangulation.cc:197-214, L2_MINIMIZATION): the eigenvector of the smallest eigenvalue of
// sum C^T C, C = (I - n n^T) P with n the normalised homogeneous pixel.  TriangulateNViewSVD (:178-194, SVD): the first
// four entries of the right singular vector of the smallest singular value of the 3N x (4 + N) matrix [-P_i | e_i x_i]:
// eliminating the N scale unknowns from its normal equations leaves, for the eigenvalue mu,
//     [ sum C^T C - mu (I + sum b_i b_i^T / (d_i (d_i - mu))) ] X = 0,   b_i = P_i^T x_i, d_i = x_i^T x_i, lambda_i = b_i.X / (d_i - mu)
// -- the L2 matrix at mu = 0 -- solved by fixed-point steps on mu (the Rayleigh quotient of the full vector) until mu stops
// moving (relative 1e-13, at most 24 steps; well-conditioned tracks take 2 - 3).  A track on which the iteration does not
// settle (two nearly equal smallest eigenvalues) keeps the L2 solution of the first step.
// The result carries the reference's normalisation (unit norm over all 4 + N entries).  The sign is arbitrary, as Eigen's is.
__device__ inline void triangulate_nview(const TrackBatch& B, int64_t beg, int64_t end, bool svd, double* X) {
  double mu = 0.0;
  double x[4] = {0.0, 0.0, 0.0, 1.0}, x_l2[4] = {0.0, 0.0, 0.0, 1.0};
  const int rounds = svd ? 24 : 1;
  bool settled = !svd;
  for (int round = 0; round < rounds; ++round) {
    double D[16];
    for (int i = 0; i < 16; ++i) D[i] = 0.0;
    for (int64_t i = beg; i < end; ++i) {
      const int c = B.obs_cam[i], grp = B.cam_group[c];
      double P[12];
      track_projection_matrix(B.group_model[grp], B.intr + (size_t)grp * THEIA_MAX_INTRINSICS, B.cam + 6 * (size_t)c, P);
      const double u = B.uv[i].x, v = B.uv[i].y;
      const double d = (u * u + v * v) + 1.0, nrm = sqrt(d);
      const double n[3] = {u / nrm, v / nrm, 1.0 / nrm};
      double t[4], C[12];
      for (int k = 0; k < 4; ++k) t[k] = (n[0] * P[k] + n[1] * P[4 + k]) + n[2] * P[8 + k];
      for (int r = 0; r < 3; ++r) for (int k = 0; k < 4; ++k) C[4 * r + k] = P[4 * r + k] - n[r] * t[k];
      for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 4; ++b) D[4 * a + b] += (C[a] * C[b] + C[4 + a] * C[4 + b]) + C[8 + a] * C[8 + b];
      if (mu != 0.0) {   // - mu b b^T / (d (d - mu)) with b = P^T x = nrm * t, i.e. - mu / (d - mu) t t^T
        const double w = mu / (d - mu);
        for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) D[4 * a + b] -= w * (t[a] * t[b]);
      }
    }
    if (mu != 0.0) for (int a = 0; a < 4; ++a) D[5 * a] -= mu;
    smallest_eigenvector4(D, x);
    if (!svd) break;
    if (round == 0) for (int k = 0; k < 4; ++k) x_l2[k] = x[k];
    // Rayleigh quotient of the full vector (X, lambda): v^T M v = sum |P X - lambda x|^2
    double num = 0.0, den = (x[0] * x[0] + x[1] * x[1]) + (x[2] * x[2] + x[3] * x[3]);
    for (int64_t i = beg; i < end; ++i) {
      const int c = B.obs_cam[i], grp = B.cam_group[c];
      double P[12];
      track_projection_matrix(B.group_model[grp], B.intr + (size_t)grp * THEIA_MAX_INTRINSICS, B.cam + 6 * (size_t)c, P);
      const double xi[3] = {B.uv[i].x, B.uv[i].y, 1.0};
      const double d = (xi[0] * xi[0] + xi[1] * xi[1]) + 1.0;
      double px[3];
      for (int r = 0; r < 3; ++r) px[r] = ((P[4 * r] * x[0] + P[4 * r + 1] * x[1]) + P[4 * r + 2] * x[2]) + P[4 * r + 3] * x[3];
      const double lam = ((xi[0] * px[0] + xi[1] * px[1]) + xi[2] * px[2]) / (d - mu);
      for (int r = 0; r < 3; ++r) { const double e = px[r] - lam * xi[r]; num += e * e; }
      den += lam * lam;
    }
    const double mu_new = num / den;
    settled = fabs(mu_new - mu) <= 1e-13 * mu_new || mu_new == mu;
    mu = mu_new;
    if (settled || round == rounds - 1) { const double sc = 1.0 / sqrt(den); for (int k = 0; k < 4; ++k) x[k] *= sc; break; }
  }
  for (int k = 0; k < 4; ++k) X[k] = settled ? x[k] : x_l2[k];
}

// Stage 1 + 2 of TrackEstimator::EstimateTrack, one thread per track: the triangulation-angle test on the
// supplied viewing rays and TriangulateMidpoint.  status: 0 = triangulated, 1 = bad angle, 2 = failed
// triangulation, 3 = skipped (constant point).
__global__ __launch_bounds__(64) void k_track_triangulate(TrackBatch B, const double* __restrict__ rays, double cos_min_angle,
                                                          int method /* 0 MIDPOINT, 1 SVD, 2 L2_MINIMIZATION */, int* __restrict__ status) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= B.num) return;
  if (B.pt_const && B.pt_const[p]) { status[p] = 3; return; }
  const int64_t beg = B.offsets[p], end = B.offsets[p + 1];
  bool ok_angle = false;
  for (int64_t i = beg; i < end && !ok_angle; ++i)
    for (int64_t j = i + 1; j < end; ++j) {
      const double d = (rays[3 * i] * rays[3 * j] + rays[3 * i + 1] * rays[3 * j + 1]) + rays[3 * i + 2] * rays[3 * j + 2];
      if (d < cos_min_angle) { ok_angle = true; break; }
    }
  if (end - beg < 2 || !ok_angle) { status[p] = 1; return; }
  if (method == 1 || method == 2) {   // TriangulateNViewSVD / TriangulateNView: always "true" (triangulation.cc:193,213)
    triangulate_nview(B, beg, end, method == 1, B.pts + 4 * (size_t)p);
    status[p] = 0;
    return;
  }
  // A = sum (I - d d^T), b = sum (I - d d^T) o  (the 4th row / column of the reference's 4 x 4 system is n X_w = n)
  double A[6] = {0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0};   // A lower: 00 10 11 20 21 22
  for (int64_t i = beg; i < end; ++i) {
    const double* d = rays + 3 * i;
    const double* o = B.cam + 6 * (size_t)B.obs_cam[i];
    const double t00 = 1.0 - d[0] * d[0], t10 = -(d[1] * d[0]), t11 = 1.0 - d[1] * d[1];
    const double t20 = -(d[2] * d[0]), t21 = -(d[2] * d[1]), t22 = 1.0 - d[2] * d[2];
    A[0] += t00; A[1] += t10; A[2] += t11; A[3] += t20; A[4] += t21; A[5] += t22;
    b[0] += (t00 * o[0] + t10 * o[1]) + t20 * o[2];
    b[1] += (t10 * o[0] + t11 * o[1]) + t21 * o[2];
    b[2] += (t20 * o[0] + t21 * o[1]) + t22 * o[2];
  }
  // Eigen::LLT: a non-positive pivot is a NumericalIssue
  if (!(A[0] > 0.0)) { status[p] = 2; return; }
  const double l00 = sqrt(A[0]), l10 = A[1] / l00, l20 = A[3] / l00;
  const double d11 = A[2] - l10 * l10;
  if (!(d11 > 0.0)) { status[p] = 2; return; }
  const double l11 = sqrt(d11), l21 = (A[4] - l20 * l10) / l11;
  const double d22 = A[5] - (l20 * l20 + l21 * l21);
  if (!(d22 > 0.0)) { status[p] = 2; return; }
  const double l22 = sqrt(d22);
  const double y0 = b[0] / l00, y1 = (b[1] - l10 * y0) / l11, y2 = (b[2] - (l20 * y0 + l21 * y1)) / l22;
  const double x2 = y2 / l22, x1 = (y1 - l21 * x2) / l11, x0 = (y0 - (l10 * x1 + l20 * x2)) / l00;
  double* X = B.pts + 4 * (size_t)p;
  X[0] = x0; X[1] = x1; X[2] = x2; X[3] = 1.0;
  status[p] = 0;
}

template <typename T>
struct Dev {
  T* p = nullptr;
  ~Dev() { if (p) (void)hipFree(p); }
  int alloc(size_t n) {
    if (hipMalloc((void**)&p, std::max<size_t>(1, n) * sizeof(T)) != hipSuccess)
      return set_error(THEIA_HIP_ERR_OUT_OF_MEMORY, "hipMalloc(%zu) failed", n * sizeof(T));
    return 0;
  }
  int up(const void* src, size_t n) {
    int rc = alloc(n);
    if (rc) return rc;
    if (n && hipMemcpy(p, src, n * sizeof(T), hipMemcpyHostToDevice) != hipSuccess)
      return set_error(THEIA_HIP_ERR_INTERNAL, "hipMemcpy H2D failed");
    return 0;
  }
};

}  // namespace

// device-resident variant for callers inside the library (LO-RANSAC)
int views_batch_device(int num, const int64_t* d_offsets, const int* d_counts, const double* d_uv, const double* d_si, const double* d_X,
                       double* d_cam, const double* d_intr, const int* d_model, const uint8_t* d_mask,
                       const theia_ba_options* o, void* d_out /* ViewOut[num] */, hipStream_t st) {
  ViewBatch B;
  B.num = num; B.offsets = d_offsets; B.counts = d_counts; B.uv = reinterpret_cast<const double2*>(d_uv);
  B.si = reinterpret_cast<const double2*>(d_si); B.X = reinterpret_cast<const double4*>(d_X);
  B.cam = d_cam; B.intr = d_intr; B.model = d_model; B.mask = d_mask;
  B.loss_type = o->loss_function_type; B.loss_width = o->robust_loss_width; B.max_iterations = o->max_num_iterations;
  B.function_tolerance = o->function_tolerance; B.gradient_tolerance = o->gradient_tolerance;
  B.parameter_tolerance = o->parameter_tolerance; B.max_radius = o->max_trust_region_radius;
  k_view_lm<<<(num + 3) / 4, 256, 0, st>>>(B, static_cast<ViewOut*>(d_out));
  return 0;
}
size_t views_batch_out_bytes() { return sizeof(ViewOut); }
void views_batch_unpack(const void* host_out, int i, int* success, int* term, int* iters, int* nsucc, double* c0, double* c1) {
  const ViewOut& r = static_cast<const ViewOut*>(host_out)[i];
  *success = r.success; *term = r.term; *iters = r.iters; *nsucc = r.nsucc; *c0 = r.initial_cost; *c1 = r.final_cost;
}

}  // namespace thip

using namespace thip;

extern "C" int theia_hip_ba_views_batch(const theia_ba_view_batch* b, const theia_ba_options* o, theia_ba_summary* summaries) {
  if (!b || !o) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "null batch/options");
  const int num = b->num_problems;
  if (num < 0) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "negative num_problems");
  if (num == 0) return 0;
  if (!b->offsets || !b->cam_ext || !b->intrinsics || !b->model || !summaries)
    return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "null array in batch");
  if (b->offsets[0] != 0) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "offsets[0] must be 0");
  for (int i = 0; i < num; ++i) {
    if (b->offsets[i + 1] < b->offsets[i]) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "offsets must be non-decreasing");
    if (b->model[i] < THEIA_CAM_PINHOLE || b->model[i] > THEIA_CAM_ORTHOGRAPHIC)
      return set_error(THEIA_HIP_ERR_UNSUPPORTED, "camera model %d of problem %d has no HIP kernel", b->model[i], i);
  }
  const int64_t total = b->offsets[num];
  if (total > 0 && (!b->obs_uv || !b->points)) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "null observation arrays");
  if (o->loss_function_type < 0 || o->loss_function_type > THEIA_LOSS_TRUNCATED)
    return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "unknown loss function type");
  if (o->max_num_iterations < 0) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "negative max_num_iterations");
  int rc = thip::ensure_device();
  if (rc) return rc;
  std::vector<uint8_t> mask(num);
  for (int i = 0; i < num; ++i) {
    unsigned m = b->cam_const ? b->cam_const[i] : 0u;
    if (o->constant_camera_orientation) m |= THEIA_CAM_CONST_ORIENTATION;
    if (o->constant_camera_position) m |= THEIA_CAM_CONST_POSITION;
    if (o->orthographic_camera) m |= THEIA_CAM_CONST_TZ;
    unsigned cols = 0;
    if (m & THEIA_CAM_CONST_POSITION) cols |= 0x07;
    if (m & THEIA_CAM_CONST_ORIENTATION) cols |= 0x38;
    if (m & THEIA_CAM_CONST_TZ) cols |= 0x04;
    mask[i] = (uint8_t)cols;
  }
  Dev<int64_t> d_off; Dev<double> d_uv, d_si, d_X, d_cam, d_intr; Dev<int> d_model; Dev<uint8_t> d_mask; Dev<char> d_out;
  if ((rc = d_off.up(b->offsets, num + 1)) || (rc = d_uv.up(b->obs_uv, 2 * total)) || (rc = d_X.up(b->points, 4 * total)) ||
      (rc = d_cam.up(b->cam_ext, 6 * (size_t)num)) || (rc = d_intr.up(b->intrinsics, THEIA_MAX_INTRINSICS * (size_t)num)) ||
      (rc = d_model.up(b->model, num)) || (rc = d_mask.up(mask.data(), num)) || (rc = d_out.alloc(views_batch_out_bytes() * num)))
    return rc;
  if (b->obs_sqrt_info && (rc = d_si.up(b->obs_sqrt_info, 2 * total))) return rc;
  const double t0 = now_s();
  views_batch_device(num, d_off.p, nullptr, d_uv.p, b->obs_sqrt_info ? d_si.p : nullptr, d_X.p, d_cam.p, d_intr.p, d_model.p, d_mask.p, o,
                     d_out.p, nullptr);
  std::vector<char> h_out(views_batch_out_bytes() * num);
  HIP_TRY(hipMemcpy(h_out.data(), d_out.p, h_out.size(), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(b->cam_ext, d_cam.p, sizeof(double) * 6 * num, hipMemcpyDeviceToHost));
  const double dt = now_s() - t0;
  for (int i = 0; i < num; ++i) {
    theia_ba_summary& S = summaries[i];
    S.trace_size = 0;
    views_batch_unpack(h_out.data(), i, &S.success, &S.termination_type, &S.num_iterations, &S.num_successful_steps,
                       &S.initial_cost, &S.final_cost);
    S.setup_time_in_seconds = 0.0; S.solve_time_in_seconds = dt / num;
    S.time_linearize = S.time_solve_reduced = S.time_backsub = S.time_kernel_linearize = 0.0;
    S.num_linearize_launches = 0;
  }
  return 0;
}

// Shared host part of the per-point batches: validation + observations grouped by point.
namespace {
struct PointGrouped {
  std::vector<int64_t> off;
  std::vector<double> uv, si;
  std::vector<int> oc;
};
int group_by_point(const theia_ba_problem* p, PointGrouped* G) {
  const int np = p->num_points;
  if (np < 0 || p->num_cameras < 0 || p->num_groups < 0 || p->num_obs < 0) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "negative sizes");
  if (np > 0 && !p->points) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "null points");
  if (p->num_obs > 0 && (!p->cam_ext || !p->intrinsics || !p->group_model || !p->cam_group || !p->obs_uv || !p->obs_cam || !p->obs_pt))
    return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "null array in problem");
  if (p->num_obs >= ((int64_t)1 << 31)) return set_error(THEIA_HIP_ERR_UNSUPPORTED, "num_obs >= 2^31");
  if (p->obs_kind)
    for (int64_t i = 0; i < p->num_obs; ++i)
      if (p->obs_kind[i] != THEIA_OBS_REPROJECTION)
        return set_error(THEIA_HIP_ERR_UNSUPPORTED, "depth-prior rows are not built for the per-track / per-view batch entry points");
  for (int c = 0; c < p->num_cameras; ++c)
    if (p->cam_group[c] < 0 || p->cam_group[c] >= p->num_groups) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "cam_group[%d] out of range", c);
  for (int g = 0; g < p->num_groups; ++g)
    if (p->group_model[g] < THEIA_CAM_PINHOLE || p->group_model[g] > THEIA_CAM_ORTHOGRAPHIC)
      return set_error(THEIA_HIP_ERR_UNSUPPORTED, "camera model %d of group %d has no HIP kernel", p->group_model[g], g);
  const int64_t nobs = p->num_obs;
  for (int64_t i = 0; i < nobs; ++i)
    if (p->obs_cam[i] < 0 || p->obs_cam[i] >= p->num_cameras || p->obs_pt[i] < 0 || p->obs_pt[i] >= np)
      return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "observation %lld indexes out of range", (long long)i);
  G->off.assign(np + 1, 0);
  for (int64_t i = 0; i < nobs; ++i) G->off[p->obs_pt[i] + 1]++;
  for (int q = 0; q < np; ++q) G->off[q + 1] += G->off[q];
  G->uv.resize(2 * (size_t)nobs); G->oc.resize((size_t)nobs);
  if (p->obs_sqrt_info) G->si.resize(2 * (size_t)nobs);
  std::vector<int64_t> fill(G->off.begin(), G->off.end() - 1);
  for (int64_t i = 0; i < nobs; ++i) {
    const int64_t s = fill[p->obs_pt[i]]++;
    G->uv[2 * s] = p->obs_uv[2 * i]; G->uv[2 * s + 1] = p->obs_uv[2 * i + 1];
    if (p->obs_sqrt_info) { G->si[2 * s] = p->obs_sqrt_info[2 * i]; G->si[2 * s + 1] = p->obs_sqrt_info[2 * i + 1]; }
    G->oc[s] = p->obs_cam[i];
  }
  return 0;
}
}  // namespace

extern "C" int theia_hip_ba_tracks_batch(const theia_ba_problem* p, const theia_ba_options* o, theia_ba_summary* summaries) {
  if (!p || !o) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "null problem/options");
  const int np = p->num_points;
  if (np == 0) return 0;
  if (!summaries) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "null summaries");
  if (o->loss_function_type < 0 || o->loss_function_type > THEIA_LOSS_TRUNCATED)
    return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "unknown loss function type");
  // observations grouped by point (stable: the reference walks a track's views in its own order; sums differ by rounding only)
  PointGrouped G;
  int rc = group_by_point(p, &G);
  if (rc) return rc;
  if ((rc = thip::ensure_device())) return rc;
  const std::vector<int64_t>& off = G.off;
  const std::vector<double>& uv = G.uv;
  const std::vector<double>& si = G.si;
  const std::vector<int>& oc = G.oc;
  Dev<int64_t> d_off; Dev<double> d_uv, d_si, d_cam, d_intr, d_pts; Dev<int> d_oc, d_gm, d_cg; Dev<uint8_t> d_pc; Dev<char> d_out;
  if ((rc = d_off.up(off.data(), np + 1)) || (rc = d_uv.up(uv.data(), uv.size())) || (rc = d_oc.up(oc.data(), oc.size())) ||
      (rc = d_cam.up(p->cam_ext, 6 * (size_t)p->num_cameras)) || (rc = d_intr.up(p->intrinsics, THEIA_MAX_INTRINSICS * (size_t)p->num_groups)) ||
      (rc = d_gm.up(p->group_model, p->num_groups)) || (rc = d_cg.up(p->cam_group, p->num_cameras)) ||
      (rc = d_pts.up(p->points, 4 * (size_t)np)) || (rc = d_out.alloc(sizeof(ViewOut) * (size_t)np)))
    return rc;
  if (p->obs_sqrt_info && (rc = d_si.up(si.data(), si.size()))) return rc;
  if (p->point_const && (rc = d_pc.up(p->point_const, np))) return rc;
  TrackBatch B;
  B.num = np; B.offsets = d_off.p; B.uv = reinterpret_cast<const double2*>(d_uv.p);
  B.si = p->obs_sqrt_info ? reinterpret_cast<const double2*>(d_si.p) : nullptr;
  B.obs_cam = d_oc.p; B.cam = d_cam.p; B.intr = d_intr.p; B.group_model = d_gm.p; B.cam_group = d_cg.p;
  B.pt_const = p->point_const ? d_pc.p : nullptr; B.pts = d_pts.p;
  B.loss_type = o->loss_function_type; B.loss_width = o->robust_loss_width; B.max_iterations = o->max_num_iterations;
  B.function_tolerance = o->function_tolerance; B.gradient_tolerance = o->gradient_tolerance;
  B.parameter_tolerance = o->parameter_tolerance; B.max_radius = o->max_trust_region_radius;
  const double t0 = now_s();
  if (o->use_homogeneous_point_parametrization) k_track_lm<3><<<(np + 63) / 64, 64>>>(B, reinterpret_cast<ViewOut*>(d_out.p));
  else k_track_lm<4><<<(np + 63) / 64, 64>>>(B, reinterpret_cast<ViewOut*>(d_out.p));
  std::vector<char> h_out(sizeof(ViewOut) * (size_t)np);
  HIP_TRY(hipMemcpy(h_out.data(), d_out.p, h_out.size(), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(p->points, d_pts.p, sizeof(double) * 4 * (size_t)np, hipMemcpyDeviceToHost));
  const double dt = now_s() - t0;
  for (int i = 0; i < np; ++i) {
    theia_ba_summary& S = summaries[i];
    S.trace_size = 0;
    views_batch_unpack(h_out.data(), i, &S.success, &S.termination_type, &S.num_iterations, &S.num_successful_steps,
                       &S.initial_cost, &S.final_cost);
    S.setup_time_in_seconds = 0.0; S.solve_time_in_seconds = dt / np;
    S.time_linearize = S.time_solve_reduced = S.time_backsub = S.time_kernel_linearize = 0.0;
    S.num_linearize_launches = 0;
  }
  return 0;
}

extern "C" int theia_hip_track_statistics(const theia_ba_problem* p, double* mean_sq_reprojection_error, int32_t* num_behind_camera,
                                          double* min_ray_cosine) {
  if (!p) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "null problem");
  const int np = p->num_points;
  if (np == 0) return 0;
  if (!mean_sq_reprojection_error || !num_behind_camera || !min_ray_cosine) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "null output");
  PointGrouped G;
  int rc = group_by_point(p, &G);
  if (rc) return rc;
  if ((rc = thip::ensure_device())) return rc;
  Dev<int64_t> d_off; Dev<double> d_uv, d_cam, d_intr, d_pts, d_err, d_cos; Dev<int> d_oc, d_gm, d_cg, d_nb;
  if ((rc = d_off.up(G.off.data(), np + 1)) || (rc = d_uv.up(G.uv.data(), G.uv.size())) || (rc = d_oc.up(G.oc.data(), G.oc.size())) ||
      (rc = d_cam.up(p->cam_ext, 6 * (size_t)p->num_cameras)) || (rc = d_intr.up(p->intrinsics, THEIA_MAX_INTRINSICS * (size_t)p->num_groups)) ||
      (rc = d_gm.up(p->group_model, p->num_groups)) || (rc = d_cg.up(p->cam_group, p->num_cameras)) ||
      (rc = d_pts.up(p->points, 4 * (size_t)np)) || (rc = d_err.alloc(np)) || (rc = d_cos.alloc(np)) || (rc = d_nb.alloc(np)))
    return rc;
  TrackBatch B;
  std::memset(&B, 0, sizeof(B));
  B.num = np; B.offsets = d_off.p; B.uv = reinterpret_cast<const double2*>(d_uv.p); B.si = nullptr;
  B.obs_cam = d_oc.p; B.cam = d_cam.p; B.intr = d_intr.p; B.group_model = d_gm.p; B.cam_group = d_cg.p; B.pts = d_pts.p;
  k_track_stats<<<(np + 63) / 64, 64>>>(B, d_err.p, d_nb.p, d_cos.p);
  HIP_TRY(hipMemcpy(mean_sq_reprojection_error, d_err.p, sizeof(double) * np, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(num_behind_camera, d_nb.p, sizeof(int) * np, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(min_ray_cosine, d_cos.p, sizeof(double) * np, hipMemcpyDeviceToHost));
  return 0;
}


extern "C" int theia_hip_estimate_tracks(const theia_ba_problem* p, const double* obs_ray_dir, const theia_ba_options* o,
                                         const theia_track_estimate_options* eo, uint8_t* estimated, int32_t counters[4]) {
  if (!p || !o || !eo) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "null problem/options");
  const int np = p->num_points;
  if (counters) for (int k = 0; k < 4; ++k) counters[k] = 0;
  if (np == 0) return 0;
  if (!estimated || !counters) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "null output");
  if (p->num_obs > 0 && !obs_ray_dir) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "null viewing rays");
  if (eo->triangulation_method < 0 || eo->triangulation_method > 2)
    return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "unknown triangulation method (0 MIDPOINT, 1 SVD, 2 L2_MINIMIZATION)");
  if (o->loss_function_type < 0 || o->loss_function_type > THEIA_LOSS_TRUNCATED)
    return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "unknown loss function type");
  PointGrouped G;
  int rc = group_by_point(p, &G);
  if (rc) return rc;
  if ((rc = thip::ensure_device())) return rc;
  // the viewing rays in the grouped order
  const int64_t nobs = p->num_obs;
  std::vector<double> rays(3 * (size_t)nobs);
  {
    std::vector<int64_t> fill(G.off.begin(), G.off.end() - 1);
    for (int64_t i = 0; i < nobs; ++i) {
      const int64_t s = fill[p->obs_pt[i]]++;
      for (int k = 0; k < 3; ++k) rays[3 * s + k] = obs_ray_dir[3 * i + k];
    }
  }
  Dev<int64_t> d_off; Dev<double> d_uv, d_si, d_cam, d_intr, d_pts, d_rays, d_err, d_cos; Dev<int> d_oc, d_gm, d_cg, d_status, d_nb;
  Dev<uint8_t> d_pc; Dev<char> d_out;
  if ((rc = d_off.up(G.off.data(), np + 1)) || (rc = d_uv.up(G.uv.data(), G.uv.size())) || (rc = d_oc.up(G.oc.data(), G.oc.size())) ||
      (rc = d_cam.up(p->cam_ext, 6 * (size_t)p->num_cameras)) || (rc = d_intr.up(p->intrinsics, THEIA_MAX_INTRINSICS * (size_t)p->num_groups)) ||
      (rc = d_gm.up(p->group_model, p->num_groups)) || (rc = d_cg.up(p->cam_group, p->num_cameras)) ||
      (rc = d_pts.up(p->points, 4 * (size_t)np)) || (rc = d_rays.up(rays.data(), rays.size())) || (rc = d_status.alloc(np)) ||
      (rc = d_out.alloc(sizeof(ViewOut) * (size_t)np)) || (rc = d_err.alloc(np)) || (rc = d_cos.alloc(np)) || (rc = d_nb.alloc(np)))
    return rc;
  if (p->obs_sqrt_info && (rc = d_si.up(G.si.data(), G.si.size()))) return rc;
  if (p->point_const && (rc = d_pc.up(p->point_const, np))) return rc;
  TrackBatch B;
  std::memset(&B, 0, sizeof(B));
  B.num = np; B.offsets = d_off.p; B.uv = reinterpret_cast<const double2*>(d_uv.p);
  B.si = p->obs_sqrt_info ? reinterpret_cast<const double2*>(d_si.p) : nullptr;
  B.obs_cam = d_oc.p; B.cam = d_cam.p; B.intr = d_intr.p; B.group_model = d_gm.p; B.cam_group = d_cg.p;
  B.pt_const = p->point_const ? d_pc.p : nullptr; B.pts = d_pts.p;
  B.loss_type = o->loss_function_type; B.loss_width = o->robust_loss_width; B.max_iterations = o->max_num_iterations;
  B.function_tolerance = o->function_tolerance; B.gradient_tolerance = o->gradient_tolerance;
  B.parameter_tolerance = o->parameter_tolerance; B.max_radius = o->max_trust_region_radius;
  const int grid = (np + 63) / 64;
  const double kPi = 3.14159265358979323846;
  k_track_triangulate<<<grid, 64>>>(B, d_rays.p, cos(eo->min_triangulation_angle_degrees * kPi / 180.0), eo->triangulation_method, d_status.p);
  std::vector<int> status(np);
  HIP_TRY(hipMemcpy(status.data(), d_status.p, sizeof(int) * np, hipMemcpyDeviceToHost));
  // the track BA and the reprojection sweep run on the triangulated tracks only: everything else is "constant"
  std::vector<uint8_t> skip(np);
  for (int i = 0; i < np; ++i) skip[i] = status[i] != 0;
  Dev<uint8_t> d_skip;
  if ((rc = d_skip.up(skip.data(), np))) return rc;
  B.pt_const = d_skip.p;
  std::vector<char> h_out;
  if (eo->bundle_adjustment) {
    if (o->use_homogeneous_point_parametrization) k_track_lm<3><<<grid, 64>>>(B, reinterpret_cast<ViewOut*>(d_out.p));
    else k_track_lm<4><<<grid, 64>>>(B, reinterpret_cast<ViewOut*>(d_out.p));
    h_out.resize(sizeof(ViewOut) * (size_t)np);
    HIP_TRY(hipMemcpy(h_out.data(), d_out.p, h_out.size(), hipMemcpyDeviceToHost));
  }
  k_track_stats<<<grid, 64>>>(B, d_err.p, d_nb.p, d_cos.p);
  std::vector<double> err(np);
  std::vector<int> nb(np);
  HIP_TRY(hipMemcpy(err.data(), d_err.p, sizeof(double) * np, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(nb.data(), d_nb.p, sizeof(int) * np, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(p->points, d_pts.p, sizeof(double) * 4 * (size_t)np, hipMemcpyDeviceToHost));
  const double sq_max = eo->max_acceptable_reprojection_error_pixels * eo->max_acceptable_reprojection_error_pixels;
  for (int i = 0; i < np; ++i) {
    estimated[i] = 0;
    if (status[i] == 3) continue;
    if (status[i] == 1) { counters[0]++; continue; }
    if (status[i] == 2) { counters[1]++; continue; }
    if (eo->bundle_adjustment) {
      int success = 0, term = 0, nit = 0, nsucc = 0;
      double c0 = 0.0, c1 = 0.0;
      views_batch_unpack(h_out.data(), i, &success, &term, &nit, &nsucc, &c0, &c1);
      if (!success) { counters[3]++; continue; }
    }
    if (nb[i] > 0 || !(err[i] < sq_max)) { counters[2]++; continue; }
    estimated[i] = 1;
  }
  return 0;
}
