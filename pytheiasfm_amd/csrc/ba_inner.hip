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

template <int N>
__device__ bool chol_solve_small(const double* H, const double* d, const double* g, double* y) {
  constexpr int NT = N * (N + 1) / 2;
  double L[NT];
  for (int i = 0; i < N; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = H[itri(i, j)] + (i == j ? d[i] : 0.0);
      for (int k = 0; k < j; ++k) s -= L[itri(i, k)] * L[itri(j, k)];
      if (i == j) { if (!(s > 0.0)) return false; L[itri(i, i)] = sqrt(s); }
      else L[itri(i, j)] = s / L[itri(j, j)];
    }
  double z[N];
  for (int i = 0; i < N; ++i) {
    double s = g[i];
    for (int k = 0; k < i; ++k) s -= L[itri(i, k)] * z[k];
    z[i] = s / L[itri(i, i)];
  }
  for (int i = N - 1; i >= 0; --i) {
    double s = z[i];
    for (int k = i + 1; k < N; ++k) s -= L[itri(k, i)] * y[k];
    y[i] = s / L[itri(i, i)];
  }
  return true;
}

// One block's LM solve (trust_region_minimizer.cc with LevenbergMarquardtStrategy, the rules of ba_solver.hip's
// lm_control_body).  Every thread of the team runs this loop with the same values; only the functors cooperate:
//   lin(x, scale, H, g, &cost, &invalid)   J'J (packed lower), J'r of the scaled Jacobian, cost, invalid functor
//   cost(xc, &invalid)                      cost only
//   plus(x, step /* tangent, already scaled */, xc)   x (+) step in the ambient space (NA doubles)
template <int N, int NA, class Lin, class Cost, class Plus>
__device__ void block_lm(double* x, Lin&& lin, Cost&& cost, Plus&& plus) {
  constexpr int NT = N * (N + 1) / 2;
  double scale[N], H[NT], g[N], x_cost = 0.0;
  bool invalid = false;
  for (int q = 0; q < N; ++q) scale[q] = 1.0;
  lin(x, scale, H, g, &x_cost, &invalid);
  if (invalid || !isfinite(x_cost)) return;   // "if the optimization is a failure ... it won't change the parameters"
  for (int q = 0; q < N; ++q) scale[q] = 1.0 / (1.0 + sqrt(H[itri(q, q)]));
  double radius = 1e4, decrease_factor = 2.0;
  bool step_successful = true, need_linearize = true;
  int iter = 0, invalid_steps = 0;
  double x_norm = 0.0, gmax = 0.0;
  for (int q = 0; q < NA; ++q) x_norm += x[q] * x[q];
  x_norm = sqrt(x_norm);
  while (true) {
    if (need_linearize) {
      lin(x, scale, H, g, &x_cost, &invalid);
      gmax = 0.0;
      for (int q = 0; q < N; ++q) gmax = fmax(gmax, fabs(g[q] / scale[q]));
      need_linearize = false;
    }
    if (iter >= kInnerMaxIterations) break;
    if (step_successful && gmax <= kInnerGradientTolerance) break;
    if (radius <= 1e-32) break;
    ++iter;
    double d[N], y[N];
    for (int q = 0; q < N; ++q) d[q] = fmin(fmax(H[itri(q, q)], 1e-6), 1e32) / radius;
    const bool pd = chol_solve_small<N>(H, d, g, y);
    double yg = 0.0, yHy = 0.0;
    for (int a = 0; a < N; ++a) {
      yg += y[a] * g[a];
      double row = 0.0;
      for (int b = 0; b < N; ++b) row += H[a >= b ? itri(a, b) : itri(b, a)] * y[b];
      yHy += y[a] * row;
    }
    const double mcc = yg - 0.5 * yHy;
    double step[N], xc[NA], stepsq = 0.0, xnormsq = 0.0;
    for (int q = 0; q < N; ++q) step[q] = -y[q] * scale[q];
    plus(x, step, xc);
    for (int q = 0; q < NA; ++q) { stepsq += (x[q] - xc[q]) * (x[q] - xc[q]); xnormsq += xc[q] * xc[q]; }
    if (!(pd && isfinite(mcc) && isfinite(stepsq) && mcc > 0.0)) {
      if (++invalid_steps >= 5) break;
      radius /= decrease_factor; decrease_factor *= 2.0; step_successful = false;
      continue;
    }
    invalid_steps = 0;
    bool cinv = false;
    double cand_cost = cost(xc, &cinv);
    if (cinv || !isfinite(cand_cost)) cand_cost = DBL_MAX;
    if (sqrt(stepsq) <= kInnerParameterTolerance * (x_norm + kInnerParameterTolerance)) break;
    const double cost_change = x_cost - cand_cost;
    if (fabs(cost_change) <= kInnerFunctionTolerance * x_cost) break;
    const double rho = cost_change / mcc;
    if (rho > 1e-3) {
      for (int q = 0; q < NA; ++q) x[q] = xc[q];
      x_norm = sqrt(xnormsq);
      const double t = 2.0 * rho - 1.0;
      radius = fmin(kInnerMaxRadius, radius / fmax(1.0 / 3.0, 1.0 - t * t * t));
      decrease_factor = 2.0; step_successful = true; need_linearize = true;
    } else {
      radius /= decrease_factor; decrease_factor *= 2.0; step_successful = false;
    }
  }
}

__device__ __forceinline__ double wave_sum64(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
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

// ------------------------------------------------------------------ cameras
__global__ __launch_bounds__(256) void k_inner_views(InnerArgs A) {
  if (!*A.gate) return;
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= A.P.nc || A.P.cam_red[c] < 0) return;
  if (A.own_world > 1 && c % A.own_world != A.own_rank) return;   // another rank sweeps this camera
  const unsigned mask = A.P.cam_mask[c];
  if ((mask & 0x3fu) == 0x3fu) return;
  const int beg = A.cam_obs_off[c], end = A.cam_obs_off[c + 1];
  const int grp = A.P.cam_group[c];
  const int model = A.P.group_model[grp];
  double intr[THEIA_MAX_INTRINSICS];
  for (int q = 0; q < THEIA_MAX_INTRINSICS; ++q) intr[q] = A.intr[(size_t)grp * THEIA_MAX_INTRINSICS + q];
  double x[6];
  for (int q = 0; q < 6; ++q) x[q] = A.cam[6 * (size_t)c + q];

  auto accumulate = [&](const double* ext, const double* scale, bool want_jac, double* H, double* g, double* cost_out, bool* invalid) {
    double acc[28];
    for (int k = 0; k < 28; ++k) acc[k] = 0.0;
    double inv = 0.0;
    RotTerms rt;   // one camera, one rotation: the terms once per evaluation, not once per observation (observe() = these + observe_rot())
    rotation_terms(ext + 3, rt);
    for (int i = beg + lane; i < end; i += 64) {
      const ObsRef ob = load_obs(A, A.cam_obs_idx[i]);
      const double4 Xv = reinterpret_cast<const double4*>(A.pts)[ob.pt];
      const double X[4] = {Xv.x, Xv.y, Xv.z, Xv.w};
      ObsLin ol;
      const int m = ob.depth_row ? THIP_MODEL_DEPTH_ROW : model;
      if (want_jac) observe_rot<true, false>(m, ext, rt, intr, X, ob.uv.x, ob.uv.y, ob.six, ob.siy, ol);
      else observe_rot<false, false>(m, ext, rt, intr, X, ob.uv.x, ob.uv.y, ob.six, ob.siy, ol);
      if (!ol.valid) inv += 1.0;
      double rho1;
      acc[27] += 0.5 * obs_loss(A, ob, ol.r[0] * ol.r[0] + ol.r[1] * ol.r[1], &rho1);
      if (!want_jac) continue;
      const double sr = sqrt(rho1);
      const double r0 = sr * ol.r[0], r1 = sr * ol.r[1];
      double J[12];
      for (int q = 0; q < 6; ++q) {
        const double sc = ((mask >> q) & 1u) ? 0.0 : sr * scale[q];
        J[q] = ol.Jc[q] * sc; J[6 + q] = ol.Jc[6 + q] * sc;
      }
      int k = 0;
      for (int a = 0; a < 6; ++a) {
        for (int b = 0; b <= a; ++b) acc[k++] += J[a] * J[b] + J[6 + a] * J[6 + b];
        acc[21 + a] += J[a] * r0 + J[6 + a] * r1;
      }
    }
    // the camera's prior rows (3 residuals each, no loss): bundle_adjuster.cc:291-313
    for (int i = lane; i < A.P.n_priors; i += 64) {
      if (A.P.prior_cam[i] != c) continue;
      double r[3], Jp[18];
      camera_prior(A.P.prior_kind[i], ext, A.P.prior_vec + 3 * (size_t)i, A.P.prior_info + 9 * (size_t)i, want_jac, r, Jp);
      acc[27] += 0.5 * ((r[0] * r[0] + r[1] * r[1]) + r[2] * r[2]);
      if (!want_jac) continue;
      for (int row = 0; row < 3; ++row) {
        double J[6];
        for (int q = 0; q < 6; ++q) J[q] = ((mask >> q) & 1u) ? 0.0 : Jp[6 * row + q] * scale[q];
        int k = 0;
        for (int a = 0; a < 6; ++a) {
          for (int b = 0; b <= a; ++b) acc[k++] += J[a] * J[b];
          acc[21 + a] += J[a] * r[row];
        }
      }
    }
    for (int k = 0; k < 28; ++k) acc[k] = wave_sum64(acc[k]);
    if (H) { for (int k = 0; k < 21; ++k) H[k] = acc[k]; for (int k = 0; k < 6; ++k) g[k] = acc[21 + k]; }
    *cost_out = acc[27];
    *invalid = wave_sum64(inv) > 0.0;
  };
  block_lm<6, 6>(
      x,
      [&](const double* xx, const double* scale, double* H, double* g, double* cost, bool* invalid) { accumulate(xx, scale, true, H, g, cost, invalid); },
      [&](const double* xc, bool* invalid) { double cst; accumulate(xc, nullptr, false, nullptr, nullptr, &cst, invalid); return cst; },
      [&](const double* xx, const double* step, double* xc) { for (int q = 0; q < 6; ++q) xc[q] = ((mask >> q) & 1u) ? xx[q] : xx[q] + step[q]; });
  if (lane == 0) for (int q = 0; q < 6; ++q) A.cam[6 * (size_t)c + q] = x[q];
}

// ------------------------------------------------------------------ shared intrinsics
// grp_wgs workgroups per group (round 4; one workgroup walked all of a group's observations per LM pass: 7.7 ms per sweep
// at C4 with eight groups of 375 000 observations).  Every workgroup of a group runs the SAME LM loop on the SAME sums: a pass
// over the observations is dealt to the workgroups, each leaves its 67 partial sums in global memory, they meet at a
// per-group arrival counter, and every one of them adds the partials in workgroup order -- identical bits everywhere, so the
// control flow stays in step without any further exchange.  The workgroups of a group must be resident together: the
// launch is cooperative (launch_inner_sweep), and falls back to one workgroup per group where that is not possible.
__global__ __launch_bounds__(256) void k_inner_groups(InnerArgs A) {
  if (!*A.gate) return;
  __shared__ double red[4][68];
  __shared__ double tot[68];
  const int NB = A.grp_wgs > 1 ? A.grp_wgs : 1;
  const int grp = blockIdx.x / NB, sub = blockIdx.x - grp * NB;
  if (grp >= A.P.ng_total || A.P.grp_red[grp] < 0) return;
  if (A.own_world > 1 && grp % A.own_world != A.own_rank) return;
  const unsigned free_mask = A.P.grp_free[grp];
  if (!free_mask) return;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int model = A.P.group_model[grp];
  const int beg = A.grp_obs_off[grp], end = A.grp_obs_off[grp + 1];
  constexpr int K = THEIA_MAX_INTRINSICS;
  double x[K];
  for (int q = 0; q < K; ++q) x[q] = A.intr[(size_t)grp * K + q];
  int pass = 0;   // barriers passed so far: the same number in every workgroup of the group
  bool aborted = false;   // a partner workgroup did not arrive (see the arrival loop): leave without a result

  auto accumulate = [&](const double* kk, const double* scale, bool want_jac, double* H, double* g, double* cost_out, bool* invalid) {
    double acc[67];
    for (int k = 0; k < 67; ++k) acc[k] = 0.0;
    for (int i = beg + sub * 256 + tid; i < end; i += 256 * NB) {
      const ObsRef ob = load_obs(A, A.grp_obs_idx[i]);
      if (ob.depth_row) {   // a depth-prior row does not depend on the intrinsics: constant in this block's problem
        continue;
      }
      const double4 Xv = reinterpret_cast<const double4*>(A.pts)[ob.pt];
      const double X[4] = {Xv.x, Xv.y, Xv.z, Xv.w};
      ObsLinK ol;
      if (want_jac) observe<true, true, ObsLinK>(model, A.cam + 6 * (size_t)ob.cam, kk, X, ob.uv.x, ob.uv.y, ob.six, ob.siy, ol);
      else observe<false, false, ObsLinK>(model, A.cam + 6 * (size_t)ob.cam, kk, X, ob.uv.x, ob.uv.y, ob.six, ob.siy, ol);
      if (!ol.valid) acc[66] += 1.0;
      double rho1;
      acc[65] += 0.5 * obs_loss(A, ob, ol.r[0] * ol.r[0] + ol.r[1] * ol.r[1], &rho1);
      if (!want_jac) continue;
      const double sr = sqrt(rho1);
      const double r0 = sr * ol.r[0], r1 = sr * ol.r[1];
      double J[2 * K];
      for (int q = 0; q < K; ++q) {
        const double sc = ((free_mask >> q) & 1u) ? sr * scale[q] : 0.0;
        J[q] = ol.Jk[q] * sc; J[K + q] = ol.Jk[K + q] * sc;
      }
      int k = 0;
      for (int a = 0; a < K; ++a) {
        for (int b = 0; b <= a; ++b) acc[k++] += J[a] * J[b] + J[K + a] * J[K + b];
        acc[55 + a] += J[a] * r0 + J[K + a] * r1;
      }
    }
    __syncthreads();
    for (int k = 0; k < 67; ++k) {
      const double v = wave_sum64(acc[k]);
      if (lane == 0) red[wv][k] = v;
    }
    __syncthreads();
    if (NB > 1) {
      double* part = A.grp_part + ((size_t)grp * 2 + (pass & 1)) * NB * kInnerGroupSums;
      if (tid < 67) part[(size_t)sub * kInnerGroupSums + tid] = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
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
        red[0][67] = gone ? 1.0 : 0.0;
      }
      __syncthreads();
      if (red[0][67] != 0.0) { aborted = true; }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      if (tid < 67) {
        double v = 0.0;
        if (!aborted) for (int w = 0; w < NB; ++w) v += __builtin_nontemporal_load(part + (size_t)w * kInnerGroupSums + tid);
        tot[tid] = v;
      }
      ++pass;
      __syncthreads();
      for (int k = 0; k < 67; ++k) acc[k] = tot[k];
      __syncthreads();
    } else {
      for (int k = 0; k < 67; ++k) acc[k] = ((red[0][k] + red[1][k]) + red[2][k]) + red[3][k];
    }
    if (H) { for (int k = 0; k < 55; ++k) H[k] = acc[k]; for (int k = 0; k < K; ++k) g[k] = acc[55 + k]; }
    *cost_out = acc[65];
    *invalid = acc[66] > 0.0;
  };
  if (NB == 1 && A.grp_bar && A.grp_bar[A.P.ng_total + 1 + grp]) return;   // the follow-up launch: this group is done
  block_lm<K, K>(
      x,
      [&](const double* xx, const double* scale, double* H, double* g, double* cost, bool* invalid) {
        if (aborted) { for (int k = 0; k < 55; ++k) H[k] = 0.0; for (int k = 0; k < K; ++k) g[k] = 0.0; *cost = 0.0; *invalid = true; return; }
        accumulate(xx, scale, true, H, g, cost, invalid);
        if (aborted) *invalid = true; },
      [&](const double* xc, bool* invalid) {
        if (aborted) { *invalid = true; return 0.0; }
        double cst; accumulate(xc, nullptr, false, nullptr, nullptr, &cst, invalid);
        if (aborted) *invalid = true;
        return cst; },
      [&](const double* xx, const double* step, double* xc) { for (int q = 0; q < K; ++q) xc[q] = ((free_mask >> q) & 1u) ? xx[q] + step[q] : xx[q]; });
  if (aborted) return;
  if (tid == 0 && sub == 0) {
    for (int q = 0; q < K; ++q) A.intr[(size_t)grp * K + q] = x[q];
    if (A.grp_bar) A.grp_bar[A.P.ng_total + 1 + grp] = 1;
  }
}

// ------------------------------------------------------------------ points
// accumulate() is kept out of line on purpose: with it inlined into the LM loop, hipcc 7.2 -O2 / -O3 produced wrong
// steps for this per-thread kernel (the -O1 build, the build with the body behind a call, and the two cooperative
// kernels above all agree with the oracle to 1e-14; tests/test_inner_gpu.py with THEIA_HIP_INNER_SKIP=3).
// ROT: the cameras' rotation terms, intrinsics and model come from the 40-double blocks k_inner_cam_blocks left in
// P.camrot_cand after the camera / intrinsics sweeps (ba_device.h kCamRot) -- one gather instead of a sincos and the
// camera -> group -> intrinsics chain per observation and LM iteration; observe() = rotation_terms() + observe_rot(), so
// the bits are the same.
template <int PD, bool ROT>
struct TrackFn {
  const InnerArgs* A;
  int beg, end;
  __device__ __attribute__((noinline)) void accumulate(const double* X, const double* scale, bool want_jac, double* H, double* g, double* cost_out, bool* invalid) const {
    constexpr int NT = PD * (PD + 1) / 2;
    const InnerArgs& Ar = *A;
    if (want_jac) { for (int k = 0; k < NT; ++k) H[k] = 0.0; for (int k = 0; k < PD; ++k) g[k] = 0.0; }
    double cst = 0.0;
    bool inv = false;
    for (int o = beg; o < end; ++o) {
      const ObsRef ob = load_obs(Ar, o);
      ObsLin ol;
      if constexpr (ROT) {
        const double* cr = Ar.P.camrot_cand + (size_t)kCamRot * ob.cam;
        double ext[6], kc[12];
        RotTerms rt;
        camrot_load(cr, ext, rt);
        load_d2<12>(cr + kCamRotIntr, kc);   // intrinsics (10) | model | reduced index
        const int m = ob.depth_row ? THIP_MODEL_DEPTH_ROW : (int)kc[kCamRotModel - kCamRotIntr];
        if (want_jac) observe_rot<true, false>(m, ext, rt, kc, X, ob.uv.x, ob.uv.y, ob.six, ob.siy, ol);
        else observe_rot<false, false>(m, ext, rt, kc, X, ob.uv.x, ob.uv.y, ob.six, ob.siy, ol);
      } else {
        const int grp = Ar.P.cam_group[ob.cam];
        const int m = ob.depth_row ? THIP_MODEL_DEPTH_ROW : Ar.P.group_model[grp];
        const double* ext = Ar.cam + 6 * (size_t)ob.cam;
        const double* kk = Ar.intr + (size_t)grp * THEIA_MAX_INTRINSICS;
        if (want_jac) observe<true, false>(m, ext, kk, X, ob.uv.x, ob.uv.y, ob.six, ob.siy, ol);
        else observe<false, false>(m, ext, kk, X, ob.uv.x, ob.uv.y, ob.six, ob.siy, ol);
      }
      if (!ol.valid) inv = true;
      double rho1;
      cst += 0.5 * obs_loss(Ar, ob, ol.r[0] * ol.r[0] + ol.r[1] * ol.r[1], &rho1);
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
    *cost_out = cst; *invalid = inv;
  }
};
template <int PD, bool ROT> struct TrackLin {
  TrackFn<PD, ROT> f;
  __device__ void operator()(const double* x, const double* scale, double* H, double* g, double* cost, bool* invalid) const { f.accumulate(x, scale, true, H, g, cost, invalid); }
};
template <int PD, bool ROT> struct TrackCost {
  TrackFn<PD, ROT> f;
  __device__ double operator()(const double* xc, bool* invalid) const { double c; f.accumulate(xc, nullptr, false, nullptr, nullptr, &c, invalid); return c; }
};
template <int PD> struct TrackPlus {
  __device__ void operator()(const double* x, const double* step, double* xc) const {
    if (PD == 3) { const double d3[3] = {step[0], step[1], step[2]}; sphere_plus(x, d3, xc); }
    else for (int q = 0; q < 4; ++q) xc[q] = x[q] + step[q];
  }
};

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

#ifndef THIP_INNER_TRACK_WAVES
#define THIP_INNER_TRACK_WAVES 2   // measured at C4: 1.70 (one wave per SIMD), 1.67 (two), 1.73 (three), 1.80 ms (four) per LM iteration
#endif
template <int PD, bool ROT>
__global__ __launch_bounds__(64, THIP_INNER_TRACK_WAVES) void k_inner_tracks(InnerArgs A) {
  if (!*A.gate) return;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= A.ntracks) return;
  const int beg = A.trk_off[t], end = A.trk_off[t + 1];
  if (end <= beg) return;
  const int p = A.P.obs_pt[beg];
  if (A.P.pt_const[p]) return;
  double x[4];
  for (int q = 0; q < 4; ++q) x[q] = A.pts[4 * (size_t)p + q];
  const TrackFn<PD, ROT> fn{&A, beg, end};
  block_lm<PD, 4>(x, TrackLin<PD, ROT>{fn}, TrackCost<PD, ROT>{fn}, TrackPlus<PD>{});
  for (int q = 0; q < 4; ++q) A.pts[4 * (size_t)p + q] = x[q];
}

// |x - x_inner|^2 and |x_inner|^2 over the variable blocks (ParameterToleranceReached measures the step to the point
// the inner iterations ended at): out[0] = step^2, out[1] = |x_inner|^2.  Two fixed-order stages: kInnerCostBlocks
// workgroups over contiguous slices of the blocks (one workgroup over 500k points took 0.44 ms), then one workgroup
// over their partial sums.
__global__ __launch_bounds__(256) void k_inner_norms(InnerArgs A, const double* __restrict__ cam0, const double* __restrict__ pts0,
                                                     const double* __restrict__ intr0, double* __restrict__ part, int which) {
  // which: 0 = every variable block, 1 = the points only (a track shard's share), 2 = cameras + intrinsics only
  __shared__ double s1[256], s2[256];
  double a = 0.0, b = 0.0;
  const int tid = threadIdx.x, nb = gridDim.x, blk = blockIdx.x;
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
__global__ __launch_bounds__(256) void k_inner_norms_reduce(int nparts, const double* __restrict__ part, double* __restrict__ out) {
  __shared__ double s1[256], s2[256];
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

// cost at the inner-iteration point: every observation row (tiles or not), then the camera priors; two fixed-order stages
__global__ __launch_bounds__(256) void k_inner_cost(InnerArgs A, double* __restrict__ part) {
  if (!*A.gate) return;
  __shared__ double s1[256], s2[256];
  double cst = 0.0, inv = 0.0;
  for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < A.nobs; o += (int64_t)gridDim.x * 256) {
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
  if (threadIdx.x == 0) { part[2 * blockIdx.x] = s1[0]; part[2 * blockIdx.x + 1] = s2[0]; }
}
__global__ __launch_bounds__(256) void k_inner_cost_reduce(InnerArgs A, const double* __restrict__ part, int nblocks, double* __restrict__ out) {
  if (!*A.gate) return;
  __shared__ double s1[256], s2[256];
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
  if (A.P.nc > 0 && (stages & 1) && !(skip & 1)) k_inner_views<<<(A.P.nc + 3) / 4, 256, 0, st>>>(A);
  if (A.P.ni > 0 && A.P.ng_total > 0 && (stages & 2) && !(skip & 2)) {
    // several workgroups per group need all of them resident at once: a cooperative launch guarantees that (or fails);
    // not inside a stream capture, and one workgroup per group whenever it is refused
    bool done = false;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(st, &cap);
    static const bool single = getenv("THEIA_HIP_INNER_GROUPS_SINGLE") != nullptr;
    if (A.grp_wgs > 1 && A.grp_part && A.grp_bar && cap == hipStreamCaptureStatusNone && !single) {
      if (hipMemsetAsync(A.grp_bar, 0, sizeof(int) * (2 * (size_t)A.P.ng_total + 2), st) == hipSuccess) {
        InnerArgs Ac = A;
        // (development / test switch: THEIA_HIP_INNER_GROUPS_MAX_POLLS=0 makes the first waiting workgroup give up at once,
        // which exercises the give-up-and-redo path of the follow-up launch)
        static const int max_polls = [] { const char* e = getenv("THEIA_HIP_INNER_GROUPS_MAX_POLLS"); return e ? atoi(e) : 200000; }();
        Ac.grp_max_polls = max_polls;
        void* args[] = {&Ac};
        const hipError_t e = hipLaunchCooperativeKernel((const void*)k_inner_groups, dim3((unsigned)(A.P.ng_total * A.grp_wgs)), dim3(256), args, 0, st);
        if (e == hipSuccess) done = true; else (void)hipGetLastError();
      }
    }
    // one workgroup per group: the whole sweep when the cooperative launch was not possible, otherwise the safety net that
    // returns at once for every group the cooperative launch marked done (all of them, unless it had to give up)
    { InnerArgs A1 = A; A1.grp_wgs = 1; if (!done) A1.grp_bar = nullptr; k_inner_groups<<<A.P.ng_total, 256, 0, st>>>(A1); }
  }
  if (A.ntracks > 0 && (stages & 4) && !(skip & 4)) {
    const int nb = (A.ntracks + 63) / 64;
    if (A.P.camrot_cand && !getenv("THEIA_HIP_INNER_NO_ROT")) {   // fused path: per-camera blocks (free between the trial step and the next one)
      k_inner_cam_blocks<<<(A.P.nc + 255) / 256, 256, 0, st>>>(A);
      if (A.P.pd == 3) k_inner_tracks<3, true><<<nb, 64, 0, st>>>(A); else k_inner_tracks<4, true><<<nb, 64, 0, st>>>(A);
    } else {
      if (A.P.pd == 3) k_inner_tracks<3, false><<<nb, 64, 0, st>>>(A); else k_inner_tracks<4, false><<<nb, 64, 0, st>>>(A);
    }
  }
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
