// twoview_lm.hip -- batches of INDEPENDENT two-view angular adjustments, one whole
// Levenberg-Marquardt solve per wavefront.
//
// Replaces N calls of BundleAdjustTwoViewsAngular (bundle_adjust_two_views.cc:189-246): the relative
// rotation of view 2 (angle-axis, 3 dofs) and its unit position (SphereManifold<3>, 2 dofs) against
// the angular epipolar error (angular_epipolar_error.h:54-91) of every correspondence, one robustified
// residual each.  Its caller on the hot path is RefineModel of the relative-pose estimator
// (estimate_relative_pose.cc:111-138: TRUNCATED loss of width error_thresh, <= 15 iterations,
// linear_solver_type CGNR + JACOBI) at every LO-RANSAC event, thousands of times per view graph.
//
// The trust-region rules are those of ba_batch.hip (Ceres 2.2 TrustRegionMinimizer +
// LevenbergMarquardtStrategy, Jacobi scaling, loss corrector).  The LINEAR solver is what differs:
// CGNR is conjugate gradients on the normal equations, preconditioned by their block diagonal over
// the two parameter blocks and stopped by Ceres' q-tolerance rule (eta = 0.1) -- an INEXACT step,
// which is part of the behaviour and is restated (cgnr5).  The 5 x 5 normal matrix is formed
// explicitly (wave reduction, fixed order), so  A p = H p + D^2 p  where Ceres evaluates J'(J p).
// With the default options (SPARSE_SCHUR; RefineModel of the uncalibrated relative-pose estimator,
// estimate_uncalibrated_relative_pose.cc:142-176) the step is the exact solution (solve5).
#include "ba_device.h"
#include "wave_reduce.h"
#include "ransac_device.h"
#include "theia_hip_internal.h"

#include <chrono>
#include <cmath>
#include <vector>

#define HIP_TRY(expr)                                                                             \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess) return thip::set_error(THEIA_HIP_ERR_INTERNAL, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

namespace thip {
namespace {

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// the XOR butterfly over the 64 lanes, on permlane swaps + DPP (wave_reduce.h: the bits of the __shfl_xor loop)
__device__ __forceinline__ double wsum(double v) { return wave_sum_butterfly(v); }
__device__ __forceinline__ int tri(int a, int b) { return a * (a + 1) / 2 + b; }

struct TwoViewBatch {
  int num;
  const int64_t* offsets;
  const int* counts;        // optional: problem p = [offsets[p], offsets[p] + counts[p])
  const double4* corr;      // (x1, y1, x2, y2)
  double* pose;             // [num][6] in/out: rotation_2 (angle-axis) | position_2
  int loss_type;
  double loss_width;
  int max_iterations;
  double function_tolerance, gradient_tolerance, parameter_tolerance, max_radius;
  int cgnr;                 // 1: the CGNR + JACOBI inexact step; 0: exact solve of the normal equations
};

struct TwoViewOut {   // = ba_batch.hip ViewOut
  int success, term, iters, nsucc;
  double initial_cost, final_cost;
};

// ceres SphereManifold<3>: householder_vector.h + sphere_manifold_functions.h (2.2)
__device__ void householder3(const double x[3], double v[3], double& beta) {
  const double sigma = x[0] * x[0] + x[1] * x[1];
  v[0] = x[0]; v[1] = x[1]; v[2] = 1.0;
  beta = 0.0;
  if (sigma <= DBL_EPSILON) { if (x[2] < 0.0) beta = 2.0; return; }
  const double mu = sqrt(x[2] * x[2] + sigma);
  const double vp = (x[2] <= 0.0) ? x[2] - mu : -sigma / (x[2] + mu);
  beta = 2.0 * vp * vp / (sigma + vp * vp);
  v[0] /= vp; v[1] /= vp;
}
__device__ void sphere3_plus(const double x[3], const double d[2], double out[3]) {
  const double nd = sqrt(d[0] * d[0] + d[1] * d[1]);
  if (nd == 0.0) { out[0] = x[0]; out[1] = x[1]; out[2] = x[2]; return; }
  double v[3], beta;
  householder3(x, v, beta);
  const double nx = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
  double s, c;
  sincos(nd, &s, &c);
  const double sbd = s / nd;
  const double y[3] = {sbd * d[0], sbd * d[1], c};
  const double vty = v[0] * y[0] + v[1] * y[1] + v[2] * y[2];
  for (int i = 0; i < 3; ++i) out[i] = nx * (y[i] - v[i] * (beta * vty));
}
// 3 x 2 row-major: |x| (I - beta v v')[:, 0:2]
__device__ void sphere3_plus_jacobian(const double x[3], double J[6]) {
  double v[3], beta;
  householder3(x, v, beta);
  const double nx = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 2; ++c) J[r * 2 + c] = nx * ((r == c ? 1.0 : 0.0) - beta * v[r] * v[c]);
}

// ceres AngleAxisToRotationMatrix (rotation.h), operation order kept; R row-major
__device__ void aa_to_rot(const double* aa, double* R) {
  const double theta2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
  if (theta2 > DBL_EPSILON) {
    const double theta = sqrt(theta2);
    const double wx = aa[0] / theta, wy = aa[1] / theta, wz = aa[2] / theta;
    double s, c;
    sincos(theta, &s, &c);
    R[0] = c + wx * wx * (1.0 - c);      R[3] = wz * s + wx * wy * (1.0 - c);  R[6] = -wy * s + wx * wz * (1.0 - c);
    R[1] = wx * wy * (1.0 - c) - wz * s; R[4] = c + wy * wy * (1.0 - c);       R[7] = wx * s + wy * wz * (1.0 - c);
    R[2] = wy * s + wx * wz * (1.0 - c); R[5] = -wx * s + wy * wz * (1.0 - c); R[8] = c + wz * wz * (1.0 - c);
  } else {
    R[0] = 1.0; R[3] = aa[2]; R[6] = -aa[1];
    R[1] = -aa[2]; R[4] = 1.0; R[7] = aa[0];
    R[2] = aa[1]; R[5] = -aa[0]; R[8] = 1.0;
  }
}

// what a linearisation point needs once: R, the rotation-derivative terms of w and of -w, I - t t', the 3 x 2 plus Jacobian
struct PoseTerms {
  double R[9], M[9], PJ[6];
  RotTerms rp, rn;
  double wn[3];
};
__device__ void pose_terms(const double x[6], bool want_jac, PoseTerms& T) {
  aa_to_rot(x, T.R);
  const double* t = x + 3;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) T.M[3 * i + j] = (i == j ? 1.0 : 0.0) - t[i] * t[j];
  if (!want_jac) return;
  rotation_terms(x, T.rp);
  T.wn[0] = -x[0]; T.wn[1] = -x[1]; T.wn[2] = -x[2];
  rotation_terms(T.wn, T.rn);
  sphere3_plus_jacobian(t, T.PJ);
}

__device__ __forceinline__ void mat3_vec(const double* A, const double* v, double* o) {
  for (int i = 0; i < 3; ++i) o[i] = (A[3 * i] * v[0] + A[3 * i + 1] * v[1]) + A[3 * i + 2] * v[2];
}
__device__ __forceinline__ void mat3t_vec(const double* A, const double* v, double* o) {
  for (int i = 0; i < 3; ++i) o[i] = (A[i] * v[0] + A[3 + i] * v[1]) + A[6 + i] * v[2];
}
__device__ __forceinline__ double dot3(const double* a, const double* b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }

// AngularEpipolarError::operator() (angular_epipolar_error.h:54-91); J = d r / d [rotation (3) | position tangent (2)]
template <bool WANT_JAC>
__device__ double angular_error(const double x[6], const PoseTerms& T, const double4 c, double* J) {
  const double f1[3] = {c.x, c.y, 1.0}, f2[3] = {c.z, c.w, 1.0};
  const double* t = x + 3;
  double u[3], w[3], Mf1[3], Mw[3];
  mat3_vec(T.R, f2, u);      // R f2
  mat3t_vec(T.R, f2, w);     // R' f2
  mat3_vec(T.M, f1, Mf1);
  mat3_vec(T.M, w, Mw);
  const double a = dot3(f1, Mf1) + dot3(u, Mw);
  const double cr[3] = {f1[1] * w[2] - f1[2] * w[1], f1[2] * w[0] - f1[0] * w[2], f1[0] * w[1] - f1[1] * w[0]};
  const double b = dot3(t, cr);
  const double s = (a * a) / 4.0 - b * b;
  if (s < 0.0) {
    if (WANT_JAC) { for (int k = 0; k < 5; ++k) J[k] = 0.0; }
    return 1000.0;
  }
  const double rs = sqrt(s);
  if (WANT_JAC) {
    double Mu[3], Du[9], Dw[9];
    mat3_vec(T.M, u, Mu);
    rotation_dq_dw(x, f2, T.rp, Du);          // d(R(w) f2)/dw
    rotation_dq_dw(T.wn, f2, T.rn, Dw);       // d(R(-w) f2)/d(-w)
    const double tf1 = dot3(t, f1), tu = dot3(t, u), tw = dot3(t, w);
    const double tx[3] = {t[1] * f1[2] - t[2] * f1[1], t[2] * f1[0] - t[0] * f1[2], t[0] * f1[1] - t[1] * f1[0]};   // t x f1
    double dr[6];
    for (int k = 0; k < 3; ++k) {
      // rotation: da = (M w)' du/dw_k + (M u)' dw/dw_k ; db = (t x f1)' dw/dw_k ; dw/dw_k = -Dw[:, k]
      const double da = ((Mw[0] * Du[k] + Mw[1] * Du[3 + k]) + Mw[2] * Du[6 + k]) -
                        ((Mu[0] * Dw[k] + Mu[1] * Dw[3 + k]) + Mu[2] * Dw[6 + k]);
      const double db = -((tx[0] * Dw[k] + tx[1] * Dw[3 + k]) + tx[2] * Dw[6 + k]);
      const double ds = (a / 2.0) * da - 2.0 * b * db;
      dr[k] = da / 2.0 - ds / (2.0 * rs);
      // position (ambient)
      const double dat = -2.0 * tf1 * f1[k] - tw * u[k] - tu * w[k];
      const double dst = (a / 2.0) * dat - 2.0 * b * cr[k];
      dr[3 + k] = dat / 2.0 - dst / (2.0 * rs);
    }
    J[0] = dr[0]; J[1] = dr[1]; J[2] = dr[2];
    J[3] = (dr[3] * T.PJ[0] + dr[4] * T.PJ[2]) + dr[5] * T.PJ[4];
    J[4] = (dr[3] * T.PJ[1] + dr[4] * T.PJ[3]) + dr[5] * T.PJ[5];
  }
  return a / 2.0 - rs;
}

__device__ double twoview_cost(const TwoViewBatch& B, int p, const double x[6], int lane) {
  const int64_t beg = B.offsets[p], end = B.counts ? beg + B.counts[p] : B.offsets[p + 1];
  PoseTerms T;
  pose_terms(x, false, T);
  double cost = 0.0;
  for (int64_t o = beg + lane; o < end; o += 64) {
    const double r = angular_error<false>(x, T, B.corr[o], nullptr);
    double rho1;
    cost += 0.5 * loss_eval(B.loss_type, B.loss_width, r * r, &rho1);
  }
  return wsum(cost);
}

// H (packed lower 15) = J'J, g = J'r with the loss corrector and the column scaling applied
__device__ void twoview_linearize(const TwoViewBatch& B, int p, const double x[6], const double* scale, int lane,
                                  double* H, double* g, double* cost) {
  const int64_t beg = B.offsets[p], end = B.counts ? beg + B.counts[p] : B.offsets[p + 1];
  PoseTerms T;
  pose_terms(x, true, T);
  double acc[21];
#pragma unroll
  for (int k = 0; k < 21; ++k) acc[k] = 0.0;
  for (int64_t o = beg + lane; o < end; o += 64) {
    double J[5];
    const double r = angular_error<true>(x, T, B.corr[o], J);
    double rho1;
    const double rho = loss_eval(B.loss_type, B.loss_width, r * r, &rho1);
    const double sr = sqrt(rho1);
    const double rr = sr * r;
#pragma unroll
    for (int q = 0; q < 5; ++q) J[q] *= sr * scale[q];
    int k = 0;
#pragma unroll
    for (int a = 0; a < 5; ++a) {
#pragma unroll
      for (int b = 0; b <= a; ++b) acc[k++] += J[a] * J[b];
      acc[15 + a] += J[a] * rr;
    }
    acc[20] += 0.5 * rho;
  }
#pragma unroll
  for (int k = 0; k < 21; ++k) acc[k] = wsum(acc[k]);
#pragma unroll
  for (int k = 0; k < 15; ++k) H[k] = acc[k];
#pragma unroll
  for (int k = 0; k < 5; ++k) g[k] = acc[15 + k];
  *cost = acc[20];
}

__device__ __forceinline__ bool zero_or_inf(double v) { return v == 0.0 || isinf(v); }

// inverse of an n x n SPD block (n <= 3) as BlockRandomAccessDiagonalMatrix::Invert does it:
// llt().solve(Identity)
__device__ void spd_inverse(int n, const double* A /* n x n row-major */, double* Ai) {
  double L[9];
  for (int j = 0; j < n; ++j) {
    double s = A[j * n + j];
    for (int k = 0; k < j; ++k) s -= L[j * n + k] * L[j * n + k];
    const double d = sqrt(s);
    L[j * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double v = A[i * n + j];
      for (int k = 0; k < j; ++k) v -= L[i * n + k] * L[j * n + k];
      L[i * n + j] = v / d;
    }
  }
  for (int c = 0; c < n; ++c) {
    double z[3];
    for (int i = 0; i < n; ++i) {
      double s = (i == c) ? 1.0 : 0.0;
      for (int k = 0; k < i; ++k) s -= L[i * n + k] * z[k];
      z[i] = s / L[i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
      double s = z[i];
      for (int k = i + 1; k < n; ++k) s -= L[k * n + i] * Ai[k * n + c];
      Ai[i * n + c] = s / L[i * n + i];
    }
  }
}

// CgnrSolver::SolveImpl + ConjugateGradientsSolver (cgnr_solver.cc, conjugate_gradients_solver.h, 2.2):
// A = H + diag(D2), b = g, x0 = 0, JACOBI preconditioner, q_tolerance 0.1, r_tolerance -1,
// max_num_iterations 500, residual_reset_period 10.  Returns false on FAILURE (the LM step is then invalid);
// NO_CONVERGENCE ("matrix is indefinite", iteration cap) leaves a usable y like Ceres does.
__device__ bool cgnr5(const double* H, const double* D2, const double* g, double* y) {
  auto Amul = [&](const double* v, double* o) {
    for (int a = 0; a < 5; ++a) {
      double s = 0.0;
      for (int b = 0; b < 5; ++b) s += H[a >= b ? tri(a, b) : tri(b, a)] * v[b];
      o[a] = s + D2[a] * v[a];
    }
  };
  for (int q = 0; q < 5; ++q) y[q] = 0.0;
  double nb = 0.0;
  for (int q = 0; q < 5; ++q) nb += g[q] * g[q];
  if (sqrt(nb) == 0.0) return true;
  // preconditioner: inverse of the (rotation, position) diagonal blocks of A
  double B3[9], B2[4], I3[9], I2[4];
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) B3[a * 3 + b] = H[a >= b ? tri(a, b) : tri(b, a)] + (a == b ? D2[a] : 0.0);
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 2; ++b) B2[a * 2 + b] = H[a >= b ? tri(3 + a, 3 + b) : tri(3 + b, 3 + a)] + (a == b ? D2[3 + a] : 0.0);
  spd_inverse(3, B3, I3);
  spd_inverse(2, B2, I2);
  double r[5], z[5], pv[5], qv[5], tmp[5];
  for (int q = 0; q < 5; ++q) r[q] = g[q];
  double rho = 1.0, Q0 = -0.0;
  for (int it = 1;; ++it) {
    for (int a = 0; a < 3; ++a) z[a] = (I3[a * 3] * r[0] + I3[a * 3 + 1] * r[1]) + I3[a * 3 + 2] * r[2];
    for (int a = 0; a < 2; ++a) z[3 + a] = I2[a * 2] * r[3] + I2[a * 2 + 1] * r[4];
    const double last_rho = rho;
    rho = 0.0;
    for (int q = 0; q < 5; ++q) rho += r[q] * z[q];
    if (zero_or_inf(rho)) return false;
    if (it == 1) { for (int q = 0; q < 5; ++q) pv[q] = z[q]; }
    else {
      const double beta = rho / last_rho;
      if (zero_or_inf(beta)) return false;
      for (int q = 0; q < 5; ++q) pv[q] = z[q] + beta * pv[q];
    }
    Amul(pv, qv);
    double pq = 0.0;
    for (int q = 0; q < 5; ++q) pq += pv[q] * qv[q];
    if (pq <= 0.0 || isinf(pq)) return true;       // NO_CONVERGENCE: the iterate so far is the step
    const double alpha = rho / pq;
    if (isinf(alpha)) return false;
    for (int q = 0; q < 5; ++q) y[q] = y[q] + alpha * pv[q];
    if (it % 10 == 0) { Amul(y, tmp); for (int q = 0; q < 5; ++q) r[q] = g[q] - tmp[q]; }
    else { for (int q = 0; q < 5; ++q) r[q] = r[q] - alpha * qv[q]; }
    double Q1 = 0.0;
    for (int q = 0; q < 5; ++q) Q1 += y[q] * (g[q] + r[q]);
    Q1 = -Q1;
    const double zeta = it * (Q1 - Q0) / Q1;
    if (zeta < 0.1) return true;
    Q0 = Q1;
    if (it >= 500) return true;
  }
}

// (H + diag(d)) y = g by Cholesky (the direct solvers: SPARSE_SCHUR / DENSE_* of the caller's options); false if not PD
__device__ bool solve5(const double* H, const double* d, const double* g, double* y) {
  double L[15];
  for (int i = 0; i < 5; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = H[tri(i, j)] + (i == j ? d[i] : 0.0);
      for (int k = 0; k < j; ++k) s -= L[tri(i, k)] * L[tri(j, k)];
      if (i == j) { if (!(s > 0.0)) return false; L[tri(i, i)] = sqrt(s); }
      else L[tri(i, j)] = s / L[tri(j, j)];
    }
  double z[5];
  for (int i = 0; i < 5; ++i) {
    double s = g[i];
    for (int k = 0; k < i; ++k) s -= L[tri(i, k)] * z[k];
    z[i] = s / L[tri(i, i)];
  }
  for (int i = 4; i >= 0; --i) {
    double s = z[i];
    for (int k = i + 1; k < 5; ++k) s -= L[tri(k, i)] * y[k];
    y[i] = s / L[tri(i, i)];
  }
  return true;
}

__global__ __launch_bounds__(256) void k_twoview_lm(TwoViewBatch B, TwoViewOut* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= B.num) return;
  double x[6];
#pragma unroll
  for (int q = 0; q < 6; ++q) x[q] = B.pose[(size_t)p * 6 + q];
  TwoViewOut R;
  R.success = 0; R.term = THEIA_TERM_NO_CONVERGENCE; R.iters = 0; R.nsucc = 0; R.initial_cost = 0.0; R.final_cost = 0.0;
  double scale[5] = {1.0, 1.0, 1.0, 1.0, 1.0};
  double H[15], g[5], x_cost;
  // Jacobi scaling from the column norms at the initial point (once per solve)
  twoview_linearize(B, p, x, scale, lane, H, g, &x_cost);
#pragma unroll
  for (int q = 0; q < 5; ++q) scale[q] = 1.0 / (1.0 + sqrt(H[tri(q, q)]));
  double radius = 1e4, decrease_factor = 2.0;
  bool step_successful = true, need_linearize = true, first = true;
  int iter = 0, invalid_steps = 0, term = THEIA_TERM_NO_CONVERGENCE;
  double x_norm = 0.0, minimum_cost = 0.0, gmax = 0.0;
#pragma unroll
  for (int q = 0; q < 6; ++q) x_norm += x[q] * x[q];
  x_norm = sqrt(x_norm);
  while (true) {
    if (need_linearize) {
      twoview_linearize(B, p, x, scale, lane, H, g, &x_cost);
      // gradient_max_norm = |x - Plus(x, -gradient)|_inf (TrustRegionMinimizer::EvaluateGradientAndJacobian)
      double ng[5], xp[3];
#pragma unroll
      for (int q = 0; q < 5; ++q) ng[q] = -(g[q] / scale[q]);
      sphere3_plus(x + 3, ng + 3, xp);
      gmax = 0.0;
#pragma unroll
      for (int q = 0; q < 3; ++q) gmax = fmax(gmax, fmax(fabs(x[q] - (x[q] + ng[q])), fabs(x[3 + q] - xp[q])));
      need_linearize = false;
    }
    if (first) {
      first = false;
      R.initial_cost = x_cost;
      minimum_cost = x_cost;
      if (!isfinite(x_cost)) { term = THEIA_TERM_FAILURE; R.final_cost = x_cost; break; }
    }
    if (iter >= B.max_iterations) { term = THEIA_TERM_NO_CONVERGENCE; break; }
    if (step_successful && gmax <= B.gradient_tolerance) { term = THEIA_TERM_CONVERGENCE; break; }
    if (radius <= 1e-32) { term = THEIA_TERM_CONVERGENCE; break; }
    ++iter;
    double d[5], y[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) d[q] = fmin(fmax(H[tri(q, q)], 1e-6), 1e32) / radius;
    const bool solved = B.cgnr ? cgnr5(H, d, g, y) : solve5(H, d, g, y);
    // model cost change of the step -y:  y'g - y'Hy/2  (H without the LM diagonal)
    double yg = 0.0, yHy = 0.0;
    for (int a = 0; a < 5; ++a) {
      yg += y[a] * g[a];
      double row = 0.0;
      for (int b = 0; b < 5; ++b) row += H[a >= b ? tri(a, b) : tri(b, a)] * y[b];
      yHy += y[a] * row;
    }
    const double mcc = yg - 0.5 * yHy;
    double cand[6], dl[5], stepsq = 0.0, xnormsq = 0.0;
#pragma unroll
    for (int q = 0; q < 5; ++q) dl[q] = -y[q] * scale[q];
#pragma unroll
    for (int q = 0; q < 3; ++q) cand[q] = x[q] + dl[q];
    sphere3_plus(x + 3, dl + 3, cand + 3);
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      stepsq += (x[q] - cand[q]) * (x[q] - cand[q]);
      xnormsq += cand[q] * cand[q];
    }
    const bool step_valid = solved && isfinite(mcc) && isfinite(stepsq) && mcc > 0.0;
    if (!step_valid) {
      if (++invalid_steps >= 5) { term = THEIA_TERM_FAILURE; break; }
      radius /= decrease_factor; decrease_factor *= 2.0; step_successful = false;
      continue;
    }
    invalid_steps = 0;
    double cand_cost = twoview_cost(B, p, cand, lane);
    if (!isfinite(cand_cost)) cand_cost = DBL_MAX;
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
  R.final_cost = term != THEIA_TERM_FAILURE ? minimum_cost : x_cost;
  if (lane == 0) {
    out[p] = R;
#pragma unroll
    for (int q = 0; q < 6; ++q) B.pose[(size_t)p * 6 + q] = x[q];
  }
}

// ------------------------------------------------------------------ homography refinement
// N independent OptimizeHomography problems (bundle_adjust_two_views.cc:298-358): the nine entries of H (no manifold)
// against the symmetric geometric distance of every correspondence (homography_error.h:45-95: forward H x1 - x2 and
// backward H^-1 x2 - x1, four residuals, one loss over them), default (direct) linear solver; finally H /= H(2, 2).
// RefineModel of the homography estimator (estimate_homography.cc:89-104) calls it with the TRUNCATED loss of width
// error_thresh and at most 15 iterations.  Parameters in Eigen's storage order: k = i + 3 j for H(i, j).
struct HomographyBatch {
  int num;
  const int64_t* offsets;
  const int* counts;
  const double4* corr;      // (x1, y1, x2, y2)
  double* H;                // [num][9] in/out, column-major
  int loss_type;
  double loss_width;
  int max_iterations;
  double function_tolerance, gradient_tolerance, parameter_tolerance, max_radius;
};

// Eigen's 3 x 3 inverse (InverseImpl.h compute_inverse<.., 3>): cofactors and the determinant along column 0
__device__ __forceinline__ double cof3(const double* m, int i, int j) {   // m column-major
  const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
  return m[i1 + 3 * j1] * m[i2 + 3 * j2] - m[i1 + 3 * j2] * m[i2 + 3 * j1];
}
__device__ void inverse3_cm(const double* m, double* r) {
  const double c0 = cof3(m, 0, 0), c1 = cof3(m, 1, 0), c2 = cof3(m, 2, 0);
  const double det = (c0 * m[0] + c1 * m[1]) + c2 * m[2];
  const double invdet = 1.0 / det;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r[i + 3 * j] = (i == 0) ? ((j == 0 ? c0 : (j == 1 ? c1 : c2)) * invdet) : cof3(m, j, i) * invdet;
}

// the four residuals and (WANT_JAC) their 4 x 9 Jacobian (row-major)
template <bool WANT_JAC>
__device__ void homography_residuals(const double* H, const double* G, const double4 c, double* r, double* J) {
  const double x[3] = {c.x, c.y, 1.0}, y[3] = {c.z, c.w, 1.0};
  double p[3], q[3];
  for (int i = 0; i < 3; ++i) {
    p[i] = (H[i] * x[0] + H[i + 3] * x[1]) + H[i + 6] * x[2];
    q[i] = (G[i] * y[0] + G[i + 3] * y[1]) + G[i + 6] * y[2];
  }
  const double u = p[0] / p[2], v = p[1] / p[2], a = q[0] / q[2], b = q[1] / q[2];
  r[0] = u - y[0]; r[1] = v - y[1]; r[2] = a - x[0]; r[3] = b - x[1];
  if (!WANT_JAC) return;
  for (int j = 0; j < 3; ++j)
    for (int i = 0; i < 3; ++i) {
      const int k = i + 3 * j;
      // forward: d(p_i)/dH(i, j) = x_j
      J[k] = (i == 0 ? x[j] / p[2] : 0.0) - (i == 2 ? u * x[j] / p[2] : 0.0);
      J[9 + k] = (i == 1 ? x[j] / p[2] : 0.0) - (i == 2 ? v * x[j] / p[2] : 0.0);
      // backward: dq/dH(i, j) = -G[:, i] q_j
      J[18 + k] = -q[j] * (G[0 + 3 * i] - a * G[2 + 3 * i]) / q[2];
      J[27 + k] = -q[j] * (G[1 + 3 * i] - b * G[2 + 3 * i]) / q[2];
    }
}

__device__ double homography_cost(const HomographyBatch& B, int p, const double* H, int lane) {
  const int64_t beg = B.offsets[p], end = B.counts ? beg + B.counts[p] : B.offsets[p + 1];
  double G[9];
  inverse3_cm(H, G);
  double cost = 0.0;
  for (int64_t o = beg + lane; o < end; o += 64) {
    double r[4];
    homography_residuals<false>(H, G, B.corr[o], r, nullptr);
    double rho1;
    cost += 0.5 * loss_eval(B.loss_type, B.loss_width, (r[0] * r[0] + r[1] * r[1]) + (r[2] * r[2] + r[3] * r[3]), &rho1);
  }
  return wsum(cost);
}

__device__ void homography_linearize(const HomographyBatch& B, int p, const double* H, const double* scale, int lane,
                                     double* A, double* g, double* cost) {
  const int64_t beg = B.offsets[p], end = B.counts ? beg + B.counts[p] : B.offsets[p + 1];
  double G[9];
  inverse3_cm(H, G);
  double acc[55];
#pragma unroll
  for (int k = 0; k < 55; ++k) acc[k] = 0.0;
  for (int64_t o = beg + lane; o < end; o += 64) {
    double r[4], J[36];
    homography_residuals<true>(H, G, B.corr[o], r, J);
    double rho1;
    const double rho = loss_eval(B.loss_type, B.loss_width, (r[0] * r[0] + r[1] * r[1]) + (r[2] * r[2] + r[3] * r[3]), &rho1);
    const double sr = sqrt(rho1);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      r[e] *= sr;
#pragma unroll
      for (int q = 0; q < 9; ++q) J[9 * e + q] *= sr * scale[q];
    }
    int k = 0;
#pragma unroll
    for (int a = 0; a < 9; ++a) {
#pragma unroll
      for (int b = 0; b <= a; ++b) acc[k++] += (J[a] * J[b] + J[9 + a] * J[9 + b]) + (J[18 + a] * J[18 + b] + J[27 + a] * J[27 + b]);
      acc[45 + a] += (J[a] * r[0] + J[9 + a] * r[1]) + (J[18 + a] * r[2] + J[27 + a] * r[3]);
    }
    acc[54] += 0.5 * rho;
  }
#pragma unroll
  for (int k = 0; k < 55; ++k) acc[k] = wsum(acc[k]);
  for (int k = 0; k < 45; ++k) A[k] = acc[k];
  for (int k = 0; k < 9; ++k) g[k] = acc[45 + k];
  *cost = acc[54];
}

__device__ bool solve9(const double* A, const double* d, const double* g, double* y) {
  double L[45];
  for (int i = 0; i < 9; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = A[tri(i, j)] + (i == j ? d[i] : 0.0);
      for (int k = 0; k < j; ++k) s -= L[tri(i, k)] * L[tri(j, k)];
      if (i == j) { if (!(s > 0.0)) return false; L[tri(i, i)] = sqrt(s); }
      else L[tri(i, j)] = s / L[tri(j, j)];
    }
  double z[9];
  for (int i = 0; i < 9; ++i) {
    double s = g[i];
    for (int k = 0; k < i; ++k) s -= L[tri(i, k)] * z[k];
    z[i] = s / L[tri(i, i)];
  }
  for (int i = 8; i >= 0; --i) {
    double s = z[i];
    for (int k = i + 1; k < 9; ++k) s -= L[tri(k, i)] * y[k];
    y[i] = s / L[tri(i, i)];
  }
  return true;
}

__global__ __launch_bounds__(256) void k_homography_lm(HomographyBatch B, TwoViewOut* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= B.num) return;
  double x[9];
  for (int q = 0; q < 9; ++q) x[q] = B.H[(size_t)p * 9 + q];
  TwoViewOut R;
  R.success = 0; R.term = THEIA_TERM_NO_CONVERGENCE; R.iters = 0; R.nsucc = 0; R.initial_cost = 0.0; R.final_cost = 0.0;
  double scale[9];
  for (int q = 0; q < 9; ++q) scale[q] = 1.0;
  double A[45], g[9], x_cost;
  homography_linearize(B, p, x, scale, lane, A, g, &x_cost);
  for (int q = 0; q < 9; ++q) scale[q] = 1.0 / (1.0 + sqrt(A[tri(q, q)]));
  double radius = 1e4, decrease_factor = 2.0;
  bool step_successful = true, need_linearize = true, first = true;
  int iter = 0, invalid_steps = 0, term = THEIA_TERM_NO_CONVERGENCE;
  double x_norm = 0.0, minimum_cost = 0.0, gmax = 0.0;
  for (int q = 0; q < 9; ++q) x_norm += x[q] * x[q];
  x_norm = sqrt(x_norm);
  while (true) {
    if (need_linearize) {
      homography_linearize(B, p, x, scale, lane, A, g, &x_cost);
      gmax = 0.0;
      for (int q = 0; q < 9; ++q) gmax = fmax(gmax, fabs(g[q] / scale[q]));
      need_linearize = false;
    }
    if (first) {
      first = false;
      R.initial_cost = x_cost;
      minimum_cost = x_cost;
      if (!isfinite(x_cost)) { term = THEIA_TERM_FAILURE; break; }
    }
    if (iter >= B.max_iterations) { term = THEIA_TERM_NO_CONVERGENCE; break; }
    if (step_successful && gmax <= B.gradient_tolerance) { term = THEIA_TERM_CONVERGENCE; break; }
    if (radius <= 1e-32) { term = THEIA_TERM_CONVERGENCE; break; }
    ++iter;
    double d[9], y[9];
    for (int q = 0; q < 9; ++q) d[q] = fmin(fmax(A[tri(q, q)], 1e-6), 1e32) / radius;
    const bool solved = solve9(A, d, g, y);
    double yg = 0.0, yAy = 0.0;
    for (int a = 0; a < 9; ++a) {
      yg += y[a] * g[a];
      double row = 0.0;
      for (int b = 0; b < 9; ++b) row += A[a >= b ? tri(a, b) : tri(b, a)] * y[b];
      yAy += y[a] * row;
    }
    const double mcc = yg - 0.5 * yAy;
    double cand[9], stepsq = 0.0, xnormsq = 0.0;
    for (int q = 0; q < 9; ++q) {
      cand[q] = x[q] - y[q] * scale[q];
      stepsq += (x[q] - cand[q]) * (x[q] - cand[q]);
      xnormsq += cand[q] * cand[q];
    }
    const bool step_valid = solved && isfinite(mcc) && isfinite(stepsq) && mcc > 0.0;
    if (!step_valid) {
      if (++invalid_steps >= 5) { term = THEIA_TERM_FAILURE; break; }
      radius /= decrease_factor; decrease_factor *= 2.0; step_successful = false;
      continue;
    }
    invalid_steps = 0;
    double cand_cost = homography_cost(B, p, cand, lane);
    if (!isfinite(cand_cost)) cand_cost = DBL_MAX;
    const double step_norm = sqrt(stepsq);
    if (step_norm <= B.parameter_tolerance * (x_norm + B.parameter_tolerance)) { term = THEIA_TERM_CONVERGENCE; break; }
    const double cost_change = x_cost - cand_cost;
    if (fabs(cost_change) <= B.function_tolerance * x_cost) { term = THEIA_TERM_CONVERGENCE; break; }
    const double rho = cost_change / mcc;
    if (rho > 1e-3) {
      for (int q = 0; q < 9; ++q) x[q] = cand[q];
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
  R.final_cost = term != THEIA_TERM_FAILURE ? minimum_cost : x_cost;
  if (lane == 0) {
    out[p] = R;
    const double h22 = x[8];   // (*homography) /= (*homography)(2, 2): the divisor is the last element in storage order
    for (int q = 0; q < 9; ++q) B.H[(size_t)p * 9 + q] = x[q] / h22;
  }
}

// ------------------------------------------------------------------ fundamental matrix refinement
// N independent OptimizeFundamentalMatrix problems (bundle_adjust_two_views.cc:248-296): F moves on the 7-dof manifold
// of fundamental_matrix_parameterization.h:15-75 -- F = U diag(1, sigma, 0) V', Plus(F, d) = (U R(d[0:3])) diag(1,
// s1/s0 + d[6], 0) (V R(d[3:6]))' from the SVD of the CURRENT F (so every accepted step also re-normalises F) -- against
// the squared Sampson distance of each correspondence as ONE residual (sampson_error.h:21-33), no loss, direct solver.
// RefineModel of the fundamental-matrix estimator (estimate_fundamental_matrix.cc:53-90) runs it for 2 iterations.
// The Jacobian of Plus at d = 0 (Ceres gets it by autodiff) is closed form in the columns u_i, v_i of U, V.
struct FundBatch {
  int num;
  const int64_t* offsets;
  const int* counts;
  const double4* corr;
  double* F;                // [num][9] row-major in/out
  int max_iterations;
  double function_tolerance, gradient_tolerance, parameter_tolerance, max_radius;
};

__device__ void fund_plus(const double* F, const double* d, double* Fo) {
  double U[9], S[3], V[9], R1[9], R2[9], Un[9], Vn[9];
  rsc::svd3(F, U, S, V);
  aa_to_rot(d, R1);
  aa_to_rot(d + 3, R2);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      Un[3 * i + j] = (U[3 * i] * R1[j] + U[3 * i + 1] * R1[3 + j]) + U[3 * i + 2] * R1[6 + j];
      Vn[3 * i + j] = (V[3 * i] * R2[j] + V[3 * i + 1] * R2[3 + j]) + V[3 * i + 2] * R2[6 + j];
    }
  const double sigma = S[1] / S[0] + d[6];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Fo[3 * i + j] = Un[3 * i] * Vn[3 * j] + sigma * Un[3 * i + 1] * Vn[3 * j + 1];
}
// PJ[k][q] = d Plus(F, d)[k] / d d[q] at d = 0, k = 3 i + j
__device__ void fund_plus_jacobian(const double* F, double* PJ) {
  double U[9], S[3], V[9];
  rsc::svd3(F, U, S, V);
  const double s0 = S[1] / S[0];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      const double* u = U + 3 * i;   // u[c] = U(i, c)
      const double* v = V + 3 * j;   // v[c] = V(j, c)
      double* o = PJ + 7 * (3 * i + j);
      o[0] = s0 * u[2] * v[1];
      o[1] = -u[2] * v[0];
      o[2] = u[1] * v[0] - s0 * u[0] * v[1];
      o[3] = s0 * u[1] * v[2];
      o[4] = -u[0] * v[2];
      o[5] = u[0] * v[1] - s0 * u[1] * v[0];
      o[6] = u[1] * v[1];
    }
}
// SampsonError::operator() and (WANT_JAC) its gradient wrt the nine entries of F (row-major)
template <bool WANT_JAC>
__device__ double sampson_residual(const double* F, const double4 c, double* J) {
  const double x1[3] = {c.x, c.y, 1.0}, x2[3] = {c.z, c.w, 1.0};
  double e[3];
  for (int i = 0; i < 3; ++i) e[i] = (F[3 * i] * x1[0] + F[3 * i + 1] * x1[1]) + F[3 * i + 2] * x1[2];
  const double N = (x2[0] * e[0] + x2[1] * e[1]) + x2[2] * e[2];
  const double d0 = (x2[0] * F[0] + x2[1] * F[3]) + x2[2] * F[6];
  const double d1 = (x2[0] * F[1] + x2[1] * F[4]) + x2[2] * F[7];
  const double D = ((d0 * d0 + d1 * d1) + e[0] * e[0]) + e[1] * e[1];
  if (WANT_JAC) {
    const double a = 2.0 * N / D, b = N * N / (D * D);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        double dD = 0.0;
        if (j == 0) dD += 2.0 * d0 * x2[i];
        if (j == 1) dD += 2.0 * d1 * x2[i];
        if (i == 0) dD += 2.0 * e[0] * x1[j];
        if (i == 1) dD += 2.0 * e[1] * x1[j];
        J[3 * i + j] = a * x2[i] * x1[j] - b * dD;
      }
  }
  return N * N / D;
}
__device__ double fund_cost(const FundBatch& B, int p, const double* F, int lane) {
  const int64_t beg = B.offsets[p], end = B.counts ? beg + B.counts[p] : B.offsets[p + 1];
  double cost = 0.0;
  for (int64_t o = beg + lane; o < end; o += 64) {
    const double r = sampson_residual<false>(F, B.corr[o], nullptr);
    cost += 0.5 * (r * r);
  }
  return wsum(cost);
}
__device__ void fund_linearize(const FundBatch& B, int p, const double* F, const double* scale, int lane,
                               double* A, double* g, double* cost) {
  const int64_t beg = B.offsets[p], end = B.counts ? beg + B.counts[p] : B.offsets[p + 1];
  double PJ[63];
  fund_plus_jacobian(F, PJ);
  double acc[36];
#pragma unroll
  for (int k = 0; k < 36; ++k) acc[k] = 0.0;
  for (int64_t o = beg + lane; o < end; o += 64) {
    double Ja[9], J[7];
    const double r = sampson_residual<true>(F, B.corr[o], Ja);
#pragma unroll
    for (int q = 0; q < 7; ++q) {
      double t = 0.0;
#pragma unroll
      for (int k = 0; k < 9; ++k) t += Ja[k] * PJ[7 * k + q];
      J[q] = t * scale[q];
    }
    int k = 0;
#pragma unroll
    for (int a = 0; a < 7; ++a) {
#pragma unroll
      for (int b = 0; b <= a; ++b) acc[k++] += J[a] * J[b];
      acc[28 + a] += J[a] * r;
    }
    acc[35] += 0.5 * (r * r);
  }
#pragma unroll
  for (int k = 0; k < 36; ++k) acc[k] = wsum(acc[k]);
  for (int k = 0; k < 28; ++k) A[k] = acc[k];
  for (int k = 0; k < 7; ++k) g[k] = acc[28 + k];
  *cost = acc[35];
}
__device__ bool solve7(const double* A, const double* d, const double* g, double* y) {
  double L[28];
  for (int i = 0; i < 7; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = A[tri(i, j)] + (i == j ? d[i] : 0.0);
      for (int k = 0; k < j; ++k) s -= L[tri(i, k)] * L[tri(j, k)];
      if (i == j) { if (!(s > 0.0)) return false; L[tri(i, i)] = sqrt(s); }
      else L[tri(i, j)] = s / L[tri(j, j)];
    }
  double z[7];
  for (int i = 0; i < 7; ++i) {
    double s = g[i];
    for (int k = 0; k < i; ++k) s -= L[tri(i, k)] * z[k];
    z[i] = s / L[tri(i, i)];
  }
  for (int i = 6; i >= 0; --i) {
    double s = z[i];
    for (int k = i + 1; k < 7; ++k) s -= L[tri(k, i)] * y[k];
    y[i] = s / L[tri(i, i)];
  }
  return true;
}

__global__ __launch_bounds__(256) void k_fundamental_lm(FundBatch B, TwoViewOut* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= B.num) return;
  double x[9];
  for (int q = 0; q < 9; ++q) x[q] = B.F[(size_t)p * 9 + q];
  TwoViewOut R;
  R.success = 0; R.term = THEIA_TERM_NO_CONVERGENCE; R.iters = 0; R.nsucc = 0; R.initial_cost = 0.0; R.final_cost = 0.0;
  double scale[7];
  for (int q = 0; q < 7; ++q) scale[q] = 1.0;
  double A[28], g[7], x_cost;
  fund_linearize(B, p, x, scale, lane, A, g, &x_cost);
  for (int q = 0; q < 7; ++q) scale[q] = 1.0 / (1.0 + sqrt(A[tri(q, q)]));
  double radius = 1e4, decrease_factor = 2.0;
  bool step_successful = true, need_linearize = true, first = true;
  int iter = 0, invalid_steps = 0, term = THEIA_TERM_NO_CONVERGENCE;
  double x_norm = 0.0, minimum_cost = 0.0, gmax = 0.0;
  for (int q = 0; q < 9; ++q) x_norm += x[q] * x[q];
  x_norm = sqrt(x_norm);
  while (true) {
    if (need_linearize) {
      fund_linearize(B, p, x, scale, lane, A, g, &x_cost);
      // gradient_max_norm = |x - Plus(x, -gradient)|_inf: Plus re-normalises F, so this is rarely small
      double ng[7], xp[9];
      for (int q = 0; q < 7; ++q) ng[q] = -(g[q] / scale[q]);
      fund_plus(x, ng, xp);
      gmax = 0.0;
      for (int q = 0; q < 9; ++q) gmax = fmax(gmax, fabs(x[q] - xp[q]));
      need_linearize = false;
    }
    if (first) {
      first = false;
      R.initial_cost = x_cost;
      minimum_cost = x_cost;
      if (!isfinite(x_cost)) { term = THEIA_TERM_FAILURE; break; }
    }
    if (iter >= B.max_iterations) { term = THEIA_TERM_NO_CONVERGENCE; break; }
    if (step_successful && gmax <= B.gradient_tolerance) { term = THEIA_TERM_CONVERGENCE; break; }
    if (radius <= 1e-32) { term = THEIA_TERM_CONVERGENCE; break; }
    ++iter;
    double d[7], y[7];
    for (int q = 0; q < 7; ++q) d[q] = fmin(fmax(A[tri(q, q)], 1e-6), 1e32) / radius;
    const bool solved = solve7(A, d, g, y);
    double yg = 0.0, yAy = 0.0;
    for (int a = 0; a < 7; ++a) {
      yg += y[a] * g[a];
      double row = 0.0;
      for (int b = 0; b < 7; ++b) row += A[a >= b ? tri(a, b) : tri(b, a)] * y[b];
      yAy += y[a] * row;
    }
    const double mcc = yg - 0.5 * yAy;
    double cand[9], dl[7], stepsq = 0.0, xnormsq = 0.0;
    for (int q = 0; q < 7; ++q) dl[q] = -y[q] * scale[q];
    fund_plus(x, dl, cand);
    for (int q = 0; q < 9; ++q) {
      stepsq += (x[q] - cand[q]) * (x[q] - cand[q]);
      xnormsq += cand[q] * cand[q];
    }
    const bool step_valid = solved && isfinite(mcc) && isfinite(stepsq) && mcc > 0.0;
    if (!step_valid) {
      if (++invalid_steps >= 5) { term = THEIA_TERM_FAILURE; break; }
      radius /= decrease_factor; decrease_factor *= 2.0; step_successful = false;
      continue;
    }
    invalid_steps = 0;
    double cand_cost = fund_cost(B, p, cand, lane);
    if (!isfinite(cand_cost)) cand_cost = DBL_MAX;
    const double step_norm = sqrt(stepsq);
    if (step_norm <= B.parameter_tolerance * (x_norm + B.parameter_tolerance)) { term = THEIA_TERM_CONVERGENCE; break; }
    const double cost_change = x_cost - cand_cost;
    if (fabs(cost_change) <= B.function_tolerance * x_cost) { term = THEIA_TERM_CONVERGENCE; break; }
    const double rho = cost_change / mcc;
    if (rho > 1e-3) {
      for (int q = 0; q < 9; ++q) x[q] = cand[q];
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
  R.final_cost = term != THEIA_TERM_FAILURE ? minimum_cost : x_cost;
  if (lane == 0) {
    out[p] = R;
    for (int q = 0; q < 9; ++q) B.F[(size_t)p * 9 + q] = x[q];
  }
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

// device-resident variant for callers inside the library (LO-RANSAC); d_out = views_batch_out_bytes() per problem
int twoview_batch_device(int num, const int64_t* d_offsets, const int* d_counts, const double* d_corr, double* d_pose,
                         const theia_ba_options* o, int cgnr, void* d_out, hipStream_t st) {
  static_assert(sizeof(TwoViewOut) == 32, "layout shared with ba_batch.hip ViewOut");
  TwoViewBatch B;
  B.num = num; B.offsets = d_offsets; B.counts = d_counts; B.corr = reinterpret_cast<const double4*>(d_corr); B.pose = d_pose;
  B.loss_type = o->loss_function_type; B.loss_width = o->robust_loss_width; B.max_iterations = o->max_num_iterations;
  B.function_tolerance = o->function_tolerance; B.gradient_tolerance = o->gradient_tolerance;
  B.parameter_tolerance = o->parameter_tolerance; B.max_radius = o->max_trust_region_radius;
  B.cgnr = cgnr;
  k_twoview_lm<<<(num + 3) / 4, 256, 0, st>>>(B, static_cast<TwoViewOut*>(d_out));
  return 0;
}

// device-resident OptimizeFundamentalMatrix batch; d_F = [num][9] row-major in/out
int fundamental_batch_device(int num, const int64_t* d_offsets, const int* d_counts, const double* d_corr, double* d_F,
                             const theia_ba_options* o, void* d_out, hipStream_t st) {
  FundBatch B;
  B.num = num; B.offsets = d_offsets; B.counts = d_counts; B.corr = reinterpret_cast<const double4*>(d_corr); B.F = d_F;
  B.max_iterations = o->max_num_iterations;
  B.function_tolerance = o->function_tolerance; B.gradient_tolerance = o->gradient_tolerance;
  B.parameter_tolerance = o->parameter_tolerance; B.max_radius = o->max_trust_region_radius;
  k_fundamental_lm<<<(num + 3) / 4, 256, 0, st>>>(B, static_cast<TwoViewOut*>(d_out));
  return 0;
}

// device-resident OptimizeHomography batch; d_H = [num][9] column-major in/out
int homography_batch_device(int num, const int64_t* d_offsets, const int* d_counts, const double* d_corr, double* d_H,
                            const theia_ba_options* o, void* d_out, hipStream_t st) {
  HomographyBatch B;
  B.num = num; B.offsets = d_offsets; B.counts = d_counts; B.corr = reinterpret_cast<const double4*>(d_corr); B.H = d_H;
  B.loss_type = o->loss_function_type; B.loss_width = o->robust_loss_width; B.max_iterations = o->max_num_iterations;
  B.function_tolerance = o->function_tolerance; B.gradient_tolerance = o->gradient_tolerance;
  B.parameter_tolerance = o->parameter_tolerance; B.max_radius = o->max_trust_region_radius;
  k_homography_lm<<<(num + 3) / 4, 256, 0, st>>>(B, static_cast<TwoViewOut*>(d_out));
  return 0;
}

}  // namespace thip

using namespace thip;

extern "C" int theia_hip_ba_two_views_angular_batch(const theia_ba_two_view_batch* b, const theia_ba_options* o,
                                                    theia_ba_summary* summaries) {
  if (!b || !o) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "null batch/options");
  const int num = b->num_problems;
  if (num < 0) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "negative num_problems");
  if (num == 0) return 0;
  if (!b->offsets || !b->rotation_position || !summaries) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "null array in batch");
  if (b->offsets[0] != 0) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "offsets[0] must be 0");
  for (int i = 0; i < num; ++i)
    if (b->offsets[i + 1] < b->offsets[i]) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "offsets must be non-decreasing");
  const int64_t total = b->offsets[num];
  if (total > 0 && !b->correspondences) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "null correspondences");
  if (o->loss_function_type < 0 || o->loss_function_type > THEIA_LOSS_TRUNCATED)
    return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "unknown loss function type");
  if (o->max_num_iterations < 0) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "negative max_num_iterations");
  if (b->linear_solver != THEIA_TWO_VIEW_EXACT && b->linear_solver != THEIA_TWO_VIEW_CGNR)
    return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "unknown linear_solver %d", b->linear_solver);
  int rc = thip::ensure_device();
  if (rc) return rc;
  Dev<int64_t> d_off; Dev<double> d_corr, d_pose; Dev<char> d_out;
  if ((rc = d_off.up(b->offsets, num + 1)) || (rc = d_corr.up(b->correspondences, 4 * total)) ||
      (rc = d_pose.up(b->rotation_position, 6 * (size_t)num)) || (rc = d_out.alloc(sizeof(TwoViewOut) * num)))
    return rc;
  const double t0 = now_s();
  twoview_batch_device(num, d_off.p, nullptr, d_corr.p, d_pose.p, o, b->linear_solver == THEIA_TWO_VIEW_CGNR, d_out.p, nullptr);
  std::vector<TwoViewOut> h_out(num);
  HIP_TRY(hipMemcpy(h_out.data(), d_out.p, sizeof(TwoViewOut) * num, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(b->rotation_position, d_pose.p, sizeof(double) * 6 * num, hipMemcpyDeviceToHost));
  const double dt = now_s() - t0;
  for (int i = 0; i < num; ++i) {
    theia_ba_summary& S = summaries[i];
    S.trace_size = 0;
    S.success = h_out[i].success; S.termination_type = h_out[i].term; S.num_iterations = h_out[i].iters;
    S.num_successful_steps = h_out[i].nsucc; S.initial_cost = h_out[i].initial_cost; S.final_cost = h_out[i].final_cost;
    S.setup_time_in_seconds = 0.0; S.solve_time_in_seconds = dt / num;
    S.time_linearize = S.time_solve_reduced = S.time_backsub = S.time_kernel_linearize = 0.0;
    S.num_linearize_launches = 0;
  }
  return 0;
}

extern "C" int theia_hip_optimize_homography_batch(int32_t num_problems, const int64_t* offsets, const double* correspondences,
                                                   double* homographies, const theia_ba_options* o, theia_ba_summary* summaries) {
  if (!o) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "null options");
  const int num = num_problems;
  if (num < 0) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "negative num_problems");
  if (num == 0) return 0;
  if (!offsets || !homographies || !summaries) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "null array");
  if (offsets[0] != 0) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "offsets[0] must be 0");
  for (int i = 0; i < num; ++i)
    if (offsets[i + 1] < offsets[i]) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "offsets must be non-decreasing");
  const int64_t total = offsets[num];
  if (total > 0 && !correspondences) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "null correspondences");
  if (o->loss_function_type < 0 || o->loss_function_type > THEIA_LOSS_TRUNCATED)
    return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "unknown loss function type");
  if (o->max_num_iterations < 0) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "negative max_num_iterations");
  int rc = thip::ensure_device();
  if (rc) return rc;
  std::vector<double> hcm((size_t)num * 9);   // row-major at the boundary, Eigen's column-major storage order inside
  for (int p = 0; p < num; ++p)
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) hcm[(size_t)p * 9 + i + 3 * j] = homographies[(size_t)p * 9 + 3 * i + j];
  Dev<int64_t> d_off; Dev<double> d_corr, d_H; Dev<char> d_out;
  if ((rc = d_off.up(offsets, num + 1)) || (rc = d_corr.up(correspondences, 4 * total)) || (rc = d_H.up(hcm.data(), hcm.size())) ||
      (rc = d_out.alloc(sizeof(TwoViewOut) * num)))
    return rc;
  const double t0 = now_s();
  homography_batch_device(num, d_off.p, nullptr, d_corr.p, d_H.p, o, d_out.p, nullptr);
  std::vector<TwoViewOut> h_out(num);
  HIP_TRY(hipMemcpy(h_out.data(), d_out.p, sizeof(TwoViewOut) * num, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(hcm.data(), d_H.p, sizeof(double) * hcm.size(), hipMemcpyDeviceToHost));
  const double dt = now_s() - t0;
  for (int p = 0; p < num; ++p)
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) homographies[(size_t)p * 9 + 3 * i + j] = hcm[(size_t)p * 9 + i + 3 * j];
  for (int i = 0; i < num; ++i) {
    theia_ba_summary& S = summaries[i];
    S.trace_size = 0;
    S.success = h_out[i].success; S.termination_type = h_out[i].term; S.num_iterations = h_out[i].iters;
    S.num_successful_steps = h_out[i].nsucc; S.initial_cost = h_out[i].initial_cost; S.final_cost = h_out[i].final_cost;
    S.setup_time_in_seconds = 0.0; S.solve_time_in_seconds = dt / num;
    S.time_linearize = S.time_solve_reduced = S.time_backsub = S.time_kernel_linearize = 0.0;
    S.num_linearize_launches = 0;
  }
  return 0;
}

extern "C" int theia_hip_optimize_fundamental_matrix_batch(int32_t num_problems, const int64_t* offsets, const double* correspondences,
                                                           double* fundamental_matrices, const theia_ba_options* o,
                                                           theia_ba_summary* summaries) {
  if (!o) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "null options");
  const int num = num_problems;
  if (num < 0) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "negative num_problems");
  if (num == 0) return 0;
  if (!offsets || !fundamental_matrices || !summaries) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "null array");
  if (offsets[0] != 0) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "offsets[0] must be 0");
  for (int i = 0; i < num; ++i)
    if (offsets[i + 1] < offsets[i]) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "offsets must be non-decreasing");
  const int64_t total = offsets[num];
  if (total > 0 && !correspondences) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "null correspondences");
  if (o->loss_function_type != THEIA_LOSS_TRIVIAL)
    return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "OptimizeFundamentalMatrix adds its residuals without a loss function");
  if (o->max_num_iterations < 0) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "negative max_num_iterations");
  int rc = thip::ensure_device();
  if (rc) return rc;
  Dev<int64_t> d_off; Dev<double> d_corr, d_F; Dev<char> d_out;
  if ((rc = d_off.up(offsets, num + 1)) || (rc = d_corr.up(correspondences, 4 * total)) ||
      (rc = d_F.up(fundamental_matrices, 9 * (size_t)num)) || (rc = d_out.alloc(sizeof(TwoViewOut) * num)))
    return rc;
  const double t0 = now_s();
  fundamental_batch_device(num, d_off.p, nullptr, d_corr.p, d_F.p, o, d_out.p, nullptr);
  std::vector<TwoViewOut> h_out(num);
  HIP_TRY(hipMemcpy(h_out.data(), d_out.p, sizeof(TwoViewOut) * num, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(fundamental_matrices, d_F.p, sizeof(double) * 9 * num, hipMemcpyDeviceToHost));
  const double dt = now_s() - t0;
  for (int i = 0; i < num; ++i) {
    theia_ba_summary& S = summaries[i];
    S.trace_size = 0;
    S.success = h_out[i].success; S.termination_type = h_out[i].term; S.num_iterations = h_out[i].iters;
    S.num_successful_steps = h_out[i].nsucc; S.initial_cost = h_out[i].initial_cost; S.final_cost = h_out[i].final_cost;
    S.setup_time_in_seconds = 0.0; S.solve_time_in_seconds = dt / num;
    S.time_linearize = S.time_solve_reduced = S.time_backsub = S.time_kernel_linearize = 0.0;
    S.num_linearize_launches = 0;
  }
  return 0;
}
