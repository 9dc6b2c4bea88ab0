// twoview_ba.hip -- batches of INDEPENDENT two-view bundle adjustments, one whole Levenberg-Marquardt solve per
// wavefront, no host round trips.
//
// Replaces N calls of BundleAdjustTwoViews (bundle_adjust_two_views.cc:110-185, the refinement step of
// TwoViewMatchGeometricVerification::VerifyMatches, two_view_match_geometric_verification.cc:259-289): camera 1 is held
// constant, camera 2 moves (6), the focal length of each camera moves unless it is held constant, every triangulated
// point moves as an XYZW vector without a manifold; reprojection residuals of both views, trivial loss.  Through the
// general solver (theia_hip_ba_solve per pair) such a call costs ~3 ms of launch latency for a few hundred residuals;
// the verification stage of a match graph makes thousands of them.
//
// Same arithmetic as ba_solver.hip on the same flat problem (Ceres 2.2 TrustRegionMinimizer + LevenbergMarquardtStrategy
// restated: Jacobi scaling of every column from the norms at the initial point, D = clamp(colnorm^2) / radius, Schur
// elimination of the 4 x 4 point blocks, an 8 x 8 reduced system [camera 2 (6) | focal 1 | focal 2], model-cost change
// from the linearised residuals, step acceptance rho > 1e-3, focal lengths projected onto >= 1).  Lane = point
// (stride 64); the reduced system is wave-reduced; every lane keeps the same scalar state.
#include "ba_device.h"
#include "wave_reduce.h"
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

// the XOR butterfly over the 64 lanes, on permlane swaps + DPP (wave_reduce.h: the bits of the __shfl_xor loop)
__device__ __forceinline__ double wsum(double v) { return wave_sum_butterfly(v); }
__device__ __forceinline__ double wmax(double v) { return wave_max_butterfly(v); }

struct TvBatch {
  int num;
  const int64_t* offsets;
  const double4* corr;     // (x1, y1, x2, y2) pixels
  double* cam;             // [num][2][6]
  double* intr;            // [num][2][THEIA_MAX_INTRINSICS]
  const int* model;        // [num][2]
  const uint8_t* kconst;   // [num][2] focal length held constant
  double4* X;              // [total] points, in/out
  double4* Xc;             // [total] candidate points (scratch)
  double4* sp;             // [total] Jacobi scaling of the point columns (scratch)
  int max_iterations;
  double function_tolerance, gradient_tolerance, parameter_tolerance, max_radius;
};
struct TvOut {
  int success, term, iters, nsucc;
  double initial_cost, final_cost;
};

constexpr int NC = 8;                               // reduced columns: camera 2 (6), focal 1, focal 2
__device__ __forceinline__ int intrinsics_count(int model) {   // kIntrinsicsSize of the eight models (the whole block counts in |x|)
  const int K[8] = {7, 10, 9, 5, 5, 7, 7, 7};
  return (model >= 0 && model < 8) ? K[model] : 0;
}
__device__ __forceinline__ int tri(int a, int b) { return a * (a + 1) / 2 + b; }   // a >= b

// Everything one point contributes at a linearisation point.
struct PointLin {
  double r[4];          // residuals (u1, v1, u2, v2)
  double E[4][4];       // d r / d X, scaled
  double F[4][NC];      // d r / d [cam2 | f1 | f2], scaled (rows 0-1 only touch f1, rows 2-3 cam2 and f2)
  bool valid;
};

// residuals (and scaled Jacobians) of point X seen by the two cameras
template <bool JAC>
__device__ __attribute__((noinline)) void tv_point(const int model[2], const double* ext1, const double* ext2, const double* k1, const double* k2,
                                                   const double X[4], const double4& c, const double sc[NC], const double sp[4],
                                                   PointLin& L) {
  ObsLinK o1, o2;
  observe<JAC, JAC, ObsLinK>(model[0], ext1, k1, X, c.x, c.y, 1.0, 1.0, o1);
  observe<JAC, JAC, ObsLinK>(model[1], ext2, k2, X, c.z, c.w, 1.0, 1.0, o2);
  L.valid = o1.valid && o2.valid;
  L.r[0] = o1.r[0]; L.r[1] = o1.r[1]; L.r[2] = o2.r[0]; L.r[3] = o2.r[1];
  if (JAC) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int q = 0; q < 4; ++q) { L.E[a][q] = o1.Jx[4 * a + q] * sp[q]; L.E[2 + a][q] = o2.Jx[4 * a + q] * sp[q]; }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int q = 0; q < NC; ++q) L.F[a][q] = 0.0;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
      for (int q = 0; q < 6; ++q) L.F[2 + a][q] = o2.Jc[6 * a + q] * sc[q];
      L.F[a][6] = o1.Jk[THEIA_MAX_INTRINSICS * a] * sc[6];        // focal length = intrinsics slot 0
      L.F[2 + a][7] = o2.Jk[THEIA_MAX_INTRINSICS * a] * sc[7];
    }
  }
}

// 4 x 4 SPD inverse through Cholesky (packed lower in, packed lower out); false if not positive definite
__device__ bool inv4(const double* V, double* Vi) {
  double Lm[4][4], Li[4][4];
  bool ok = true;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      double s = V[tri(i, j)];
#pragma unroll
      for (int k = 0; k < j; ++k) s -= Lm[i][k] * Lm[j][k];
      if (i == j) { if (!(s > 0.0)) ok = false; Lm[i][i] = sqrt(s); }
      else Lm[i][j] = s / Lm[j][j];
    }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    Li[i][i] = 1.0 / Lm[i][i];
#pragma unroll
    for (int j = 0; j < i; ++j) {
      double s = 0.0;
#pragma unroll
      for (int k = j; k < i; ++k) s -= Lm[i][k] * Li[k][j];
      Li[i][j] = s / Lm[i][i];
    }
  }
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b <= a; ++b) {
      double s = 0.0;
#pragma unroll
      for (int k = a; k < 4; ++k) s += Li[k][a] * Li[k][b];
      Vi[tri(a, b)] = s;
    }
  return ok;
}
__device__ __forceinline__ double sym4(const double* V, int a, int b) { return a >= b ? V[tri(a, b)] : V[tri(b, a)]; }

// (H + diag d) y = g over the free columns (frozen columns: y = 0); false if not positive definite
__device__ bool solve8(const double* H, const double* d, const double* g, unsigned frozen, double* y) {
  double L[NC][NC];
  bool ok = true;
  for (int i = 0; i < NC; ++i)
    for (int j = 0; j <= i; ++j) {
      double s;
      if (((frozen >> i) & 1u) || ((frozen >> j) & 1u)) s = (i == j) ? 1.0 : 0.0;
      else s = H[tri(i, j)] + (i == j ? d[i] : 0.0);
      for (int k = 0; k < j; ++k) s -= L[i][k] * L[j][k];
      if (i == j) { if (!(s > 0.0)) ok = false; L[i][i] = sqrt(s); }
      else L[i][j] = s / L[j][j];
    }
  double z[NC];
  for (int i = 0; i < NC; ++i) {
    double s = ((frozen >> i) & 1u) ? 0.0 : g[i];
    for (int k = 0; k < i; ++k) s -= L[i][k] * z[k];
    z[i] = s / L[i][i];
  }
  for (int i = NC - 1; i >= 0; --i) {
    double s = z[i];
    for (int k = i + 1; k < NC; ++k) s -= L[k][i] * y[k];
    y[i] = s / L[i][i];
  }
  return ok;
}

struct TvState {
  double ext1[6], ext2[6], k1[THEIA_MAX_INTRINSICS], k2[THEIA_MAX_INTRINSICS];
};

// point-side normal equations of one point: V (packed), gp, W = E^T F (4 x 8), H += F^T F, rc += F^T r
struct PointNe {
  double V[10], gp[4], W[4][NC];
};
__device__ void point_ne(const PointLin& L, PointNe& N) {
#pragma unroll
  for (int a = 0; a < 4; ++a) {
#pragma unroll
    for (int b = 0; b <= a; ++b) {
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < 4; ++k) s += L.E[k][a] * L.E[k][b];
      N.V[tri(a, b)] = s;
    }
    double g = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) g += L.E[k][a] * L.r[k];
    N.gp[a] = g;
#pragma unroll
    for (int q = 0; q < NC; ++q) {
      double w = 0.0;
#pragma unroll
      for (int k = 0; k < 4; ++k) w += L.E[k][a] * L.F[k][q];
      N.W[a][q] = w;
    }
  }
}

__global__ __launch_bounds__(64) void k_two_view_ba(TvBatch B, TvOut* __restrict__ out) {
  const int lane = threadIdx.x;
  const int p = blockIdx.x;
  if (p >= B.num) return;
  const int64_t o0 = B.offsets[p];
  const int n = (int)(B.offsets[p + 1] - o0);
  int model[2] = {B.model[2 * p], B.model[2 * p + 1]};
  TvState S;
  for (int q = 0; q < 6; ++q) { S.ext1[q] = B.cam[(size_t)p * 12 + q]; S.ext2[q] = B.cam[(size_t)p * 12 + 6 + q]; }
  for (int q = 0; q < THEIA_MAX_INTRINSICS; ++q) {
    S.k1[q] = B.intr[(size_t)p * 2 * THEIA_MAX_INTRINSICS + q];
    S.k2[q] = B.intr[(size_t)p * 2 * THEIA_MAX_INTRINSICS + THEIA_MAX_INTRINSICS + q];
  }
  const int nk1 = intrinsics_count(model[0]), nk2 = intrinsics_count(model[1]);
  unsigned frozen = 0u;
  if (B.kconst[2 * p]) frozen |= 1u << 6;
  if (B.kconst[2 * p + 1]) frozen |= 1u << 7;
  TvOut R;
  R.success = 0; R.term = THEIA_TERM_NO_CONVERGENCE; R.iters = 0; R.nsucc = 0; R.initial_cost = 0.0; R.final_cost = 0.0;
  double sc[NC];
  const double one4[4] = {1.0, 1.0, 1.0, 1.0};

  // ---- Jacobi scaling from the column norms at the initial point (once per solve)
  {
    double ones[NC];
    for (int q = 0; q < NC; ++q) ones[q] = 1.0;
    double cn[NC];
    for (int q = 0; q < NC; ++q) cn[q] = 0.0;
    for (int i = lane; i < n; i += 64) {
      const double4 Xv = B.X[o0 + i];
      const double X[4] = {Xv.x, Xv.y, Xv.z, Xv.w};
      PointLin L;
      tv_point<true>(model, S.ext1, S.ext2, S.k1, S.k2, X, B.corr[o0 + i], ones, one4, L);
      double pn[4] = {0.0, 0.0, 0.0, 0.0};
      for (int k = 0; k < 4; ++k) {
        for (int q = 0; q < 4; ++q) pn[q] += L.E[k][q] * L.E[k][q];
        for (int q = 0; q < NC; ++q) cn[q] += L.F[k][q] * L.F[k][q];
      }
      B.sp[o0 + i] = make_double4(1.0 / (1.0 + sqrt(pn[0])), 1.0 / (1.0 + sqrt(pn[1])), 1.0 / (1.0 + sqrt(pn[2])), 1.0 / (1.0 + sqrt(pn[3])));
    }
    for (int q = 0; q < NC; ++q) { cn[q] = wsum(cn[q]); sc[q] = ((frozen >> q) & 1u) ? 0.0 : 1.0 / (1.0 + sqrt(cn[q])); }
  }

  double H[36], g[NC], x_cost = 0.0, gmax = 0.0;
  double radius = 1e4, decrease_factor = 2.0;
  bool step_successful = true, need_linearize = true, first = true;
  int iter = 0, invalid_steps = 0, term = THEIA_TERM_NO_CONVERGENCE;
  double minimum_cost = 0.0, invalid = 0.0;

  // |x| over the variable blocks: camera 2, the intrinsics blocks with a free focal length (whole block), the points
  auto x_norm_sq = [&](const TvState& T, const double4* Xs) {
    double s = 0.0;
    for (int q = 0; q < 6; ++q) s += T.ext2[q] * T.ext2[q];
    if (!((frozen >> 6) & 1u)) for (int q = 0; q < nk1; ++q) s += T.k1[q] * T.k1[q];
    if (!((frozen >> 7) & 1u)) for (int q = 0; q < nk2; ++q) s += T.k2[q] * T.k2[q];
    double ps = 0.0;
    for (int i = lane; i < n; i += 64) { const double4 v = Xs[o0 + i]; ps += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w; }
    return s + wsum(ps);
  };
  double x_norm = sqrt(x_norm_sq(S, B.X));

  while (true) {
    if (need_linearize) {
      // reduced system H (8 x 8, FtF - Wt Vinv W without the LM diagonals of the camera side), g, cost, gradient max
      double hl[36], gl[NC], cost = 0.0, gm = 0.0, inv = 0.0;
      for (int k = 0; k < 36; ++k) hl[k] = 0.0;
      for (int q = 0; q < NC; ++q) gl[q] = 0.0;
      // the point blocks depend on the radius: they are rebuilt in the step loop below; here cost / gradient only
      for (int i = lane; i < n; i += 64) {
        const double4 Xv = B.X[o0 + i], spv = B.sp[o0 + i];
        const double X[4] = {Xv.x, Xv.y, Xv.z, Xv.w}, sp4[4] = {spv.x, spv.y, spv.z, spv.w};
        PointLin L;
        tv_point<true>(model, S.ext1, S.ext2, S.k1, S.k2, X, B.corr[o0 + i], sc, sp4, L);
        if (!L.valid) inv += 1.0;
        for (int k = 0; k < 4; ++k) {
          cost += 0.5 * L.r[k] * L.r[k];
          for (int q = 0; q < NC; ++q) gl[q] += L.F[k][q] * L.r[k];
        }
        for (int q = 0; q < 4; ++q) {
          double gp = 0.0;
          for (int k = 0; k < 4; ++k) gp += L.E[k][q] * L.r[k];
          gm = fmax(gm, fabs(gp / sp4[q]));
        }
      }
      x_cost = wsum(cost); invalid = wsum(inv);
      gm = wmax(gm);
      for (int q = 0; q < NC; ++q) { g[q] = wsum(gl[q]); if (!((frozen >> q) & 1u)) gm = fmax(gm, fabs(g[q] / sc[q])); }
      gmax = gm;
      (void)hl;
      need_linearize = false;
    }
    if (first) {
      first = false;
      R.initial_cost = x_cost; minimum_cost = x_cost;
      if (invalid > 0.0 || !isfinite(x_cost)) { term = THEIA_TERM_FAILURE; R.final_cost = x_cost; break; }
    }
    if (iter >= B.max_iterations) { term = THEIA_TERM_NO_CONVERGENCE; break; }
    if (step_successful && gmax <= B.gradient_tolerance) { term = THEIA_TERM_CONVERGENCE; break; }
    if (radius <= 1e-32) { term = THEIA_TERM_CONVERGENCE; break; }
    ++iter;
    // ---- reduced camera system at this radius
    double hl[36], rl[NC], cd[NC], npd = 0.0;
    for (int k = 0; k < 36; ++k) hl[k] = 0.0;
    for (int q = 0; q < NC; ++q) { rl[q] = 0.0; cd[q] = 0.0; }
    for (int i = lane; i < n; i += 64) {
      const double4 Xv = B.X[o0 + i], spv = B.sp[o0 + i];
      const double X[4] = {Xv.x, Xv.y, Xv.z, Xv.w}, sp4[4] = {spv.x, spv.y, spv.z, spv.w};
      PointLin L;
      tv_point<true>(model, S.ext1, S.ext2, S.k1, S.k2, X, B.corr[o0 + i], sc, sp4, L);
      PointNe N;
      point_ne(L, N);
      double Vd[10], Vi[10];
      for (int k = 0; k < 10; ++k) Vd[k] = N.V[k];
      for (int a = 0; a < 4; ++a) Vd[tri(a, a)] += fmin(fmax(N.V[tri(a, a)], 1e-6), 1e32) / radius;
      if (!inv4(Vd, Vi)) { npd += 1.0; for (int k = 0; k < 10; ++k) Vi[k] = 0.0; }
      double T[4][NC];   // Vinv W
      for (int a = 0; a < 4; ++a)
        for (int q = 0; q < NC; ++q) {
          double s = 0.0;
          for (int b = 0; b < 4; ++b) s += sym4(Vi, a, b) * N.W[b][q];
          T[a][q] = s;
        }
      double tg[4];      // Vinv gp
      for (int a = 0; a < 4; ++a) { double s = 0.0; for (int b = 0; b < 4; ++b) s += sym4(Vi, a, b) * N.gp[b]; tg[a] = s; }
      for (int a = 0; a < NC; ++a) {
        for (int b = 0; b <= a; ++b) {
          double ff = 0.0, ww = 0.0;
          for (int k = 0; k < 4; ++k) { ff += L.F[k][a] * L.F[k][b]; ww += N.W[k][a] * T[k][b]; }
          hl[tri(a, b)] += ff - ww;
          if (a == b) cd[a] += ff;
        }
        double fr = 0.0, wg = 0.0;
        for (int k = 0; k < 4; ++k) { fr += L.F[k][a] * L.r[k]; wg += N.W[k][a] * tg[k]; }
        rl[a] += fr - wg;
      }
    }
    double rhs[NC], d[NC];
    for (int k = 0; k < 36; ++k) H[k] = wsum(hl[k]);
    for (int q = 0; q < NC; ++q) { rhs[q] = wsum(rl[q]); d[q] = fmin(fmax(wsum(cd[q]), 1e-6), 1e32) / radius; }
    npd = wsum(npd);
    double y[NC];
    const bool pd = solve8(H, d, rhs, frozen, y) && npd == 0.0;
    // ---- back-substitution, candidate, model cost change
    TvState C = S;
    double stepsq = 0.0;
    for (int q = 0; q < 6; ++q) { C.ext2[q] = S.ext2[q] - y[q] * sc[q]; stepsq += (S.ext2[q] - C.ext2[q]) * (S.ext2[q] - C.ext2[q]); }
    if (!((frozen >> 6) & 1u)) { C.k1[0] = fmax(1.0, S.k1[0] - y[6] * sc[6]); stepsq += (S.k1[0] - C.k1[0]) * (S.k1[0] - C.k1[0]); }
    if (!((frozen >> 7) & 1u)) { C.k2[0] = fmax(1.0, S.k2[0] - y[7] * sc[7]); stepsq += (S.k2[0] - C.k2[0]) * (S.k2[0] - C.k2[0]); }
    double mcc = 0.0, pstep = 0.0;
    for (int i = lane; i < n; i += 64) {
      const double4 Xv = B.X[o0 + i], spv = B.sp[o0 + i];
      const double X[4] = {Xv.x, Xv.y, Xv.z, Xv.w}, sp4[4] = {spv.x, spv.y, spv.z, spv.w};
      PointLin L;
      tv_point<true>(model, S.ext1, S.ext2, S.k1, S.k2, X, B.corr[o0 + i], sc, sp4, L);
      PointNe N;
      point_ne(L, N);
      double Vd[10], Vi[10];
      for (int k = 0; k < 10; ++k) Vd[k] = N.V[k];
      for (int a = 0; a < 4; ++a) Vd[tri(a, a)] += fmin(fmax(N.V[tri(a, a)], 1e-6), 1e32) / radius;
      if (!inv4(Vd, Vi)) for (int k = 0; k < 10; ++k) Vi[k] = 0.0;
      double t[4], yp[4];
      for (int a = 0; a < 4; ++a) { double s = N.gp[a]; for (int q = 0; q < NC; ++q) s -= N.W[a][q] * y[q]; t[a] = s; }
      for (int a = 0; a < 4; ++a) { double s = 0.0; for (int b = 0; b < 4; ++b) s += sym4(Vi, a, b) * t[b]; yp[a] = s; }
      for (int k = 0; k < 4; ++k) {   // model residual of the step -y
        double m = 0.0;
        for (int q = 0; q < NC; ++q) m -= L.F[k][q] * y[q];
        for (int q = 0; q < 4; ++q) m -= L.E[k][q] * yp[q];
        mcc -= m * (L.r[k] + m / 2.0);
      }
      double Xn[4];
      for (int q = 0; q < 4; ++q) { Xn[q] = X[q] - yp[q] * sp4[q]; pstep += (X[q] - Xn[q]) * (X[q] - Xn[q]); }
      B.Xc[o0 + i] = make_double4(Xn[0], Xn[1], Xn[2], Xn[3]);
    }
    mcc = wsum(mcc); stepsq += wsum(pstep);
    const bool step_valid = pd && isfinite(mcc) && isfinite(stepsq) && mcc > 0.0;
    if (!step_valid) {
      if (++invalid_steps >= 5) { term = THEIA_TERM_FAILURE; break; }
      radius /= decrease_factor; decrease_factor *= 2.0; step_successful = false;
      continue;
    }
    invalid_steps = 0;
    // ---- cost at the candidate
    double cc = 0.0, ci = 0.0;
    for (int i = lane; i < n; i += 64) {
      const double4 Xv = B.Xc[o0 + i];
      const double X[4] = {Xv.x, Xv.y, Xv.z, Xv.w};
      PointLin L;
      tv_point<false>(model, C.ext1, C.ext2, C.k1, C.k2, X, B.corr[o0 + i], sc, one4, L);
      if (!L.valid) ci += 1.0;
      for (int k = 0; k < 4; ++k) cc += 0.5 * L.r[k] * L.r[k];
    }
    double cand_cost = wsum(cc);
    if (wsum(ci) > 0.0 || !isfinite(cand_cost)) cand_cost = DBL_MAX;
    const double step_norm = sqrt(stepsq);
    if (step_norm <= B.parameter_tolerance * (x_norm + B.parameter_tolerance)) { term = THEIA_TERM_CONVERGENCE; break; }
    const double cost_change = x_cost - cand_cost;
    if (fabs(cost_change) <= B.function_tolerance * x_cost) { term = THEIA_TERM_CONVERGENCE; break; }
    const double rho = cost_change / mcc;
    if (rho > 1e-3) {
      S = C;
      for (int i = lane; i < n; i += 64) B.X[o0 + i] = B.Xc[o0 + i];
      x_norm = sqrt(x_norm_sq(S, B.X));
      const double tt = 2.0 * rho - 1.0;
      radius = radius / fmax(1.0 / 3.0, 1.0 - tt * tt * tt);
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
    for (int q = 0; q < 6; ++q) B.cam[(size_t)p * 12 + 6 + q] = S.ext2[q];
    B.intr[(size_t)p * 2 * THEIA_MAX_INTRINSICS] = S.k1[0];
    B.intr[(size_t)p * 2 * THEIA_MAX_INTRINSICS + THEIA_MAX_INTRINSICS] = S.k2[0];
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
}  // namespace thip

using namespace thip;

extern "C" int theia_hip_ba_two_views_batch(const theia_ba_two_view_full_batch* b, const theia_ba_options* o, theia_ba_summary* summaries) {
  if (!b || !o) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "null batch/options");
  const int num = b->num_problems;
  if (num < 0) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "negative num_problems");
  if (num == 0) return 0;
  if (!b->offsets || !b->cam_ext || !b->intrinsics || !b->model || !b->const_intrinsics || !summaries)
    return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "null array in batch");
  if (b->offsets[0] != 0) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "offsets[0] must be 0");
  for (int i = 0; i < num; ++i) {
    if (b->offsets[i + 1] < b->offsets[i]) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "offsets must be non-decreasing");
    for (int k = 0; k < 2; ++k)
      if (b->model[2 * i + k] < THEIA_CAM_PINHOLE || b->model[2 * i + k] > THEIA_CAM_ORTHOGRAPHIC)
        return set_error(THEIA_HIP_ERR_UNSUPPORTED, "camera model %d of pair %d has no HIP kernel", b->model[2 * i + k], i);
  }
  const int64_t total = b->offsets[num];
  if (total > 0 && (!b->correspondences || !b->points)) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "null correspondence / point arrays");
  if (o->loss_function_type != THEIA_LOSS_TRIVIAL)
    return set_error(THEIA_HIP_ERR_UNSUPPORTED, "BundleAdjustTwoViews runs with the trivial loss (bundle_adjust_two_views.cc:61-72)");
  if (o->max_num_iterations < 0) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "negative max_num_iterations");
  int rc = thip::ensure_device();
  if (rc) return rc;
  Dev<int64_t> d_off; Dev<double> d_corr, d_cam, d_intr, d_X, d_Xc, d_sp; Dev<int> d_model; Dev<uint8_t> d_kc; Dev<TvOut> d_out;
  if ((rc = d_off.up(b->offsets, num + 1)) || (rc = d_corr.up(b->correspondences, 4 * total)) || (rc = d_X.up(b->points, 4 * total)) ||
      (rc = d_cam.up(b->cam_ext, 12 * (size_t)num)) || (rc = d_intr.up(b->intrinsics, 2 * THEIA_MAX_INTRINSICS * (size_t)num)) ||
      (rc = d_model.up(b->model, 2 * (size_t)num)) || (rc = d_kc.up(b->const_intrinsics, 2 * (size_t)num)) ||
      (rc = d_Xc.alloc(4 * total)) || (rc = d_sp.alloc(4 * total)) || (rc = d_out.alloc(num)))
    return rc;
  TvBatch B;
  B.num = num; B.offsets = d_off.p; B.corr = reinterpret_cast<const double4*>(d_corr.p); B.cam = d_cam.p; B.intr = d_intr.p;
  B.model = d_model.p; B.kconst = d_kc.p; B.X = reinterpret_cast<double4*>(d_X.p); B.Xc = reinterpret_cast<double4*>(d_Xc.p);
  B.sp = reinterpret_cast<double4*>(d_sp.p);
  B.max_iterations = o->max_num_iterations; B.function_tolerance = o->function_tolerance; B.gradient_tolerance = o->gradient_tolerance;
  B.parameter_tolerance = o->parameter_tolerance; B.max_radius = o->max_trust_region_radius;
  const double t0 = now_s();
  k_two_view_ba<<<num, 64>>>(B, d_out.p);
  HIP_TRY(hipGetLastError());
  std::vector<TvOut> h_out(num);
  HIP_TRY(hipMemcpy(h_out.data(), d_out.p, sizeof(TvOut) * num, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(b->cam_ext, d_cam.p, sizeof(double) * 12 * num, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(b->intrinsics, d_intr.p, sizeof(double) * 2 * THEIA_MAX_INTRINSICS * num, hipMemcpyDeviceToHost));
  if (total) HIP_TRY(hipMemcpy(b->points, d_X.p, sizeof(double) * 4 * total, hipMemcpyDeviceToHost));
  const double dt = now_s() - t0;
  for (int i = 0; i < num; ++i) {
    theia_ba_summary& S = summaries[i];
    const TvOut& r = h_out[i];
    S.trace_size = 0;
    S.success = r.success; S.termination_type = r.term; S.num_iterations = r.iters; S.num_successful_steps = r.nsucc;
    S.initial_cost = r.initial_cost; S.final_cost = r.final_cost;
    S.setup_time_in_seconds = 0.0; S.solve_time_in_seconds = dt / num;
    S.time_linearize = S.time_solve_reduced = S.time_backsub = S.time_kernel_linearize = 0.0;
    S.num_linearize_launches = 0;
  }
  return 0;
}
