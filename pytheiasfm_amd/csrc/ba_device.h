// ba_device.h -- per-observation device math of the BA path (gfx950, FP64).
//
// Hand-derived analytic derivatives of the reference's residual functor
//   ReprojectionError<Model>::operator()   src/theia/sfm/camera/reprojection_error.h:54-110
// (the reference differentiates the same code path with Ceres Jets; the oracle
// restates that, this file is the closed form).  Branches are differentiated
// on the branch taken, exactly like autodiff does.
#pragma once
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cstdint>

#include "theia_hip.h"

namespace thip {

#define THIP_DEV __device__ __forceinline__

// sqrt of the BA kernels.  THIP_LEAN_SQRT (the large-solve kernels, ba_fused.hip / ba_kernels.hip): v_rsq_f64, one
// Goldschmidt step and two residual corrections -- the compiler's own expansion without its range scaling (arguments
// below 2^-767), its class selects (inf) and 12 of its 22 instructions; arguments here are sums of squares of scene
// quantities.  0 -> 0, negative -> NaN as sqrt().  Elsewhere (micro-BA batches, RANSAC refinement): sqrt().
#ifdef THIP_LEAN_SQRT
__device__ __forceinline__ double fsqrt(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  const double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  g = __builtin_fma(__builtin_fma(-g, g, x), h, g);
  g = __builtin_fma(__builtin_fma(-g, g, x), h, g);
  return x == 0.0 ? 0.0 : g;
}
#else
__device__ __forceinline__ double fsqrt(double x) { return sqrt(x); }
#endif

struct ObsLin {        // linearisation of one observation (unscaled, uncorrected)
  double r[2];         // residual
  double Jc[12];       // 2x6 wrt [position | angle-axis]
  double Jx[8];        // 2x4 wrt homogeneous point (ambient)
  double dc0, dc1;     // DIRC only: J_cam * delta_cam (rows u, v), the directional derivative along one camera step (Jc is not formed)
  bool valid;          // functor return value
};
struct ObsLinK : ObsLin {
  double Jk[2 * THEIA_MAX_INTRINSICS];  // 2xK wrt the intrinsics block (already sqrt-information weighted)
};

// Rotation terms of one camera: R and the scalars the d/d(omega) needs.
struct RotTerms {
  double R[9];
  double A, B, cA, cB;  // A=sin/th, B=(1-cos)/th^2, cA=(cos-A)/th^2, cB=(A-2B)/th^2
  bool small;
};

// ceres/rotation.h AngleAxisRotatePoint: Rodrigues if theta^2 > DBL_EPSILON,
// else first order p + w x p.
THIP_DEV void rotation_terms(const double w[3], RotTerms& t) {
  const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  if (th2 > DBL_EPSILON) {
    const double th = fsqrt(th2);
    double s, c;
    sincos(th, &s, &c);
    const double A = s / th;
    const double B = (1.0 - c) / th2;
    t.A = A; t.B = B; t.cA = (c - A) / th2; t.cB = (A - 2.0 * B) / th2; t.small = false;
    t.R[0] = c + B * w[0] * w[0];        t.R[1] = B * w[0] * w[1] - A * w[2]; t.R[2] = B * w[0] * w[2] + A * w[1];
    t.R[3] = B * w[0] * w[1] + A * w[2]; t.R[4] = c + B * w[1] * w[1];        t.R[5] = B * w[1] * w[2] - A * w[0];
    t.R[6] = B * w[0] * w[2] - A * w[1]; t.R[7] = B * w[1] * w[2] + A * w[0]; t.R[8] = c + B * w[2] * w[2];
  } else {
    t.A = 1.0; t.B = 0.0; t.cA = 0.0; t.cB = 0.0; t.small = true;
    t.R[0] = 1.0;   t.R[1] = -w[2]; t.R[2] = w[1];
    t.R[3] = w[2];  t.R[4] = 1.0;   t.R[5] = -w[0];
    t.R[6] = -w[1]; t.R[7] = w[0];  t.R[8] = 1.0;
  }
}

// d(R(w) p)/dw, 3x3 row-major.
THIP_DEV void rotation_dq_dw(const double w[3], const double p[3], const RotTerms& t, double M[9]) {
  // column k = A (e_k x p) + B (p_k w + (w.p) e_k) + w_k h
  const double c0 = w[1] * p[2] - w[2] * p[1];
  const double c1 = w[2] * p[0] - w[0] * p[2];
  const double c2 = w[0] * p[1] - w[1] * p[0];
  const double d = w[0] * p[0] + w[1] * p[1] + w[2] * p[2];
  double h0 = 0.0, h1 = 0.0, h2 = 0.0;
  if (!t.small) {
    h0 = -t.A * p[0] + t.cA * c0 + t.cB * d * w[0];
    h1 = -t.A * p[1] + t.cA * c1 + t.cB * d * w[1];
    h2 = -t.A * p[2] + t.cA * c2 + t.cB * d * w[2];
  }
  const double A = t.A, B = t.B, Bd = t.B * d;
  // k = 0: e0 x p = (0, -p2, p1)
  M[0] = B * p[0] * w[0] + Bd + w[0] * h0;
  M[3] = -A * p[2] + B * p[0] * w[1] + w[0] * h1;
  M[6] = A * p[1] + B * p[0] * w[2] + w[0] * h2;
  // k = 1: e1 x p = (p2, 0, -p0)
  M[1] = A * p[2] + B * p[1] * w[0] + w[1] * h0;
  M[4] = B * p[1] * w[1] + Bd + w[1] * h1;
  M[7] = -A * p[0] + B * p[1] * w[2] + w[1] * h2;
  // k = 2: e2 x p = (-p1, p0, 0)
  M[2] = -A * p[1] + B * p[2] * w[0] + w[2] * h0;
  M[5] = A * p[0] + B * p[2] * w[1] + w[2] * h1;
  M[8] = B * p[2] * w[2] + Bd + w[2] * h2;
}

// The step of one camera as {D, v}: d(R(w) p)/dw dw = D p for every p (the columns of rotation_dq_dw summed with the
// weights dw: D = A [dw]x + B (w dw^T + dw w^T) + (w . dw) H,  H = -A I + cA [w]x + cB w w^T, 0 for small angles),
// v = R dC.  out: 12 doubles.
THIP_DEV void camera_step_direction(const double w[3], const RotTerms& t, const double dC[3], const double dw[3], double out[12]) {
  const double wd = w[0] * dw[0] + w[1] * dw[1] + w[2] * dw[2];
  const double hA = t.small ? 0.0 : -t.A * wd, hcA = t.small ? 0.0 : t.cA * wd, hcB = t.small ? 0.0 : t.cB * wd;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      out[3 * i + j] = t.B * (w[i] * dw[j] + dw[i] * w[j]) + hcB * w[i] * w[j] + (i == j ? hA : 0.0);
  // skew parts: [a]x = (0, -a2, a1; a2, 0, -a0; -a1, a0, 0) for a = A dw + hcA w
  const double a0 = t.A * dw[0] + hcA * w[0], a1 = t.A * dw[1] + hcA * w[1], a2 = t.A * dw[2] + hcA * w[2];
  out[1] -= a2; out[2] += a1; out[3] += a2; out[5] -= a0; out[6] -= a1; out[7] += a0;
  for (int i = 0; i < 3; ++i) out[9 + i] = t.R[3 * i] * dC[0] + t.R[3 * i + 1] * dC[1] + t.R[3 * i + 2] * dC[2];
}

// Projection pi(k, q) and its 2x3 Jacobian wrt q for the eight camera models of
// create_reprojection_error_cost_function.h:61-135.  Returns the model's
// validity boolean.  Derivatives are those of the branch taken.
// WANT_KJAC additionally returns Jk (2 x THEIA_MAX_INTRINSICS, row-major): the
// derivatives wrt the intrinsics block (zero beyond the model's K parameters).
// Pseudo camera model of an observation row that is a depth prior (DepthPriorError, depth_prior_error.h):
// "pixel" = (q_z, 0), so that the generic residual sqrt_info * (pixel - obs_uv) is
// (sqrt_information * (rotated_point[2] - depth_prior), 0) with the matching Jacobian rows.
constexpr int THIP_MODEL_DEPTH_ROW = 1000;

// MODELS: compile-time set of camera models a kernel instance carries (bit = THEIA_CAM_*).  The FOV and fisheye
// models evaluate tan / atan / atan2, whose polynomial constants the compiler hoists into registers for the whole
// kernel; instances for problems without them (kModelsNoTrig) shed that register pressure and code.
constexpr unsigned kModelsAll = 0xffu;
constexpr unsigned kModelsNoTrig = 0xffu & ~((1u << THEIA_CAM_FOV) | (1u << THEIA_CAM_FISHEYE));
template <bool WANT_JAC, bool WANT_KJAC = false, unsigned MODELS = kModelsAll>
THIP_DEV bool project(int model, const double* k, const double q[3], double uv[2], double Jq[6], double* Jk = nullptr) {
  if constexpr (MODELS != kModelsAll) {
    if (model >= 0 && model < 8 && !((MODELS >> model) & 1u)) model = -1;   // not in this instance: the default branch
  }
  if (model == THIP_MODEL_DEPTH_ROW) {
    uv[0] = q[2]; uv[1] = 0.0;
    if (WANT_JAC) { Jq[0] = 0.0; Jq[1] = 0.0; Jq[2] = 1.0; Jq[3] = 0.0; Jq[4] = 0.0; Jq[5] = 0.0; }
    if (WANT_KJAC) {
#pragma unroll
      for (int i = 0; i < 2 * THEIA_MAX_INTRINSICS; ++i) Jk[i] = 0.0;
    }
    return true;
  }
  double dx = 0.0, dy = 0.0;                                  // distorted normalised point
  double ddx[3] = {0.0, 0.0, 0.0}, ddy[3] = {0.0, 0.0, 0.0};  // d(dx)/dq, d(dy)/dq
  // d(dx)/d(distortion parameter p), d(dy)/d(p) for p >= 5 (or 4 for the no-skew layouts)
  double pdx[THEIA_MAX_INTRINSICS], pdy[THEIA_MAX_INTRINSICS];
  if (WANT_KJAC) {
#pragma unroll
    for (int i = 0; i < THEIA_MAX_INTRINSICS; ++i) { pdx[i] = 0.0; pdy[i] = 0.0; }
#pragma unroll
    for (int i = 0; i < 2 * THEIA_MAX_INTRINSICS; ++i) Jk[i] = 0.0;
  }
  bool ok = true;
  // models whose distortion acts on (x, y) = (q0/q2, q1/q2): fill dxx.. then chain
  bool planar = false;
  double x = 0.0, y = 0.0, dxx = 0.0, dxy = 0.0, dyx = 0.0, dyy = 0.0;
  switch (model) {
    case THEIA_CAM_PINHOLE: {
      // pinhole_camera_model.h:181-211,243-260
      planar = true;
      x = q[0] / q[2]; y = q[1] / q[2];
      const double r2 = x * x + y * y;
      const double d = 1.0 + r2 * (k[5] + k[6] * r2);
      dx = x * d; dy = y * d;
      if (WANT_JAC) {
        const double e = 2.0 * (k[5] + 2.0 * k[6] * r2);
        dxx = d + x * x * e; dxy = x * y * e; dyx = dxy; dyy = d + y * y * e;
      }
      if (WANT_KJAC) { pdx[5] = x * r2; pdy[5] = y * r2; pdx[6] = x * r2 * r2; pdy[6] = y * r2 * r2; }
      break; }
    case THEIA_CAM_PINHOLE_RADIAL_TANGENTIAL: {
      // pinhole_radial_tangential_camera_model.h:191-296  [.., k1 k2 k3 t1 t2]
      planar = true;
      x = q[0] / q[2]; y = q[1] / q[2];
      const double r2 = x * x + y * y;
      const double rd = 1.0 + k[5] * r2 + k[6] * r2 * r2 + k[7] * r2 * r2 * r2;
      const double t1 = k[8], t2 = k[9];
      dx = x * rd + (t2 * (r2 + 2.0 * x * x) + 2.0 * t1 * x * y);
      dy = y * rd + (t1 * (r2 + 2.0 * y * y) + 2.0 * t2 * x * y);
      if (WANT_JAC) {
        const double e = 2.0 * (k[5] + 2.0 * k[6] * r2 + 3.0 * k[7] * r2 * r2);
        dxx = rd + x * x * e + 6.0 * t2 * x + 2.0 * t1 * y;
        dxy = x * y * e + 2.0 * t2 * y + 2.0 * t1 * x;
        dyx = x * y * e + 2.0 * t1 * x + 2.0 * t2 * y;
        dyy = rd + y * y * e + 6.0 * t1 * y + 2.0 * t2 * x;
      }
      if (WANT_KJAC) {
        pdx[5] = x * r2; pdy[5] = y * r2; pdx[6] = x * r2 * r2; pdy[6] = y * r2 * r2;
        pdx[7] = x * r2 * r2 * r2; pdy[7] = y * r2 * r2 * r2;
        pdx[8] = 2.0 * x * y; pdy[8] = r2 + 2.0 * y * y;
        pdx[9] = r2 + 2.0 * x * x; pdy[9] = 2.0 * x * y;
      }
      break; }
    case THEIA_CAM_FOV: {
      if constexpr ((MODELS >> THEIA_CAM_FOV) & 1u) {
      // fov_camera_model.h:156-258  [f a cx cy omega]
      planar = true;
      x = q[0] / q[2]; y = q[1] / q[2];
      const double omega = k[4];
      const double ru2 = x * x + y * y;
      double rd, g, gw;  // g = d rd / d(ru2), gw = d rd / d(omega)
      if (omega < 1e-3) {
        rd = (omega * omega * ru2) / 3.0 - omega * omega / 12.0 + 1.0;
        g = omega * omega / 3.0;
        gw = 2.0 * omega * ru2 / 3.0 - omega / 6.0;
      } else if (ru2 < 1e-3) {
        const double th = tan(omega / 2.0);
        const double dth = 0.5 * (1.0 + th * th);
        rd = (-2.0 * th * (4.0 * ru2 * th * th - 3.0)) / (3.0 * omega);
        g = -8.0 * th * th * th / (3.0 * omega);
        gw = (-2.0 * dth * (4.0 * ru2 * th * th - 3.0) - 2.0 * th * (8.0 * ru2 * th * dth)) / (3.0 * omega) - rd / omega;
      } else {
        const double ru = fsqrt(ru2);
        const double th = tan(omega / 2.0);
        const double dth = 0.5 * (1.0 + th * th);
        const double at = atan(2.0 * ru * th);
        rd = at / (ru * omega);
        const double drd_dru = ((2.0 * th / (1.0 + 4.0 * ru2 * th * th)) * ru - at) / (ru2 * omega);
        g = drd_dru / (2.0 * ru);
        gw = (2.0 * ru * dth / (1.0 + 4.0 * ru2 * th * th)) / (ru * omega) - rd / omega;
      }
      dx = rd * x; dy = rd * y;
      if (WANT_KJAC) { pdx[4] = gw * x; pdy[4] = gw * y; }
      if (WANT_JAC) { dxx = rd + 2.0 * x * x * g; dxy = 2.0 * x * y * g; dyx = dxy; dyy = rd + 2.0 * y * y * g; }
      }
      break; }
    case THEIA_CAM_ORTHOGRAPHIC: {
      // orthographic_camera_model.h:162-240: distortion on (q0, q1), no depth division
      const double r2 = q[0] * q[0] + q[1] * q[1];
      const double d = 1.0 + r2 * (k[5] + k[6] * r2);
      dx = q[0] * d; dy = q[1] * d;
      if (WANT_JAC) {
        const double e = 2.0 * (k[5] + 2.0 * k[6] * r2);
        ddx[0] = d + q[0] * q[0] * e; ddx[1] = q[0] * q[1] * e;
        ddy[0] = ddx[1]; ddy[1] = d + q[1] * q[1] * e;
      }
      if (WANT_KJAC) { pdx[5] = q[0] * r2; pdy[5] = q[1] * r2; pdx[6] = q[0] * r2 * r2; pdy[6] = q[1] * r2 * r2; }
      break; }
    case THEIA_CAM_FISHEYE: {
      if constexpr ((MODELS >> THEIA_CAM_FISHEYE) & 1u) {
      // fisheye_camera_model.h:163-272  [.., k1 k2 k3 k4]
      const double r2 = q[0] * q[0] + q[1] * q[1];
      if (r2 < 1e-8) {
        dx = q[0]; dy = q[1];
        if (WANT_JAC) { ddx[0] = 1.0; ddy[1] = 1.0; }
      } else {
        const double r = fsqrt(r2);
        const double az = fabs(q[2]);
        const double th = atan2(r, az);
        const double t2 = th * th;
        const double thd = th * (1.0 + k[5] * t2 + k[6] * t2 * t2 + k[7] * t2 * t2 * t2 + k[8] * t2 * t2 * t2 * t2);
        const double sgn = (q[2] < 0.0) ? -1.0 : 1.0;
        const double s = thd / r;
        dx = sgn * s * q[0]; dy = sgn * s * q[1];
        if (WANT_JAC) {
          const double P = 1.0 + 3.0 * k[5] * t2 + 5.0 * k[6] * t2 * t2 + 7.0 * k[7] * t2 * t2 * t2 + 9.0 * k[8] * t2 * t2 * t2 * t2;
          const double den = r2 + q[2] * q[2];
          const double dth_dr = az / den;
          const double dth_dz = -r * ((q[2] < 0.0) ? -1.0 : 1.0) / den;
          const double ds_dr = (P * dth_dr * r - thd) / r2;
          const double ds0 = ds_dr * q[0] / r, ds1 = ds_dr * q[1] / r, ds2 = P * dth_dz / r;
          ddx[0] = sgn * (s + q[0] * ds0); ddx[1] = sgn * (q[0] * ds1); ddx[2] = sgn * (q[0] * ds2);
          ddy[0] = sgn * (q[1] * ds0); ddy[1] = sgn * (s + q[1] * ds1); ddy[2] = sgn * (q[1] * ds2);
        }
        if (WANT_KJAC) {
          double tp = th * t2;  // theta^3, ^5, ^7, ^9
#pragma unroll
          for (int i = 0; i < 4; ++i) { pdx[5 + i] = sgn * tp * q[0] / r; pdy[5 + i] = sgn * tp * q[1] / r; tp *= t2; }
        }
      }
      }
      break; }
    case THEIA_CAM_DOUBLE_SPHERE: {
      // double_sphere_camera_model.h:160-249
      const double alpha = k[6], xi = k[5];
      const double r2 = q[0] * q[0] + q[1] * q[1];
      const double d1 = fsqrt(r2 + q[2] * q[2]);
      const double w1 = alpha > 0.5 ? (1.0 - alpha) / alpha : alpha / (1.0 - alpha);
      const double w2 = (w1 + xi) / fsqrt(2.0 * w1 * xi + xi * xi + 1.0);
      if (q[2] <= -w2 * d1) ok = false;
      const double kk = xi * d1 + q[2];
      const double d2 = fsqrt(r2 + kk * kk);
      const double n = alpha * d2 + (1.0 - alpha) * kk;
      dx = q[0] / n; dy = q[1] / n;
      if (WANT_JAC) {
        const double id1 = 1.0 / d1, id2 = 1.0 / d2, in = 1.0 / n;
        const double dk[3] = {xi * q[0] * id1, xi * q[1] * id1, xi * q[2] * id1 + 1.0};
        const double dd2[3] = {(q[0] + kk * dk[0]) * id2, (q[1] + kk * dk[1]) * id2, (kk * dk[2]) * id2};
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const double dn = alpha * dd2[i] + (1.0 - alpha) * dk[i];
          ddx[i] = -dx * dn * in;
          ddy[i] = -dy * dn * in;
        }
        ddx[0] += in; ddy[1] += in;
      }
      if (WANT_KJAC) {
        // n = alpha d2 + (1 - alpha) kk, kk = xi d1 + z, d2 = sqrt(r2 + kk^2)
        const double dn_dxi = alpha * kk * d1 / d2 + (1.0 - alpha) * d1;
        const double dn_dal = d2 - kk;
        pdx[5] = -dx * dn_dxi / n; pdy[5] = -dy * dn_dxi / n;
        pdx[6] = -dx * dn_dal / n; pdy[6] = -dy * dn_dal / n;
      }
      break; }
    case THEIA_CAM_EXTENDED_UNIFIED: {
      // extended_unified_camera_model.h:161-249  [.., alpha, beta]
      const double alpha = k[5], beta = k[6];
      const double r2 = q[0] * q[0] + q[1] * q[1];
      const double rho = fsqrt(beta * r2 + q[2] * q[2]);
      const double n = alpha * rho + (1.0 - alpha) * q[2];
      bool zero = n < 1e-3;
      if (!zero && alpha > 0.5) zero = (q[2] / n) < (alpha - 1.0) / (alpha + alpha - 1.0);
      if (!zero) {
        dx = q[0] / n; dy = q[1] / n;
        if (WANT_JAC) {
          const double in = 1.0 / n, ir = 1.0 / rho;
          const double dn[3] = {alpha * beta * q[0] * ir, alpha * beta * q[1] * ir, alpha * q[2] * ir + (1.0 - alpha)};
#pragma unroll
          for (int i = 0; i < 3; ++i) { ddx[i] = -dx * dn[i] * in; ddy[i] = -dy * dn[i] * in; }
          ddx[0] += in; ddy[1] += in;
        }
        if (WANT_KJAC) {
          const double dn_dal = rho - q[2], dn_dbe = alpha * r2 / (2.0 * rho);
          pdx[5] = -dx * dn_dal / n; pdy[5] = -dy * dn_dal / n;
          pdx[6] = -dx * dn_dbe / n; pdy[6] = -dy * dn_dbe / n;
        }
      }
      break; }
    case THEIA_CAM_DIVISION_UNDISTORTION: {
      // division_undistortion_camera_model.h:173-231,263-297  [f a cx cy k]:
      // distortion AFTER focal scaling, no skew -> own affine stage
      const double iz = 1.0 / q[2];
      const double fx = k[0], fy = k[0] * k[1];
      const double ux = fx * (q[0] / q[2]), uy = fy * (q[1] / q[2]);
      const double ru2 = ux * ux + uy * uy;
      const double denom = 2.0 * k[4] * ru2;
      const double inner = 1.0 - 4.0 * k[4] * ru2;
      double scale = 1.0, g = 0.0;  // g = d scale / d(ru2)
      if (!(fabs(denom) < DBL_EPSILON || inner < 0.0)) {
        const double sq = fsqrt(inner);
        scale = (1.0 - sq) / denom;
        g = ((2.0 * k[4] / sq) * denom - (1.0 - sq) * (2.0 * k[4])) / (denom * denom);
      }
      uv[0] = ux * scale + k[2];
      uv[1] = uy * scale + k[3];
      if (WANT_JAC) {
        const double pxx = scale + 2.0 * ux * ux * g, pxy = 2.0 * ux * uy * g, pyy = scale + 2.0 * uy * uy * g;
        // d(ux)/dq = fx (1/z, 0, -x/z), d(uy)/dq = fy (0, 1/z, -y/z)
        const double xx = q[0] * iz, yy = q[1] * iz;
        Jq[0] = pxx * fx * iz; Jq[1] = pxy * fy * iz; Jq[2] = -(pxx * fx * xx + pxy * fy * yy) * iz;
        Jq[3] = pxy * fx * iz; Jq[4] = pyy * fy * iz; Jq[5] = -(pxy * fx * xx + pyy * fy * yy) * iz;
      }
      if (WANT_KJAC) {
        const double xn = q[0] * iz, yn = q[1] * iz;
        const double pxx = scale + 2.0 * ux * ux * g, pxy = 2.0 * ux * uy * g, pyy = scale + 2.0 * uy * uy * g;
        // f: ux = f xn, uy = f a yn ; a: uy = f a yn
        Jk[0] = pxx * xn + pxy * k[1] * yn;                         Jk[THEIA_MAX_INTRINSICS + 0] = pxy * xn + pyy * k[1] * yn;
        Jk[1] = pxy * k[0] * yn;                                    Jk[THEIA_MAX_INTRINSICS + 1] = pyy * k[0] * yn;
        Jk[2] = 1.0;                                                Jk[THEIA_MAX_INTRINSICS + 3] = 1.0;
        double gk = 0.0;  // d scale / d k  at fixed ru2
        if (!(fabs(denom) < DBL_EPSILON || inner < 0.0)) {
          const double sq = fsqrt(inner);
          gk = ((2.0 * ru2 / sq) * denom - (1.0 - sq) * (2.0 * ru2)) / (denom * denom);
        }
        Jk[4] = ux * gk;                                            Jk[THEIA_MAX_INTRINSICS + 4] = uy * gk;
      }
      return true; }
    default:
      uv[0] = 0.0; uv[1] = 0.0;
      if (WANT_JAC) for (int i = 0; i < 6; ++i) Jq[i] = 0.0;
      return false;
  }
  if (planar && WANT_JAC) {
    // chain through x = q0/q2, y = q1/q2
    const double iz = 1.0 / q[2];
    ddx[0] = dxx * iz; ddx[1] = dxy * iz; ddx[2] = -(dxx * x + dxy * y) * iz;
    ddy[0] = dyx * iz; ddy[1] = dyy * iz; ddy[2] = -(dyx * x + dyy * y) * iz;
  }
  // affine stage: u = f dx + s dy + cx, v = f a dy + cy; FOV has no skew slot
  const bool noskew = (model == THEIA_CAM_FOV);
  const double f = k[0], fa = k[0] * k[1];
  const double sk = noskew ? 0.0 : k[2];
  const double cx = noskew ? k[2] : k[3], cy = noskew ? k[3] : k[4];
  uv[0] = f * dx + sk * dy + cx;
  uv[1] = fa * dy + cy;
  if (WANT_JAC) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      Jq[i] = f * ddx[i] + sk * ddy[i];
      Jq[3 + i] = fa * ddy[i];
    }
  }
  if (WANT_KJAC) {
    double* Ju = Jk;
    double* Jv = Jk + THEIA_MAX_INTRINSICS;
    Ju[0] = dx;            Jv[0] = k[1] * dy;   // focal
    Jv[1] = k[0] * dy;                          // aspect ratio
    if (noskew) { Ju[2] = 1.0; Jv[3] = 1.0; }
    else { Ju[2] = dy; Ju[3] = 1.0; Jv[4] = 1.0; }
    const int p0 = noskew ? 4 : 5;
#pragma unroll
    for (int i = 4; i < THEIA_MAX_INTRINSICS; ++i)
      if (i >= p0) { Ju[i] = f * pdx[i] + sk * pdy[i]; Jv[i] = fa * pdy[i]; }
  }
  return ok;
}

// Residual (and optionally Jacobians) of one observation.
// `t` = rotation_terms(ext + 3): per-camera quantities, computed per observation by observe() or once per camera and
// iteration by the callers that keep them in HBM (ba_fused.hip: k_cam_prep).
// DIRC: `dir` = {D (3 x 3, row-major), v (3)} of the camera's step (k_cam_update): instead of the 2 x 6 camera block the
// functor returns o.dc = J_cam delta_cam = s * Jq (D p - w v)  (d(R p)/d(omega) delta_omega = D p is linear in p,
// dq/dC delta_C = -w R delta_C = -w v); o.Jc is left untouched.
template <bool WANT_JAC, bool WANT_KJAC = false, typename OL = ObsLin, unsigned MODELS = kModelsAll, bool DIRC = false>
THIP_DEV void observe_rot(int model, const double* ext, const RotTerms& t, const double* intr, const double X[4],
                          double u0, double v0, double six, double siy, OL& o, const double* dir = nullptr) {
  const double p[3] = {X[0] - X[3] * ext[0], X[1] - X[3] * ext[1], X[2] - X[3] * ext[2]};
  const double sq = p[0] * p[0] + p[1] * p[1] + p[2] * p[2];
  o.dc0 = 0.0; o.dc1 = 0.0;
  if (sq < 1e-8) {  // reprojection_error.h:78-80 -> functor returns false
    o.valid = false; o.r[0] = 0.0; o.r[1] = 0.0;
    if (WANT_JAC) {
#pragma unroll
      for (int i = 0; i < 12; ++i) o.Jc[i] = 0.0;
#pragma unroll
      for (int i = 0; i < 8; ++i) o.Jx[i] = 0.0;
    }
    if constexpr (WANT_KJAC) {
#pragma unroll
      for (int i = 0; i < 2 * THEIA_MAX_INTRINSICS; ++i) o.Jk[i] = 0.0;
    }
    return;
  }
  const double q[3] = {t.R[0] * p[0] + t.R[1] * p[1] + t.R[2] * p[2],
                       t.R[3] * p[0] + t.R[4] * p[1] + t.R[5] * p[2],
                       t.R[6] * p[0] + t.R[7] * p[1] + t.R[8] * p[2]};
  double uv[2], Jq[6];
  if constexpr (WANT_KJAC) {
    o.valid = project<WANT_JAC, true, MODELS>(model, intr, q, uv, Jq, o.Jk);
#pragma unroll
    for (int i = 0; i < THEIA_MAX_INTRINSICS; ++i) { o.Jk[i] *= six; o.Jk[THEIA_MAX_INTRINSICS + i] *= siy; }
  } else {
    o.valid = project<WANT_JAC, false, MODELS>(model, intr, q, uv, Jq);
  }
  o.r[0] = six * (uv[0] - u0);
  o.r[1] = siy * (uv[1] - v0);
  if (WANT_JAC) {
    double M[9];
    double u[3] = {0.0, 0.0, 0.0};
    if constexpr (DIRC) {
      for (int i = 0; i < 3; ++i) u[i] = (dir[3 * i] * p[0] + dir[3 * i + 1] * p[1] + dir[3 * i + 2] * p[2]) - X[3] * dir[9 + i];
    } else {
      rotation_dq_dw(ext + 3, p, t, M);
    }
    const double s[2] = {six, siy};
#pragma unroll
    for (int a = 0; a < 2; ++a) {   // (unrolled: o.dc[a] / o.Jc[6 a + ..] must be register indices, not a scratch array)
      const double* jq = Jq + 3 * a;
      // A = Jq R  (1x3)
      const double A0 = jq[0] * t.R[0] + jq[1] * t.R[3] + jq[2] * t.R[6];
      const double A1 = jq[0] * t.R[1] + jq[1] * t.R[4] + jq[2] * t.R[7];
      const double A2 = jq[0] * t.R[2] + jq[1] * t.R[5] + jq[2] * t.R[8];
      if constexpr (DIRC) {
        const double dv = s[a] * (jq[0] * u[0] + jq[1] * u[1] + jq[2] * u[2]);
        if (a == 0) o.dc0 = dv; else o.dc1 = dv;
      } else {
      // dq/dC = -w R
      o.Jc[6 * a + 0] = -s[a] * X[3] * A0;
      o.Jc[6 * a + 1] = -s[a] * X[3] * A1;
      o.Jc[6 * a + 2] = -s[a] * X[3] * A2;
      o.Jc[6 * a + 3] = s[a] * (jq[0] * M[0] + jq[1] * M[3] + jq[2] * M[6]);
      o.Jc[6 * a + 4] = s[a] * (jq[0] * M[1] + jq[1] * M[4] + jq[2] * M[7]);
      o.Jc[6 * a + 5] = s[a] * (jq[0] * M[2] + jq[1] * M[5] + jq[2] * M[8]);
      }
      // dq/dX = [R | -R C]
      o.Jx[4 * a + 0] = s[a] * A0;
      o.Jx[4 * a + 1] = s[a] * A1;
      o.Jx[4 * a + 2] = s[a] * A2;
      o.Jx[4 * a + 3] = -s[a] * (A0 * ext[0] + A1 * ext[1] + A2 * ext[2]);
    }
  }
}

template <bool WANT_JAC, bool WANT_KJAC = false, typename OL = ObsLin>
THIP_DEV void observe(int model, const double* ext, const double* intr, const double X[4],
                      double u0, double v0, double six, double siy, OL& o) {
  RotTerms t;
  rotation_terms(ext + 3, t);
  observe_rot<WANT_JAC, WANT_KJAC, OL>(model, ext, t, intr, X, u0, v0, six, siy, o);
}

// Per-camera block kept in HBM by k_cam_prep, so that an observation needs ONE dependent gather for everything keyed
// by its camera: {position (3), angle-axis (3), R (9), A, B, cA, cB, small | Jacobi scaling of the six extrinsics
// columns (0 = frozen column) | intrinsics of the camera's group (10) | model, reduced index, group, pad} = 40 doubles.
constexpr int kCamRot = 40;
constexpr int kCamRotScale = 20, kCamRotIntr = 26, kCamRotModel = 36, kCamRotRed = 37, kCamRotGroup = 38;
THIP_DEV void camrot_store(const double* ext, double* o) {
  RotTerms t;
  rotation_terms(ext + 3, t);
  for (int i = 0; i < 6; ++i) o[i] = ext[i];
  for (int i = 0; i < 9; ++i) o[6 + i] = t.R[i];
  o[15] = t.A; o[16] = t.B; o[17] = t.cA; o[18] = t.cB; o[19] = t.small ? 1.0 : 0.0;
}
template <int N>
THIP_DEV void load_d2(const double* __restrict__ src, double (&v)[N]) {   // 16-B aligned source, N even
  const double2* c2 = reinterpret_cast<const double2*>(src);
#pragma unroll
  for (int i = 0; i < N / 2; ++i) { const double2 q = c2[i]; v[2 * i] = q.x; v[2 * i + 1] = q.y; }
}
THIP_DEV void camrot_load(const double* __restrict__ cr, double ext[6], RotTerms& t) {
  double v[20];
  load_d2<20>(cr, v);
#pragma unroll
  for (int i = 0; i < 6; ++i) ext[i] = v[i];
#pragma unroll
  for (int i = 0; i < 9; ++i) t.R[i] = v[6 + i];
  t.A = v[15]; t.B = v[16]; t.cA = v[17]; t.cB = v[18]; t.small = v[19] != 0.0;
}

// ceres/loss_function.cc + theia TruncatedLoss (loss_functions.cc:40-44).
// Returns rho(s); *rho1 = rho'(s).  Every loss here has rho'' <= 0, so the
// Triggs corrector (ceres/corrector.cc) reduces to scaling by sqrt(rho').
THIP_DEV double loss_eval(int type, double a, double s, double* rho1) {
  switch (type) {
    case THEIA_LOSS_HUBER: {
      const double b = a * a;
      if (s > b) { const double r = fsqrt(s); *rho1 = fmax(DBL_MIN, a / r); return 2.0 * a * r - b; }
      *rho1 = 1.0; return s; }
    case THEIA_LOSS_SOFTLONE: {
      const double b = a * a; const double sum = 1.0 + s / b; const double tmp = fsqrt(sum);
      *rho1 = fmax(DBL_MIN, 1.0 / tmp); return 2.0 * b * (tmp - 1.0); }
    case THEIA_LOSS_CAUCHY: {
      const double b = a * a; const double sum = 1.0 + s / b;
      *rho1 = fmax(DBL_MIN, 1.0 / sum); return b * log(sum); }
    case THEIA_LOSS_ARCTAN: {
      const double sum = 1.0 + s * s / (a * a);
      *rho1 = fmax(DBL_MIN, 1.0 / sum); return a * atan2(s, a); }
    case THEIA_LOSS_TUKEY: {
      const double a2 = a * a;
      if (s <= a2) { const double v = 1.0 - s / a2; const double v2 = v * v; *rho1 = v2;
        return a2 / 3.0 * (1.0 - v2 * v); }
      *rho1 = 0.0; return a2 / 3.0; }
    case THEIA_LOSS_TRUNCATED: {
      const double se = a * a; *rho1 = s < se ? 1.0 : 0.0; return fmin(s, se); }
    default: *rho1 = 1.0; return s;
  }
}

// ceres SphereManifold<4> (bundle_adjuster.cc:538-545): Householder vector.
THIP_DEV void householder4(const double x[4], double v[4], double& beta) {
  const double sigma = x[0] * x[0] + x[1] * x[1] + x[2] * x[2];
  v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = 1.0;
  beta = 0.0;
  if (sigma <= DBL_EPSILON) { if (x[3] < 0.0) beta = 2.0; return; }
  const double mu = fsqrt(x[3] * x[3] + sigma);
  const double vp = (x[3] <= 0.0) ? x[3] - mu : -sigma / (x[3] + mu);
  beta = 2.0 * vp * vp / (sigma + vp * vp);
  v[0] /= vp; v[1] /= vp; v[2] /= vp;
}

// Tangent-space point Jacobian: Jt(2x3) = Jx(2x4) * |x| (I - beta v v^T)[:,0:3]
THIP_DEV void to_tangent(const double X[4], const double Jx[8], double Jt[6]) {
  double v[4], beta;
  householder4(X, v, beta);
  const double nx = fsqrt(X[0] * X[0] + X[1] * X[1] + X[2] * X[2] + X[3] * X[3]);
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const double* j = Jx + 4 * a;
    const double jv = j[0] * v[0] + j[1] * v[1] + j[2] * v[2] + j[3] * v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) Jt[3 * a + c] = nx * (j[c] - beta * v[c] * jv);
  }
}

// sin(t) / t and cos(t) from t^2 for t^2 <= 0.5 (Taylor to t^20: truncation < 2e-21); THIP_LEAN_SINCOS (the large-solve
// kernels): SphereManifold::Plus on a tangent step needs exactly these two, a BA point step is tiny, and libm's sincos brings
// its range reduction, ~100 instructions and a set of hoisted constants into kernels that are issue and register bound.
THIP_DEV void sinc_cos_small(double t2, double& sinc, double& cosv) {
  double a = 1.0 / 51090942171709440000.0, b = 1.0 / 2432902008176640000.0;   // 1/21!, 1/20!
  a = __builtin_fma(-a, t2, 1.0 / 121645100408832000.0);  b = __builtin_fma(-b, t2, 1.0 / 6402373705728000.0);    // 1/19!, 1/18!
  a = __builtin_fma(-a, t2, 1.0 / 355687428096000.0);     b = __builtin_fma(-b, t2, 1.0 / 20922789888000.0);      // 1/17!, 1/16!
  a = __builtin_fma(-a, t2, 1.0 / 1307674368000.0);       b = __builtin_fma(-b, t2, 1.0 / 87178291200.0);         // 1/15!, 1/14!
  a = __builtin_fma(-a, t2, 1.0 / 6227020800.0);          b = __builtin_fma(-b, t2, 1.0 / 479001600.0);           // 1/13!, 1/12!
  a = __builtin_fma(-a, t2, 1.0 / 39916800.0);            b = __builtin_fma(-b, t2, 1.0 / 3628800.0);             // 1/11!, 1/10!
  a = __builtin_fma(-a, t2, 1.0 / 362880.0);              b = __builtin_fma(-b, t2, 1.0 / 40320.0);               // 1/9!, 1/8!
  a = __builtin_fma(-a, t2, 1.0 / 5040.0);                b = __builtin_fma(-b, t2, 1.0 / 720.0);                 // 1/7!, 1/6!
  a = __builtin_fma(-a, t2, 1.0 / 120.0);                 b = __builtin_fma(-b, t2, 1.0 / 24.0);                  // 1/5!, 1/4!
  a = __builtin_fma(-a, t2, 1.0 / 6.0);                   b = __builtin_fma(-b, t2, 0.5);                         // 1/3!, 1/2!
  sinc = __builtin_fma(-a, t2, 1.0);                      cosv = __builtin_fma(-b, t2, 1.0);
}
THIP_DEV void sphere_plus(const double x[4], const double d[3], double out[4]) {
  const double nd2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
#ifdef THIP_LEAN_SINCOS
  if (nd2 == 0.0) { out[0] = x[0]; out[1] = x[1]; out[2] = x[2]; out[3] = x[3]; return; }
  double v[4], beta;
  householder4(x, v, beta);
  const double nx = fsqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3]);
  double sbd, c;
  if (nd2 <= 0.5) sinc_cos_small(nd2, sbd, c);
  else { const double nd = fsqrt(nd2); double s; sincos(nd, &s, &c); sbd = s / nd; }
#else
  const double nd = fsqrt(nd2);
  if (nd == 0.0) { out[0] = x[0]; out[1] = x[1]; out[2] = x[2]; out[3] = x[3]; return; }
  double v[4], beta;
  householder4(x, v, beta);
  const double nx = fsqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3]);
  double s, c;
  sincos(nd, &s, &c);
  const double sbd = s / nd;
#endif
  const double y[4] = {sbd * d[0], sbd * d[1], sbd * d[2], c};
  const double vty = v[0] * y[0] + v[1] * y[1] + v[2] * y[2] + v[3] * y[3];
  for (int i = 0; i < 4; ++i) out[i] = nx * (y[i] - v[i] * (beta * vty));
}

// 3x3 SPD inverse through Cholesky (ceres InvertPSDMatrix, full-rank branch).
// V = [v00 v10 v11 v20 v21 v22] lower; returns false if not positive definite.
THIP_DEV bool invert_spd3(const double V[6], double Vi[6]) {
  if (!(V[0] > 0.0)) return false;
  const double l00 = fsqrt(V[0]);
  const double l10 = V[1] / l00;
  const double d11 = V[2] - l10 * l10;
  if (!(d11 > 0.0)) return false;
  const double l11 = fsqrt(d11);
  const double l20 = V[3] / l00;
  const double l21 = (V[4] - l20 * l10) / l11;
  const double d22 = V[5] - l20 * l20 - l21 * l21;
  if (!(d22 > 0.0)) return false;
  const double l22 = fsqrt(d22);
  // inverse of L (lower)
  const double i00 = 1.0 / l00, i11 = 1.0 / l11, i22 = 1.0 / l22;
  const double i10 = -l10 * i00 * i11;
  const double i21 = -l21 * i11 * i22;
  const double i20 = -(l20 * i00 + l21 * i10) * i22;
  // Vi = Li^T Li
  Vi[0] = i00 * i00 + i10 * i10 + i20 * i20;
  Vi[1] = i10 * i11 + i20 * i21;
  Vi[2] = i11 * i11 + i21 * i21;
  Vi[3] = i20 * i22;
  Vi[4] = i21 * i22;
  Vi[5] = i22 * i22;
  return true;
}

}  // namespace thip
