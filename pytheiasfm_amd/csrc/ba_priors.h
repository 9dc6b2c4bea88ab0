// ba_priors.h -- camera priors of the bundle adjustment (device side).
//
// Three optional residual blocks per view, each 3 residuals on the camera extrinsics, no loss function
// (bundle_adjuster.cc:291-313,624-658):
//   position     r = S (p_prior - position)                                  position_error.h:51-59
//   gravity      r = S (R(w) (0,0,-1) - g_prior)                             gravity_error.h:51-65
//   orientation  r = S log(exp(w) exp(w_prior)^-1)   (Sophus SO3)            orientation_error.h:53-64
// The reference differentiates them with Jets; here the Jacobians are closed form:
//   d r_pos / d position = -S;   d r_grav / d w = S d(R g)/dw;
//   d r_ori / d w = S Jl^-1(phi) Jl(w),  phi = log(exp(w) exp(w_prior)^-1),  Jl = left Jacobian of SO(3).
#pragma once
#include "ba_device.h"

namespace thip {

// quaternion [w, x, y, z] of exp(omega) (Sophus SO3::expAndTheta)
THIP_DEV void so3_exp_quat(const double w[3], double q[4]) {
  const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  double imag, real;
  if (th2 < 1e-10 * 1e-10) {
    const double th4 = th2 * th2;
    imag = 0.5 - th2 * (1.0 / 48.0) + th4 * (1.0 / 3840.0);
    real = 1.0 - th2 * (1.0 / 8.0) + th4 * (1.0 / 384.0);
  } else {
    const double th = sqrt(th2), half = th * 0.5;
    imag = sin(half) / th;
    real = cos(half);
  }
  q[0] = real; q[1] = imag * w[0]; q[2] = imag * w[1]; q[3] = imag * w[2];
}
THIP_DEV void so3_log_quat(const double q[4], double t[3]) {   // Sophus SO3::logAndTheta
  const double sn = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  const double w = q[0];
  double f;
  if (sn < 1e-10 * 1e-10) {
    f = 2.0 / w - (sn * (2.0 / 3.0)) / (w * w * w);
  } else {
    const double n = sqrt(sn);
    const double at = (w < 0.0) ? atan2(-n, -w) : atan2(n, w);
    f = (at * 2.0) / n;
  }
  t[0] = f * q[1]; t[1] = f * q[2]; t[2] = f * q[3];
}
THIP_DEV void quat_mul_so3(const double a[4], const double b[4], double o[4]) {   // SO3Base::operator*
  o[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  o[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  o[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  o[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  const double sq = o[0] * o[0] + o[1] * o[1] + o[2] * o[2] + o[3] * o[3];
  if (sq != 1.0) {
    const double s = 2.0 / (1.0 + sq);
    for (int k = 0; k < 4; ++k) o[k] *= s;
  }
}
// M = I + a [v]x + b [v]x^2   (row-major)
THIP_DEV void skew_poly(const double v[3], double a, double b, double M[9]) {
  const double xx = v[0] * v[0], yy = v[1] * v[1], zz = v[2] * v[2];
  const double xy = v[0] * v[1], xz = v[0] * v[2], yz = v[1] * v[2];
  M[0] = 1.0 - b * (yy + zz); M[1] = -a * v[2] + b * xy;     M[2] = a * v[1] + b * xz;
  M[3] = a * v[2] + b * xy;   M[4] = 1.0 - b * (xx + zz);    M[5] = -a * v[0] + b * yz;
  M[6] = -a * v[1] + b * xz;  M[7] = a * v[0] + b * yz;      M[8] = 1.0 - b * (xx + yy);
}
THIP_DEV void so3_left_jacobian(const double w[3], double J[9]) {
  const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  double B, C;
  if (th2 < 1e-8) { B = 0.5 - th2 / 24.0; C = 1.0 / 6.0 - th2 / 120.0; }
  else { const double th = sqrt(th2); B = (1.0 - cos(th)) / th2; C = (th - sin(th)) / (th2 * th); }
  skew_poly(w, B, C, J);
}
THIP_DEV void so3_left_jacobian_inverse(const double p[3], double J[9]) {
  const double th2 = p[0] * p[0] + p[1] * p[1] + p[2] * p[2];
  double D;
  if (th2 < 1e-8) D = 1.0 / 12.0 + th2 / 720.0;
  else { const double th = sqrt(th2); D = 1.0 / th2 - (1.0 + cos(th)) / (2.0 * th * sin(th)); }
  skew_poly(p, -0.5, D, J);
}

// kind: THEIA_PRIOR_* bit.  r[3]; J[3][6] (unscaled, unmasked) when want_jac.
THIP_DEV void camera_prior(int kind, const double* ext, const double* prior, const double* S, bool want_jac,
                           double r[3], double J[18]) {
  double v[3], D[9];   // v = the residual before weighting, D = d v / d (the 3 parameters it depends on)
  int col0 = 0;
  if (kind == THEIA_PRIOR_POSITION) {
    for (int k = 0; k < 3; ++k) v[k] = prior[k] - ext[k];
    for (int k = 0; k < 9; ++k) D[k] = (k % 4 == 0) ? -1.0 : 0.0;
  } else if (kind == THEIA_PRIOR_GRAVITY) {
    col0 = 3;
    RotTerms rt;
    rotation_terms(ext + 3, rt);
    const double g[3] = {0.0, 0.0, -1.0};
    double gc[3];
    if (rt.small) { gc[0] = g[0] + (ext[4] * g[2] - ext[5] * g[1]); gc[1] = g[1] + (ext[5] * g[0] - ext[3] * g[2]); gc[2] = g[2] + (ext[3] * g[1] - ext[4] * g[0]); }
    else for (int a = 0; a < 3; ++a) gc[a] = rt.R[3 * a] * g[0] + rt.R[3 * a + 1] * g[1] + rt.R[3 * a + 2] * g[2];
    for (int k = 0; k < 3; ++k) v[k] = gc[k] - prior[k];
    if (want_jac) rotation_dq_dw(ext + 3, g, rt, D);
  } else {
    col0 = 3;
    double qc[4], qp[4], qe[4];
    so3_exp_quat(ext + 3, qc);
    so3_exp_quat(prior, qp);
    qp[1] = -qp[1]; qp[2] = -qp[2]; qp[3] = -qp[3];
    quat_mul_so3(qc, qp, qe);
    so3_log_quat(qe, v);
    if (want_jac) {
      double Ji[9], Jl[9];
      so3_left_jacobian_inverse(v, Ji);
      so3_left_jacobian(ext + 3, Jl);
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) D[3 * a + b] = (Ji[3 * a] * Jl[b] + Ji[3 * a + 1] * Jl[3 + b]) + Ji[3 * a + 2] * Jl[6 + b];
    }
  }
  for (int a = 0; a < 3; ++a) r[a] = (v[0] * S[3 * a] + v[1] * S[3 * a + 1]) + v[2] * S[3 * a + 2];
  if (!want_jac) return;
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) {   // (constant indices: J[6 a + col0 + b] with a run-time col0 made J a scratch array in its callers)
      const double val = (S[3 * a] * D[b] + S[3 * a + 1] * D[3 + b]) + S[3 * a + 2] * D[6 + b];
      J[6 * a + b] = col0 == 0 ? val : 0.0;
      J[6 * a + 3 + b] = col0 == 3 ? val : 0.0;
    }
}

}  // namespace thip
