// ba_oracle.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Scalar FP64 CPU restatement of the reference's bundle-adjustment hot path,
// used as the parity checker for the HIP kernels and as the "port" CPU baseline
// in bench.py.  Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may load it.
//
// PARITY STATUS: the per-observation functor (residual + forward-mode "Jet"
// Jacobian, exactly how Ceres differentiates ReprojectionError) is pinned by
// finite differences / scipy in tests/.  The LM trajectory and the Schur
// solve restate Ceres Solver 2.2 semantics from its public documentation --
// Ceres is NOT in /root/reference and not installed here, and the reference's
// own tests hold no golden LM traces -> "parity unpinned" for
// the LM trajectory; pinned only through the reference's threshold tests
// (bundle_adjustment_test.cc:108-114,200-206) re-stated in tests/.
//
// What each part follows (paths relative to the reference root):
//   reprojection_error()      src/theia/sfm/camera/reprojection_error.h:54-110
//   angle_axis_rotate_point() ceres/rotation.h AngleAxisRotatePoint (Ceres 2.x)
//   pinhole_project()         camera/pinhole_camera_model.h:181-211,243-260
//   double_sphere_project()   camera/double_sphere_camera_model.h:160-249
//   other models              camera/{fisheye,fov,division_undistortion,
//                             extended_unified,pinhole_radial_tangential,
//                             orthographic}_camera_model.h  DistortPoint bodies
//   loss_evaluate()           bundle_adjustment/create_loss_function.cc:44-76,
//                             loss_functions.cc:40-44, ceres/loss_function.cc
//   sphere manifold           ceres/sphere_manifold.h (bundle_adjuster.cc:538-545)
//   problem structure         bundle_adjuster.cc:116-221,357-460,547-577
//   LM loop                   ceres trust_region_minimizer.cc +
//                             levenberg_marquardt_strategy.cc (via
//                             bundle_adjuster.cc:63-89,339)
//   Schur elimination         ceres schur_eliminator_impl.h, group 0 = points
//                             (bundle_adjuster.cc:565-577)
#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <type_traits>
#include <vector>

#include <omp.h>

namespace {
int g_armijo_checks = 0, g_armijo_failures = 0;   // LM iterations with free (= bounded) intrinsics / those whose full step fails Armijo's test

// Threads of the "all host cores" CPU baseline (bench.py cpu_baseline): 1 = the serial code path every test and
// golden fixture runs (bitwise unchanged); > 1 parallelises the per-observation / per-point loops with OpenMP.
// Sums are then taken in a different (still fixed) order, i.e. results agree with the serial ones to rounding.
int g_threads = 1;
#define OBA_PAR_FOR _Pragma("omp parallel for schedule(static) num_threads(g_threads) if (g_threads > 1)")

// ------------------------------------------------------------------ Jets
// Forward-mode dual number, the same construction Ceres' AutoDiffCostFunction
// uses (ceres/jet.h): value + N partial derivatives.
template <int N>
struct Jet {
  double a;
  double v[N];
  Jet() : a(0.0) { for (int i = 0; i < N; ++i) v[i] = 0.0; }
  Jet(double s) : a(s) { for (int i = 0; i < N; ++i) v[i] = 0.0; }  // NOLINT
  Jet(double s, int k) : a(s) { for (int i = 0; i < N; ++i) v[i] = 0.0; v[k] = 1.0; }
};
template <int N> inline Jet<N> operator+(const Jet<N>& f, const Jet<N>& g) {
  Jet<N> h; h.a = f.a + g.a; for (int i = 0; i < N; ++i) h.v[i] = f.v[i] + g.v[i]; return h; }
template <int N> inline Jet<N> operator-(const Jet<N>& f, const Jet<N>& g) {
  Jet<N> h; h.a = f.a - g.a; for (int i = 0; i < N; ++i) h.v[i] = f.v[i] - g.v[i]; return h; }
template <int N> inline Jet<N> operator-(const Jet<N>& f) {
  Jet<N> h; h.a = -f.a; for (int i = 0; i < N; ++i) h.v[i] = -f.v[i]; return h; }
template <int N> inline Jet<N> operator*(const Jet<N>& f, const Jet<N>& g) {
  Jet<N> h; h.a = f.a * g.a; for (int i = 0; i < N; ++i) h.v[i] = f.a * g.v[i] + f.v[i] * g.a; return h; }
template <int N> inline Jet<N> operator/(const Jet<N>& f, const Jet<N>& g) {
  Jet<N> h; const double gi = 1.0 / g.a; const double fg = f.a * gi; h.a = fg;
  for (int i = 0; i < N; ++i) h.v[i] = (f.v[i] - fg * g.v[i]) * gi; return h; }
template <int N> inline Jet<N> operator+(const Jet<N>& f, double s) { Jet<N> h = f; h.a += s; return h; }
template <int N> inline Jet<N> operator+(double s, const Jet<N>& f) { Jet<N> h = f; h.a += s; return h; }
template <int N> inline Jet<N> operator-(const Jet<N>& f, double s) { Jet<N> h = f; h.a -= s; return h; }
template <int N> inline Jet<N> operator-(double s, const Jet<N>& f) { Jet<N> h = -f; h.a += s; return h; }
template <int N> inline Jet<N> operator*(const Jet<N>& f, double s) {
  Jet<N> h; h.a = f.a * s; for (int i = 0; i < N; ++i) h.v[i] = f.v[i] * s; return h; }
template <int N> inline Jet<N> operator*(double s, const Jet<N>& f) { return f * s; }
template <int N> inline Jet<N> operator/(const Jet<N>& f, double s) { return f * (1.0 / s); }
template <int N> inline Jet<N> operator/(double s, const Jet<N>& g) { return Jet<N>(s) / g; }
template <int N> inline bool operator<(const Jet<N>& f, const Jet<N>& g) { return f.a < g.a; }
template <int N> inline bool operator<(const Jet<N>& f, double g) { return f.a < g; }
template <int N> inline bool operator>(const Jet<N>& f, const Jet<N>& g) { return f.a > g.a; }
template <int N> inline bool operator>(const Jet<N>& f, double g) { return f.a > g; }
template <int N> inline bool operator<=(const Jet<N>& f, const Jet<N>& g) { return f.a <= g.a; }
template <int N> inline bool operator<=(const Jet<N>& f, double g) { return f.a <= g; }
template <int N> inline bool operator>=(const Jet<N>& f, double g) { return f.a >= g; }
template <int N> inline Jet<N> jsqrt(const Jet<N>& f) {
  Jet<N> h; h.a = std::sqrt(f.a); const double d = 1.0 / (2.0 * h.a);
  for (int i = 0; i < N; ++i) h.v[i] = f.v[i] * d; return h; }
template <int N> inline Jet<N> jsin(const Jet<N>& f) {
  Jet<N> h; h.a = std::sin(f.a); const double c = std::cos(f.a);
  for (int i = 0; i < N; ++i) h.v[i] = c * f.v[i]; return h; }
template <int N> inline Jet<N> jcos(const Jet<N>& f) {
  Jet<N> h; h.a = std::cos(f.a); const double s = -std::sin(f.a);
  for (int i = 0; i < N; ++i) h.v[i] = s * f.v[i]; return h; }
template <int N> inline Jet<N> jtan(const Jet<N>& f) {
  Jet<N> h; h.a = std::tan(f.a); const double d = 1.0 + h.a * h.a;
  for (int i = 0; i < N; ++i) h.v[i] = d * f.v[i]; return h; }
template <int N> inline Jet<N> jatan(const Jet<N>& f) {
  Jet<N> h; h.a = std::atan(f.a); const double d = 1.0 / (1.0 + f.a * f.a);
  for (int i = 0; i < N; ++i) h.v[i] = d * f.v[i]; return h; }
template <int N> inline Jet<N> jatan2(const Jet<N>& g, const Jet<N>& f) {
  // atan2(g, f): d = (f dg - g df) / (f^2+g^2)   (ceres/jet.h)
  Jet<N> h; h.a = std::atan2(g.a, f.a); const double d = 1.0 / (f.a * f.a + g.a * g.a);
  for (int i = 0; i < N; ++i) h.v[i] = d * (f.a * g.v[i] - g.a * f.v[i]); return h; }
template <int N> inline Jet<N> jabs(const Jet<N>& f) { return f.a < 0.0 ? -f : f; }
inline double jsqrt(double x) { return std::sqrt(x); }
inline double jsin(double x) { return std::sin(x); }
inline double jcos(double x) { return std::cos(x); }
inline double jtan(double x) { return std::tan(x); }
inline double jatan(double x) { return std::atan(x); }
inline double jatan2(double y, double x) { return std::atan2(y, x); }
inline double jabs(double x) { return std::fabs(x); }
inline double scalar_of(double x) { return x; }
template <int N> inline double scalar_of(const Jet<N>& x) { return x.a; }

// ------------------------------------------------------- rotation (Ceres)
// ceres/rotation.h AngleAxisRotatePoint: Rodrigues for theta^2 > DBL_EPSILON,
// first-order  p + w x p  otherwise.
template <typename T>
inline void angle_axis_rotate_point(const T aa[3], const T pt[3], T result[3]) {
  const T theta2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
  if (theta2 > std::numeric_limits<double>::epsilon()) {
    const T theta = jsqrt(theta2);
    const T costheta = jcos(theta);
    const T sintheta = jsin(theta);
    const T theta_inverse = 1.0 / theta;
    const T w[3] = {aa[0] * theta_inverse, aa[1] * theta_inverse, aa[2] * theta_inverse};
    const T w_cross_pt[3] = {w[1] * pt[2] - w[2] * pt[1], w[2] * pt[0] - w[0] * pt[2],
                             w[0] * pt[1] - w[1] * pt[0]};
    const T tmp = (w[0] * pt[0] + w[1] * pt[1] + w[2] * pt[2]) * (1.0 - costheta);
    result[0] = pt[0] * costheta + w_cross_pt[0] * sintheta + w[0] * tmp;
    result[1] = pt[1] * costheta + w_cross_pt[1] * sintheta + w[1] * tmp;
    result[2] = pt[2] * costheta + w_cross_pt[2] * sintheta + w[2] * tmp;
  } else {
    const T w_cross_pt[3] = {aa[1] * pt[2] - aa[2] * pt[1], aa[2] * pt[0] - aa[0] * pt[2],
                             aa[0] * pt[1] - aa[1] * pt[0]};
    result[0] = pt[0] + w_cross_pt[0];
    result[1] = pt[1] + w_cross_pt[1];
    result[2] = pt[2] + w_cross_pt[2];
  }
}

// --------------------------------------------------------- camera models
enum {
  CAM_PINHOLE = 0, CAM_PINHOLE_RADIAL_TANGENTIAL = 1, CAM_FISHEYE = 2, CAM_FOV = 3,
  CAM_DIVISION_UNDISTORTION = 4, CAM_DOUBLE_SPHERE = 5, CAM_EXTENDED_UNIFIED = 6,
  CAM_ORTHOGRAPHIC = 7
};
const int kMaxIntr = 10;

inline int intrinsics_size(int model) {
  switch (model) {
    case CAM_PINHOLE: return 7;                    // pinhole_camera_model.h:84
    case CAM_PINHOLE_RADIAL_TANGENTIAL: return 10; // pinhole_radial_tangential_camera_model.h
    case CAM_FISHEYE: return 9;                    // fisheye_camera_model.h
    case CAM_FOV: return 5;                        // fov_camera_model.h
    case CAM_DIVISION_UNDISTORTION: return 5;      // division_undistortion_camera_model.h
    case CAM_DOUBLE_SPHERE: return 7;              // double_sphere_camera_model.h:64
    case CAM_EXTENDED_UNIFIED: return 7;           // extended_unified_camera_model.h
    case CAM_ORTHOGRAPHIC: return 7;               // orthographic_camera_model.h
  }
  return -1;
}

// Common affine stage: u = f x' + s y' + cx ; v = f a y' + cy
// (pinhole_camera_model.h:205-208).  Layout [f, aspect, skew, cx, cy, ...].
template <typename T>
inline void affine_stage(const T* k, const T d[2], T* pixel) {
  pixel[0] = k[0] * d[0] + k[2] * d[1] + k[3];
  pixel[1] = k[0] * k[1] * d[1] + k[4];
}

// pinhole_camera_model.h:181-211 + DistortPoint :243-260
template <typename T>
inline bool pinhole_project(const T* k, const T* p, T* pixel) {
  const T n[2] = {p[0] / p[2], p[1] / p[2]};
  const T r_sq = n[0] * n[0] + n[1] * n[1];
  const T d = 1.0 + r_sq * (k[5] + k[6] * r_sq);
  const T dp[2] = {n[0] * d, n[1] * d};
  affine_stage(k, dp, pixel);
  return true;
}

// double_sphere_camera_model.h:160-249  ([.., xi(5), alpha(6)])
template <typename T>
inline bool double_sphere_project(const T* k, const T* p, T* pixel) {
  const T& alpha = k[6];
  const T& xi = k[5];
  const T xx = p[0] * p[0], yy = p[1] * p[1], zz = p[2] * p[2];
  const T r2 = xx + yy;
  const T d1_2 = r2 + zz;
  const T d1 = jsqrt(d1_2);
  const T w1 = alpha > 0.5 ? (1.0 - alpha) / alpha : alpha / (1.0 - alpha);
  const T w2 = (w1 + xi) / jsqrt(2.0 * w1 * xi + xi * xi + 1.0);
  bool ok = true;
  if (p[2] <= -w2 * d1) ok = false;  // reference returns before writing the pixel
  const T kk_ = xi * d1 + p[2];
  const T d2 = jsqrt(r2 + kk_ * kk_);
  const T norm = alpha * d2 + (1.0 - alpha) * kk_;
  const T dp[2] = {p[0] / norm, p[1] / norm};
  affine_stage(k, dp, pixel);
  return ok;
}

// extended_unified_camera_model.h:215-249 ([.., alpha(5), beta(6)])
template <typename T>
inline bool eucm_project(const T* k, const T* p, T* pixel) {
  const T& alpha = k[5];
  const T& beta = k[6];
  const T xx = p[0] * p[0], yy = p[1] * p[1], zz = p[2] * p[2];
  const T r2 = xx + yy;
  const T rho2 = beta * r2 + zz;
  const T rho = jsqrt(rho2);
  const T norm = alpha * rho + (1.0 - alpha) * p[2];
  T dp[2];
  bool zero = false;
  if (norm < 1e-3) zero = true;
  if (!zero && alpha > 0.5) {
    const T zn = p[2] / norm;
    const T c = (alpha - 1.0) / (alpha + alpha - 1.0);
    if (zn < c) zero = true;  // reference: zero output, returns true
  }
  if (zero) { dp[0] = T(0.0); dp[1] = T(0.0); }
  else { dp[0] = p[0] / norm; dp[1] = p[1] / norm; }
  affine_stage(k, dp, pixel);
  return true;
}

// fisheye_camera_model.h:163-272 ([f,a,s,cx,cy,k1..k4])
template <typename T>
inline bool fisheye_project(const T* k, const T* p, T* pixel) {
  const T r_sq = p[0] * p[0] + p[1] * p[1];
  T dp[2];
  if (r_sq < 1e-8) {
    dp[0] = p[0]; dp[1] = p[1];
    // reference DistortPoint: identity for tiny radius (uses x, y directly)
  } else {
    const T r = jsqrt(r_sq);
    const T theta = jatan2(r, jabs(p[2]));
    const T t2 = theta * theta, t4 = t2 * t2, t6 = t4 * t2, t8 = t4 * t4;
    const T theta_d = theta * (1.0 + k[5] * t2 + k[6] * t4 + k[7] * t6 + k[8] * t8);
    dp[0] = theta_d * p[0] / r;
    dp[1] = theta_d * p[1] / r;
    if (p[2] < 0.0) { dp[0] = -dp[0]; dp[1] = -dp[1]; }
  }
  affine_stage(k, dp, pixel);
  return true;
}

// pinhole_radial_tangential_camera_model.h:191-296 ([f,a,s,cx,cy,k1,k2,k3,t1,t2])
template <typename T>
inline bool radtan_project(const T* k, const T* p, T* pixel) {
  const T x = p[0] / p[2], y = p[1] / p[2];
  const T r_sq = x * x + y * y;
  const T d = 1.0 + r_sq * (k[5] + r_sq * (k[6] + k[7] * r_sq));
  const T dp[2] = {x * d + 2.0 * k[8] * x * y + k[9] * (r_sq + 2.0 * x * x),
                   y * d + 2.0 * k[9] * x * y + k[8] * (r_sq + 2.0 * y * y)};
  affine_stage(k, dp, pixel);
  return true;
}

// fov_camera_model.h:156-258 ([f, a, cx, cy, omega]; no skew)
template <typename T>
inline bool fov_project(const T* k, const T* p, T* pixel) {
  const T x = p[0] / p[2], y = p[1] / p[2];
  const T& omega = k[4];
  const T r_u_sq = x * x + y * y;
  T r_d;
  if (omega < 1e-3) {
    r_d = (omega * omega * r_u_sq) / 3.0 - omega * omega / 12.0 + 1.0;
  } else if (r_u_sq < 1e-3) {
    const T th = jtan(omega / 2.0);
    r_d = (-2.0 * th * (4.0 * r_u_sq * th * th - 3.0)) / (3.0 * omega);
  } else {
    const T r_u = jsqrt(r_u_sq);
    r_d = jatan(2.0 * r_u * jtan(omega / 2.0)) / (r_u * omega);
  }
  pixel[0] = k[0] * (r_d * x) + k[2];
  pixel[1] = k[0] * k[1] * (r_d * y) + k[3];
  return true;
}

// division_undistortion_camera_model.h:173-231,263-297 ([f, a, cx, cy, k]):
// the distortion acts on focal-scaled coordinates.
template <typename T>
inline bool division_project(const T* k, const T* p, T* pixel) {
  const T ux = k[0] * (p[0] / p[2]);
  const T uy = k[0] * k[1] * (p[1] / p[2]);
  const T r_u_sq = ux * ux + uy * uy;
  const T denom = 2.0 * k[4] * r_u_sq;
  const T inner_sqrt = 1.0 - 4.0 * k[4] * r_u_sq;
  if (scalar_of(jabs(denom)) < std::numeric_limits<double>::epsilon() || inner_sqrt < 0.0) {
    pixel[0] = ux + k[2]; pixel[1] = uy + k[3];
  } else {
    const T scale = (1.0 - jsqrt(inner_sqrt)) / denom;
    pixel[0] = ux * scale + k[2]; pixel[1] = uy * scale + k[3];
  }
  return true;
}

// orthographic_camera_model.h:162-240 ([f, a, s, cx, cy, k1, k2]): no depth division
template <typename T>
inline bool ortho_project(const T* k, const T* p, T* pixel) {
  const T r_sq = p[0] * p[0] + p[1] * p[1];
  const T d = 1.0 + r_sq * (k[5] + k[6] * r_sq);
  const T dp[2] = {p[0] * d, p[1] * d};
  affine_stage(k, dp, pixel);
  return true;
}

template <typename T>
inline bool project(int model, const T* k, const T* p, T* pixel) {
  switch (model) {
    case CAM_PINHOLE: return pinhole_project(k, p, pixel);
    case CAM_DOUBLE_SPHERE: return double_sphere_project(k, p, pixel);
    case CAM_EXTENDED_UNIFIED: return eucm_project(k, p, pixel);
    case CAM_FISHEYE: return fisheye_project(k, p, pixel);
    case CAM_PINHOLE_RADIAL_TANGENTIAL: return radtan_project(k, p, pixel);
    case CAM_FOV: return fov_project(k, p, pixel);
    case CAM_DIVISION_UNDISTORTION: return division_project(k, p, pixel);
    case CAM_ORTHOGRAPHIC: return ortho_project(k, p, pixel);
    default: pixel[0] = T(0.0); pixel[1] = T(0.0); return false;
  }
}

const int kDepthRow = 1000;   // pseudo camera model of an observation row that is a depth prior

// reprojection_error.h:54-110.  Returns the functor's boolean; the residual is
// written even when the model reports "invalid" (:100-109).
template <typename T>
inline bool reprojection_error(int model, const T* ext, const T* intr, const T* X,
                               const double uv[2], const double sqrt_info[2], T* res) {
  const T adj[3] = {X[0] - X[3] * ext[0], X[1] - X[3] * ext[1], X[2] - X[3] * ext[2]};
  const T sq = adj[0] * adj[0] + adj[1] * adj[1] + adj[2] * adj[2];
  if (sq < 1e-8) return false;
  T rot[3];
  angle_axis_rotate_point(ext + 3, adj, rot);
  if (model == kDepthRow) {
    // DepthPriorError (depth_prior_error.h): sqrt_information * (rotated_point[2] - depth_prior); one residual,
    // carried as the first of the two rows of an observation (the second is identically zero)
    res[0] = sqrt_info[0] * (rot[2] - uv[0]);
    res[1] = T(0.0);
    return true;
  }
  T pix[2];
  const bool ok = project(model, intr, rot, pix);
  res[0] = sqrt_info[0] * (pix[0] - uv[0]);
  res[1] = sqrt_info[1] * (pix[1] - uv[1]);
  return ok;
}

// ----------------------------------------------------------------- losses
enum { LOSS_TRIVIAL = 0, LOSS_HUBER, LOSS_SOFTLONE, LOSS_CAUCHY, LOSS_ARCTAN, LOSS_TUKEY,
       LOSS_TRUNCATED };
// ceres/loss_function.cc (Ceres 2.x) + theia TruncatedLoss (loss_functions.cc:40-44)
inline void loss_evaluate(int type, double a, double s, double rho[3]) {
  const double kMin = std::numeric_limits<double>::min();
  switch (type) {
    case LOSS_HUBER: {
      const double b = a * a;
      if (s > b) { const double r = std::sqrt(s); rho[0] = 2.0 * a * r - b;
        rho[1] = std::max(kMin, a / r); rho[2] = -rho[1] / (2.0 * s); }
      else { rho[0] = s; rho[1] = 1.0; rho[2] = 0.0; }
      break; }
    case LOSS_SOFTLONE: {
      const double b = a * a, c = 1.0 / b;
      const double sum = 1.0 + s * c, tmp = std::sqrt(sum);
      rho[0] = 2.0 * b * (tmp - 1.0); rho[1] = std::max(kMin, 1.0 / tmp);
      rho[2] = -(c * rho[1]) / (2.0 * sum);
      break; }
    case LOSS_CAUCHY: {
      const double b = a * a, c = 1.0 / b;
      const double sum = 1.0 + s * c, inv = 1.0 / sum;
      rho[0] = b * std::log(sum); rho[1] = std::max(kMin, inv); rho[2] = -c * (inv * inv);
      break; }
    case LOSS_ARCTAN: {
      const double b = 1.0 / (a * a);
      const double sum = 1.0 + s * s * b, inv = 1.0 / sum;
      rho[0] = a * std::atan2(s, a); rho[1] = std::max(kMin, inv);
      rho[2] = -2.0 * s * b * (inv * inv);
      break; }
    case LOSS_TUKEY: {
      const double a2 = a * a;
      if (s <= a2) { const double value = 1.0 - s / a2, value_sq = value * value;
        rho[0] = a2 / 3.0 * (1.0 - value_sq * value); rho[1] = value_sq;
        rho[2] = -2.0 / a2 * value; }
      else { rho[0] = a2 / 3.0; rho[1] = 0.0; rho[2] = 0.0; }
      break; }
    case LOSS_TRUNCATED: {
      const double se = a * a;
      rho[0] = std::min(s, se); rho[1] = s < se ? 1.0 : 0.0; rho[2] = 0.0;
      break; }
    default: rho[0] = s; rho[1] = 1.0; rho[2] = 0.0;
  }
}

// ------------------------------------------------------ sphere manifold<4>
// ceres/internal/householder_vector.h + sphere_manifold_functions.h (2.2).
inline void householder4(const double x[4], double v[4], double* beta) {
  const double sigma = x[0] * x[0] + x[1] * x[1] + x[2] * x[2];
  v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = 1.0;
  *beta = 0.0;
  if (sigma <= std::numeric_limits<double>::epsilon()) { if (x[3] < 0.0) *beta = 2.0; return; }
  const double mu = std::sqrt(x[3] * x[3] + sigma);
  double v_pivot = 1.0;
  if (x[3] <= 0.0) v_pivot = x[3] - mu; else v_pivot = -sigma / (x[3] + mu);
  *beta = 2.0 * v_pivot * v_pivot / (sigma + v_pivot * v_pivot);
  v[0] /= v_pivot; v[1] /= v_pivot; v[2] /= v_pivot;
}
// 4x3 row-major: J = |x| * (I - beta v v^T)[:, 0:3]
inline void sphere_plus_jacobian(const double x[4], double J[12]) {
  double v[4], beta; householder4(x, v, &beta);
  const double nx = std::sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3]);
  for (int r = 0; r < 4; ++r) for (int c = 0; c < 3; ++c)
    J[r * 3 + c] = nx * ((r == c ? 1.0 : 0.0) - beta * v[r] * v[c]);
}
inline void sphere_plus(const double x[4], const double d[3], double out[4]) {
  const double nd = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  if (nd == 0.0) { for (int i = 0; i < 4; ++i) out[i] = x[i]; return; }
  double v[4], beta; householder4(x, v, &beta);
  const double nx = std::sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3]);
  const double sbd = std::sin(nd) / nd;
  const double y[4] = {sbd * d[0], sbd * d[1], sbd * d[2], std::cos(nd)};
  const double vty = v[0] * y[0] + v[1] * y[1] + v[2] * y[2] + v[3] * y[3];
  for (int i = 0; i < 4; ++i) out[i] = nx * (y[i] - v[i] * (beta * vty));
}

// ----------------------------------------------------------------- camera priors
// position_error.h:51-59, gravity_error.h:51-65, orientation_error.h:53-64 as one templated function;
// the SO(3) exp / log / product follow Sophus (so3.hpp: expAndTheta, logAndTheta, operator* with its
// first-order renormalisation), which is what the reference's orientation prior is autodiff'ed through
// (Sophus is not under /root/reference: restated from its published source).
template <typename T>
inline void so3_exp_quat(const T w[3], T q[4]) {   // q = [w, x, y, z]
  const T theta_sq = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  T imag, real;
  if (scalar_of(theta_sq) < 1e-10 * 1e-10) {
    const T theta_po4 = theta_sq * theta_sq;
    imag = T(0.5) - theta_sq * (1.0 / 48.0) + theta_po4 * (1.0 / 3840.0);
    real = T(1.0) - theta_sq * (1.0 / 8.0) + theta_po4 * (1.0 / 384.0);
  } else {
    const T theta = jsqrt(theta_sq);
    const T half = theta * 0.5;
    imag = jsin(half) / theta;
    real = jcos(half);
  }
  q[0] = real; q[1] = imag * w[0]; q[2] = imag * w[1]; q[3] = imag * w[2];
}
template <typename T>
inline void so3_log_quat(const T q[4], T t[3]) {
  const T sn = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  const T w = q[0];
  T two_atan_nbyw_by_n;
  if (scalar_of(sn) < 1e-10 * 1e-10) {
    two_atan_nbyw_by_n = T(2.0) / w - (sn * (2.0 / 3.0)) / (w * w * w);
  } else {
    const T n = jsqrt(sn);
    const T at = (scalar_of(w) < 0.0) ? jatan2(-n, -w) : jatan2(n, w);
    two_atan_nbyw_by_n = (at * 2.0) / n;
  }
  t[0] = two_atan_nbyw_by_n * q[1]; t[1] = two_atan_nbyw_by_n * q[2]; t[2] = two_atan_nbyw_by_n * q[3];
}
template <typename T>
inline void quat_mul_so3(const T a[4], const T b[4], T o[4]) {   // Sophus SO3Base::operator*
  o[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  o[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  o[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  o[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  const T sq = o[0] * o[0] + o[1] * o[1] + o[2] * o[2] + o[3] * o[3];
  if (scalar_of(sq) != 1.0) {
    const T scale = T(2.0) / (T(1.0) + sq);
    for (int k = 0; k < 4; ++k) o[k] = o[k] * scale;
  }
}
// kind: 1 position, 2 gravity, 4 orientation.  ext = [position(3) | angle-axis(3)]; r = 3 residuals.
template <typename T>
inline void camera_prior_residual(int kind, const T* ext, const double* prior, const double* S, T* r) {
  T v[3];
  if (kind == 1) {
    for (int k = 0; k < 3; ++k) v[k] = T(prior[k]) - ext[k];
  } else if (kind == 2) {
    const T gw[3] = {T(0.0), T(0.0), T(-1.0)};
    T gc[3];
    angle_axis_rotate_point(ext + 3, gw, gc);
    for (int k = 0; k < 3; ++k) v[k] = gc[k] - prior[k];
  } else {
    T qc[4], qp[4], qe[4];
    const T wp[3] = {T(prior[0]), T(prior[1]), T(prior[2])};
    so3_exp_quat(ext + 3, qc);
    so3_exp_quat(wp, qp);
    qp[1] = -qp[1]; qp[2] = -qp[2]; qp[3] = -qp[3];   // inverse = conjugate
    quat_mul_so3(qc, qp, qe);
    so3_log_quat(qe, v);
  }
  for (int a = 0; a < 3; ++a) r[a] = v[0] * S[3 * a] + v[1] * S[3 * a + 1] + v[2] * S[3 * a + 2];
}

}  // namespace

// ===================================================================== C API
extern "C" {

// Same field layout as theia_ba_problem (include/theia_hip.h) so the tests can
// hand one ctypes structure to both sides; defined independently here.
struct oba_problem {
  int32_t num_cameras, num_groups, num_points, flags;
  int64_t num_obs;
  double* cam_ext; double* intrinsics; const int32_t* group_model; const int32_t* cam_group;
  const uint8_t* cam_const; const uint8_t* group_const;
  double* points; const uint8_t* point_const;
  const double* obs_uv; const double* obs_sqrt_info; const int32_t* obs_cam; const int32_t* obs_pt;
  const uint8_t* cam_prior_mask;
  const double* cam_position_prior; const double* cam_position_prior_sqrt_info;
  const double* cam_gravity_prior; const double* cam_gravity_prior_sqrt_info;
  const double* cam_orientation_prior; const double* cam_orientation_prior_sqrt_info;
  const uint8_t* obs_kind;   // 1 = DepthPriorError row (depth_prior_error.h), NULL / 0 = reprojection error
};
struct oba_options {
  int32_t loss_function_type, intrinsics_to_optimize, max_num_iterations,
      use_homogeneous_point_parametrization, constant_camera_orientation,
      constant_camera_position, orthographic_camera, use_inner_iterations, verbose, prior_mask;
  double robust_loss_width, function_tolerance, gradient_tolerance, parameter_tolerance,
      max_trust_region_radius, max_solver_time_in_seconds, robust_loss_width_depth_prior;
};
struct oba_summary {
  int32_t success, termination_type, num_iterations, num_successful_steps;
  double initial_cost, final_cost, setup_time_in_seconds, solve_time_in_seconds;
  int32_t trace_capacity, trace_size;
  double* trace_cost; double* trace_gradient_max_norm; double* trace_step_norm;
  double* trace_radius; int32_t* trace_accepted;
  double time_linearize, time_solve_reduced, time_backsub;
  double time_kernel_linearize; int32_t num_linearize_launches, reserved1;
};

// Single observation: residual + ambient Jacobians (what Ceres' autodiff
// returns): J_ext 2x6, J_intr 2xK, J_pt 2x4, all row-major.  Returns the
// functor's boolean (0/1).
int oracle_reprojection_error(int model, const double* ext, const double* intr, const double* X,
                              const double* uv, const double* sqrt_info, double* res,
                              double* J_ext, double* J_intr, double* J_pt) {
  const int K = intrinsics_size(model);
  if (K < 0) return -1;
  const int N = 6 + kMaxIntr + 4;
  typedef Jet<N> J;
  J e[6], k[kMaxIntr], x[4], r[2];
  for (int i = 0; i < 6; ++i) e[i] = J(ext[i], i);
  for (int i = 0; i < K; ++i) k[i] = J(intr[i], 6 + i);
  for (int i = 0; i < 4; ++i) x[i] = J(X[i], 6 + kMaxIntr + i);
  const double one[2] = {1.0, 1.0};
  const bool ok = reprojection_error<J>(model, e, k, x, uv, sqrt_info ? sqrt_info : one, r);
  for (int a = 0; a < 2; ++a) {
    res[a] = r[a].a;
    if (J_ext) for (int i = 0; i < 6; ++i) J_ext[a * 6 + i] = r[a].v[i];
    if (J_intr) for (int i = 0; i < K; ++i) J_intr[a * K + i] = r[a].v[6 + i];
    if (J_pt) for (int i = 0; i < 4; ++i) J_pt[a * 4 + i] = r[a].v[6 + kMaxIntr + i];
  }
  return ok ? 1 : 0;
}

// Camera::ProjectPoint (src/theia/sfm/camera/camera.cc:206-216): adjusted = X.head<3>() - X[3] position, rotated by the
// angle-axis, pixel = CameraToPixelCoordinates(rotated); returns the depth rotated.z / X[3].  Used by the RANSAC oracle's
// TriangulationEstimator::Error (ransac_oracle.cpp).
double oracle_camera_project_point(int model, const double* ext, const double* intr, const double* X, double* pixel) {
  const double adj[3] = {X[0] - X[3] * ext[0], X[1] - X[3] * ext[1], X[2] - X[3] * ext[2]};
  double rot[3];
  angle_axis_rotate_point(ext + 3, adj, rot);
  if (pixel) project(model, intr, rot, pixel);
  return rot[2] / X[3];
}

void oracle_loss_evaluate(int type, double a, double s, double* rho) { loss_evaluate(type, a, s, rho); }
void oracle_sphere_plus(const double* x, const double* d, double* out) { sphere_plus(x, d, out); }
void oracle_sphere_plus_jacobian(const double* x, double* J) { sphere_plus_jacobian(x, J); }

}  // extern "C"

// ============================================================ BA solver state
namespace {

// Which intrinsics are FREE for a model under an OptimizeIntrinsicsType mask:
// GetSubsetFromOptimizeIntrinsicsType of every *_camera_model.cc (e.g.
// pinhole_camera_model.cc:132-162).  Returns a bit mask over the K parameters.
inline unsigned intrinsics_free_mask(int model, int opt) {
  const bool noskew = (model == CAM_FOV || model == CAM_DIVISION_UNDISTORTION);
  unsigned m = 0;
  if (opt & 0x01) m |= 1u << 0;                                   // FOCAL_LENGTH
  if (opt & 0x02) m |= 1u << 1;                                   // ASPECT_RATIO
  if ((opt & 0x04) && !noskew) m |= 1u << 2;                      // SKEW
  if (opt & 0x08) m |= noskew ? (3u << 2) : (3u << 3);            // PRINCIPAL_POINTS
  if (opt & 0x10) {                                               // RADIAL_DISTORTION
    switch (model) {
      case CAM_PINHOLE: case CAM_DOUBLE_SPHERE: case CAM_EXTENDED_UNIFIED: case CAM_ORTHOGRAPHIC: m |= 3u << 5; break;
      case CAM_PINHOLE_RADIAL_TANGENTIAL: m |= 7u << 5; break;
      case CAM_FISHEYE: m |= 15u << 5; break;
      case CAM_FOV: case CAM_DIVISION_UNDISTORTION: m |= 1u << 4; break;
    }
  }
  if ((opt & 0x20) && model == CAM_PINHOLE_RADIAL_TANGENTIAL) m |= 3u << 8;  // TANGENTIAL_DISTORTION
  return m;
}

// bundle_adjuster.cc:406-427: focal >= 1; double sphere xi in [-1,1], alpha in
// [0,1]; EUCM alpha in [0,1], beta >= 0.1.  (Ceres: constrained problem; this
// restatement projects the LM step onto the box, see DESIGN.md deviations.)
inline void project_intrinsics_to_bounds(int model, double* k) {
  if (k[0] < 1.0) k[0] = 1.0;
  if (model == CAM_DOUBLE_SPHERE) { k[5] = std::min(1.0, std::max(-1.0, k[5])); k[6] = std::min(1.0, std::max(0.0, k[6])); }
  if (model == CAM_EXTENDED_UNIFIED) { k[5] = std::min(1.0, std::max(0.0, k[5])); k[6] = std::max(0.1, k[6]); }
}

const int FW = kMaxIntr + 6;  // camera-side columns of one observation: [intrinsics(10) | extrinsics(6)]

struct Oracle {
  const oba_problem* P;
  oba_options O;
  int nc, np, ng; int64_t nobs;
  int pd;                        // point tangent dofs (3 manifold / 4 plain)
  std::vector<int> cam_red;      // camera -> reduced index or -1 (whole block const)
  std::vector<int> grp_red;      // intrinsics group -> reduced index or -1
  int ncv, ngv, ni;              // ni = 10 * ngv: intrinsics come first in the reduced system
  std::vector<uint8_t> cam_mask; // per camera: 6 bits, 1 = column frozen
  std::vector<unsigned> grp_free;// per group: bit q = parameter q free
  std::vector<uint8_t> pt_const;
  std::vector<uint8_t> obs_fixed;// residual blocks with only constant blocks
  double fixed_cost;
  // state
  std::vector<double> cam, pts, intr;      // current x
  std::vector<double> ccam, cpts, cintr;   // candidate
  // linearisation at x (Jacobi-scaled, loss-corrected, tangent space)
  std::vector<double> r;             // 2*nobs
  std::vector<double> F;             // nobs * 2 * FW   camera-side Jacobian [Jk | Jc]
  std::vector<double> Jp;            // nobs*2*pd
  std::vector<double> scale_f, scale_p;  // jacobi scaling of the reduced camera-side columns / points
  std::vector<double> diag_f, diag_p;    // clamped squared column norms
  // CSR by point
  std::vector<int64_t> pt_off; std::vector<int64_t> pt_obs;
  // reduced system
  std::vector<double> S, rhs, Vinv, yp, yc;
  int n() const { return ni + 6 * ncv; }
  // camera priors of variable cameras: 3 residuals each, Jacobian 3 x 6 wrt the extrinsics (masked, scaled)
  struct Prior { int cam, kind; const double* vec; const double* sqrt_info; double r[3]; double J[18]; };
  std::vector<Prior> priors;
};

// reduced index of camera-side column q (0..9 intrinsics, 10..15 extrinsics) of an observation, or -1
inline int fcol(const Oracle& o, int c, int g, int q) {
  if (q < kMaxIntr) { const int gr = o.grp_red[g]; return (gr >= 0 && ((o.grp_free[g] >> q) & 1u)) ? 10 * gr + q : -1; }
  const int rc = o.cam_red[c]; const int e = q - kMaxIntr;
  return (rc >= 0 && !((o.cam_mask[c] >> e) & 1)) ? o.ni + 6 * rc + e : -1;
}

bool evaluate(Oracle& o, const std::vector<double>& cam, const std::vector<double>& pts, const std::vector<double>& intrv,
              bool want_jac, double* cost_out) {
  const oba_problem& P = *o.P;
  double cost = 0.0;
  int bad = 0;
  const double one[2] = {1.0, 1.0};
  const int pd = o.pd;
#pragma omp parallel for schedule(static) num_threads(g_threads) if (g_threads > 1) reduction(+ : cost) reduction(+ : bad)
  for (int64_t i = 0; i < o.nobs; ++i) {
    bool ok = true;
    if (o.obs_fixed[i]) continue;
    const int c = P.obs_cam[i], p = P.obs_pt[i];
    const int g = P.cam_group[c];
    const bool depth_row = P.obs_kind && P.obs_kind[i];
    const int model = depth_row ? kDepthRow : P.group_model[g];
    const double loss_width = depth_row ? o.O.robust_loss_width_depth_prior : o.O.robust_loss_width;
    const double* intr = &intrv[(size_t)g * kMaxIntr];
    const double* si = P.obs_sqrt_info ? P.obs_sqrt_info + 2 * i : one;
    double res[2];
    if (!want_jac) {
      if (!reprojection_error<double>(model, &cam[6 * c], intr, &pts[4 * p], P.obs_uv + 2 * i, si, res))
        ok = false;
      double rho[3];
      loss_evaluate(o.O.loss_function_type, loss_width, res[0] * res[0] + res[1] * res[1], rho);
      cost += 0.5 * rho[0];
      if (!ok) bad++;
      continue;
    }
    typedef Jet<20> J;  // 6 extrinsics + 10 intrinsics + 4 point, as AutoDiffCostFunction<.., 2, 6, K, 4>
    J e[6], k[kMaxIntr], x[4], rr[2];
    for (int q = 0; q < 6; ++q) e[q] = J(cam[6 * c + q], q);
    for (int q = 0; q < kMaxIntr; ++q) k[q] = J(intr[q], 6 + q);
    for (int q = 0; q < 4; ++q) x[q] = J(pts[4 * p + q], 16 + q);
    if (!reprojection_error<J>(model, e, k, x, P.obs_uv + 2 * i, si, rr)) ok = false;
    res[0] = rr[0].a; res[1] = rr[1].a;
    const double s = res[0] * res[0] + res[1] * res[1];
    double rho[3];
    loss_evaluate(o.O.loss_function_type, loss_width, s, rho);
    cost += 0.5 * rho[0];
    // ceres/corrector.cc: every loss here has rho'' <= 0 -> residual and
    // Jacobian are both scaled by sqrt(rho').
    const double sr = std::sqrt(rho[1]);
    double PJ[12];
    if (pd == 3) sphere_plus_jacobian(&pts[4 * p], PJ);
    for (int a = 0; a < 2; ++a) {
      o.r[2 * i + a] = sr * res[a];
      double* Fr = &o.F[((size_t)i * 2 + a) * FW];
      for (int q = 0; q < kMaxIntr; ++q) Fr[q] = (fcol(o, c, g, q) >= 0) ? sr * rr[a].v[6 + q] : 0.0;
      for (int q = 0; q < 6; ++q) Fr[kMaxIntr + q] = (fcol(o, c, g, kMaxIntr + q) >= 0) ? sr * rr[a].v[q] : 0.0;
      for (int q = 0; q < pd; ++q) {
        double v;
        if (pd == 3) { v = 0.0; for (int t = 0; t < 4; ++t) v += rr[a].v[16 + t] * PJ[t * 3 + q]; }
        else v = rr[a].v[16 + q];
        o.Jp[(size_t)i * 2 * pd + a * pd + q] = o.pt_const[p] ? 0.0 : sr * v;
      }
    }
    if (!ok) bad++;
  }
  bool ok = bad == 0;
  // camera priors (no loss function: NULL in bundle_adjuster.cc:627-657)
  for (Oracle::Prior& pr : o.priors) {
    const int c = pr.cam;
    if (!want_jac) {
      double rr3[3];
      camera_prior_residual<double>(pr.kind, &cam[6 * c], pr.vec, pr.sqrt_info, rr3);
      cost += 0.5 * ((rr3[0] * rr3[0] + rr3[1] * rr3[1]) + rr3[2] * rr3[2]);
      continue;
    }
    typedef Jet<6> J6;
    J6 e[6], rr3[3];
    for (int q = 0; q < 6; ++q) e[q] = J6(cam[6 * c + q], q);
    camera_prior_residual<J6>(pr.kind, e, pr.vec, pr.sqrt_info, rr3);
    for (int a = 0; a < 3; ++a) {
      pr.r[a] = rr3[a].a;
      for (int q = 0; q < 6; ++q) pr.J[6 * a + q] = ((o.cam_mask[c] >> q) & 1) ? 0.0 : rr3[a].v[q];
    }
    cost += 0.5 * ((pr.r[0] * pr.r[0] + pr.r[1] * pr.r[1]) + pr.r[2] * pr.r[2]);
  }
  *cost_out = cost;
  return ok;
}

// squared column norms of the current (scaled or not) Jacobian
void column_norms(Oracle& o, std::vector<double>& nf_, std::vector<double>& np_) {
  const oba_problem& P = *o.P; const int pd = o.pd;
  std::fill(nf_.begin(), nf_.end(), 0.0); std::fill(np_.begin(), np_.end(), 0.0);
  if (g_threads > 1) {
    // points: one thread per point (CSR by point); camera side: per-thread vectors added in thread order
    OBA_PAR_FOR
    for (int p = 0; p < o.np; ++p)
      for (int64_t k = o.pt_off[p]; k < o.pt_off[p + 1]; ++k) { const int64_t i = o.pt_obs[k];
        for (int a = 0; a < 2; ++a) for (int q = 0; q < pd; ++q) { const double v = o.Jp[(size_t)i * 2 * pd + a * pd + q]; np_[(size_t)pd * p + q] += v * v; } }
    std::vector<std::vector<double>> loc(g_threads, std::vector<double>(nf_.size(), 0.0));
#pragma omp parallel num_threads(g_threads)
    {
      std::vector<double>& mine = loc[omp_get_thread_num()];
#pragma omp for schedule(static)
      for (int64_t i = 0; i < o.nobs; ++i) {
        if (o.obs_fixed[i]) continue;
        const int c = P.obs_cam[i], g = P.cam_group[c];
        for (int a = 0; a < 2; ++a) { const double* Fr = &o.F[((size_t)i * 2 + a) * FW];
          for (int q = 0; q < FW; ++q) { const int col = fcol(o, c, g, q); if (col >= 0) mine[col] += Fr[q] * Fr[q]; } }
      }
    }
    for (int t = 0; t < g_threads; ++t) for (size_t d = 0; d < nf_.size(); ++d) nf_[d] += loc[t][d];
  } else
  for (int64_t i = 0; i < o.nobs; ++i) {
    if (o.obs_fixed[i]) continue;
    const int c = P.obs_cam[i], p = P.obs_pt[i], g = P.cam_group[c];
    for (int a = 0; a < 2; ++a) {
      const double* Fr = &o.F[((size_t)i * 2 + a) * FW];
      for (int q = 0; q < FW; ++q) { const int col = fcol(o, c, g, q); if (col >= 0) nf_[col] += Fr[q] * Fr[q]; }
      for (int q = 0; q < pd; ++q) { const double v = o.Jp[(size_t)i * 2 * pd + a * pd + q]; np_[(size_t)pd * p + q] += v * v; }
    }
  }
  for (const Oracle::Prior& pr : o.priors) {
    const int base = o.ni + 6 * o.cam_red[pr.cam];
    for (int a = 0; a < 3; ++a) for (int q = 0; q < 6; ++q) nf_[base + q] += pr.J[6 * a + q] * pr.J[6 * a + q];
  }
}

void apply_scaling(Oracle& o) {
  const oba_problem& P = *o.P; const int pd = o.pd;
  OBA_PAR_FOR
  for (int64_t i = 0; i < o.nobs; ++i) {
    if (o.obs_fixed[i]) continue;
    const int c = P.obs_cam[i], p = P.obs_pt[i], g = P.cam_group[c];
    for (int a = 0; a < 2; ++a) {
      double* Fr = &o.F[((size_t)i * 2 + a) * FW];
      for (int q = 0; q < FW; ++q) { const int col = fcol(o, c, g, q); if (col >= 0) Fr[q] *= o.scale_f[col]; }
      for (int q = 0; q < pd; ++q) o.Jp[(size_t)i * 2 * pd + a * pd + q] *= o.scale_p[(size_t)pd * p + q];
    }
  }
  for (Oracle::Prior& pr : o.priors) {
    const int base = o.ni + 6 * o.cam_red[pr.cam];
    for (int a = 0; a < 3; ++a) for (int q = 0; q < 6; ++q) pr.J[6 * a + q] *= o.scale_f[base + q];
  }
}

// gradient in tangent space (unscaled): g = J^T r = (Js^T r) / scale ; returns max |g|
double compute_gradient(Oracle& o) {
  const oba_problem& P = *o.P; const int pd = o.pd;
  std::vector<double> gf(o.n(), 0.0), gp((size_t)pd * o.np, 0.0);
  if (g_threads > 1) {
    OBA_PAR_FOR
    for (int p = 0; p < o.np; ++p)
      for (int64_t k = o.pt_off[p]; k < o.pt_off[p + 1]; ++k) { const int64_t i = o.pt_obs[k];
        for (int a = 0; a < 2; ++a) for (int q = 0; q < pd; ++q) gp[(size_t)pd * p + q] += o.Jp[(size_t)i * 2 * pd + a * pd + q] * o.r[2 * i + a]; }
    std::vector<std::vector<double>> loc(g_threads, std::vector<double>(gf.size(), 0.0));
#pragma omp parallel num_threads(g_threads)
    {
      std::vector<double>& mine = loc[omp_get_thread_num()];
#pragma omp for schedule(static)
      for (int64_t i = 0; i < o.nobs; ++i) {
        if (o.obs_fixed[i]) continue;
        const int c = P.obs_cam[i], g = P.cam_group[c];
        for (int a = 0; a < 2; ++a) { const double ra = o.r[2 * i + a]; const double* Fr = &o.F[((size_t)i * 2 + a) * FW];
          for (int q = 0; q < FW; ++q) { const int col = fcol(o, c, g, q); if (col >= 0) mine[col] += Fr[q] * ra; } }
      }
    }
    for (int t = 0; t < g_threads; ++t) for (size_t d = 0; d < gf.size(); ++d) gf[d] += loc[t][d];
  } else
  for (int64_t i = 0; i < o.nobs; ++i) {
    if (o.obs_fixed[i]) continue;
    const int c = P.obs_cam[i], p = P.obs_pt[i], g = P.cam_group[c];
    for (int a = 0; a < 2; ++a) {
      const double ra = o.r[2 * i + a];
      const double* Fr = &o.F[((size_t)i * 2 + a) * FW];
      for (int q = 0; q < FW; ++q) { const int col = fcol(o, c, g, q); if (col >= 0) gf[col] += Fr[q] * ra; }
      for (int q = 0; q < pd; ++q) gp[(size_t)pd * p + q] += o.Jp[(size_t)i * 2 * pd + a * pd + q] * ra;
    }
  }
  for (const Oracle::Prior& pr : o.priors) {
    const int base = o.ni + 6 * o.cam_red[pr.cam];
    for (int a = 0; a < 3; ++a) for (int q = 0; q < 6; ++q) gf[base + q] += pr.J[6 * a + q] * pr.r[a];
  }
  double gmax = 0.0;
  for (int d = 0; d < o.n(); ++d) gmax = std::max(gmax, std::fabs(gf[d] / o.scale_f[d]));
  for (int p = 0; p < o.np; ++p) if (!o.pt_const[p])
    for (int q = 0; q < pd; ++q) gmax = std::max(gmax, std::fabs(gp[(size_t)pd * p + q] / o.scale_p[(size_t)pd * p + q]));
  return gmax;
}

// small SPD inverse via Cholesky (ceres InvertPSDMatrix, full-rank branch)
bool invert_spd(int n, const double* A, double* Ainv) {
  double L[16];
  for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) {
    double s = A[i * n + j];
    for (int k = 0; k < j; ++k) s -= L[i * n + k] * L[j * n + k];
    if (i == j) { if (!(s > 0.0)) return false; L[i * n + i] = std::sqrt(s); }
    else L[i * n + j] = s / L[j * n + j];
  }
  for (int c = 0; c < n; ++c) {
    double y[4];
    for (int i = 0; i < n; ++i) { double s = (i == c) ? 1.0 : 0.0;
      for (int k = 0; k < i; ++k) s -= L[i * n + k] * y[k]; y[i] = s / L[i * n + i]; }
    for (int i = n - 1; i >= 0; --i) { double s = y[i];
      for (int k = i + 1; k < n; ++k) s -= L[k * n + i] * Ainv[k * n + c]; Ainv[i * n + c] = s / L[i * n + i]; }
  }
  return true;
}

// In-place Cholesky of the lower triangle (row-major), then solve.  The factorisation only visits the ENVELOPE of
// the matrix (row i from its first structural non-zero f(i) on; the factor has the same envelope): on the banded
// reduced systems of camera rings this is O(n b^2) instead of O(n^3 / 3).  Entries skipped are exact zeros, so the
// factor is bitwise that of the plain dense loops; the backward substitution runs row-wise (contiguous memory) and
// therefore adds its terms in descending row order.
bool dense_cholesky_solve(int n, std::vector<double>& A, std::vector<double>& b) {
  std::vector<int> first(n);
  for (int i = 0; i < n; ++i) { const double* Ai = &A[(size_t)i * n]; int f = 0; while (f < i && Ai[f] == 0.0) ++f; first[i] = f; }
  for (int j = 0; j < n; ++j) {
    double* Aj = &A[(size_t)j * n];
    double d = Aj[j];
    for (int k = first[j]; k < j; ++k) d -= Aj[k] * Aj[k];
    if (!(d > 0.0) || !std::isfinite(d)) return false;
    const double ljj = std::sqrt(d);
    Aj[j] = ljj;
    const double inv = 1.0 / ljj;
    for (int i = j + 1; i < n; ++i) {
      if (first[i] > j) continue;   // (i, j) outside the envelope: stays zero
      double* Ai = &A[(size_t)i * n];
      double s = Ai[j];
      for (int k = std::max(first[i], first[j]); k < j; ++k) s -= Ai[k] * Aj[k];
      Ai[j] = s * inv;
    }
  }
  for (int i = 0; i < n; ++i) { double s = b[i]; const double* Ai = &A[(size_t)i * n];
    for (int k = first[i]; k < i; ++k) s -= Ai[k] * b[k]; b[i] = s / Ai[i]; }
  // backward substitution by columns of L^T = rows of L: x_i known -> subtract from the rows above inside the envelope
  for (int i = n - 1; i >= 0; --i) { const double* Ai = &A[(size_t)i * n];
    b[i] /= Ai[i];
    for (int k = first[i]; k < i; ++k) b[k] -= Ai[k] * b[i]; }
  return true;
}

// Build the reduced camera system for LM radius `radius` (Jacobi-scaled space)
// following ceres SchurEliminator: per point chunk
//   ete = sum E^T E + D_p^2 ; lhs -= (F^T E) ete^-1 (E^T F) ; rhs -= (F^T E) ete^-1 (E^T b)
// with lhs initialised to F^T F + D_c^2 (groups 1 and 2 = intrinsics and
// extrinsics blocks, bundle_adjuster.cc:547-563, are NOT eliminated).
// One point's elimination (ceres SchurEliminator::ChunkDiagonalBlockAndGradient / UpdateRhs / ChunkOuterProduct):
// V^-1 into o.Vinv, then S and rhs updates.
static bool eliminate_point(Oracle& o, int p, double radius, std::vector<double>& W, std::vector<int>& wc) {
  const oba_problem& P = *o.P; const int pd = o.pd; const int n = o.n();
  const int64_t b0 = o.pt_off[p], b1 = o.pt_off[p + 1];
  if (b0 == b1) return true;
  double V[16] = {0}, gp[4] = {0};
  for (int64_t t = b0; t < b1; ++t) { const int64_t i = o.pt_obs[t];
    const double* E = &o.Jp[(size_t)i * 2 * pd];
    for (int a = 0; a < pd; ++a) { for (int b = 0; b < pd; ++b) V[a * pd + b] += E[a] * E[b] + E[pd + a] * E[pd + b];
      gp[a] += E[a] * o.r[2 * i] + E[pd + a] * o.r[2 * i + 1]; } }
  for (int a = 0; a < pd; ++a) V[a * pd + a] += o.diag_p[(size_t)pd * p + a] / radius;
  double* Vi = &o.Vinv[(size_t)p * pd * pd];
  if (!invert_spd(pd, V, Vi)) return false;
  const int L = (int)(b1 - b0);
  W.assign((size_t)L * FW * pd, 0.0); wc.assign((size_t)L * FW, -1);
  for (int t = 0; t < L; ++t) { const int64_t i = o.pt_obs[b0 + t];
    const int c = P.obs_cam[i], g = P.cam_group[c];
    const double* F0 = &o.F[((size_t)i * 2) * FW]; const double* F1 = F0 + FW; const double* E = &o.Jp[(size_t)i * 2 * pd];
    for (int a = 0; a < FW; ++a) { wc[(size_t)t * FW + a] = fcol(o, c, g, a);
      for (int b = 0; b < pd; ++b) W[((size_t)t * FW + a) * pd + b] = F0[a] * E[b] + F1[a] * E[pd + b]; } }
  double Vig[4];
  for (int a = 0; a < pd; ++a) { double s = 0; for (int b = 0; b < pd; ++b) s += Vi[a * pd + b] * gp[b]; Vig[a] = s; }
  for (int t = 0; t < L; ++t) {
    for (int a = 0; a < FW; ++a) { const int ca = wc[(size_t)t * FW + a]; if (ca < 0) continue;
      const double* Wt = &W[((size_t)t * FW + a) * pd];
      double WV[4];
      for (int b = 0; b < pd; ++b) { double s = 0; for (int k = 0; k < pd; ++k) s += Wt[k] * Vi[k * pd + b]; WV[b] = s; }
      double s0 = 0; for (int b = 0; b < pd; ++b) s0 += Wt[b] * Vig[b];
      o.rhs[ca] -= s0;
      for (int u = 0; u < L; ++u) for (int b = 0; b < FW; ++b) { const int cb = wc[(size_t)u * FW + b]; if (cb < 0) continue;
        const double* Wu = &W[((size_t)u * FW + b) * pd];
        double s = 0; for (int k = 0; k < pd; ++k) s += WV[k] * Wu[k];
        o.S[(size_t)ca * n + cb] -= s; }
    }
  }
  return true;
}

bool build_reduced(Oracle& o, double radius, bool add_cam_diag = true) {
  const oba_problem& P = *o.P;
  const int n = o.n();
  if (g_threads > 1 && o.S.size() == (size_t)n * n) {   // zero by rows in parallel (also spreads the pages over the NUMA nodes)
    OBA_PAR_FOR
    for (int r = 0; r < n; ++r) std::fill(o.S.begin() + (size_t)r * n, o.S.begin() + (size_t)(r + 1) * n, 0.0);
  } else {
    o.S.assign((size_t)n * n, 0.0);
  }
  o.rhs.assign(n, 0.0);
  o.Vinv.assign((size_t)o.np * o.pd * o.pd, 0.0);
  // F^T F and F^T r of one observation
  auto obs_term = [&](int64_t i) {
    int cols[FW];
    const int c = P.obs_cam[i], g = P.cam_group[c];
    for (int q = 0; q < FW; ++q) cols[q] = fcol(o, c, g, q);
    const double* F0 = &o.F[((size_t)i * 2) * FW]; const double* F1 = F0 + FW;
    for (int a = 0; a < FW; ++a) { if (cols[a] < 0) continue;
      for (int b = 0; b < FW; ++b) if (cols[b] >= 0)
        o.S[(size_t)cols[a] * n + cols[b]] += F0[a] * F0[b] + F1[a] * F1[b];
      o.rhs[cols[a]] += F0[a] * o.r[2 * i] + F1[a] * o.r[2 * i + 1]; }
  };
  bool ok = true;
  if (g_threads > 1 && o.ni == 0 && o.ncv > 0) {
    // All-cores baseline (extrinsics only).  Every update of a point touches the S rows of the point's own cameras.
    // Points are bucketed by their lowest reduced camera into chunks of `width` cameras; a point whose cameras span
    // less than `width` only writes rows of its own chunk and of the next one, so the even chunks can run
    // concurrently, then the odd ones; the remaining (far-spanning) points run serially at the end.  Inside a
    // chunk the order is the serial one: the result does not depend on thread scheduling.
    std::vector<int> lo(o.np, -1), hi(o.np, -1);
    for (int p = 0; p < o.np; ++p)
      for (int64_t k = o.pt_off[p]; k < o.pt_off[p + 1]; ++k) { const int rc = o.cam_red[P.obs_cam[o.pt_obs[k]]];
        if (rc < 0) continue; if (lo[p] < 0 || rc < lo[p]) lo[p] = rc; if (rc > hi[p]) hi[p] = rc; }
    int span = 1;
    { std::vector<int> sp; for (int p = 0; p < o.np; ++p) if (lo[p] >= 0) sp.push_back(hi[p] - lo[p] + 1);
      if (!sp.empty()) { std::nth_element(sp.begin(), sp.begin() + (sp.size() * 98) / 100, sp.end()); span = sp[(sp.size() * 98) / 100]; } }
    const int width = std::max(span, (o.ncv + 2 * g_threads - 1) / (2 * g_threads));
    const int nchunk = (o.ncv + width - 1) / width;
    std::vector<std::vector<int>> bucket(nchunk);
    std::vector<int> far;
    for (int p = 0; p < o.np; ++p) {
      if (o.pt_off[p] == o.pt_off[p + 1]) continue;
      if (lo[p] < 0) { far.push_back(p); continue; }   // only constant cameras: no S rows, but V^-1 is still needed
      if (hi[p] - lo[p] + 1 <= width) bucket[lo[p] / width].push_back(p);
      else far.push_back(p);
    }
    int fail = 0;
    for (int phase = 0; phase < 2; ++phase) {
#pragma omp parallel for schedule(dynamic, 1) num_threads(g_threads) reduction(+ : fail)
      for (int k = phase; k < nchunk; k += 2) {
        std::vector<double> W; std::vector<int> wc;
        for (int p : bucket[k]) {
          for (int64_t t = o.pt_off[p]; t < o.pt_off[p + 1]; ++t) if (!o.obs_fixed[o.pt_obs[t]]) obs_term(o.pt_obs[t]);
          if (!o.pt_const[p] && !eliminate_point(o, p, radius, W, wc)) fail++;
        }
      }
    }
    { std::vector<double> W; std::vector<int> wc;
      for (int p : far) {
        for (int64_t t = o.pt_off[p]; t < o.pt_off[p + 1]; ++t) if (!o.obs_fixed[o.pt_obs[t]]) obs_term(o.pt_obs[t]);
        if (!o.pt_const[p] && !eliminate_point(o, p, radius, W, wc)) fail++;
      } }
    ok = fail == 0;
  } else {
    for (int64_t i = 0; i < o.nobs; ++i) if (!o.obs_fixed[i]) obs_term(i);
  }
  for (const Oracle::Prior& pr : o.priors) {
    const int base = o.ni + 6 * o.cam_red[pr.cam];
    for (int a = 0; a < 6; ++a) {
      for (int b = 0; b < 6; ++b)
        o.S[(size_t)(base + a) * n + base + b] += (pr.J[a] * pr.J[b] + pr.J[6 + a] * pr.J[6 + b]) + pr.J[12 + a] * pr.J[12 + b];
      o.rhs[base + a] += (pr.J[a] * pr.r[0] + pr.J[6 + a] * pr.r[1]) + pr.J[12 + a] * pr.r[2];
    }
  }
  if (add_cam_diag) for (int d = 0; d < n; ++d) o.S[(size_t)d * n + d] += o.diag_f[d] / radius;
  if (g_threads > 1 && o.ni == 0 && o.ncv > 0) return ok;
  // eliminate points
  std::vector<double> W;  // per obs of the point: F^T E (FW x pd)
  std::vector<int> wc;
  for (int p = 0; p < o.np; ++p) {
    if (o.pt_const[p]) continue;
    if (!eliminate_point(o, p, radius, W, wc)) return false;
  }
  return true;
}

// back-substitution: y_p = Vinv (E^T b - E^T F y_c)
void back_substitute(Oracle& o) {
  const oba_problem& P = *o.P; const int pd = o.pd;
  o.yp.assign((size_t)o.np * pd, 0.0);
  OBA_PAR_FOR
  for (int p = 0; p < o.np; ++p) {
    if (o.pt_const[p]) continue;
    double t[4] = {0};
    for (int64_t k = o.pt_off[p]; k < o.pt_off[p + 1]; ++k) { const int64_t i = o.pt_obs[k];
      const int c = P.obs_cam[i], g = P.cam_group[c];
      const double* E = &o.Jp[(size_t)i * 2 * pd];
      for (int a = 0; a < 2; ++a) { double m = o.r[2 * i + a];
        const double* Fr = &o.F[((size_t)i * 2 + a) * FW];
        for (int q = 0; q < FW; ++q) { const int col = fcol(o, c, g, q); if (col >= 0) m -= Fr[q] * o.yc[col]; }
        for (int q = 0; q < pd; ++q) t[q] += E[a * pd + q] * m; } }
    const double* Vi = &o.Vinv[(size_t)p * pd * pd];
    for (int a = 0; a < pd; ++a) { double s = 0; for (int b = 0; b < pd; ++b) s += Vi[a * pd + b] * t[b];
      o.yp[(size_t)pd * p + a] = s; }
  }
}

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int setup(Oracle& o, const oba_problem* P, const oba_options* O) {
  o.P = P; o.O = *O; o.nc = P->num_cameras; o.np = P->num_points; o.ng = P->num_groups; o.nobs = P->num_obs;
  o.pd = O->use_homogeneous_point_parametrization ? 3 : 4;
  o.cam_red.assign(o.nc, -1); o.cam_mask.assign(o.nc, 0); o.pt_const.assign(o.np, 0);
  o.grp_red.assign(o.ng, -1); o.grp_free.assign(o.ng, 0u);
  std::vector<uint8_t> cam_used(o.nc, 0), pt_used(o.np, 0), grp_used(o.ng, 0);
  for (int64_t i = 0; i < o.nobs; ++i) { cam_used[P->obs_cam[i]] = 1; pt_used[P->obs_pt[i]] = 1; grp_used[P->cam_group[P->obs_cam[i]]] = 1; }
  // intrinsics blocks (bundle_adjuster.cc:382-460): constant when nothing is
  // optimised or when the caller marked the group constant, else a subset manifold
  o.ngv = 0;
  for (int g = 0; g < o.ng; ++g) {
    const unsigned fm = intrinsics_free_mask(P->group_model[g], O->intrinsics_to_optimize);
    const bool gconst = (P->group_const && P->group_const[g]) || fm == 0 || (!grp_used[g] && !(P->flags & 1));
    if (!gconst) { o.grp_red[g] = o.ngv++; o.grp_free[g] = fm; }
  }
  o.ni = kMaxIntr * o.ngv;
  o.ncv = 0;
  for (int c = 0; c < o.nc; ++c) {
    uint8_t m = P->cam_const ? P->cam_const[c] : 0;
    // bundle_adjuster.cc:357-380 global options
    if (O->constant_camera_orientation) m |= 2;
    if (O->constant_camera_position) m |= 1;
    if (O->orthographic_camera) m |= 4;
    uint8_t cols = 0;
    if (m & 1) cols |= 0x07; if (m & 2) cols |= 0x38; if (m & 4) cols |= 0x04;
    o.cam_mask[c] = cols;
    if (cols != 0x3f && (cam_used[c] || (P->flags & 1))) o.cam_red[c] = o.ncv++;
    else o.cam_mask[c] = 0x3f;
  }
  for (int p = 0; p < o.np; ++p) o.pt_const[p] = (P->point_const && P->point_const[p]) || !pt_used[p];
  o.obs_fixed.assign(o.nobs, 0);
  for (int64_t i = 0; i < o.nobs; ++i)
    if (o.cam_red[P->obs_cam[i]] < 0 && o.grp_red[P->cam_group[P->obs_cam[i]]] < 0 && o.pt_const[P->obs_pt[i]]) o.obs_fixed[i] = 1;
  // CSR by point
  o.pt_off.assign(o.np + 1, 0);
  for (int64_t i = 0; i < o.nobs; ++i) if (!o.obs_fixed[i]) o.pt_off[P->obs_pt[i] + 1]++;
  for (int p = 0; p < o.np; ++p) o.pt_off[p + 1] += o.pt_off[p];
  o.pt_obs.assign(o.pt_off[o.np], 0);
  { std::vector<int64_t> fill(o.pt_off.begin(), o.pt_off.end() - 1);
    for (int64_t i = 0; i < o.nobs; ++i) if (!o.obs_fixed[i]) o.pt_obs[fill[P->obs_pt[i]]++] = i; }
  o.cam.assign(P->cam_ext, P->cam_ext + 6 * (size_t)o.nc);
  o.pts.assign(P->points, P->points + 4 * (size_t)o.np);
  o.intr.assign(P->intrinsics, P->intrinsics + (size_t)kMaxIntr * o.ng);
  for (int g = 0; g < o.ng; ++g) if (o.grp_red[g] >= 0) project_intrinsics_to_bounds(P->group_model[g], &o.intr[(size_t)g * kMaxIntr]);
  o.r.assign(2 * o.nobs, 0.0); o.F.assign((size_t)2 * FW * o.nobs, 0.0); o.Jp.assign((size_t)2 * o.pd * o.nobs, 0.0);
  o.scale_f.assign(o.n(), 1.0); o.scale_p.assign((size_t)o.pd * o.np, 1.0);
  o.diag_f.assign(o.n(), 0.0); o.diag_p.assign((size_t)o.pd * o.np, 0.0);
  // fixed cost: residual blocks whose parameter blocks are all constant
  o.fixed_cost = 0.0;
  const double one[2] = {1.0, 1.0};
  for (int64_t i = 0; i < o.nobs; ++i) if (o.obs_fixed[i]) {
    const int c = P->obs_cam[i], p = P->obs_pt[i], g = P->cam_group[c];
    double res[2] = {0, 0};
    const bool depth_row = P->obs_kind && P->obs_kind[i];
    reprojection_error<double>(depth_row ? kDepthRow : P->group_model[g], &o.cam[6 * c], &o.intr[(size_t)g * kMaxIntr],
                               &o.pts[4 * p], P->obs_uv + 2 * i, P->obs_sqrt_info ? P->obs_sqrt_info + 2 * i : one, res);
    double rho[3];
    loss_evaluate(O->loss_function_type, depth_row ? O->robust_loss_width_depth_prior : O->robust_loss_width,
                  res[0] * res[0] + res[1] * res[1], rho);
    o.fixed_cost += 0.5 * rho[0];
  }
  // camera priors: used when both the camera's bit and the option's bit are set; a prior on a constant
  // camera is a residual block without variable parameters -> fixed cost
  o.priors.clear();
  if (P->cam_prior_mask && O->prior_mask) {
    const double* vecs[3] = {P->cam_position_prior, P->cam_gravity_prior, P->cam_orientation_prior};
    const double* infos[3] = {P->cam_position_prior_sqrt_info, P->cam_gravity_prior_sqrt_info, P->cam_orientation_prior_sqrt_info};
    for (int c = 0; c < o.nc; ++c)
      for (int k = 0; k < 3; ++k) {
        const int bit = 1 << k;
        if (!(P->cam_prior_mask[c] & bit) || !(O->prior_mask & bit) || !vecs[k] || !infos[k]) continue;
        if (o.cam_red[c] >= 0) {
          Oracle::Prior pr; pr.cam = c; pr.kind = bit; pr.vec = vecs[k] + 3 * c; pr.sqrt_info = infos[k] + 9 * c;
          o.priors.push_back(pr);
        } else {
          double rr3[3];
          camera_prior_residual<double>(bit, &o.cam[6 * c], vecs[k] + 3 * c, infos[k] + 9 * c, rr3);
          o.fixed_cost += 0.5 * ((rr3[0] * rr3[0] + rr3[1] * rr3[1]) + rr3[2] * rr3[2]);
        }
      }
  }
  return 0;
}

double state_norm(const Oracle& o, const std::vector<double>& cam, const std::vector<double>& pts, const std::vector<double>& intr) {
  double s = 0;
  for (int g = 0; g < o.ng; ++g) if (o.grp_red[g] >= 0) {
    const int K = intrinsics_size(o.P->group_model[g]);
    for (int q = 0; q < K; ++q) s += intr[(size_t)g * kMaxIntr + q] * intr[(size_t)g * kMaxIntr + q]; }
  for (int c = 0; c < o.nc; ++c) if (o.cam_red[c] >= 0) for (int q = 0; q < 6; ++q) s += cam[6 * c + q] * cam[6 * c + q];
  for (int p = 0; p < o.np; ++p) if (!o.pt_const[p]) for (int q = 0; q < 4; ++q) s += pts[4 * p + q] * pts[4 * p + q];
  return std::sqrt(s);
}

void trace_push(oba_summary* S, double cost, double g, double step, double radius, int acc) {
  if (!S->trace_cost || S->trace_size >= S->trace_capacity) return;
  const int k = S->trace_size++;
  S->trace_cost[k] = cost; S->trace_gradient_max_norm[k] = g; S->trace_step_norm[k] = step;
  S->trace_radius[k] = radius; S->trace_accepted[k] = acc;
}

void init_scaling(Oracle& o) {
  column_norms(o, o.diag_f, o.diag_p);  // jacobi scaling, computed once (trust_region_minimizer.cc)
  for (size_t i = 0; i < o.scale_f.size(); ++i) o.scale_f[i] = 1.0 / (1.0 + std::sqrt(o.diag_f[i]));
  for (size_t i = 0; i < o.scale_p.size(); ++i) o.scale_p[i] = 1.0 / (1.0 + std::sqrt(o.diag_p[i]));
  apply_scaling(o);
}
void lm_diagonal(Oracle& o) {
  column_norms(o, o.diag_f, o.diag_p);
  for (auto& d : o.diag_f) d = std::min(std::max(d, 1e-6), 1e32);
  for (auto& d : o.diag_p) d = std::min(std::max(d, 1e-6), 1e32);
}

}  // namespace


// ============================================================ inner iterations
// Ceres 2.2 CoordinateDescentMinimizer (coordinate_descent_minimizer.cc:113-262) as TrustRegionMinimizer runs it after
// every candidate (DoInnerIterationsIfNeeded): the reference hands it its elimination ordering REVERSED
// (bundle_adjuster.cc:329-333): independent set 0 = camera extrinsics, 1 = intrinsics groups, 2 = points.  Every
// non-constant block is minimised alone, all others fixed at their current values, by a fresh trust-region solve with
// default Minimizer::Options (LM, DENSE_QR, <= 50 iterations, 1e-6 / 1e-10 / 1e-8, radius 1e4, Jacobi scaling) over the
// residual blocks that depend on it.  Restated with normal equations + Cholesky (same step as the QR of [J; D] up to
// round-off) and Jets that carry only the block's parameters.
namespace {

template <int N>
bool inner_chol_solve(const double* H /* N x N full */, const double* d, const double* g, double* y) {
  double L[N * N];
  for (int i = 0; i < N; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = H[i * N + j] + (i == j ? d[i] : 0.0);
      for (int k = 0; k < j; ++k) s -= L[i * N + k] * L[j * N + k];
      if (i == j) { if (!(s > 0.0)) return false; L[i * N + i] = std::sqrt(s); }
      else L[i * N + j] = s / L[j * N + j];
    }
  double z[N];
  for (int i = 0; i < N; ++i) { double s = g[i]; for (int k = 0; k < i; ++k) s -= L[i * N + k] * z[k]; z[i] = s / L[i * N + i]; }
  for (int i = N - 1; i >= 0; --i) { double s = z[i]; for (int k = i + 1; k < N; ++k) s -= L[k * N + i] * y[k]; y[i] = s / L[i * N + i]; }
  return true;
}

// lin(x, J-scale, H, g, &cost) -> false if a functor failed; costf(x, &cost) likewise; plus(x, tangent step, out)
template <int N, int NA, class Lin, class CostF, class Plus>
void inner_block_lm(double* x, Lin lin, CostF costf, Plus plus) {
  double scale[N], H[N * N], g[N], x_cost = 0.0;
  for (int q = 0; q < N; ++q) scale[q] = 1.0;
  if (!lin(x, scale, H, g, &x_cost) || !std::isfinite(x_cost)) return;
  for (int q = 0; q < N; ++q) scale[q] = 1.0 / (1.0 + std::sqrt(H[q * N + q]));
  double radius = 1e4, decrease_factor = 2.0, x_norm = 0.0, gmax = 0.0;
  for (int q = 0; q < NA; ++q) x_norm += x[q] * x[q];
  x_norm = std::sqrt(x_norm);
  bool step_successful = true, need_lin = true;
  int iter = 0, invalid_steps = 0;
  while (true) {
    if (need_lin) {
      lin(x, scale, H, g, &x_cost);
      gmax = 0.0;
      for (int q = 0; q < N; ++q) gmax = std::max(gmax, std::fabs(g[q] / scale[q]));
      need_lin = false;
    }
    if (iter >= 50) break;
    if (step_successful && gmax <= 1e-10) break;
    if (radius <= 1e-32) break;
    ++iter;
    double d[N], y[N];
    for (int q = 0; q < N; ++q) d[q] = std::min(std::max(H[q * N + q], 1e-6), 1e32) / radius;
    const bool pd = inner_chol_solve<N>(H, d, g, y);
    double yg = 0.0, yHy = 0.0;
    for (int a = 0; a < N; ++a) { yg += y[a] * g[a]; double row = 0.0; for (int b = 0; b < N; ++b) row += H[a * N + b] * y[b]; yHy += y[a] * row; }
    const double mcc = yg - 0.5 * yHy;
    double step[N], xc[NA], stepsq = 0.0, xnormsq = 0.0;
    for (int q = 0; q < N; ++q) step[q] = -y[q] * scale[q];
    plus(x, step, xc);
    for (int q = 0; q < NA; ++q) { stepsq += (x[q] - xc[q]) * (x[q] - xc[q]); xnormsq += xc[q] * xc[q]; }
    if (!(pd && std::isfinite(mcc) && std::isfinite(stepsq) && mcc > 0.0)) {
      if (++invalid_steps >= 5) break;
      radius /= decrease_factor; decrease_factor *= 2.0; step_successful = false;
      continue;
    }
    invalid_steps = 0;
    double cand_cost;
    if (!costf(xc, &cand_cost) || !std::isfinite(cand_cost)) cand_cost = std::numeric_limits<double>::max();
    if (std::sqrt(stepsq) <= 1e-8 * (x_norm + 1e-8)) break;
    const double cost_change = x_cost - cand_cost;
    if (std::fabs(cost_change) <= 1e-6 * x_cost) break;
    const double rho = cost_change / mcc;
    if (rho > 1e-3) {
      for (int q = 0; q < NA; ++q) x[q] = xc[q];
      x_norm = std::sqrt(xnormsq);
      radius = std::min(1e16, radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * rho - 1.0, 3)));
      decrease_factor = 2.0; step_successful = true; need_lin = true;
    } else {
      radius /= decrease_factor; decrease_factor *= 2.0; step_successful = false;
    }
  }
}

// accumulate one residual row pair (already loss-weighted) with its N-column Jacobian rows
template <int N>
inline void inner_add_rows(int nrows, const double* res, const double* J /* nrows x N */, double* H, double* g) {
  for (int a = 0; a < nrows; ++a)
    for (int i = 0; i < N; ++i) {
      g[i] += J[a * N + i] * res[a];
      for (int j = 0; j < N; ++j) H[i * N + j] += J[a * N + i] * J[a * N + j];
    }
}

void coordinate_descent(const Oracle& o, std::vector<double>& cam, std::vector<double>& pts, std::vector<double>& intr) {
  const oba_problem& P = *o.P;
  const double one[2] = {1.0, 1.0};
  // observation lists by camera and by group (the residual blocks that depend on the block)
  std::vector<std::vector<int64_t>> by_cam(o.nc), by_grp(o.ng);
  for (int64_t i = 0; i < o.nobs; ++i) {
    if (o.obs_fixed[i]) continue;
    by_cam[P.obs_cam[i]].push_back(i);
    if (!(P.obs_kind && P.obs_kind[i])) by_grp[P.cam_group[P.obs_cam[i]]].push_back(i);
  }
  auto obs_model = [&](int64_t i) { return (P.obs_kind && P.obs_kind[i]) ? kDepthRow : P.group_model[P.cam_group[P.obs_cam[i]]]; };
  auto obs_width = [&](int64_t i) { return (P.obs_kind && P.obs_kind[i]) ? o.O.robust_loss_width_depth_prior : o.O.robust_loss_width; };
  // development switch shared with the device (THEIA_HIP_INNER_SKIP: bit 0 cameras, 1 intrinsics, 2 points)
  const char* skip_env = getenv("THEIA_HIP_INNER_SKIP");
  const int skip = skip_env ? atoi(skip_env) : 0;
  // ---- set 0: camera extrinsics
  for (int c = 0; c < o.nc; ++c) {
    if (skip & 1) break;
    if (o.cam_red[c] < 0) continue;
    const unsigned mask = o.cam_mask[c];
    if ((mask & 0x3fu) == 0x3fu) continue;
    const int g = P.cam_group[c];
    auto run = [&](const double* x, const double* scale, bool want_jac, double* H, double* gg, double* cost) {
      bool ok = true;
      double cst = 0.0;
      if (want_jac) { for (int k = 0; k < 36; ++k) H[k] = 0.0; for (int k = 0; k < 6; ++k) gg[k] = 0.0; }
      for (int64_t i : by_cam[c]) {
        const int p = P.obs_pt[i];
        const double* si = P.obs_sqrt_info ? P.obs_sqrt_info + 2 * i : one;
        double res[2], J[12];
        if (want_jac) {
          typedef Jet<6> J6;
          J6 e[6], k[kMaxIntr], X[4], rr[2];
          for (int q = 0; q < 6; ++q) e[q] = J6(x[q], q);
          for (int q = 0; q < kMaxIntr; ++q) k[q] = J6(intr[(size_t)g * kMaxIntr + q]);
          for (int q = 0; q < 4; ++q) X[q] = J6(pts[4 * (size_t)p + q]);
          if (!reprojection_error<J6>(obs_model(i), e, k, X, P.obs_uv + 2 * i, si, rr)) ok = false;
          for (int a = 0; a < 2; ++a) { res[a] = rr[a].a; for (int q = 0; q < 6; ++q) J[6 * a + q] = rr[a].v[q]; }
        } else if (!reprojection_error<double>(obs_model(i), x, &intr[(size_t)g * kMaxIntr], &pts[4 * (size_t)p], P.obs_uv + 2 * i, si, res)) ok = false;
        double rho[3];
        loss_evaluate(o.O.loss_function_type, obs_width(i), res[0] * res[0] + res[1] * res[1], rho);
        cst += 0.5 * rho[0];
        if (!want_jac) continue;
        const double sr = std::sqrt(rho[1]);
        for (int a = 0; a < 2; ++a) { res[a] *= sr; for (int q = 0; q < 6; ++q) J[6 * a + q] *= ((mask >> q) & 1u) ? 0.0 : sr * scale[q]; }
        inner_add_rows<6>(2, res, J, H, gg);
      }
      for (const Oracle::Prior& pr : o.priors) {
        if (pr.cam != c) continue;
        double r3[3], J[18];
        if (want_jac) {
          typedef Jet<6> J6;
          J6 e[6], rr[3];
          for (int q = 0; q < 6; ++q) e[q] = J6(x[q], q);
          camera_prior_residual<J6>(pr.kind, e, pr.vec, pr.sqrt_info, rr);
          for (int a = 0; a < 3; ++a) { r3[a] = rr[a].a; for (int q = 0; q < 6; ++q) J[6 * a + q] = ((mask >> q) & 1u) ? 0.0 : rr[a].v[q] * scale[q]; }
        } else camera_prior_residual<double>(pr.kind, x, pr.vec, pr.sqrt_info, r3);
        cst += 0.5 * (r3[0] * r3[0] + r3[1] * r3[1] + r3[2] * r3[2]);
        if (want_jac) inner_add_rows<6>(3, r3, J, H, gg);
      }
      *cost = cst;
      return ok;
    };
    inner_block_lm<6, 6>(&cam[6 * (size_t)c],
        [&](const double* x, const double* scale, double* H, double* gg, double* cost) { return run(x, scale, true, H, gg, cost); },
        [&](const double* x, double* cost) { return run(x, nullptr, false, nullptr, nullptr, cost); },
        [&](const double* x, const double* step, double* out) { for (int q = 0; q < 6; ++q) out[q] = ((mask >> q) & 1u) ? x[q] : x[q] + step[q]; });
  }
  // ---- set 1: intrinsics groups
  for (int g = 0; g < o.ng; ++g) {
    if (skip & 2) break;
    if (o.grp_red[g] < 0 || !o.grp_free[g]) continue;
    const unsigned fm = o.grp_free[g];
    auto run = [&](const double* x, const double* scale, bool want_jac, double* H, double* gg, double* cost) {
      bool ok = true;
      double cst = 0.0;
      if (want_jac) { for (int k = 0; k < 100; ++k) H[k] = 0.0; for (int k = 0; k < 10; ++k) gg[k] = 0.0; }
      for (int64_t i : by_grp[g]) {
        const int c = P.obs_cam[i], p = P.obs_pt[i];
        const double* si = P.obs_sqrt_info ? P.obs_sqrt_info + 2 * i : one;
        double res[2], J[20];
        if (want_jac) {
          typedef Jet<10> JK;
          JK e[6], k[kMaxIntr], X[4], rr[2];
          for (int q = 0; q < 6; ++q) e[q] = JK(cam[6 * (size_t)c + q]);
          for (int q = 0; q < kMaxIntr; ++q) k[q] = JK(x[q], q);
          for (int q = 0; q < 4; ++q) X[q] = JK(pts[4 * (size_t)p + q]);
          if (!reprojection_error<JK>(P.group_model[g], e, k, X, P.obs_uv + 2 * i, si, rr)) ok = false;
          for (int a = 0; a < 2; ++a) { res[a] = rr[a].a; for (int q = 0; q < 10; ++q) J[10 * a + q] = rr[a].v[q]; }
        } else if (!reprojection_error<double>(P.group_model[g], &cam[6 * (size_t)c], x, &pts[4 * (size_t)p], P.obs_uv + 2 * i, si, res)) ok = false;
        double rho[3];
        loss_evaluate(o.O.loss_function_type, o.O.robust_loss_width, res[0] * res[0] + res[1] * res[1], rho);
        cst += 0.5 * rho[0];
        if (!want_jac) continue;
        const double sr = std::sqrt(rho[1]);
        for (int a = 0; a < 2; ++a) { res[a] *= sr; for (int q = 0; q < 10; ++q) J[10 * a + q] *= ((fm >> q) & 1u) ? sr * scale[q] : 0.0; }
        inner_add_rows<10>(2, res, J, H, gg);
      }
      *cost = cst;
      return ok;
    };
    inner_block_lm<10, 10>(&intr[(size_t)g * kMaxIntr],
        [&](const double* x, const double* scale, double* H, double* gg, double* cost) { return run(x, scale, true, H, gg, cost); },
        [&](const double* x, double* cost) { return run(x, nullptr, false, nullptr, nullptr, cost); },
        [&](const double* x, const double* step, double* out) { for (int q = 0; q < 10; ++q) out[q] = ((fm >> q) & 1u) ? x[q] + step[q] : x[q]; });
  }
  // ---- set 2: points
  const int pd = o.pd;
  for (int p = 0; p < o.np; ++p) {
    if (skip & 4) break;
    if (o.pt_const[p] || o.pt_off[p] == o.pt_off[p + 1]) continue;
    auto run = [&](const double* x, const double* scale, bool want_jac, double* H, double* gg, double* cost, int n) {
      bool ok = true;
      double cst = 0.0;
      if (want_jac) { for (int k = 0; k < n * n; ++k) H[k] = 0.0; for (int k = 0; k < n; ++k) gg[k] = 0.0; }
      double PJ[12];
      if (want_jac && pd == 3) sphere_plus_jacobian(x, PJ);
      for (int64_t kk = o.pt_off[p]; kk < o.pt_off[p + 1]; ++kk) {
        const int64_t i = o.pt_obs[kk];
        if (o.obs_fixed[i]) continue;
        const int c = P.obs_cam[i], g = P.cam_group[c];
        const double* si = P.obs_sqrt_info ? P.obs_sqrt_info + 2 * i : one;
        double res[2], J4[8];
        if (want_jac) {
          typedef Jet<4> J4T;
          J4T e[6], k[kMaxIntr], X[4], rr[2];
          for (int q = 0; q < 6; ++q) e[q] = J4T(cam[6 * (size_t)c + q]);
          for (int q = 0; q < kMaxIntr; ++q) k[q] = J4T(intr[(size_t)g * kMaxIntr + q]);
          for (int q = 0; q < 4; ++q) X[q] = J4T(x[q], q);
          if (!reprojection_error<J4T>(obs_model(i), e, k, X, P.obs_uv + 2 * i, si, rr)) ok = false;
          for (int a = 0; a < 2; ++a) { res[a] = rr[a].a; for (int q = 0; q < 4; ++q) J4[4 * a + q] = rr[a].v[q]; }
        } else if (!reprojection_error<double>(obs_model(i), &cam[6 * (size_t)c], &intr[(size_t)g * kMaxIntr], x, P.obs_uv + 2 * i, si, res)) ok = false;
        double rho[3];
        loss_evaluate(o.O.loss_function_type, obs_width(i), res[0] * res[0] + res[1] * res[1], rho);
        cst += 0.5 * rho[0];
        if (!want_jac) continue;
        const double sr = std::sqrt(rho[1]);
        double J[8];
        for (int a = 0; a < 2; ++a) {
          res[a] *= sr;
          for (int q = 0; q < n; ++q) {
            double v;
            if (pd == 3) { v = 0.0; for (int t = 0; t < 4; ++t) v += J4[4 * a + t] * PJ[t * 3 + q]; }
            else v = J4[4 * a + q];
            J[n * a + q] = v * sr * scale[q];
          }
        }
        if (n == 3) inner_add_rows<3>(2, res, J, H, gg); else inner_add_rows<4>(2, res, J, H, gg);
      }
      *cost = cst;
      return ok;
    };
    if (pd == 3)
      inner_block_lm<3, 4>(&pts[4 * (size_t)p],
          [&](const double* x, const double* scale, double* H, double* gg, double* cost) { return run(x, scale, true, H, gg, cost, 3); },
          [&](const double* x, double* cost) { return run(x, nullptr, false, nullptr, nullptr, cost, 3); },
          [&](const double* x, const double* step, double* out) { sphere_plus(x, step, out); });
    else
      inner_block_lm<4, 4>(&pts[4 * (size_t)p],
          [&](const double* x, const double* scale, double* H, double* gg, double* cost) { return run(x, scale, true, H, gg, cost, 4); },
          [&](const double* x, double* cost) { return run(x, nullptr, false, nullptr, nullptr, cost, 4); },
          [&](const double* x, const double* step, double* out) { for (int q = 0; q < 4; ++q) out[q] = x[q] + step[q]; });
  }
}

}  // namespace

extern "C" {

// Threads of the all-cores CPU baseline (1 = the serial reference path of the tests); returns the previous value.
int oracle_ba_set_threads(int n) { const int prev = g_threads; g_threads = n < 1 ? 1 : n; return prev; }
int oracle_ba_max_threads() { return omp_get_max_threads(); }

void oracle_ba_options_default(oba_options* o) {
  std::memset(o, 0, sizeof(*o));
  o->robust_loss_width_depth_prior = 0.01;   // bundle_adjustment.h:94
  o->loss_function_type = 0; o->robust_loss_width = 2.0; o->intrinsics_to_optimize = 0;
  o->max_num_iterations = 100; o->use_homogeneous_point_parametrization = 1;
  o->use_inner_iterations = 1; o->function_tolerance = 1e-6; o->gradient_tolerance = 1e-10;
  o->parameter_tolerance = 1e-8; o->max_trust_region_radius = 1e12; o->max_solver_time_in_seconds = 3600.0;
}

// Evaluate residuals / Jacobian blocks at the problem's current parameters:
// residuals[nobs][2], jac_cam[nobs][2][6], jac_pt[nobs][2][pd], optional
// jac_intr[nobs][2][10] (loss-corrected, tangent space, unscaled; frozen
// columns are zero), returns 1 iff all functors returned true.
int oracle_ba_evaluate_ex(const oba_problem* P, const oba_options* O, double* cost, double* residuals,
                          double* jac_cam, double* jac_pt, double* jac_intr) {
  Oracle o; int rc = setup(o, P, O); if (rc) return rc;
  double c; const bool ok = evaluate(o, o.cam, o.pts, o.intr, true, &c);
  *cost = c + o.fixed_cost;
  if (residuals) std::copy(o.r.begin(), o.r.end(), residuals);
  for (int64_t i = 0; i < o.nobs; ++i) for (int a = 0; a < 2; ++a) {
    const double* Fr = &o.F[((size_t)i * 2 + a) * FW];
    if (jac_cam) for (int q = 0; q < 6; ++q) jac_cam[12 * i + 6 * a + q] = Fr[kMaxIntr + q];
    if (jac_intr) for (int q = 0; q < kMaxIntr; ++q) jac_intr[20 * i + 10 * a + q] = Fr[q];
  }
  if (jac_pt) std::copy(o.Jp.begin(), o.Jp.end(), jac_pt);
  return ok ? 1 : 0;
}
// residual (3) and Jacobian (3 x 6, row-major) of one camera prior by Jets: used to pin the closed forms
void oracle_camera_prior(int kind, const double* ext, const double* prior, const double* sqrt_info, double* r, double* J) {
  typedef Jet<6> J6;
  J6 e[6], rr3[3];
  for (int q = 0; q < 6; ++q) e[q] = J6(ext[q], q);
  camera_prior_residual<J6>(kind, e, prior, sqrt_info, rr3);
  for (int a = 0; a < 3; ++a) { r[a] = rr3[a].a; for (int q = 0; q < 6; ++q) J[6 * a + q] = rr3[a].v[q]; }
}

int oracle_ba_evaluate(const oba_problem* P, const oba_options* O, double* cost, double* residuals,
                       double* jac_cam, double* jac_pt) {
  return oracle_ba_evaluate_ex(P, O, cost, residuals, jac_cam, jac_pt, nullptr);
}

// Dense reduced camera system at the current parameters for a given radius
// (Jacobi-scaled, as the LM step solves it). S: n*n row-major, rhs: n;
// n = 10 * (#variable intrinsics groups) + 6 * (#variable cameras).
int oracle_ba_reduced_system(const oba_problem* P, const oba_options* O, double radius, int32_t* n_out,
                             double* S, double* rhs, int64_t capacity) {
  Oracle o; int rc = setup(o, P, O); if (rc) return rc;
  double c; if (!evaluate(o, o.cam, o.pts, o.intr, true, &c)) return -5;
  init_scaling(o);
  lm_diagonal(o);
  if (!build_reduced(o, radius)) return -5;
  const int n = o.n(); *n_out = n;
  if ((int64_t)n * n > capacity) return -1;
  std::copy(o.S.begin(), o.S.end(), S); std::copy(o.rhs.begin(), o.rhs.end(), rhs);
  return 0;
}

// Multi-rank protocol pieces (what each rank computes before / after the
// all-reduce of the sharded path; used by the world_size-2 gloo test):
//  1. unscaled squared camera-side column norms of this shard        [n]
//  2. with the GLOBAL (summed) norms -> Jacobi scale, this shard's partial
//     reduced system WITHOUT the camera-side LM diagonal, and this shard's scaled
//     squared column norms (summed over ranks, they give that diagonal).
int oracle_ba_colnorms(const oba_problem* P, const oba_options* O, double* colsq_c) {
  Oracle o; int rc = setup(o, P, O); if (rc) return rc;
  double c; if (!evaluate(o, o.cam, o.pts, o.intr, true, &c)) return -5;
  column_norms(o, o.diag_f, o.diag_p);
  std::fill(colsq_c, colsq_c + 6 * (size_t)o.nc, 0.0);
  for (int cc = 0; cc < o.nc; ++cc) { const int r = o.cam_red[cc]; if (r < 0) continue;
    for (int a = 0; a < 6; ++a) colsq_c[6 * cc + a] = o.diag_f[o.ni + 6 * r + a]; }
  return 0;
}

int oracle_ba_reduced_partial(const oba_problem* P, const oba_options* O, double radius, const double* colsq_c_global,
                              int32_t* n_out, double* S, double* rhs, double* colsq_scaled, int64_t capacity) {
  Oracle o; int rc = setup(o, P, O); if (rc) return rc;
  if (o.ni != 0) return -3;  // protocol test covers the default (intrinsics NONE) path
  double c; if (!evaluate(o, o.cam, o.pts, o.intr, true, &c)) return -5;
  column_norms(o, o.diag_f, o.diag_p);
  for (int cc = 0; cc < o.nc; ++cc) { const int r = o.cam_red[cc]; if (r < 0) continue;
    for (int a = 0; a < 6; ++a) o.scale_f[6 * r + a] = 1.0 / (1.0 + std::sqrt(colsq_c_global[6 * cc + a])); }
  for (size_t i = 0; i < o.scale_p.size(); ++i) o.scale_p[i] = 1.0 / (1.0 + std::sqrt(o.diag_p[i]));
  apply_scaling(o);
  column_norms(o, o.diag_f, o.diag_p);
  for (auto& d : o.diag_p) d = std::min(std::max(d, 1e-6), 1e32);
  if (!build_reduced(o, radius, false)) return -5;
  const int n = o.n(); *n_out = n;
  if ((int64_t)n * n > capacity) return -1;
  std::copy(o.S.begin(), o.S.end(), S); std::copy(o.rhs.begin(), o.rhs.end(), rhs);
  for (int d = 0; d < n; ++d) colsq_scaled[d] = o.diag_f[d];
  return 0;
}

// The LM loop: ceres::Solve as configured by bundle_adjuster.cc:63-89,315-355.
void oracle_ba_armijo_stats(int* checks, int* failures, int reset) {
  if (checks) *checks = g_armijo_checks;
  if (failures) *failures = g_armijo_failures;
  if (reset) g_armijo_checks = g_armijo_failures = 0;
}

int oracle_ba_solve(oba_problem* P, const oba_options* O, oba_summary* S) {
  const double t0 = now_s();
  Oracle o; int rc = setup(o, P, O); if (rc) return rc;
  S->trace_size = 0; S->success = 0; S->num_iterations = 0; S->num_successful_steps = 0;
  S->time_linearize = S->time_solve_reduced = S->time_backsub = 0.0;
  const int pd = o.pd;
  const double t1 = now_s();
  S->setup_time_in_seconds = t1 - t0;
  // iteration zero
  double x_cost;
  double tl = now_s();
  if (!evaluate(o, o.cam, o.pts, o.intr, true, &x_cost)) {
    S->termination_type = 2; S->initial_cost = S->final_cost = x_cost + o.fixed_cost;
    S->solve_time_in_seconds = now_s() - t1; return 0;
  }
  init_scaling(o);
  double gmax = compute_gradient(o);
  S->time_linearize += now_s() - tl;
  double x_norm = state_norm(o, o.cam, o.pts, o.intr);
  S->initial_cost = x_cost + o.fixed_cost;
  double radius = 1e4, decrease_factor = 2.0;
  bool reuse_diagonal = false, step_successful = true;
  int iter = 0, invalid_steps = 0;
  double minimum_cost = x_cost;
  trace_push(S, x_cost + o.fixed_cost, gmax, 0.0, radius, 1);
  int term = 1;
  const int n = o.n();
  bool inner_enabled = O->use_inner_iterations != 0;
  while (true) {
    // FinalizeIterationAndCheckIfMinimizerCanContinue
    if (now_s() - t1 >= O->max_solver_time_in_seconds) { term = 1; break; }
    if (iter >= O->max_num_iterations) { term = 1; break; }
    if (step_successful && gmax <= O->gradient_tolerance) { term = 0; break; }
    if (radius <= 1e-32) { term = 0; break; }
    ++iter;
    // LevenbergMarquardtStrategy::ComputeStep
    double ts = now_s();
    if (!reuse_diagonal) lm_diagonal(o);
    reuse_diagonal = true;
    bool solved = build_reduced(o, radius);
    S->time_linearize += now_s() - ts; ts = now_s();
    if (solved) { o.yc = o.rhs; solved = n == 0 ? true : dense_cholesky_solve(n, o.S, o.yc);
      for (double v : o.yc) if (!std::isfinite(v)) solved = false; }
    S->time_solve_reduced += now_s() - ts; ts = now_s();
    double model_cost_change = 0.0, gdot = 0.0;   // gdot = gradient . step = sum r . (J step): the slope Ceres' projected line search starts from
    bool step_valid = solved;
    if (solved) {
      back_substitute(o);
      // step = -y ; model_residuals = Js * step ; mcc = -m.(r + m/2)
#pragma omp parallel for schedule(static) num_threads(g_threads) if (g_threads > 1) reduction(+ : model_cost_change, gdot)
      for (int64_t i = 0; i < o.nobs; ++i) { if (o.obs_fixed[i]) continue;
        const int c = P->obs_cam[i], p = P->obs_pt[i], g = P->cam_group[c];
        for (int a = 0; a < 2; ++a) { double m = 0;
          const double* Fr = &o.F[((size_t)i * 2 + a) * FW];
          for (int q = 0; q < FW; ++q) { const int col = fcol(o, c, g, q); if (col >= 0) m -= Fr[q] * o.yc[col]; }
          for (int q = 0; q < pd; ++q) m -= o.Jp[(size_t)i * 2 * pd + a * pd + q] * o.yp[(size_t)pd * p + q];
          model_cost_change -= m * (o.r[2 * i + a] + m / 2.0); gdot += m * o.r[2 * i + a]; } }
      for (const Oracle::Prior& pr : o.priors) {
        const int base = o.ni + 6 * o.cam_red[pr.cam];
        for (int a = 0; a < 3; ++a) { double m = 0;
          for (int q = 0; q < 6; ++q) m -= pr.J[6 * a + q] * o.yc[base + q];
          model_cost_change -= m * (pr.r[a] + m / 2.0); gdot += m * pr.r[a]; } }
      step_valid = model_cost_change > 0.0;
    }
    if (!step_valid) {
      S->time_backsub += now_s() - ts;
      if (++invalid_steps >= 5) { term = 2; break; }
      radius /= decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true; step_successful = false;
      trace_push(S, x_cost + o.fixed_cost, gmax, 0.0, radius, 0);
      continue;
    }
    invalid_steps = 0;
    // candidate = Plus(x, delta), delta = step .* scale (intrinsics: projected onto their bounds)
    o.ccam = o.cam; o.cpts = o.pts; o.cintr = o.intr;
    for (int g = 0; g < o.ng; ++g) { const int gr = o.grp_red[g]; if (gr < 0) continue;
      for (int q = 0; q < kMaxIntr; ++q) if ((o.grp_free[g] >> q) & 1u)
        o.cintr[(size_t)g * kMaxIntr + q] = o.intr[(size_t)g * kMaxIntr + q] + (-o.yc[10 * gr + q]) * o.scale_f[10 * gr + q];
      project_intrinsics_to_bounds(P->group_model[g], &o.cintr[(size_t)g * kMaxIntr]); }
    for (int c = 0; c < o.nc; ++c) { const int rc2 = o.cam_red[c]; if (rc2 < 0) continue;
      for (int q = 0; q < 6; ++q) if (!((o.cam_mask[c] >> q) & 1))
        o.ccam[6 * c + q] = o.cam[6 * c + q] + (-o.yc[o.ni + 6 * rc2 + q]) * o.scale_f[o.ni + 6 * rc2 + q]; }
    OBA_PAR_FOR
    for (int p = 0; p < o.np; ++p) { if (o.pt_const[p]) continue;
      double d[4]; for (int q = 0; q < pd; ++q) d[q] = -o.yp[(size_t)pd * p + q] * o.scale_p[(size_t)pd * p + q];
      if (pd == 3) sphere_plus(&o.pts[4 * p], d, &o.cpts[4 * p]);
      else for (int q = 0; q < 4; ++q) o.cpts[4 * p + q] = o.pts[4 * p + q] + d[q]; }
    double cand_cost;
    if (!evaluate(o, o.ccam, o.cpts, o.cintr, false, &cand_cost)) cand_cost = std::numeric_limits<double>::max();
    // Ceres' TrustRegionMinimizer runs a projected Armijo line search along the step whenever a parameter block has bounds
    // (trust_region_minimizer.cc DoLineSearch; the reference bounds the intrinsics, bundle_adjuster.cc:406-427) and shortens the
    // step when cost(x + step) > cost(x) + 1e-4 gradient . step.  Not restated (box projection + the rho test); what IS
    // recorded is whether the search would have left the step alone, so that the tests can tell the trajectories it cannot touch.
    if (o.ni > 0) { ++g_armijo_checks; if (!(cand_cost <= x_cost + 1e-4 * gdot)) ++g_armijo_failures; }
    // TrustRegionMinimizer::DoInnerIterationsIfNeeded (trust_region_minimizer.cc)
    bool inner_useful = false;
    if (inner_enabled && cand_cost < std::numeric_limits<double>::max()) {
      std::vector<double> icam = o.ccam, ipts = o.cpts, iintr = o.cintr;
      coordinate_descent(o, icam, ipts, iintr);
      double inner_cost;
      if (evaluate(o, icam, ipts, iintr, false, &inner_cost)) {
        o.ccam.swap(icam); o.cpts.swap(ipts); o.cintr.swap(iintr);
        model_cost_change += cand_cost - inner_cost;
        inner_useful = inner_cost < x_cost;
        const double progress = 1.0 - inner_cost / cand_cost;
        inner_enabled = progress > 1e-3;   // inner_iteration_tolerance
        cand_cost = inner_cost;
      }
    }
    S->time_backsub += now_s() - ts;
    // ParameterToleranceReached
    double sn = 0;
    for (int g = 0; g < o.ng; ++g) if (o.grp_red[g] >= 0) for (int q = 0; q < kMaxIntr; ++q) { const double d = o.intr[(size_t)g * kMaxIntr + q] - o.cintr[(size_t)g * kMaxIntr + q]; sn += d * d; }
    for (int c = 0; c < o.nc; ++c) if (o.cam_red[c] >= 0) for (int q = 0; q < 6; ++q) { const double d = o.cam[6 * c + q] - o.ccam[6 * c + q]; sn += d * d; }
    for (int p = 0; p < o.np; ++p) if (!o.pt_const[p]) for (int q = 0; q < 4; ++q) { const double d = o.pts[4 * p + q] - o.cpts[4 * p + q]; sn += d * d; }
    const double step_norm = std::sqrt(sn);
    if (step_norm <= O->parameter_tolerance * (x_norm + O->parameter_tolerance)) {
      trace_push(S, cand_cost + o.fixed_cost, gmax, step_norm, radius, 0); term = 0; break; }
    // FunctionToleranceReached
    const double cost_change = x_cost - cand_cost;
    if (std::fabs(cost_change) <= O->function_tolerance * x_cost) {
      trace_push(S, cand_cost + o.fixed_cost, gmax, step_norm, radius, 0); term = 0; break; }
    const double relative_decrease = cost_change / model_cost_change;
    if (inner_useful || relative_decrease > 1e-3) {
      o.cam.swap(o.ccam); o.pts.swap(o.cpts); o.intr.swap(o.cintr);
      x_norm = state_norm(o, o.cam, o.pts, o.intr);
      double tl2 = now_s();
      evaluate(o, o.cam, o.pts, o.intr, true, &x_cost);
      apply_scaling(o);
      gmax = compute_gradient(o);
      S->time_linearize += now_s() - tl2;
      radius = radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * relative_decrease - 1.0, 3));
      radius = std::min(O->max_trust_region_radius, radius);
      decrease_factor = 2.0; reuse_diagonal = false; step_successful = true;
      S->num_successful_steps++;
      if (x_cost < minimum_cost) minimum_cost = x_cost;
      trace_push(S, x_cost + o.fixed_cost, gmax, step_norm, radius, 1);
    } else {
      radius /= decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true; step_successful = false;
      trace_push(S, cand_cost + o.fixed_cost, gmax, step_norm, radius, 0);
    }
  }
  S->num_iterations = iter; S->termination_type = term; S->success = term != 2;
  S->final_cost = minimum_cost + o.fixed_cost;
  std::copy(o.cam.begin(), o.cam.end(), P->cam_ext);
  std::copy(o.pts.begin(), o.pts.end(), P->points);
  std::copy(o.intr.begin(), o.intr.end(), P->intrinsics);
  S->solve_time_in_seconds = now_s() - t1;
  if (O->verbose) std::fprintf(stderr, "[oracle] iters=%d term=%d cost %.6e -> %.6e\n", iter, term, S->initial_cost, S->final_cost);
  return 0;
}

}  // extern "C"

// ============================================================ two views, angular epipolar error
// BundleAdjustTwoViewsAngular (bundle_adjust_two_views.cc:189-246) as RefineModel of the relative-pose
// estimator calls it (estimate_relative_pose.cc:111-138): parameter blocks rotation_2 (3) and position_2
// (3, SphereManifold<3>), one AngularEpipolarError residual (angular_epipolar_error.h:54-91) per
// correspondence under one loss, TRUST_REGION / LEVENBERG_MARQUARDT with linear_solver_type CGNR and the
// JACOBI preconditioner.  Ceres is not under /root/reference: CgnrSolver / ConjugateGradientsSolver /
// BlockSparseJacobiPreconditioner are restated from the published 2.2 sources (cgnr_solver.cc,
// conjugate_gradients_solver.h, block_random_access_diagonal_matrix.cc: Invert = llt().solve(I)).
// Inner iterations are not restated (as everywhere in this oracle).
namespace {

template <typename T>
inline void angle_axis_to_rotation_matrix(const T aa[3], T R[9]) {   // ceres/rotation.h; R(i, j) = R[3 i + j]
  const T theta2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
  if (theta2 > std::numeric_limits<double>::epsilon()) {
    const T theta = jsqrt(theta2);
    const T wx = aa[0] / theta, wy = aa[1] / theta, wz = aa[2] / theta;
    const T costheta = jcos(theta), sintheta = jsin(theta);
    R[0] = costheta + wx * wx * (1.0 - costheta);
    R[3] = wz * sintheta + wx * wy * (1.0 - costheta);
    R[6] = -wy * sintheta + wx * wz * (1.0 - costheta);
    R[1] = wx * wy * (1.0 - costheta) - wz * sintheta;
    R[4] = costheta + wy * wy * (1.0 - costheta);
    R[7] = wx * sintheta + wy * wz * (1.0 - costheta);
    R[2] = wy * sintheta + wx * wz * (1.0 - costheta);
    R[5] = -wx * sintheta + wy * wz * (1.0 - costheta);
    R[8] = costheta + wz * wz * (1.0 - costheta);
  } else {
    R[0] = T(1.0); R[3] = aa[2]; R[6] = -aa[1];
    R[1] = -aa[2]; R[4] = T(1.0); R[7] = aa[0];
    R[2] = aa[1]; R[5] = -aa[0]; R[8] = T(1.0);
  }
}

template <typename T>
inline T dot3t(const T* a, const T* b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }

// AngularEpipolarError::operator() -- always "true"; 1000 when the square root would be imaginary
template <typename T>
inline T angular_epipolar_error(const T rotation[3], const T translation[3], const double* c) {
  const T f1[3] = {T(c[0]), T(c[1]), T(1.0)}, f2[3] = {T(c[2]), T(c[3]), T(1.0)};
  T R[9];
  angle_axis_to_rotation_matrix(rotation, R);
  T M[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) M[3 * i + j] = T(i == j ? 1.0 : 0.0) - translation[i] * translation[j];
  T u[3], w[3], Mf1[3], Mw[3];
  for (int i = 0; i < 3; ++i) u[i] = (R[3 * i] * f2[0] + R[3 * i + 1] * f2[1]) + R[3 * i + 2] * f2[2];
  for (int i = 0; i < 3; ++i) w[i] = (R[i] * f2[0] + R[3 + i] * f2[1]) + R[6 + i] * f2[2];
  for (int i = 0; i < 3; ++i) Mf1[i] = (M[3 * i] * f1[0] + M[3 * i + 1] * f1[1]) + M[3 * i + 2] * f1[2];
  for (int i = 0; i < 3; ++i) Mw[i] = (M[3 * i] * w[0] + M[3 * i + 1] * w[1]) + M[3 * i + 2] * w[2];
  const T a = dot3t(f1, Mf1) + dot3t(u, Mw);
  const T cr[3] = {f1[1] * w[2] - f1[2] * w[1], f1[2] * w[0] - f1[0] * w[2], f1[0] * w[1] - f1[1] * w[0]};
  const T b_sqrt = dot3t(translation, cr);
  const T sqrt_term = (a * a) / 4.0 - b_sqrt * b_sqrt;
  if (sqrt_term < 0.0) return T(1000.0);
  return a / 2.0 - jsqrt(sqrt_term);
}

inline void householder3(const double x[3], double v[3], double* beta) {
  const double sigma = x[0] * x[0] + x[1] * x[1];
  v[0] = x[0]; v[1] = x[1]; v[2] = 1.0;
  *beta = 0.0;
  if (sigma <= std::numeric_limits<double>::epsilon()) { if (x[2] < 0.0) *beta = 2.0; return; }
  const double mu = std::sqrt(x[2] * x[2] + sigma);
  const double v_pivot = (x[2] <= 0.0) ? x[2] - mu : -sigma / (x[2] + mu);
  *beta = 2.0 * v_pivot * v_pivot / (sigma + v_pivot * v_pivot);
  v[0] /= v_pivot; v[1] /= v_pivot;
}
inline void sphere3_plus(const double x[3], const double d[2], double out[3]) {
  const double nd = std::sqrt(d[0] * d[0] + d[1] * d[1]);
  if (nd == 0.0) { for (int i = 0; i < 3; ++i) out[i] = x[i]; return; }
  double v[3], beta; householder3(x, v, &beta);
  const double nx = std::sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
  const double sbd = std::sin(nd) / nd;
  const double y[3] = {sbd * d[0], sbd * d[1], std::cos(nd)};
  const double vty = v[0] * y[0] + v[1] * y[1] + v[2] * y[2];
  for (int i = 0; i < 3; ++i) out[i] = nx * (y[i] - v[i] * (beta * vty));
}
inline void sphere3_plus_jacobian(const double x[3], double J[6]) {
  double v[3], beta; householder3(x, v, &beta);
  const double nx = std::sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 2; ++c) J[r * 2 + c] = nx * ((r == c ? 1.0 : 0.0) - beta * v[r] * v[c]);
}

struct TwoViewLin {
  std::vector<double> J;   // n x 5, loss-corrected, column-scaled
  std::vector<double> r;   // n, loss-corrected
  double g[5];             // J_unscaled' r
  double colsq[5];         // squared column norms of the loss-corrected UNSCALED Jacobian
  double cost;
};

inline double two_view_cost(int64_t n, const double* corr, int loss_type, double loss_width, const double x[6]) {
  double cost = 0.0;
  for (int64_t i = 0; i < n; ++i) {
    const double r = angular_epipolar_error<double>(x, x + 3, corr + 4 * i);
    double rho[3]; loss_evaluate(loss_type, loss_width, r * r, rho);
    cost += 0.5 * rho[0];
  }
  return cost;
}

inline void two_view_linearize(int64_t n, const double* corr, int loss_type, double loss_width, const double x[6],
                               const double* scale, TwoViewLin& L) {
  typedef Jet<6> J6;
  L.J.assign((size_t)n * 5, 0.0); L.r.assign(n, 0.0); L.cost = 0.0;
  for (int q = 0; q < 5; ++q) { L.g[q] = 0.0; L.colsq[q] = 0.0; }
  double PJ[6]; sphere3_plus_jacobian(x + 3, PJ);
  J6 rot[3], tr[3];
  for (int q = 0; q < 3; ++q) { rot[q] = J6(x[q], q); tr[q] = J6(x[3 + q], 3 + q); }
  for (int64_t i = 0; i < n; ++i) {
    const J6 e = angular_epipolar_error<J6>(rot, tr, corr + 4 * i);
    double rho[3]; loss_evaluate(loss_type, loss_width, e.a * e.a, rho);
    const double sr = std::sqrt(rho[1]);
    L.cost += 0.5 * rho[0];
    double j[5];
    for (int q = 0; q < 3; ++q) j[q] = e.v[q];
    for (int q = 0; q < 2; ++q) j[3 + q] = (e.v[3] * PJ[q] + e.v[4] * PJ[2 + q]) + e.v[5] * PJ[4 + q];
    const double rr = sr * e.a;
    L.r[i] = rr;
    for (int q = 0; q < 5; ++q) {
      const double ju = sr * j[q];
      L.g[q] += ju * rr; L.colsq[q] += ju * ju;
      L.J[(size_t)i * 5 + q] = ju * scale[q];
    }
  }
}

inline bool zero_or_inf(double v) { return v == 0.0 || std::isinf(v); }

inline void spd_inverse(int n, const double* A, double* Ai) {   // llt().solve(Identity)
  double Lm[9];
  for (int j = 0; j < n; ++j) {
    double s = A[j * n + j];
    for (int k = 0; k < j; ++k) s -= Lm[j * n + k] * Lm[j * n + k];
    const double d = std::sqrt(s);
    Lm[j * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double v = A[i * n + j];
      for (int k = 0; k < j; ++k) v -= Lm[i * n + k] * Lm[j * n + k];
      Lm[i * n + j] = v / d;
    }
  }
  for (int c = 0; c < n; ++c) {
    double z[3];
    for (int i = 0; i < n; ++i) {
      double s = (i == c) ? 1.0 : 0.0;
      for (int k = 0; k < i; ++k) s -= Lm[i * n + k] * z[k];
      z[i] = s / Lm[i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
      double s = z[i];
      for (int k = i + 1; k < n; ++k) s -= Lm[k * n + i] * Ai[k * n + c];
      Ai[i * n + c] = s / Lm[i * n + i];
    }
  }
}

// y ~ argmin |J y - r|^2 + |D y|^2 by preconditioned CG on the normal equations; false = FAILURE
inline bool cgnr_solve(int64_t n, const std::vector<double>& J, const std::vector<double>& res, const double* D2, double* y) {
  auto Amul = [&](const double* v, double* o) {   // CgnrLinearOperator: J'(J v) + D^2 v
    for (int q = 0; q < 5; ++q) o[q] = 0.0;
    for (int64_t i = 0; i < n; ++i) {
      const double* row = &J[(size_t)i * 5];
      double z = 0.0;
      for (int q = 0; q < 5; ++q) z += row[q] * v[q];
      for (int q = 0; q < 5; ++q) o[q] += row[q] * z;
    }
    for (int q = 0; q < 5; ++q) o[q] += D2[q] * v[q];
  };
  double b[5] = {0, 0, 0, 0, 0};
  for (int64_t i = 0; i < n; ++i) for (int q = 0; q < 5; ++q) b[q] += J[(size_t)i * 5 + q] * res[i];
  for (int q = 0; q < 5; ++q) y[q] = 0.0;
  double nb = 0.0;
  for (int q = 0; q < 5; ++q) nb += b[q] * b[q];
  if (std::sqrt(nb) == 0.0) return true;
  double B3[9] = {0}, B2[4] = {0}, I3[9], I2[4];
  for (int64_t i = 0; i < n; ++i) {
    const double* row = &J[(size_t)i * 5];
    for (int a = 0; a < 3; ++a) for (int c = 0; c < 3; ++c) B3[a * 3 + c] += row[a] * row[c];
    for (int a = 0; a < 2; ++a) for (int c = 0; c < 2; ++c) B2[a * 2 + c] += row[3 + a] * row[3 + c];
  }
  for (int a = 0; a < 3; ++a) B3[a * 3 + a] += D2[a];
  for (int a = 0; a < 2; ++a) B2[a * 2 + a] += D2[3 + a];
  spd_inverse(3, B3, I3);
  spd_inverse(2, B2, I2);
  double r[5], z[5], pv[5], qv[5], tmp[5];
  for (int q = 0; q < 5; ++q) r[q] = b[q];
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
    if (pq <= 0.0 || std::isinf(pq)) return true;   // NO_CONVERGENCE ("matrix is indefinite"): usable
    const double alpha = rho / pq;
    if (std::isinf(alpha)) return false;
    for (int q = 0; q < 5; ++q) y[q] = y[q] + alpha * pv[q];
    if (it % 10 == 0) { Amul(y, tmp); for (int q = 0; q < 5; ++q) r[q] = b[q] - tmp[q]; }
    else { for (int q = 0; q < 5; ++q) r[q] = r[q] - alpha * qv[q]; }
    double Q1 = 0.0;
    for (int q = 0; q < 5; ++q) Q1 += y[q] * (b[q] + r[q]);
    Q1 = -Q1;
    const double zeta = it * (Q1 - Q0) / Q1;
    if (zeta < 0.1) return true;
    Q0 = Q1;
    if (it >= 500) return true;
  }
}

}  // namespace

extern "C" {

// pose = rotation_2 | position_2 (in/out); out_int = {success, termination, iterations, successful steps};
// out_cost = {initial, final}
// linear_solver: 0 = exact solve of the normal equations (the direct solver types), 1 = CGNR + JACOBI
int oracle_two_views_angular(int64_t n, const double* corr, int loss_type, double loss_width, int max_num_iterations,
                             double function_tolerance, double gradient_tolerance, double parameter_tolerance,
                             double max_trust_region_radius, int linear_solver, double* pose, int* out_int, double* out_cost) {
  double x[6];
  for (int q = 0; q < 6; ++q) x[q] = pose[q];
  double scale[5] = {1, 1, 1, 1, 1};
  TwoViewLin L;
  two_view_linearize(n, corr, loss_type, loss_width, x, scale, L);
  for (int q = 0; q < 5; ++q) scale[q] = 1.0 / (1.0 + std::sqrt(L.colsq[q]));
  double radius = 1e4, decrease_factor = 2.0;
  bool step_successful = true, need_linearize = true, first = true;
  int iter = 0, invalid_steps = 0, term = 1, nsucc = 0;
  double x_norm = 0.0, minimum_cost = 0.0, gmax = 0.0, x_cost = 0.0, initial_cost = 0.0;
  for (int q = 0; q < 6; ++q) x_norm += x[q] * x[q];
  x_norm = std::sqrt(x_norm);
  while (true) {
    if (need_linearize) {
      two_view_linearize(n, corr, loss_type, loss_width, x, scale, L);
      x_cost = L.cost;
      double ng[5], xp[3];
      for (int q = 0; q < 5; ++q) ng[q] = -L.g[q];
      sphere3_plus(x + 3, ng + 3, xp);
      gmax = 0.0;
      for (int q = 0; q < 3; ++q) gmax = std::max(gmax, std::max(std::fabs(x[q] - (x[q] + ng[q])), std::fabs(x[3 + q] - xp[q])));
      need_linearize = false;
    }
    if (first) {
      first = false;
      initial_cost = minimum_cost = x_cost;
      if (!std::isfinite(x_cost)) { term = 2; break; }
    }
    if (iter >= max_num_iterations) { term = 1; break; }
    if (step_successful && gmax <= gradient_tolerance) { term = 0; break; }
    if (radius <= 1e-32) { term = 0; break; }
    ++iter;
    double d[5], y[5];
    for (int q = 0; q < 5; ++q) d[q] = std::min(std::max(L.colsq[q] * scale[q] * scale[q], 1e-6), 1e32) / radius;
    bool solved;
    if (linear_solver == 1) solved = cgnr_solve(n, L.J, L.r, d, y);
    else {   // (J'J + D^2) y = J'r by dense Cholesky
      std::vector<double> A(25, 0.0), b(5, 0.0);
      for (int64_t i = 0; i < n; ++i) {
        const double* row = &L.J[(size_t)i * 5];
        for (int a2 = 0; a2 < 5; ++a2) { b[a2] += row[a2] * L.r[i]; for (int c2 = 0; c2 <= a2; ++c2) A[a2 * 5 + c2] += row[a2] * row[c2]; }
      }
      for (int a2 = 0; a2 < 5; ++a2) A[a2 * 5 + a2] += d[a2];
      solved = dense_cholesky_solve(5, A, b);
      for (int q = 0; q < 5; ++q) { y[q] = b[q]; if (!std::isfinite(y[q])) solved = false; }
    }
    // model_cost_change = -m'(r + m/2), m = J (-y)
    double mcc = 0.0;
    for (int64_t i = 0; i < n; ++i) {
      double m = 0.0;
      for (int q = 0; q < 5; ++q) m -= L.J[(size_t)i * 5 + q] * y[q];
      mcc -= m * (L.r[i] + m / 2.0);
    }
    double cand[6], dl[5], stepsq = 0.0, xnormsq = 0.0;
    for (int q = 0; q < 5; ++q) dl[q] = -y[q] * scale[q];
    for (int q = 0; q < 3; ++q) cand[q] = x[q] + dl[q];
    sphere3_plus(x + 3, dl + 3, cand + 3);
    for (int q = 0; q < 6; ++q) { stepsq += (x[q] - cand[q]) * (x[q] - cand[q]); xnormsq += cand[q] * cand[q]; }
    const bool step_valid = solved && std::isfinite(mcc) && std::isfinite(stepsq) && mcc > 0.0;
    if (!step_valid) {
      if (++invalid_steps >= 5) { term = 2; break; }
      radius /= decrease_factor; decrease_factor *= 2.0; step_successful = false;
      continue;
    }
    invalid_steps = 0;
    double cand_cost = two_view_cost(n, corr, loss_type, loss_width, cand);
    if (!std::isfinite(cand_cost)) cand_cost = std::numeric_limits<double>::max();
    const double step_norm = std::sqrt(stepsq);
    if (step_norm <= parameter_tolerance * (x_norm + parameter_tolerance)) { term = 0; break; }
    const double cost_change = x_cost - cand_cost;
    if (std::fabs(cost_change) <= function_tolerance * x_cost) { term = 0; break; }
    const double relative_decrease = cost_change / mcc;
    if (relative_decrease > 1e-3) {
      for (int q = 0; q < 6; ++q) x[q] = cand[q];
      x_norm = std::sqrt(xnormsq);
      radius = radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * relative_decrease - 1.0, 3));
      radius = std::min(max_trust_region_radius, radius);
      decrease_factor = 2.0; step_successful = true; need_linearize = true;
      nsucc++;
      if (cand_cost < minimum_cost) minimum_cost = cand_cost;
    } else {
      radius /= decrease_factor; decrease_factor *= 2.0; step_successful = false;
    }
  }
  for (int q = 0; q < 6; ++q) pose[q] = x[q];
  out_int[0] = term != 2; out_int[1] = term; out_int[2] = iter; out_int[3] = nsucc;
  out_cost[0] = initial_cost; out_cost[1] = term != 2 ? minimum_cost : x_cost;
  return 0;
}

// OptimizeHomography (bundle_adjust_two_views.cc:298-358) on H in Eigen's column-major storage order (in/out,
// divided by H(2,2) at the end); SymmetricGeometricDistanceTerms (homography_error.h:45-73) through Jets, with
// Eigen's cofactor 3 x 3 inverse; dense Cholesky for the direct solver.
int oracle_optimize_homography(int64_t n, const double* corr, int loss_type, double loss_width, int max_num_iterations,
                               double function_tolerance, double gradient_tolerance, double parameter_tolerance,
                               double max_trust_region_radius, double* Hcm, int* out_int, double* out_cost) {
  typedef Jet<9> J9;
  auto residuals = [&](const auto* H, const double* c, auto* r) {
    typedef typename std::remove_cv<typename std::remove_reference<decltype(H[0])>::type>::type T;
    auto cof = [&](int i, int j) {
      const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      return H[i1 + 3 * j1] * H[i2 + 3 * j2] - H[i1 + 3 * j2] * H[i2 + 3 * j1];
    };
    const T c0 = cof(0, 0), c1 = cof(1, 0), c2 = cof(2, 0);
    const T det = (c0 * H[0] + c1 * H[1]) + c2 * H[2];
    const T invdet = 1.0 / det;
    T G[9];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) G[i + 3 * j] = (i == 0) ? ((j == 0 ? c0 : (j == 1 ? c1 : c2)) * invdet) : cof(j, i) * invdet;
    const double x[3] = {c[0], c[1], 1.0}, y[3] = {c[2], c[3], 1.0};
    T p[3], q[3];
    for (int i = 0; i < 3; ++i) {
      p[i] = (H[i] * x[0] + H[i + 3] * x[1]) + H[i + 6] * x[2];
      q[i] = (G[i] * y[0] + G[i + 3] * y[1]) + G[i + 6] * y[2];
    }
    r[0] = p[0] / p[2] - y[0]; r[1] = p[1] / p[2] - y[1];
    r[2] = q[0] / q[2] - x[0]; r[3] = q[1] / q[2] - x[1];
  };
  auto cost_of = [&](const double* H) {
    double cost = 0.0;
    for (int64_t i = 0; i < n; ++i) {
      double r[4]; residuals(H, corr + 4 * i, r);
      double rho[3]; loss_evaluate(loss_type, loss_width, (r[0] * r[0] + r[1] * r[1]) + (r[2] * r[2] + r[3] * r[3]), rho);
      cost += 0.5 * rho[0];
    }
    return cost;
  };
  std::vector<double> Jm, rm;   // 4n x 9 (loss-corrected, scaled), 4n
  double g[9], colsq[9], scale[9];
  for (int q = 0; q < 9; ++q) scale[q] = 1.0;
  auto linearize = [&](const double* H, double* cost) {
    Jm.assign((size_t)n * 36, 0.0); rm.assign((size_t)n * 4, 0.0); *cost = 0.0;
    for (int q = 0; q < 9; ++q) { g[q] = 0.0; colsq[q] = 0.0; }
    J9 Hj[9];
    for (int q = 0; q < 9; ++q) Hj[q] = J9(H[q], q);
    for (int64_t i = 0; i < n; ++i) {
      J9 r[4]; residuals(Hj, corr + 4 * i, r);
      double rho[3]; loss_evaluate(loss_type, loss_width, (r[0].a * r[0].a + r[1].a * r[1].a) + (r[2].a * r[2].a + r[3].a * r[3].a), rho);
      const double sr = std::sqrt(rho[1]);
      *cost += 0.5 * rho[0];
      for (int e = 0; e < 4; ++e) {
        const double rr = sr * r[e].a;
        rm[(size_t)i * 4 + e] = rr;
        for (int q = 0; q < 9; ++q) {
          const double ju = sr * r[e].v[q];
          g[q] += ju * rr; colsq[q] += ju * ju;
          Jm[((size_t)i * 4 + e) * 9 + q] = ju * scale[q];
        }
      }
    }
  };
  double x[9];
  for (int q = 0; q < 9; ++q) x[q] = Hcm[q];
  double x_cost = 0.0;
  linearize(x, &x_cost);
  for (int q = 0; q < 9; ++q) scale[q] = 1.0 / (1.0 + std::sqrt(colsq[q]));
  double radius = 1e4, decrease_factor = 2.0;
  bool step_successful = true, need_linearize = true, first = true;
  int iter = 0, invalid_steps = 0, term = 1, nsucc = 0;
  double x_norm = 0.0, minimum_cost = 0.0, gmax = 0.0, initial_cost = 0.0;
  for (int q = 0; q < 9; ++q) x_norm += x[q] * x[q];
  x_norm = std::sqrt(x_norm);
  while (true) {
    if (need_linearize) {
      linearize(x, &x_cost);
      gmax = 0.0;
      for (int q = 0; q < 9; ++q) gmax = std::max(gmax, std::fabs(g[q]));
      need_linearize = false;
    }
    if (first) { first = false; initial_cost = minimum_cost = x_cost; if (!std::isfinite(x_cost)) { term = 2; break; } }
    if (iter >= max_num_iterations) { term = 1; break; }
    if (step_successful && gmax <= gradient_tolerance) { term = 0; break; }
    if (radius <= 1e-32) { term = 0; break; }
    ++iter;
    std::vector<double> A(81, 0.0), b(9, 0.0);
    for (size_t i = 0; i < rm.size(); ++i) {
      const double* row = &Jm[i * 9];
      for (int a2 = 0; a2 < 9; ++a2) { b[a2] += row[a2] * rm[i]; for (int c2 = 0; c2 <= a2; ++c2) A[a2 * 9 + c2] += row[a2] * row[c2]; }
    }
    for (int q = 0; q < 9; ++q) A[q * 9 + q] += std::min(std::max(colsq[q] * scale[q] * scale[q], 1e-6), 1e32) / radius;
    bool solved = dense_cholesky_solve(9, A, b);
    double y[9];
    for (int q = 0; q < 9; ++q) { y[q] = b[q]; if (!std::isfinite(y[q])) solved = false; }
    double mcc = 0.0;
    for (size_t i = 0; i < rm.size(); ++i) {
      double m = 0.0;
      for (int q = 0; q < 9; ++q) m -= Jm[i * 9 + q] * y[q];
      mcc -= m * (rm[i] + m / 2.0);
    }
    double cand[9], stepsq = 0.0, xnormsq = 0.0;
    for (int q = 0; q < 9; ++q) { cand[q] = x[q] - y[q] * scale[q]; stepsq += (x[q] - cand[q]) * (x[q] - cand[q]); xnormsq += cand[q] * cand[q]; }
    const bool step_valid = solved && std::isfinite(mcc) && std::isfinite(stepsq) && mcc > 0.0;
    if (!step_valid) {
      if (++invalid_steps >= 5) { term = 2; break; }
      radius /= decrease_factor; decrease_factor *= 2.0; step_successful = false;
      continue;
    }
    invalid_steps = 0;
    double cand_cost = cost_of(cand);
    if (!std::isfinite(cand_cost)) cand_cost = std::numeric_limits<double>::max();
    const double step_norm = std::sqrt(stepsq);
    if (step_norm <= parameter_tolerance * (x_norm + parameter_tolerance)) { term = 0; break; }
    const double cost_change = x_cost - cand_cost;
    if (std::fabs(cost_change) <= function_tolerance * x_cost) { term = 0; break; }
    const double relative_decrease = cost_change / mcc;
    if (relative_decrease > 1e-3) {
      for (int q = 0; q < 9; ++q) x[q] = cand[q];
      x_norm = std::sqrt(xnormsq);
      radius = radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * relative_decrease - 1.0, 3));
      radius = std::min(max_trust_region_radius, radius);
      decrease_factor = 2.0; step_successful = true; need_linearize = true;
      nsucc++;
      if (cand_cost < minimum_cost) minimum_cost = cand_cost;
    } else {
      radius /= decrease_factor; decrease_factor *= 2.0; step_successful = false;
    }
  }
  const double h22 = x[8];
  for (int q = 0; q < 9; ++q) Hcm[q] = x[q] / h22;
  out_int[0] = term != 2; out_int[1] = term; out_int[2] = iter; out_int[3] = nsucc;
  out_cost[0] = initial_cost; out_cost[1] = term != 2 ? minimum_cost : x_cost;
  return 0;
}

double oracle_angular_epipolar_error(const double* rotation, const double* position, const double* corr) {
  return angular_epipolar_error<double>(rotation, position, corr);
}

}  // extern "C"

// ============================================================ inverse-depth parametrisation
// BundleAdjustmentOptions::use_inverse_depth_parametrization (bundle_adjuster.cc:223-289, 594-622): the variable of a
// track is Track::InverseDepth() along Track::ReferenceBearingVector() in the frame of its reference view; a residual
// block is InvReprojectionError (reprojection_error.h:173-229: reference extrinsics, other extrinsics, other intrinsics,
// inverse depth) or, for the reference view's own feature, InvReprojectionPoseError (:232-286).  Restated with Jets over
// exactly those parameters and solved as ONE dense Levenberg-Marquardt problem (normal equations of the full scaled
// Jacobian, no Schur complement): the same steps Ceres' exact SPARSE_SCHUR solve produces, reached another way than
// the device's per-track elimination.  The intrinsics of a group are a parameter block of every residual of its cameras
// (optimised on the subset of BundleAdjustmentOptions::intrinsics_to_optimize, bundle_adjuster.cc:382-460, bounds as the
// main path: box projection of the step); AddViewPriors (bundle_adjuster.cc:289-313) adds the position / gravity /
// orientation prior rows of the optimised views (no loss function).  No inner iterations (the reference passes no
// inner ordering in this mode and leaves the choice to Ceres' graph heuristic).
namespace {

template <typename T>
inline bool inv_reprojection_error(int model, const T* ext_ref, const T* ext_oth, bool same_view, const T* k,
                                   const double* bearing, const T& rho, const double uv[2], const double si[2], T* res) {
  T p_ref[3] = {T(bearing[0]) / rho, T(bearing[1]) / rho, T(bearing[2]) / rho};
  T pc[3];
  if (same_view) {
    // T_world_ref.inverse() * T_world_ref * point_ref: the identity up to round-off
    pc[0] = p_ref[0]; pc[1] = p_ref[1]; pc[2] = p_ref[2];
  } else {
    const T minus_w[3] = {-ext_ref[3], -ext_ref[4], -ext_ref[5]};
    T pw[3];
    angle_axis_rotate_point(minus_w, p_ref, pw);                       // R_ref^T p
    T d[3] = {pw[0] + ext_ref[0] - ext_oth[0], pw[1] + ext_ref[1] - ext_oth[1], pw[2] + ext_ref[2] - ext_oth[2]};
    angle_axis_rotate_point(ext_oth + 3, d, pc);                       // R_oth (X_w - c_oth)
  }
  T pix[2];
  const bool ok = project(model, k, pc, pix);
  res[0] = si[0] * (pix[0] - uv[0]);
  res[1] = si[1] * (pix[1] - uv[1]);
  return ok;
}

}  // namespace

extern "C" int oracle_ba_solve_inverse_depth(oba_problem* P, const int32_t* point_ref_cam, const double* point_ref_bearing,
                                              double* point_inverse_depth, const oba_options* O, oba_summary* S) {
  const int nc = P->num_cameras, np = P->num_points, ng = P->num_groups;
  const int64_t nobs = P->num_obs;
  S->trace_size = 0; S->success = 0; S->num_iterations = 0; S->num_successful_steps = 0;
  const double one[2] = {1.0, 1.0};
  // variable blocks: cameras that see a variable track (as reference or as observer) and are not constant
  std::vector<uint8_t> cam_mask(nc, 0), cam_used(nc, 0), grp_used(ng, 0);
  for (int c = 0; c < nc; ++c) {
    unsigned m = 0;
    const int cc = P->cam_const ? P->cam_const[c] : 0;
    if ((cc & 1) || O->constant_camera_position) m |= 0x07;
    if ((cc & 2) || O->constant_camera_orientation) m |= 0x38;
    cam_mask[c] = (uint8_t)m;
  }
  for (int64_t i = 0; i < nobs; ++i) { cam_used[P->obs_cam[i]] = 1; cam_used[point_ref_cam[P->obs_pt[i]]] = 1; grp_used[P->cam_group[P->obs_cam[i]]] = 1; }
  std::vector<int> cam_col(nc, -1), pt_col(np, -1), grp_col((size_t)ng * kMaxIntr, -1);
  std::vector<uint8_t> grp_var(ng, 0);
  int ncol = 0;
  for (int c = 0; c < nc; ++c) if (cam_used[c] && (cam_mask[c] & 0x3f) != 0x3f) { cam_col[c] = ncol; ncol += 6; }
  for (int g = 0; g < ng; ++g) {   // one column per free parameter of a variable group (SubsetManifold)
    const unsigned fm = intrinsics_free_mask(P->group_model[g], O->intrinsics_to_optimize);
    if ((P->group_const && P->group_const[g]) || fm == 0 || !grp_used[g]) continue;
    grp_var[g] = 1;
    for (int q = 0; q < kMaxIntr; ++q) if ((fm >> q) & 1u) grp_col[(size_t)g * kMaxIntr + q] = ncol++;
  }
  std::vector<uint8_t> pt_used(np, 0);
  for (int64_t i = 0; i < nobs; ++i) pt_used[P->obs_pt[i]] = 1;
  for (int p = 0; p < np; ++p) if (pt_used[p] && !(P->point_const && P->point_const[p])) pt_col[p] = ncol++;
  std::vector<double> cam(P->cam_ext, P->cam_ext + 6 * (size_t)nc), rho(point_inverse_depth, point_inverse_depth + np);
  std::vector<double> intr(P->intrinsics, P->intrinsics + (size_t)kMaxIntr * ng);
  for (int g = 0; g < ng; ++g) if (grp_var[g]) project_intrinsics_to_bounds(P->group_model[g], &intr[(size_t)g * kMaxIntr]);
  // camera priors (AddViewPriors): rows of the optimised views; on a constant camera a fixed cost
  struct Prior { int cam, kind; const double* vec; const double* sqrt_info; };
  std::vector<Prior> priors;
  double fixed_cost = 0.0;
  if (P->cam_prior_mask && O->prior_mask) {
    const double* vecs[3] = {P->cam_position_prior, P->cam_gravity_prior, P->cam_orientation_prior};
    const double* infos[3] = {P->cam_position_prior_sqrt_info, P->cam_gravity_prior_sqrt_info, P->cam_orientation_prior_sqrt_info};
    for (int c = 0; c < nc; ++c)
      for (int k = 0; k < 3; ++k) {
        const int bit = 1 << k;
        if (!cam_used[c] || !(P->cam_prior_mask[c] & bit) || !(O->prior_mask & bit) || !vecs[k] || !infos[k]) continue;
        if (cam_col[c] >= 0) priors.push_back(Prior{c, bit, vecs[k] + 3 * c, infos[k] + 9 * c});
        else {
          double r3[3];
          camera_prior_residual<double>(bit, &cam[6 * (size_t)c], vecs[k] + 3 * c, infos[k] + 9 * c, r3);
          fixed_cost += 0.5 * ((r3[0] * r3[0] + r3[1] * r3[1]) + r3[2] * r3[2]);
        }
      }
  }
  const int64_t nrow = 2 * nobs + 3 * (int64_t)priors.size();
  std::vector<double> J, r((size_t)nrow);

  auto evaluate = [&](const std::vector<double>& cm, const std::vector<double>& rh, const std::vector<double>& kk, bool want_jac, double* cost) {
    bool ok = true;
    double cst = 0.0;
    if (want_jac) J.assign((size_t)nrow * ncol, 0.0);
    for (int64_t i = 0; i < nobs; ++i) {
      const int c = P->obs_cam[i], p = P->obs_pt[i], cr = point_ref_cam[p], g = P->cam_group[c];
      const int model = P->group_model[g];
      const double* ki = &kk[(size_t)g * kMaxIntr];
      const double* si = P->obs_sqrt_info ? P->obs_sqrt_info + 2 * i : one;
      const bool same = c == cr;
      double res[2];
      typedef Jet<13 + kMaxIntr> JT;
      JT rr[2];
      if (want_jac) {
        JT er[6], eo[6], kj[kMaxIntr];
        for (int q = 0; q < 6; ++q) { er[q] = JT(cm[6 * (size_t)cr + q], q); eo[q] = same ? er[q] : JT(cm[6 * (size_t)c + q], 6 + q); }
        for (int q = 0; q < kMaxIntr; ++q) kj[q] = JT(ki[q], 13 + q);
        const JT rj(rh[p], 12);
        if (!inv_reprojection_error<JT>(model, er, eo, same, kj, point_ref_bearing + 3 * (size_t)p, rj, P->obs_uv + 2 * i, si, rr)) ok = false;
        res[0] = rr[0].a; res[1] = rr[1].a;
      } else {
        if (!inv_reprojection_error<double>(model, &cm[6 * (size_t)cr], &cm[6 * (size_t)c], same, ki, point_ref_bearing + 3 * (size_t)p,
                                            rh[p], P->obs_uv + 2 * i, si, res)) ok = false;
      }
      double lr[3];
      loss_evaluate(O->loss_function_type, O->robust_loss_width, res[0] * res[0] + res[1] * res[1], lr);
      cst += 0.5 * lr[0];
      if (!want_jac) continue;
      const double sr = std::sqrt(lr[1]);
      for (int a = 0; a < 2; ++a) {
        r[2 * i + a] = sr * res[a];
        double* row = &J[((size_t)2 * i + a) * ncol];
        if (cam_col[cr] >= 0) for (int q = 0; q < 6; ++q) if (!((cam_mask[cr] >> q) & 1)) row[cam_col[cr] + q] += sr * rr[a].v[q];
        if (!same && cam_col[c] >= 0) for (int q = 0; q < 6; ++q) if (!((cam_mask[c] >> q) & 1)) row[cam_col[c] + q] += sr * rr[a].v[6 + q];
        for (int q = 0; q < kMaxIntr; ++q) { const int gc = grp_col[(size_t)g * kMaxIntr + q]; if (gc >= 0) row[gc] += sr * rr[a].v[13 + q]; }
        if (pt_col[p] >= 0) row[pt_col[p]] += sr * rr[a].v[12];
      }
    }
    for (size_t k = 0; k < priors.size(); ++k) {
      const Prior& pr = priors[k];
      typedef Jet<6> J6;
      J6 e[6], r3[3];
      for (int q = 0; q < 6; ++q) e[q] = J6(cm[6 * (size_t)pr.cam + q], q);
      camera_prior_residual<J6>(pr.kind, e, pr.vec, pr.sqrt_info, r3);
      for (int a = 0; a < 3; ++a) {
        cst += 0.5 * r3[a].a * r3[a].a;
        if (!want_jac) continue;
        const int64_t rw = 2 * nobs + 3 * (int64_t)k + a;
        r[rw] = r3[a].a;
        for (int q = 0; q < 6; ++q) if (!((cam_mask[pr.cam] >> q) & 1)) J[(size_t)rw * ncol + cam_col[pr.cam] + q] += r3[a].v[q];
      }
    }
    *cost = cst;
    return ok;
  };
  auto state_norm = [&](const std::vector<double>& cm, const std::vector<double>& rh, const std::vector<double>& kk) {
    double sn = 0.0;
    for (int g = 0; g < ng; ++g) if (grp_var[g]) { const int K = intrinsics_size(P->group_model[g]); for (int q = 0; q < K; ++q) sn += kk[(size_t)g * kMaxIntr + q] * kk[(size_t)g * kMaxIntr + q]; }
    for (int c = 0; c < nc; ++c) if (cam_col[c] >= 0) for (int q = 0; q < 6; ++q) sn += cm[6 * (size_t)c + q] * cm[6 * (size_t)c + q];
    for (int p = 0; p < np; ++p) if (pt_col[p] >= 0) sn += rh[p] * rh[p];
    return std::sqrt(sn);
  };
  double x_cost;
  if (!evaluate(cam, rho, intr, true, &x_cost)) { S->termination_type = 2; S->initial_cost = S->final_cost = x_cost + fixed_cost; return 0; }
  // Jacobi scaling from the column norms at the start
  std::vector<double> scale(ncol, 1.0), g(ncol), H, y(ncol), diag(ncol);
  for (int k = 0; k < ncol; ++k) { double sq = 0.0; for (int64_t rw = 0; rw < nrow; ++rw) sq += J[(size_t)rw * ncol + k] * J[(size_t)rw * ncol + k]; scale[k] = 1.0 / (1.0 + std::sqrt(sq)); }
  auto apply_scale_and_gradient = [&]() {
    for (int64_t rw = 0; rw < nrow; ++rw) for (int k = 0; k < ncol; ++k) J[(size_t)rw * ncol + k] *= scale[k];
    double gm = 0.0;
    for (int k = 0; k < ncol; ++k) { double s2 = 0.0; for (int64_t rw = 0; rw < nrow; ++rw) s2 += J[(size_t)rw * ncol + k] * r[rw]; g[k] = s2; gm = std::max(gm, std::fabs(s2 / scale[k])); }
    return gm;
  };
  double gmax = apply_scale_and_gradient();
  double x_norm = state_norm(cam, rho, intr);
  S->initial_cost = x_cost + fixed_cost;
  double radius = 1e4, decrease_factor = 2.0, minimum_cost = x_cost;
  bool step_successful = true, reuse = false;
  int iter = 0, invalid_steps = 0, term = 1;
  trace_push(S, x_cost + fixed_cost, gmax, 0.0, radius, 1);
  std::vector<double> ccam, crho, cintr;
  while (true) {
    if (iter >= O->max_num_iterations) { term = 1; break; }
    if (step_successful && gmax <= O->gradient_tolerance) { term = 0; break; }
    if (radius <= 1e-32) { term = 0; break; }
    ++iter;
    if (!reuse) {
      H.assign((size_t)ncol * ncol, 0.0);
      for (int64_t rw = 0; rw < nrow; ++rw) {
        const double* row = &J[(size_t)rw * ncol];
        for (int a = 0; a < ncol; ++a) { if (row[a] == 0.0) continue; for (int b = 0; b <= a; ++b) H[(size_t)a * ncol + b] += row[a] * row[b]; }
      }
      for (int a = 0; a < ncol; ++a) diag[a] = std::min(std::max(H[(size_t)a * ncol + a], 1e-6), 1e32);
    }
    reuse = true;
    // (H + D) y = g, Cholesky
    std::vector<double> L(H);
    for (int a = 0; a < ncol; ++a) L[(size_t)a * ncol + a] += diag[a] / radius;
    bool solved = true;
    for (int i = 0; i < ncol && solved; ++i)
      for (int j = 0; j <= i; ++j) {
        double s2 = L[(size_t)i * ncol + j];
        for (int k = 0; k < j; ++k) s2 -= L[(size_t)i * ncol + k] * L[(size_t)j * ncol + k];
        if (i == j) { if (!(s2 > 0.0)) { solved = false; break; } L[(size_t)i * ncol + i] = std::sqrt(s2); }
        else L[(size_t)i * ncol + j] = s2 / L[(size_t)j * ncol + j];
      }
    double model_cost_change = 0.0;
    if (solved) {
      for (int i = 0; i < ncol; ++i) { double s2 = g[i]; for (int k = 0; k < i; ++k) s2 -= L[(size_t)i * ncol + k] * y[k]; y[i] = s2 / L[(size_t)i * ncol + i]; }
      for (int i = ncol - 1; i >= 0; --i) { double s2 = y[i]; for (int k = i + 1; k < ncol; ++k) s2 -= L[(size_t)k * ncol + i] * y[k]; y[i] = s2 / L[(size_t)i * ncol + i]; }
      for (int64_t rw = 0; rw < nrow; ++rw) {
        double m = 0.0;
        const double* row = &J[(size_t)rw * ncol];
        for (int k = 0; k < ncol; ++k) m -= row[k] * y[k];
        model_cost_change -= m * (r[rw] + m / 2.0);
      }
    }
    if (!(solved && std::isfinite(model_cost_change) && model_cost_change > 0.0)) {
      if (++invalid_steps >= 5) { term = 2; break; }
      radius /= decrease_factor; decrease_factor *= 2.0; step_successful = false;
      trace_push(S, x_cost + fixed_cost, gmax, 0.0, radius, 0);
      continue;
    }
    invalid_steps = 0;
    ccam = cam; crho = rho; cintr = intr;
    for (int c = 0; c < nc; ++c) if (cam_col[c] >= 0) for (int q = 0; q < 6; ++q) if (!((cam_mask[c] >> q) & 1))
      ccam[6 * (size_t)c + q] = cam[6 * (size_t)c + q] - y[cam_col[c] + q] * scale[cam_col[c] + q];
    for (int gq = 0; gq < ng * kMaxIntr; ++gq) if (grp_col[gq] >= 0) cintr[gq] = intr[gq] - y[grp_col[gq]] * scale[grp_col[gq]];
    for (int gi = 0; gi < ng; ++gi) if (grp_var[gi]) project_intrinsics_to_bounds(P->group_model[gi], &cintr[(size_t)gi * kMaxIntr]);
    for (int p = 0; p < np; ++p) if (pt_col[p] >= 0) crho[p] = rho[p] - y[pt_col[p]] * scale[pt_col[p]];
    double cand_cost;
    if (!evaluate(ccam, crho, cintr, false, &cand_cost)) cand_cost = std::numeric_limits<double>::max();
    double sn = 0.0;
    for (int c = 0; c < nc; ++c) if (cam_col[c] >= 0) for (int q = 0; q < 6; ++q) { const double d = cam[6 * (size_t)c + q] - ccam[6 * (size_t)c + q]; sn += d * d; }
    for (int gq = 0; gq < ng * kMaxIntr; ++gq) if (grp_var[gq / kMaxIntr]) { const double d = intr[gq] - cintr[gq]; sn += d * d; }
    for (int p = 0; p < np; ++p) if (pt_col[p] >= 0) { const double d = rho[p] - crho[p]; sn += d * d; }
    const double step_norm = std::sqrt(sn);
    if (step_norm <= O->parameter_tolerance * (x_norm + O->parameter_tolerance)) { trace_push(S, cand_cost + fixed_cost, gmax, step_norm, radius, 0); term = 0; break; }
    const double cost_change = x_cost - cand_cost;
    if (std::fabs(cost_change) <= O->function_tolerance * x_cost) { trace_push(S, cand_cost + fixed_cost, gmax, step_norm, radius, 0); term = 0; break; }
    const double relative_decrease = cost_change / model_cost_change;
    if (relative_decrease > 1e-3) {
      cam.swap(ccam); rho.swap(crho); intr.swap(cintr);
      x_norm = state_norm(cam, rho, intr);
      evaluate(cam, rho, intr, true, &x_cost);
      gmax = apply_scale_and_gradient();
      radius = std::min(O->max_trust_region_radius, radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * relative_decrease - 1.0, 3)));
      decrease_factor = 2.0; reuse = false; step_successful = true;
      S->num_successful_steps++;
      if (x_cost < minimum_cost) minimum_cost = x_cost;
      trace_push(S, x_cost + fixed_cost, gmax, step_norm, radius, 1);
    } else {
      radius /= decrease_factor; decrease_factor *= 2.0; step_successful = false;
      trace_push(S, cand_cost + fixed_cost, gmax, step_norm, radius, 0);
    }
  }
  S->num_iterations = iter; S->termination_type = term; S->success = term != 2;
  S->final_cost = minimum_cost + fixed_cost;
  std::copy(cam.begin(), cam.end(), P->cam_ext);
  std::copy(intr.begin(), intr.end(), P->intrinsics);
  std::copy(rho.begin(), rho.end(), point_inverse_depth);
  return 0;
}
