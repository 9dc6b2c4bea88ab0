// Exercises the C++ shim end to end on the device: a synthetic pinhole scene built with plain C++ (no Theia types),
// perturbed, adjusted through BundleAdjuster::Optimize(); then a batch of relative-pose RANSAC problems.
// Prints one "ok" line per check and exits non-zero on failure (tests/test_shim_gpu.py runs it).
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

#include "bundle_adjuster_hip.h"

using namespace theia_hip_shim;

static void rotate(const double w[3], const double p[3], double out[3]) {   // Rodrigues
  const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  if (th2 < 1e-30) { for (int i = 0; i < 3; ++i) out[i] = p[i]; return; }
  const double th = std::sqrt(th2), c = std::cos(th), s = std::sin(th);
  const double k[3] = {w[0] / th, w[1] / th, w[2] / th};
  const double kxp[3] = {k[1] * p[2] - k[2] * p[1], k[2] * p[0] - k[0] * p[2], k[0] * p[1] - k[1] * p[0]};
  const double kp = k[0] * p[0] + k[1] * p[1] + k[2] * p[2];
  for (int i = 0; i < 3; ++i) out[i] = p[i] * c + kxp[i] * s + k[i] * kp * (1 - c);
}

int main() {
  std::mt19937 gen(7);
  std::uniform_real_distribution<double> U(-1.0, 1.0);
  std::normal_distribution<double> N(0.0, 1.0);
  const int nv = 12, nt = 600;
  // the caller's "Reconstruction": cameras, one shared intrinsics block, tracks
  std::vector<double> cam(6 * nv), cam_true(6 * nv), pts(4 * nt), pts_true(4 * nt);
  double intr[7] = {800.0, 1.0, 0.0, 500.0, 400.0, 0.0, 0.0};   // PinholeCameraModel: f, aspect, skew, px, py, k1, k2
  for (int v = 0; v < nv; ++v) {
    double* e = &cam_true[6 * v];
    e[0] = 4.0 * std::cos(0.5 * v) + 0.2 * U(gen); e[1] = 0.3 * U(gen); e[2] = -8.0 + 0.5 * U(gen);
    e[3] = 0.05 * U(gen); e[4] = -0.12 * std::cos(0.5 * v); e[5] = 0.05 * U(gen);
  }
  for (int t = 0; t < nt; ++t) { pts_true[4 * t] = 3 * U(gen); pts_true[4 * t + 1] = 2 * U(gen); pts_true[4 * t + 2] = 2 * U(gen); pts_true[4 * t + 3] = 1.0; }
  cam = cam_true; pts = pts_true;
  for (int v = 1; v < nv; ++v) for (int q = 0; q < 6; ++q) cam[6 * v + q] += (q < 3 ? 0.05 : 0.01) * N(gen);
  for (int t = 0; t < nt; ++t) for (int q = 0; q < 3; ++q) pts[4 * t + q] += 0.05 * N(gen);

  BundleAdjustmentOptions opts;
  opts.max_num_iterations = 30;
  BundleAdjuster ba(opts);
  ba.AddIntrinsicsGroup(0, THEIA_CAM_PINHOLE, intr, 7);
  for (int v = 0; v < nv; ++v) ba.AddCamera(v, &cam[6 * v], 0);
  for (int t = 0; t < nt; ++t) ba.AddTrack(t, &pts[4 * t]);
  int nobs = 0;
  const double cov[2] = {1.0, 1.0};
  for (int t = 0; t < nt; ++t)
    for (int v = 0; v < nv; ++v) {
      if (((t * 31 + v * 17) % 5) > 2) continue;   // each track in ~60 % of the views
      const double* e = &cam_true[6 * v];
      const double d[3] = {pts_true[4 * t] - e[0], pts_true[4 * t + 1] - e[1], pts_true[4 * t + 2] - e[2]};
      double q[3];
      rotate(e + 3, d, q);
      if (q[2] <= 0.1) continue;
      const double uv[2] = {intr[0] * q[0] / q[2] + intr[3] + 0.3 * N(gen), intr[0] * intr[1] * q[1] / q[2] + intr[4] + 0.3 * N(gen)};
      ba.AddObservation(v, t, uv, cov);
      ++nobs;
    }
  ba.SetCameraExtrinsicsConstant(0);                 // gauge
  ba.SetCameraPositionConstant(nv / 2);
  const BundleAdjustmentSummary s = ba.Optimize();
  if (!ba.error().empty()) { std::printf("FAIL: %s\n", ba.error().c_str()); return 1; }
  double cam_err = 0.0;
  for (int v = 0; v < nv; ++v) for (int q = 0; q < 6; ++q) cam_err = std::fmax(cam_err, std::fabs(cam[6 * v + q] - cam_true[6 * v + q]));
  std::printf("BA: %d observations, cost %.6e -> %.6e, success %d, max camera error %.3e\n", nobs, s.initial_cost, s.final_cost, (int)s.success, cam_err);
  // 0.5 * sum r^2 at 0.3 px noise: ~0.09 per observation
  if (!s.success || !(s.final_cost < 0.01 * s.initial_cost) || !(s.final_cost < 0.2 * nobs) || !(cam_err < 0.15)) { std::printf("FAIL: bundle adjustment\n"); return 1; }
  std::printf("ok bundle adjustment\n");

  // ---- relative pose RANSAC, 6 pairs as one batch
  std::vector<std::vector<double>> corr(6);
  std::vector<std::vector<double>> true_rot(6);
  for (int p = 0; p < 6; ++p) {
    const double w[3] = {0.1 * U(gen), 0.2 * U(gen), 0.1 * U(gen)};
    double t[3] = {U(gen), 0.3 * U(gen), 0.2 * U(gen)};
    const double tn = std::sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
    for (double& x : t) x /= tn;
    true_rot[p] = {w[0], w[1], w[2]};
    for (int i = 0; i < 400; ++i) {
      const double X[3] = {2 * U(gen), 2 * U(gen), 6 + 2 * U(gen)};
      const double d[3] = {X[0] - t[0], X[1] - t[1], X[2] - t[2]};
      double q[3];
      rotate(w, d, q);
      double x1 = X[0] / X[2], y1 = X[1] / X[2], x2 = q[0] / q[2], y2 = q[1] / q[2];
      if (i % 4 == 0) { x2 = U(gen); y2 = U(gen); }   // 25 % outliers
      else { x1 += 5e-4 * N(gen); y1 += 5e-4 * N(gen); x2 += 5e-4 * N(gen); y2 += 5e-4 * N(gen); }
      corr[p].insert(corr[p].end(), {x1, y1, x2, y2});
    }
  }
  RansacParameters rp;
  rp.error_thresh = 2.5e-3 * 2.5e-3; rp.min_iterations = 200; rp.max_iterations = 2000; rp.seed = 5;
  std::vector<bool> ok; std::vector<RelativePose> poses; std::vector<RansacSummary> sums; std::string err;
  if (!EstimateRelativePoseBatch(rp, corr, &ok, &poses, &sums, &err)) { std::printf("FAIL: %s\n", err.c_str()); return 1; }
  for (int p = 0; p < 6; ++p) {
    // rotation error through trace(R_est * R_true^T)
    double Rt[9];
    for (int c = 0; c < 3; ++c) { double e[3] = {0, 0, 0}; e[c] = 1.0; double col[3]; rotate(true_rot[p].data(), e, col); for (int r = 0; r < 3; ++r) Rt[3 * r + c] = col[r]; }
    double tr = 0.0;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) tr += poses[p].rotation[3 * i + j] * Rt[3 * i + j];
    const double ang = std::acos(std::fmin(1.0, std::fmax(-1.0, (tr - 1.0) / 2.0))) * 180.0 / M_PI;
    std::printf("pair %d: success %d, %zu inliers of 400, %d iterations, rotation error %.3f deg\n", p, (int)ok[p], sums[p].inliers.size(), sums[p].num_iterations, ang);
    if (!ok[p] || sums[p].inliers.size() < 250 || !(ang < 1.0)) { std::printf("FAIL: relative pose\n"); return 1; }
  }
  std::printf("ok relative pose batch\n");
  return 0;
}
