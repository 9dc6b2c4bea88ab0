// Exercises the C++ shim end to end on the device: a synthetic pinhole scene built with plain C++ (no Theia types),
// perturbed, adjusted through BundleAdjuster::Optimize(); then batches of relative-pose, absolute-pose (KNEIP / DLS / SQPnP) and
// fundamental-matrix RANSAC problems, BundleAdjustView for every camera as one launch, and track covariances.
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

  // ---- calibrated absolute pose, the three PnP types, 5 problems each as one batch; the same data through the
  // fundamental-matrix / homography front ends of the generic routine would be meaningless, so those get their own data
  {
    std::vector<std::vector<double>> c23(5);
    std::vector<std::vector<double>> wtrue(5), ctrue(5);
    for (int p = 0; p < 5; ++p) {
      const double w[3] = {0.2 * U(gen), 0.2 * U(gen), 0.2 * U(gen)};
      const double c[3] = {U(gen), U(gen), -6.0 + U(gen)};
      wtrue[p] = {w[0], w[1], w[2]}; ctrue[p] = {c[0], c[1], c[2]};
      for (int i = 0; i < 300; ++i) {
        const double X[3] = {2 * U(gen), 2 * U(gen), 2 * U(gen)};
        const double d[3] = {X[0] - c[0], X[1] - c[1], X[2] - c[2]};
        double q[3];
        rotate(w, d, q);
        double u = q[0] / q[2] + 5e-4 * N(gen), v = q[1] / q[2] + 5e-4 * N(gen);
        if (i % 5 == 0) { u = U(gen); v = U(gen); }   // 20 % outliers
        c23[p].insert(c23[p].end(), {u, v, X[0], X[1], X[2]});
      }
    }
    RansacParameters ap;
    ap.error_thresh = 3e-3 * 3e-3; ap.min_iterations = 100; ap.max_iterations = 1000; ap.seed = 11;
    const PnPType types[3] = {PnPType::KNEIP, PnPType::DLS, PnPType::SQPnP};
    const char* names[3] = {"KNEIP", "DLS", "SQPnP"};
    for (int k = 0; k < 3; ++k) {
      std::vector<bool> aok; std::vector<CalibratedAbsolutePose> ap_out; std::vector<RansacSummary> asum; std::string aerr;
      if (!EstimateCalibratedAbsolutePoseBatch(ap, THEIA_RANSAC_RANSAC, types[k], c23, &aok, &ap_out, &asum, &aerr)) { std::printf("FAIL: %s\n", aerr.c_str()); return 1; }
      for (int p = 0; p < 5; ++p) {
        double perr = 0.0;
        for (int i = 0; i < 3; ++i) perr = std::fmax(perr, std::fabs(ap_out[p].position[i] - ctrue[p][i]));
        std::printf("absolute pose %s %d: success %d, %zu inliers of 300, position error %.2e\n", names[k], p, (int)aok[p], asum[p].inliers.size(), perr);
        // (SQPnP answers a minimal sample with ONE SQP iteration, sqpnp.cc:33-34: approximate models, fewer inliers)
        const bool sq = types[k] == PnPType::SQPnP;
        if (!aok[p] || asum[p].inliers.size() < (sq ? 120u : 200u) || !(perr < (sq ? 0.15 : 0.05))) { std::printf("FAIL: absolute pose %s\n", names[k]); return 1; }
      }
    }
    std::printf("ok calibrated absolute pose batches\n");
  }
  // ---- EstimateRigidTransformation2D3D (UPnP), the central overload: [u v X Y Z] as 26-double rows of identity pinhole cameras.
  // No outliers: the reference's estimator accumulates its cost parameters over the samples (include/theia_hip.h), so it is not robust
  {
    std::vector<std::vector<double>> c26(4);
    std::vector<std::vector<double>> ttrue(4);
    for (int p = 0; p < 4; ++p) {
      const double w[3] = {0.2 * U(gen), 0.2 * U(gen), 0.2 * U(gen)};
      const double c[3] = {U(gen), U(gen), -6.0 + U(gen)};
      const double mc[3] = {-c[0], -c[1], -c[2]};
      double t[3];
      rotate(w, mc, t);                        // R X + t with t = -R c
      ttrue[p] = {t[0], t[1], t[2]};
      for (int i = 0; i < 200; ++i) {
        const double X[3] = {2 * U(gen), 2 * U(gen), 2 * U(gen)};
        const double d[3] = {X[0] - c[0], X[1] - c[1], X[2] - c[2]};
        double q[3];
        rotate(w, d, q);
        const double u = q[0] / q[2] + 2e-4 * N(gen), v = q[1] / q[2] + 2e-4 * N(gen);
        const double n = std::sqrt(u * u + v * v + 1.0);
        double row[26] = {u / n, v / n, 1.0 / n, X[0], X[1], X[2], 1.0, u, v, 0, 0, 0, 0, 0, 0, (double)THEIA_CAM_PINHOLE, 1.0, 1.0, 0, 0, 0, 0, 0, 0, 0, 0};
        c26[p].insert(c26[p].end(), row, row + 26);
      }
    }
    RansacParameters up;
    up.error_thresh = 2e-3 * 2e-3; up.min_iterations = 100; up.max_iterations = 500; up.seed = 21;
    EstimatorBatchResult ur; std::string uerr;
    if (!EstimateRigidTransformation2D3DBatch(up, THEIA_RANSAC_RANSAC, c26, &ur, &uerr)) { std::printf("FAIL: %s\n", uerr.c_str()); return 1; }
    for (int p = 0; p < 4; ++p) {
      double terr = 0.0;
      for (int i = 0; i < 3; ++i) terr = std::fmax(terr, std::fabs(ur.models[p][9 + i] - ttrue[p][i]));
      std::printf("rigid transformation (UPnP) %d: success %d, %zu inliers of 200, translation error %.2e\n", p, (int)ur.success[p], ur.summaries[p].inliers.size(), terr);
      if (!ur.success[p] || ur.summaries[p].inliers.size() < 150 || !(terr < 0.05)) { std::printf("FAIL: rigid transformation\n"); return 1; }
    }
    std::printf("ok rigid transformation batch\n");
  }
  // ---- EstimateRadialDistUncalibratedAbsolutePose (P4Pfr): pixels of a camera with focal length 900 and division-model distortion -2e-7
  {
    const double f = 900.0, kd = -2e-7;
    std::vector<std::vector<double>> c5(4);
    for (int p = 0; p < 4; ++p) {
      const double w[3] = {0.2 * U(gen), 0.2 * U(gen), 0.2 * U(gen)};
      const double t[3] = {0.5 * U(gen), 0.5 * U(gen), 0.2};
      for (int i = 0; i < 200; ++i) {
        const double X[3] = {3 * U(gen), 3 * U(gen), 7.0 + 2 * U(gen)};
        double q[3];
        rotate(w, X, q);
        for (int k = 0; k < 3; ++k) q[k] += t[k];
        double u = f * q[0] / q[2], v = f * q[1] / q[2];
        if (i % 5 == 4) { u = 400.0 * U(gen); v = 400.0 * U(gen); }            // 20 % outliers
        else {                                                                   // DistortPoint (:56-74) + 0.3 px of noise
          const double r2 = u * u + v * v, den = 2.0 * kd * r2, inner = 1.0 - 4.0 * kd * r2;
          if (!(std::fabs(den) < 1e-15 || inner < 0.0)) { const double sc = (1.0 - std::sqrt(inner)) / den; u *= sc; v *= sc; }
          u += 0.3 * N(gen); v += 0.3 * N(gen);
        }
        const double row[5] = {u, v, X[0], X[1], X[2]};
        c5[p].insert(c5[p].end(), row, row + 5);
      }
    }
    RansacParameters rp;
    rp.error_thresh = 1.5 * 1.5; rp.min_iterations = 200; rp.max_iterations = 2000; rp.seed = 11;
    const double meta[5] = {2000.0, 100.0, -1e-5, -1e-9, 0.0};
    EstimatorBatchResult rr; std::string rerr;
    if (!EstimateRadialDistUncalibratedAbsolutePoseBatch(rp, THEIA_RANSAC_RANSAC, meta, c5, &rr, &rerr)) { std::printf("FAIL: %s\n", rerr.c_str()); return 1; }
    for (int p = 0; p < 4; ++p) {
      std::printf("radial-distortion absolute pose (P4Pfr) %d: success %d, %zu inliers of 200, focal length %.1f, distortion %.2e\n", p, (int)rr.success[p],
                  rr.summaries[p].inliers.size(), rr.models[p][12], rr.models[p][13]);
      if (!rr.success[p] || rr.summaries[p].inliers.size() < 140 || !(std::fabs(rr.models[p][12] - f) < 0.05 * f)) { std::printf("FAIL: radial-distortion absolute pose\n"); return 1; }
    }
    std::printf("ok radial-distortion absolute pose batch\n");
  }
  // ---- the generic front end under the reference's other names: fundamental matrix on the relative-pose pairs (pixels)
  {
    std::vector<std::vector<double>> px = corr;
    for (auto& c : px) for (double& x : c) x = 800.0 * x;   // pixels, principal point removed
    RansacParameters fp;
    fp.error_thresh = 2.0 * 2.0; fp.min_iterations = 200; fp.max_iterations = 2000; fp.seed = 3;
    EstimatorBatchResult fr; std::string ferr;
    if (!EstimateFundamentalMatrixBatch(fp, THEIA_RANSAC_RANSAC, px, &fr, &ferr)) { std::printf("FAIL: %s\n", ferr.c_str()); return 1; }
    for (int p = 0; p < 6; ++p) {
      std::printf("fundamental matrix %d: success %d, %zu inliers of 400\n", p, (int)fr.success[p], fr.summaries[p].inliers.size());
      if (!fr.success[p] || fr.summaries[p].inliers.size() < 250) { std::printf("FAIL: fundamental matrix\n"); return 1; }
    }
    std::printf("ok fundamental matrix batch\n");
  }
  // ---- BundleAdjustView for every camera as one launch: perturbed cameras against the (now adjusted) points
  {
    std::vector<double> cam2 = cam;
    std::vector<ViewProblem> vp(nv);
    for (int v = 0; v < nv; ++v) {
      vp[v].extrinsics = &cam2[6 * v]; vp[v].intrinsics = intr; vp[v].num_intrinsics = 7; vp[v].camera_model = THEIA_CAM_PINHOLE;
      for (int t = 0; t < nt; ++t) {
        const double* e = &cam[6 * v];
        const double d[3] = {pts[4 * t] / pts[4 * t + 3] - e[0], pts[4 * t + 1] / pts[4 * t + 3] - e[1], pts[4 * t + 2] / pts[4 * t + 3] - e[2]};
        double q[3];
        rotate(e + 3, d, q);
        if (q[2] <= 0.1) continue;
        vp[v].features.insert(vp[v].features.end(), {intr[0] * q[0] / q[2] + intr[3], intr[0] * intr[1] * q[1] / q[2] + intr[4]});
        vp[v].points.insert(vp[v].points.end(), {pts[4 * t], pts[4 * t + 1], pts[4 * t + 2], pts[4 * t + 3]});
      }
      for (int q = 0; q < 6; ++q) cam2[6 * v + q] += (q < 3 ? 0.03 : 0.005) * N(gen);
    }
    BundleAdjustmentOptions vo; vo.max_num_iterations = 30;
    std::vector<BundleAdjustmentSummary> vs; std::string verr;
    if (!BundleAdjustViews(vo, &vp, &vs, &verr)) { std::printf("FAIL: %s\n", verr.c_str()); return 1; }
    double worst = 0.0;
    for (int v = 0; v < nv; ++v) for (int q = 0; q < 6; ++q) worst = std::fmax(worst, std::fabs(cam2[6 * v + q] - cam[6 * v + q]));
    std::printf("BundleAdjustViews: %d views, largest distance from the generating camera %.2e\n", nv, worst);
    if (!(worst < 1e-6)) { std::printf("FAIL: BundleAdjustViews\n"); return 1; }
    std::printf("ok view batch\n");
  }
  // ---- GetCovarianceForTracks: every camera constant (BundleAdjustTracksWithCov, bundle_adjustment.cc:420-460)
  {
    BundleAdjustmentOptions co; co.max_num_iterations = 5;
    BundleAdjuster cb(co);
    cb.AddIntrinsicsGroup(0, THEIA_CAM_PINHOLE, intr, 7);
    for (int v = 0; v < nv; ++v) { cb.AddCamera(v, &cam[6 * v], 0); }
    for (int t = 0; t < 50; ++t) cb.AddTrack(t, &pts[4 * t]);
    const double cov1[2] = {1.0, 1.0};
    for (int t = 0; t < 50; ++t)
      for (int v = 0; v < nv; ++v) {
        const double* e = &cam[6 * v];
        const double d[3] = {pts[4 * t] / pts[4 * t + 3] - e[0], pts[4 * t + 1] / pts[4 * t + 3] - e[1], pts[4 * t + 2] / pts[4 * t + 3] - e[2]};
        double q[3];
        rotate(e + 3, d, q);
        if (q[2] <= 0.1) continue;
        const double uv[2] = {intr[0] * q[0] / q[2] + intr[3], intr[0] * intr[1] * q[1] / q[2] + intr[4]};
        cb.AddObservation(v, t, uv, cov1);
      }
    for (int v = 0; v < nv; ++v) cb.SetCameraExtrinsicsConstant(v);
    std::vector<std::vector<double>> covs;
    if (!cb.GetCovarianceForTracks({0, 7, 49}, &covs)) { std::printf("FAIL: %s\n", cb.error().c_str()); return 1; }
    for (const auto& c : covs) {
      const bool sym = std::fabs(c[1] - c[3]) < 1e-12 * std::fabs(c[0]) + 1e-30 && std::fabs(c[2] - c[6]) < 1e-12 * std::fabs(c[0]) + 1e-30;
      std::printf("track covariance diag %.3e %.3e %.3e\n", c[0], c[4], c[8]);
      if (c.size() != 9 || !(c[0] > 0 && c[4] > 0 && c[8] > 0) || !sym) { std::printf("FAIL: covariance\n"); return 1; }
    }
    std::printf("ok track covariances\n");
  }
  return 0;
}
