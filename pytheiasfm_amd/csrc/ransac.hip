// ransac.hip -- many-hypothesis RANSAC on gfx950: K5 (minimal-solver fit) and
// K6 (score every correspondence under every model) plus the host driver that
// replaces SampleConsensusEstimator<E>::Estimate
//   (src/theia/solvers/sample_consensus_estimator.h:300-415)
// for Ransac<E> + RandomSampler (ransac.h:57-61, random_sampler.cc:53-72).
//
// Split of work (DESIGN.md "RANSAC"):
//  host   : the std::mt19937 sample stream (bit-exact restatement, util/random.cc)
//           for a ROUND of iterations up front; sequential replay of the
//           accept / adaptive-termination rules in sample order.
//  device : k_fit   -- one thread per hypothesis: minimal solver -> <= 10 models
//           k_score -- one thread per (hypothesis, model): walks all N
//                      correspondences (staged in LDS, broadcast reads) in data
//                      order, so the MLE sum keeps the reference's
//                      left-to-right FP64 order.
//  Hypotheses of a round are independent given the pre-generated samples, so a
//  batch of problems x iterations fills the chip; results are identical to the
//  sequential loop because acceptance is replayed in order.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

#include "ba_device.h"      // the camera models' projection (TriangulationEstimator::Error)
#include "ransac_device.h"
#include "dls_device.h"
#include "eig_team.h"
#include "fit5_team.h"
#include "svd_team.h"
#include "p4pf_device.h"
#include "theia_hip.h"
#include <atomic>
#include <mutex>
#include <functional>
#include <thread>

#include "theia_hip_internal.h"
#include "host_team.h"
#include "pools.h"

namespace thip {
namespace {

constexpr int kMaxCap = 18;   // largest EstimateModel output of the thread-per-hypothesis solvers (SQPnP: 18 solutions)
// models per sample an estimator can return = slot stride of the per-hypothesis arrays
__host__ __device__ inline int max_models(int est) {
  if (est == THEIA_EST_RADIAL_HOMOGRAPHY) return 2;
  if (est == THEIA_EST_SIMILARITY_2D3D) return dlsdev::kMaxSolutions;
  if (est == THEIA_EST_UNCALIBRATED_ABSOLUTE_POSE) return 10;
  if (est == THEIA_EST_RIGID_TRANSFORMATION_2D3D) return 8;
  if (est == THEIA_EST_RADIAL_DIST_UNCALIBRATED_ABSOLUTE_POSE) return 13;
  if (est >= THEIA_EST_FUNDAMENTAL_MATRIX) return 1;
  if (est == THEIA_EST_ABSOLUTE_POSE_DLS) return dlsdev::kMaxSolutions;
  return est == THEIA_EST_ABSOLUTE_POSE_SQPNP ? 18 : (est == THEIA_EST_ABSOLUTE_POSE_KNEIP ? 4 : 10);
}
constexpr int kStride = THEIA_RANSAC_MODEL_STRIDE;

__host__ __device__ inline int sample_size(int est) {
  switch (est) {
    case THEIA_EST_RELATIVE_POSE: case THEIA_EST_ESSENTIAL_MATRIX: return 5;
    case THEIA_EST_FUNDAMENTAL_MATRIX: case THEIA_EST_UNCALIBRATED_RELATIVE_POSE: return 8;
    case THEIA_EST_HOMOGRAPHY: case THEIA_EST_SIMILARITY_2D3D: case THEIA_EST_UNCALIBRATED_ABSOLUTE_POSE:
    case THEIA_EST_RIGID_TRANSFORMATION_2D3D: case THEIA_EST_RADIAL_DIST_UNCALIBRATED_ABSOLUTE_POSE: return 4;
    case THEIA_EST_RADIAL_HOMOGRAPHY: return 6;
    case THEIA_EST_RELATIVE_POSE_KNOWN_ORIENTATION: case THEIA_EST_ABSOLUTE_POSE_KNOWN_ORIENTATION:
    case THEIA_EST_TRIANGULATION: return 2;
    default: return 3;
  }
}
// meaningful doubles of a model row (layouts in theia_hip.h)
inline int model_doubles(int est) {
  switch (est) {
    case THEIA_EST_RELATIVE_POSE: return 21;
    case THEIA_EST_UNCALIBRATED_RELATIVE_POSE: return 23;
    case THEIA_EST_ABSOLUTE_POSE_KNEIP: case THEIA_EST_ABSOLUTE_POSE_DLS: case THEIA_EST_ABSOLUTE_POSE_SQPNP: return 12;
    case THEIA_EST_UNCALIBRATED_ABSOLUTE_POSE: return 12;   // projection matrix, row-major 3 x 4
    case THEIA_EST_RIGID_TRANSFORMATION_2D3D: return 12;    // rotation | translation
    case THEIA_EST_RADIAL_DIST_UNCALIBRATED_ABSOLUTE_POSE: return 14;   // rotation | translation | focal length | radial distortion
    case THEIA_EST_DOMINANT_PLANE: return 6;
    case THEIA_EST_RELATIVE_POSE_KNOWN_ORIENTATION: case THEIA_EST_ABSOLUTE_POSE_KNOWN_ORIENTATION: return 3;
    case THEIA_EST_TRIANGULATION: return 4;
    case THEIA_EST_RADIAL_HOMOGRAPHY: return 20;   // H | l1 | l2 | H^-1
    case THEIA_EST_SIMILARITY_2D3D: return 13;     // rotation | translation | scale
    default: return 9;   // essential / fundamental matrix, homography
  }
}
constexpr int kTriDatum = 33;   // PointObservation row of THEIA_EST_TRIANGULATION (theia_hip.h)
constexpr int kSimDatum = 26;   // CameraAndFeatureCorrespondence2D3D row of THEIA_EST_SIMILARITY_2D3D: dir (3) | point (4) | pixel (2) | extrinsics (6) | model | intrinsics (10)
__host__ __device__ inline int datum_size(int est) {
  switch (est) {
    case THEIA_EST_ABSOLUTE_POSE_KNEIP: case THEIA_EST_ABSOLUTE_POSE_DLS: case THEIA_EST_ABSOLUTE_POSE_SQPNP:
    case THEIA_EST_ABSOLUTE_POSE_KNOWN_ORIENTATION: case THEIA_EST_UNCALIBRATED_ABSOLUTE_POSE:
    case THEIA_EST_RADIAL_DIST_UNCALIBRATED_ABSOLUTE_POSE: return 5;
    case THEIA_EST_DOMINANT_PLANE: return 3;
    case THEIA_EST_TRIANGULATION: return kTriDatum;
    case THEIA_EST_RADIAL_HOMOGRAPHY: return rsc::kRadHomDatum;
    case THEIA_EST_SIMILARITY_2D3D: case THEIA_EST_RIGID_TRANSFORMATION_2D3D: return kSimDatum;
    default: return 4;
  }
}
constexpr int kMaxSample = 8;            // largest minimal sample (8-point fundamental matrix)
constexpr int kMaxSampleDoubles = 6 * rsc::kRadHomDatum;   // 8 correspondences x 4, two observations with their cameras (66), six radial-distortion correspondences (72)
static_assert(kMaxSampleDoubles >= 2 * kTriDatum, "sample buffer");
// the sample buffer of a k_fit instance: 32 doubles for the correspondence estimators (their kernels keep the frame they had)
template <int EST> constexpr int sample_doubles() { return (EST == THEIA_EST_TRIANGULATION || EST == THEIA_EST_RADIAL_HOMOGRAPHY || EST < 0) ? kMaxSampleDoubles : 32; }

// EstimateModel of the three estimators (estimate_relative_pose.cc:75-109,
// estimate_essential_matrix.cc:62-73, estimate_calibrated_absolute_pose.cc:76-118).
// EST >= 0: compile-time estimator (k_fit: each instantiation carries only its own solver's scratch frame);
// EST < 0: runtime dispatch on `est`.
// estimator constants that are not data (UncalibratedRelativePoseEstimator's min / max focal length)
struct EstParams { double min_focal, max_focal; };

template <int EST = -1>
__device__ int estimate_models(int est, const double* subset, double* models, EstParams ep = EstParams{0.0, 0.0}) {
  if (EST >= 0) est = EST;
  constexpr bool any = EST < 0;
  if ((any || EST == THEIA_EST_RELATIVE_POSE || EST == THEIA_EST_ESSENTIAL_MATRIX) &&
      (est == THEIA_EST_RELATIVE_POSE || est == THEIA_EST_ESSENTIAL_MATRIX)) {
    double E[90];
    const int ne = rsc::five_point(subset, E);
    if (ne == 0) return 0;
    int nm = 0;
    for (int i = 0; i < ne; ++i) {
      double* m = models + kStride * nm;
      for (int k = 0; k < 9; ++k) m[k] = E[9 * i + k];
      if (est == THEIA_EST_ESSENTIAL_MATRIX) { for (int k = 9; k < kStride; ++k) m[k] = 0.0; nm++; continue; }
      const int nfront = rsc::best_pose_from_E(E + 9 * i, subset, 5, m + 9, m + 18);
      if (nfront >= 4) nm++;
    }
    return nm;
  }
  if ((any || EST == THEIA_EST_ABSOLUTE_POSE_KNEIP) && est == THEIA_EST_ABSOLUTE_POSE_KNEIP) {
    double Rs[36], ts[12];
    const int n = rsc::p3p(subset, Rs, ts);
    for (int i = 0; i < n; ++i) {
      double* m = models + kStride * i;
      const double* R = Rs + 9 * i;
      const double* t = ts + 3 * i;
      for (int k = 0; k < 9; ++k) m[k] = R[k];
      for (int c = 0; c < 3; ++c) m[9 + c] = -((R[c] * t[0] + R[3 + c] * t[1]) + R[6 + c] * t[2]);
      for (int k = 12; k < kStride; ++k) m[k] = 0.0;
    }
    return n;
  }
  if ((any || EST == THEIA_EST_ABSOLUTE_POSE_SQPNP) && est == THEIA_EST_ABSOLUTE_POSE_SQPNP) {
    // SQPnP on the 3 sampled correspondences; quaternion -> matrix as the estimator does
    // (estimate_calibrated_absolute_pose.cc:99-106)
    double feat[6], world[9], quats[72], ts[54];
    for (int i = 0; i < 3; ++i) {
      feat[2 * i] = subset[5 * i]; feat[2 * i + 1] = subset[5 * i + 1];
      for (int k = 0; k < 3; ++k) world[3 * i + k] = subset[5 * i + 2 + k];
    }
    const int n = rsc::sqpnp(3, feat, world, quats, ts);
    for (int i = 0; i < n; ++i) {
      double* m = models + kStride * i;
      double R[9];
      rsc::quat_to_rot(quats + 4 * i, R);
      const double* t = ts + 3 * i;
      for (int k = 0; k < 9; ++k) m[k] = R[k];
      for (int c = 0; c < 3; ++c) m[9 + c] = -((R[c] * t[0] + R[3 + c] * t[1]) + R[6 + c] * t[2]);
      for (int k = 12; k < kStride; ++k) m[k] = 0.0;
    }
    return n;
  }
  if ((any || EST == THEIA_EST_RADIAL_HOMOGRAPHY) && est == THEIA_EST_RADIAL_HOMOGRAPHY)   // estimate_radial_distortion_homography.cc:62-77
    return rsc::radial_homography_six_point(subset, models, kStride);
  // single-model estimators (estimate_fundamental_matrix.cc:64-78, estimate_homography.cc:72-88,
  // estimate_dominant_plane_from_points.cc:62-81, estimate_relative_pose_with_known_orientation.cc:31-45)
  if ((any || EST >= THEIA_EST_FUNDAMENTAL_MATRIX) && est >= THEIA_EST_FUNDAMENTAL_MATRIX) {
    bool ok = false;
    for (int k = 0; k < kStride; ++k) models[k] = 0.0;
    if (est == THEIA_EST_UNCALIBRATED_RELATIVE_POSE) {
      const double mm[2] = {ep.min_focal, ep.max_focal};
      ok = rsc::uncalibrated_relative_pose(subset, mm, models);
    }
    else if (est == THEIA_EST_ABSOLUTE_POSE_KNOWN_ORIENTATION) ok = rsc::position_from_two_rays(subset, models);
    else if (est == THEIA_EST_TRIANGULATION) ok = rsc::triangulate_two_views(subset, models);
    else if (est == THEIA_EST_FUNDAMENTAL_MATRIX) ok = rsc::eight_point_fundamental(subset, models);
    else if (est == THEIA_EST_HOMOGRAPHY) ok = rsc::four_point_homography(subset, models);
    else if (est == THEIA_EST_DOMINANT_PLANE) ok = rsc::plane_from_three_points(subset, models);
    else if (est == THEIA_EST_RELATIVE_POSE_KNOWN_ORIENTATION) ok = rsc::two_point_relative_position(subset, models);
    return ok ? 1 : 0;
  }
  return 0;
}

// TriangulationEstimator::Error (estimate_triangulation.cc:92-101): Camera::ProjectPoint (camera.cc:206-216) of the
// homogeneous point -- adjusted = X.head<3>() - X[3] c, rotated by the angle-axis, pixel through the camera model (the
// BA kernels' projection, ba_device.h), depth = rotated.z / X[3] -- then the squared pixel error, DBL_MAX behind the camera.
__device__ inline double triangulation_error(const double* X, const double* d) {
  const double* ext = d + 16;
  const double p[3] = {X[0] - X[3] * ext[0], X[1] - X[3] * ext[1], X[2] - X[3] * ext[2]};
  RotTerms rt;
  rotation_terms(ext + 3, rt);
  const double q[3] = {(rt.R[0] * p[0] + rt.R[1] * p[1]) + rt.R[2] * p[2], (rt.R[3] * p[0] + rt.R[4] * p[1]) + rt.R[5] * p[2],
                       (rt.R[6] * p[0] + rt.R[7] * p[1]) + rt.R[8] * p[2]};
  const double depth = q[2] / X[3];
  if (depth <= 0.0) return DBL_MAX;
  double uv[2], Jq[6];
  project<false>((int)d[22], d + 23, q, uv, Jq);
  const double ex = d[14] - uv[0], ey = d[15] - uv[1];
  return ex * ex + ey * ey;
}

// Estimator::Error (estimate_relative_pose.cc:142-151, estimate_essential_matrix.cc:77-83,
// estimate_calibrated_absolute_pose.cc:158-167)
// GdlsSimilarityTransformationEstimator::Error (estimate_similarity_transformation_2d_3d.cc:137-155) with TransformCamera
// (:52-70): new position = s R c + t, new orientation = R_cam R^T, then Camera::ProjectPoint; DBL_MAX at negative depth.
__device__ inline double similarity_error(const double* m, const double* d) {
  const double* X = d + 3; const double* c = d + 9;
  double np[3], a[3], b[3];
  for (int r = 0; r < 3; ++r) np[r] = m[12] * ((m[3 * r] * c[0] + m[3 * r + 1] * c[1]) + m[3 * r + 2] * c[2]) + m[9 + r];
  for (int r = 0; r < 3; ++r) a[r] = X[r] - X[3] * np[r];
  for (int r = 0; r < 3; ++r) b[r] = (m[r] * a[0] + m[3 + r] * a[1]) + m[6 + r] * a[2];   // R^T a
  RotTerms rt;
  rotation_terms(d + 12, rt);
  const double q[3] = {(rt.R[0] * b[0] + rt.R[1] * b[1]) + rt.R[2] * b[2], (rt.R[3] * b[0] + rt.R[4] * b[1]) + rt.R[5] * b[2],
                       (rt.R[6] * b[0] + rt.R[7] * b[1]) + rt.R[8] * b[2]};
  if (q[2] / X[3] < 0.0) return DBL_MAX;
  double uv[2], Jq[6];
  project<false>((int)d[15], d + 16, q, uv, Jq);
  const double ex = d[7] - uv[0], ey = d[8] - uv[1];
  return ex * ex + ey * ey;
}

// NonCentralCameraPoseEstimator::Error (estimate_rigid_transformation_2d_3d.cc:116-129): R X + t through the datum's camera
// (Camera::ProjectPoint), DBL_MAX at negative depth.
__device__ inline double rigid_error(const double* m, const double* d) {
  const double* X = d + 3; const double* c = d + 9;
  const double h[3] = {X[0] / X[3], X[1] / X[3], X[2] / X[3]};
  double a[3];
  for (int r = 0; r < 3; ++r) a[r] = (((m[3 * r] * h[0] + m[3 * r + 1] * h[1]) + m[3 * r + 2] * h[2]) + m[9 + r]) - 1.0 * c[r];
  RotTerms rt;
  rotation_terms(d + 12, rt);
  const double q[3] = {(rt.R[0] * a[0] + rt.R[1] * a[1]) + rt.R[2] * a[2], (rt.R[3] * a[0] + rt.R[4] * a[1]) + rt.R[5] * a[2],
                       (rt.R[6] * a[0] + rt.R[7] * a[1]) + rt.R[8] * a[2]};
  if (q[2] / 1.0 < 0.0) return DBL_MAX;
  double uv[2], Jq[6];
  project<false>((int)d[15], d + 16, q, uv, Jq);
  const double ex = d[7] - uv[0], ey = d[8] - uv[1];
  return ex * ex + ey * ey;
}

// RadialDistUncalibratedAbsolutePoseEstimator::Error (estimate_radial_dist_uncalibrated_absolute_pose.cc:130-147) with DistortPoint
// (:56-74): the projection with the model's focal length, distorted by the division model, against the feature; 1e10 when the
// translation's z is negative.  m: rotation (9) | translation (3) | focal length | distortion; d: [u v X Y Z]
__device__ inline double radial_dist_error(const double* m, const double* d) {
  if (m[11] < 0.0) return 1.0e10;
  double p[3];
  for (int r = 0; r < 3; ++r) p[r] = ((m[3 * r] * d[2] + m[3 * r + 1] * d[3]) + m[3 * r + 2] * d[4]) + m[9 + r];
  const double kp[3] = {m[12] * p[0], m[12] * p[1], 1.0 * p[2]};
  const double x = kp[0] / kp[2], y = kp[1] / kp[2];
  const double r2 = x * x + y * y;
  const double denom = 2.0 * m[13] * r2, inner = 1.0 - 4.0 * m[13] * r2;
  double dx = x, dy = y;
  if (!(fabs(denom) < 1e-15 || inner < 0.0)) {
    const double sc = (1.0 - sqrt(inner)) / denom;
    dx = x * sc; dy = y * sc;
  }
  const double ex = dx - d[0], ey = dy - d[1];
  return ex * ex + ey * ey;
}

__device__ inline double model_error(int est, const double* m, const double* d) {
  if (est == THEIA_EST_RADIAL_DIST_UNCALIBRATED_ABSOLUTE_POSE) return radial_dist_error(m, d);
  if (est == THEIA_EST_UNCALIBRATED_ABSOLUTE_POSE) return p4pfdev::reprojection_error(m, d);
  if (est == THEIA_EST_SIMILARITY_2D3D) return similarity_error(m, d);
  if (est == THEIA_EST_RIGID_TRANSFORMATION_2D3D) return rigid_error(m, d);
  if (est == THEIA_EST_TRIANGULATION) return triangulation_error(m, d);
  if (est == THEIA_EST_RADIAL_HOMOGRAPHY) return rsc::radial_homography_error(m, d);
  if (est == THEIA_EST_RELATIVE_POSE) {
    if (rsc::in_front(d, m + 9, m + 18)) return rsc::sampson(m, d);
    return DBL_MAX;
  }
  if (est == THEIA_EST_ESSENTIAL_MATRIX || est == THEIA_EST_FUNDAMENTAL_MATRIX) return rsc::sampson(m, d);
  if (est == THEIA_EST_HOMOGRAPHY) return rsc::homography_error(m, d);
  if (est == THEIA_EST_DOMINANT_PLANE) return rsc::plane_error(m, d);
  if (est == THEIA_EST_RELATIVE_POSE_KNOWN_ORIENTATION) return rsc::known_orientation_error(m, d);
  if (est == THEIA_EST_UNCALIBRATED_RELATIVE_POSE) return rsc::uncalibrated_relative_pose_error(m, d);
  if (est == THEIA_EST_ABSOLUTE_POSE_KNOWN_ORIENTATION) return rsc::known_orientation_abs_error(m, d);
  const double dx = d[2] - m[9], dy = d[3] - m[10], dz = d[4] - m[11];
  const double px = (m[0] * dx + m[1] * dy) + m[2] * dz;
  const double py = (m[3] * dx + m[4] * dy) + m[5] * dz;
  const double pz = (m[6] * dx + m[7] * dy) + m[8] * dz;
  const double ex = px / pz - d[0], ey = py / pz - d[1];
  return ex * ex + ey * ey;
}

// K5: one thread per (problem, iteration of the round).
//   samples : [nprob][B][m] indices into the problem's data
//   models  : [nprob][B][mm][kStride], mm = max_models(est)
//   counts  : [nprob][B]
template <int EST>
__global__ __launch_bounds__(64) void k_fit(int est, int nprob, int B, const int64_t* __restrict__ offsets,
                                            const double* __restrict__ data, const int* __restrict__ samples,
                                            const int* __restrict__ active_iters, double* __restrict__ models,
                                            int* __restrict__ counts, int* __restrict__ dense_count,
                                            int* __restrict__ tags, int* __restrict__ hyp_base, EstParams ep) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const int p = blockIdx.y;
  if (b >= B || p >= nprob) return;
  const size_t hyp = (size_t)p * B + b;
  const int mm = max_models(est);
  if (b >= active_iters[p]) { counts[hyp] = 0; return; }
  const int m = sample_size(est), ds = datum_size(est);
  const double* pd = data + (size_t)offsets[p] * ds;
  double subset[sample_doubles<EST>()];
  for (int i = 0; i < m; ++i) {
    const int idx = samples[hyp * m + i];
    for (int k = 0; k < ds; ++k) subset[i * ds + k] = pd[(size_t)idx * ds + k];
  }
  double mloc[kMaxCap * kStride];
  const int nm = estimate_models<EST>(est, subset, mloc, ep);
  counts[hyp] = nm;
  if (nm == 0) return;
  // append to the problem's DENSE model list (most of the 10 slots per hypothesis
  // are empty: scoring a dense list keeps every lane of k_score busy); the tag
  // remembers (iteration, slot) so the host replays in sample order.
  const int base = atomicAdd(&dense_count[p], nm);
  hyp_base[hyp] = base;   // where this hypothesis' models start in the dense list (LO events fetch their model from there)
  double* mo = models + ((size_t)p * B * mm + base) * (size_t)kStride;
  int* tg = tags + (size_t)p * B * mm + base;
  for (int j = 0; j < nm; ++j) {
    for (int k = 0; k < kStride; ++k) mo[j * kStride + k] = mloc[j * kStride + k];
    tg[j] = b * mm + j;
  }
}

// K6: one thread per (problem, iteration, model slot); block = 256 slots of ONE
// problem, whose correspondences are staged in LDS when they fit.
// EST >= 0: the estimator is a compile-time constant (the error functions of the others drop out: the run-time dispatch
// keeps all of them -- the eight camera models of the triangulation error included -- in one 192-VGPR kernel at two waves
// per SIMD); EST < 0: run-time dispatch on `est`.
// The specialised instances run 512 slots per workgroup: the LDS copy of a problem's correspondences (64 - 80 KB at
// 2000 per pair) allows two workgroups per CU, i.e. four waves per SIMD instead of two.
constexpr int score_threads(int est) { return est >= 0 ? 512 : 256; }
template <bool USE_LDS, int EST = -1>
__global__ __launch_bounds__(score_threads(EST)) void k_score(int est_rt, int nprob, int B, const int64_t* __restrict__ offsets,
                                               const double* __restrict__ data, const double* __restrict__ models,
                                               const int* __restrict__ dense_count, const int* __restrict__ tags,
                                               double thresh, int use_mle, double* __restrict__ cost,
                                               int* __restrict__ ninl) {
  extern __shared__ __attribute__((aligned(16))) double sdata[];
  const int est = EST >= 0 ? EST : est_rt;
  const int p = blockIdx.y;
  const int nmodels = dense_count[p];
  if ((int)(blockIdx.x * blockDim.x) >= nmodels) return;  // whole block beyond the dense list
  const int ds = datum_size(est);
  const int64_t n64 = offsets[p + 1] - offsets[p];
  const int n = (int)n64;
  const double* pd = data + (size_t)offsets[p] * ds;
  if (USE_LDS) {
    for (int i = threadIdx.x; i < n * ds; i += blockDim.x) sdata[i] = pd[i];
    __syncthreads();
    pd = sdata;
  }
  const int slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= nmodels) return;
  const int mm = max_models(est);
  const size_t dense = (size_t)p * B * mm + slot;
  const size_t out = dense;   // scores stay in the dense order of the models (hyp_base[hyp] + slot): the host downloads them packed
  double m[kStride];
  const double* mo = models + dense * (size_t)kStride;
#pragma unroll
  for (int k = 0; k < kStride; ++k) m[k] = mo[k];
  int cnt = 0;
  double mle = 0.0;
  if (EST == THEIA_EST_RELATIVE_POSE || EST == THEIA_EST_ESSENTIAL_MATRIX) {
    // Sampson error without its division wherever the comparison is decided by the margin (ransac_device.h quotient_below)
    for (int i = 0; i < n; ++i) {
      const double* d = pd + (size_t)i * ds;
      bool in = false;
      double r = 0.0;
      if (EST == THEIA_EST_ESSENTIAL_MATRIX || rsc::in_front(d, m + 9, m + 18)) {
        double a, den;
        rsc::sampson_parts(m, d, &a, &den);
        in = rsc::quotient_below(a, den, thresh, use_mle != 0, &r);
      }
      if (in) { cnt++; mle += r; }
      else mle += thresh;
    }
  } else if (EST == THEIA_EST_ABSOLUTE_POSE_KNEIP || EST == THEIA_EST_ABSOLUTE_POSE_DLS || EST == THEIA_EST_ABSOLUTE_POSE_SQPNP) {
    // | hnormalized(R (X - c)) - x |^2 < thresh  (estimate_calibrated_absolute_pose.cc:158-167) decided WITHOUT its two divisions
    // wherever the margin allows: 1 / z from v_rcp_f64 + one Newton step (relative error < 2^-48: scripts/ubench/rcp_acc), the
    // error of the squared distance computed with it bounded by  r 2^-20 + (|x / z| + |y / z|)^2 2^-60  (a cancellation in
    // x / z - u costs absolute, not relative accuracy); inside that band -- and for an inlier whose value the MLE score adds --
    // the reference's arithmetic runs.  Same decisions, same sums: the two divisions were ~half of this loop's instructions.
    for (int i = 0; i < n; ++i) {
      const double* d = pd + (size_t)i * ds;
      const double dx = d[2] - m[9], dy = d[3] - m[10], dz = d[4] - m[11];
      const double px = (m[0] * dx + m[1] * dy) + m[2] * dz;
      const double py = (m[3] * dx + m[4] * dy) + m[5] * dz;
      const double pz = (m[6] * dx + m[7] * dy) + m[8] * dz;
      double iz = __builtin_amdgcn_rcp(pz);
      iz = __builtin_fma(iz, __builtin_fma(-pz, iz, 1.0), iz);
      const double qx = px * iz, qy = py * iz;
      const double ax = qx - d[0], ay = qy - d[1];
      const double ra = ax * ax + ay * ay;
      const double qq = fabs(qx) + fabs(qy);
      const double bound = ra * 0x1p-20 + (qq * qq) * 0x1p-60;
      if (ra - bound > thresh) { mle += thresh; continue; }          // surely an outlier (an outlier's value is never used)
      if (!use_mle && ra + bound < thresh) { cnt++; continue; }      // surely an inlier, and nobody asks for the value
      const double ex = px / pz - d[0], ey = py / pz - d[1];
      const double r = ex * ex + ey * ey;
      if (r < thresh) { cnt++; mle += r; }
      else mle += thresh;
    }
  } else {
    for (int i = 0; i < n; ++i) {
      const double r = model_error(est, m, pd + (size_t)i * ds);
      if (r < thresh) { cnt++; mle += r; }
      else mle += thresh;
    }
  }
  cost[out] = use_mle ? mle : (double)(n - cnt);
  ninl[out] = cnt;
}

// ---- LMED (lmed.h:64-70, lmed_quality_measurement.h:58-130): cost = median of the squared residuals, inliers by the
// 2.5 * 1.4826 * (1 + 5 / (n - m)) * sqrt(median) rule.  One workgroup per model: squared residuals in LDS, the order
// statistics by an 8-bit radix select on the IEEE bit patterns (exact: the median is an element, or the mean of two).
// sq == nullptr (more data than the LDS holds: > 19 456 per problem): the squared residuals are re-evaluated in every pass --
// the same bits each time, ten evaluations per datum instead of one; lmed_quality_measurement.h:56-63 has no size limit.
__device__ __forceinline__ double lmed_sq(const double* sq, int est, const double* m, const double* pd, int ds, int i) {
  if (sq) return sq[i];
  const double r = model_error(est, m, pd + (size_t)i * ds);
  return r * r;
}
__device__ double lmed_select(const double* sq, int n, int k, int* hist, int* sel, int est, const double* m, const double* pd, int ds) {
  unsigned long long prefix = 0ull, mask = 0ull;
  for (int shift = 56; shift >= 0; shift -= 8) {
    for (int b = threadIdx.x; b < 256; b += blockDim.x) hist[b] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const unsigned long long key = (unsigned long long)__double_as_longlong(lmed_sq(sq, est, m, pd, ds, i));
      if ((key & mask) == prefix) atomicAdd(&hist[(int)((key >> shift) & 255ull)], 1);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int cum = 0, b = 0;
      for (; b < 255; ++b) { if (cum + hist[b] > k) break; cum += hist[b]; }
      sel[0] = b; sel[1] = k - cum;
    }
    __syncthreads();
    prefix |= (unsigned long long)sel[0] << shift;
    mask |= 255ull << shift;
    k = sel[1];
    __syncthreads();
  }
  return __longlong_as_double((long long)prefix);
}

// returns the median (cost); *ninl = inlier count; mask (optional, global) receives the inlier flags
__device__ double lmed_block(int est, const double* m, const double* pd, int n, int ds, int min_samples, double* sq,
                             int* hist, int* sel, int* ninl, uint8_t* mask, double* sqt_out = nullptr) {
  if (sq) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) { const double r = model_error(est, m, pd + (size_t)i * ds); sq[i] = r * r; }
  }
  __syncthreads();
  double median = lmed_select(sq, n, n / 2, hist, sel, est, m, pd, ds);
  if ((n % 2) != 0) median = 0.5 * (lmed_select(sq, n, n / 2 - 1, hist, sel, est, m, pd, ds) + median);
  const double thr = 2.5 * 1.4826 * (1 + 5.0 / (double)((size_t)n - (size_t)min_samples)) * sqrt(median);
  const double sqt = thr * thr;
  if (sqt_out && threadIdx.x == 0) *sqt_out = sqt;
  if (threadIdx.x == 0) sel[2] = 0;
  __syncthreads();
  int cnt = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const bool in = lmed_sq(sq, est, m, pd, ds, i) < sqt;
    cnt += in ? 1 : 0;
    if (mask) mask[i] = in ? 1 : 0;
  }
  atomicAdd(&sel[2], cnt);
  __syncthreads();
  *ninl = sel[2];
  return median;
}

__global__ __launch_bounds__(256) void k_score_lmed(int est, int nprob, int B, const int64_t* __restrict__ offsets,
                                                    const double* __restrict__ data, const double* __restrict__ models,
                                                    const int* __restrict__ dense_count, const int* __restrict__ tags,
                                                    double* __restrict__ cost, int* __restrict__ ninl, int in_lds) {
  extern __shared__ __attribute__((aligned(16))) double sdata[];
  __shared__ int hist[256], sel[4];
  __shared__ double m[kStride];
  const int p = blockIdx.y, slot = blockIdx.x;
  if (slot >= dense_count[p]) return;
  const int mm = max_models(est), ds = datum_size(est);
  const int n = (int)(offsets[p + 1] - offsets[p]);
  const size_t dense = (size_t)p * B * mm + slot;
  if (threadIdx.x < kStride) m[threadIdx.x] = models[dense * (size_t)kStride + threadIdx.x];
  __syncthreads();
  int cnt;
  const double med = lmed_block(est, m, data + (size_t)offsets[p] * ds, n, ds, sample_size(est), in_lds ? sdata : nullptr, hist, sel, &cnt, nullptr);
  if (threadIdx.x == 0) {
    const size_t out = dense;
    cost[out] = med; ninl[out] = cnt;
  }
}

__global__ __launch_bounds__(256) void k_inlier_mask_lmed(int est, int nprob, const int64_t* __restrict__ offsets,
                                                          const double* __restrict__ data, const double* __restrict__ best_models,
                                                          uint8_t* __restrict__ mask, int in_lds) {
  extern __shared__ __attribute__((aligned(16))) double sdata[];
  __shared__ int hist[256], sel[4];
  __shared__ double m[kStride];
  const int p = blockIdx.x;
  const int ds = datum_size(est);
  const int n = (int)(offsets[p + 1] - offsets[p]);
  if (threadIdx.x < kStride) m[threadIdx.x] = best_models[(size_t)p * kStride + threadIdx.x];
  __syncthreads();
  int cnt;
  lmed_block(est, m, data + (size_t)offsets[p] * ds, n, ds, sample_size(est), in_lds ? sdata : nullptr, hist, sel, &cnt, mask + offsets[p]);
}

// LO under LMED: the inliers RefineModel sees are the quality measurement's own (r^2 < its median-derived bound): the bound of
// every LO event's model, for k_lo_gather
__global__ __launch_bounds__(256) void k_lo_lmed_bound(int est, const int* __restrict__ ev_prob, const int64_t* __restrict__ offsets,
                                                       const double* __restrict__ data, const double* __restrict__ ev_model,
                                                       double* __restrict__ ev_sqt, int in_lds) {
  extern __shared__ __attribute__((aligned(16))) double sdata[];
  __shared__ int hist[256], sel[4];
  __shared__ double m[kStride];
  const int e = blockIdx.x, p = ev_prob[e];
  const int ds = datum_size(est);
  const int n = (int)(offsets[p + 1] - offsets[p]);
  if (threadIdx.x < kStride) m[threadIdx.x] = ev_model[(size_t)e * kStride + threadIdx.x];
  __syncthreads();
  int cnt;
  lmed_block(est, m, data + (size_t)offsets[p] * ds, n, ds, sample_size(est), in_lds ? sdata : nullptr, hist, sel, &cnt, nullptr, ev_sqt + e);
}

// final pass: refit the winning hypothesis (deterministic -> identical model)
// and mark the inliers of every datum (sample_consensus_estimator.h:396-399).
__global__ void k_refit(int est, int nprob, const int64_t* __restrict__ offsets, const double* __restrict__ data,
                        const int* __restrict__ best_samples, const int* __restrict__ best_slot,
                        double* __restrict__ out_models, EstParams ep) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= nprob) return;
  double* mo = out_models + (size_t)p * kStride;
  for (int k = 0; k < kStride; ++k) mo[k] = 0.0;
  if (best_slot[p] < 0) return;
  const int m = sample_size(est), ds = datum_size(est);
  const double* pd = data + (size_t)offsets[p] * ds;
  double subset[kMaxSampleDoubles];
  for (int i = 0; i < m; ++i) {
    const int idx = best_samples[p * kMaxSample + i];
    for (int k = 0; k < ds; ++k) subset[i * ds + k] = pd[(size_t)idx * ds + k];
  }
  double mloc[kMaxCap * kStride];
  const int nm = estimate_models(est, subset, mloc, ep);
  const int j = best_slot[p];
  if (j < nm) for (int k = 0; k < kStride; ++k) mo[k] = mloc[j * kStride + k];
}

// after a round's replay: the problems whose best model changed in this round copy it out of the round's dense model
// list (the final models then need no refit: a single-thread five-point solve is ~2 ms of latency)
__global__ void k_save_best(int est, int n, const int* __restrict__ prob, const int* __restrict__ hyp, const int* __restrict__ slot,
                            int round_B, const double* __restrict__ round_models, const int* __restrict__ hyp_base,
                            double* __restrict__ best_models) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * kStride) return;
  const int e = i / kStride, k = i % kStride;
  const int h = hyp[e], q = h / round_B;
  best_models[(size_t)prob[e] * kStride + k] =
      round_models[((size_t)q * round_B * max_models(est) + hyp_base[h] + slot[e]) * (size_t)kStride + k];
}

// scores of a round, packed over the problems of the chunk for the download: problem q's dense_count[q] entries (stride
// B * mm on the device) go to [prefix[q], prefix[q + 1]).  The host reads model j of hypothesis h at prefix[q] + hyp_base[h] + j.
__global__ void k_pack_scores(int B, int mm, const int* __restrict__ dense_count, const int* __restrict__ prefix,
                              const double* __restrict__ cost, const int* __restrict__ ninl, double* __restrict__ pcost,
                              int* __restrict__ pninl) {
  const int q = blockIdx.y;
  const int n = dense_count[q];
  const size_t src = (size_t)q * B * mm;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    pcost[prefix[q] + i] = cost[src + i];
    pninl[prefix[q] + i] = ninl[src + i];
  }
}

__global__ void k_inlier_mask(int est, int nprob, const int64_t* __restrict__ offsets, const double* __restrict__ data,
                              const double* __restrict__ best_models, double thresh, uint8_t* __restrict__ mask) {
  const int p = blockIdx.y;
  const int ds = datum_size(est);
  const int n = (int)(offsets[p + 1] - offsets[p]);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double m[kStride];
  for (int k = 0; k < kStride; ++k) m[k] = best_models[(size_t)p * kStride + k];
  const double r = model_error(est, m, data + ((size_t)offsets[p] + i) * ds);
  mask[offsets[p] + i] = (r < thresh) ? 1 : 0;
}

// batched minimal solvers (directly bound entry points)
__global__ void k_five_point(int num, const double* __restrict__ corr, double* __restrict__ E, int* __restrict__ nsol) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= num) return;
  double c[20], e[90];
  for (int k = 0; k < 20; ++k) c[k] = corr[(size_t)i * 20 + k];
  const int n = rsc::five_point(c, e);
  nsol[i] = n;
  for (int k = 0; k < 90; ++k) E[(size_t)i * 90 + k] = (k < 9 * n) ? e[k] : 0.0;
}

__global__ void k_p3p(int num, const double* __restrict__ corr, double* __restrict__ R, double* __restrict__ t, int* __restrict__ nsol) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= num) return;
  double c[15], r[36], tt[12];
  for (int k = 0; k < 15; ++k) c[k] = corr[(size_t)i * 15 + k];
  const int n = rsc::p3p(c, r, tt);
  nsol[i] = n;
  for (int k = 0; k < 36; ++k) R[(size_t)i * 36 + k] = (k < 9 * n) ? r[k] : 0.0;
  for (int k = 0; k < 12; ++k) t[(size_t)i * 12 + k] = (k < 3 * n) ? tt[k] : 0.0;
}

// ---- LO-RANSAC refinement of the absolute-pose estimator (RefineModel,
// estimate_calibrated_absolute_pose.cc:120-153): the events of all problems of a
// replay round are refined as ONE batch of single-view LM solves (ba_batch.hip).
// An event = (problem, model source); source: refit from (samples, slot), or the
// problem's current best model.
__global__ void k_lo_prepare(int est, int nev, const int* __restrict__ ev_prob, const int* __restrict__ ev_samples,
                             const int* __restrict__ ev_slot, const int* __restrict__ ev_hyp, int round_B,
                             const double* __restrict__ round_models, const int* __restrict__ hyp_base,
                             const int64_t* __restrict__ offsets,
                             const double* __restrict__ data, const double* __restrict__ cur_models,
                             double* __restrict__ ev_model, double* __restrict__ ev_cam, EstParams ep) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nev) return;
  const int p = ev_prob[e];
  double mo[kStride];
  for (int k = 0; k < kStride; ++k) mo[k] = 0.0;
  if (ev_slot[e] < 0) {
    for (int k = 0; k < kStride; ++k) mo[k] = cur_models[(size_t)p * kStride + k];
  } else if (ev_hyp[e] >= 0) {
    // an event of the round being replayed: its model is still in the round's dense list (k_fit), the same bits a
    // refit would give -- a single-thread five-point refit costs ~2 ms of latency per LO batch
    const int h = ev_hyp[e];
    const int q = h / round_B;   // problem inside the chunk
    const double* src = round_models + ((size_t)q * round_B * max_models(est) + hyp_base[h] + ev_slot[e]) * (size_t)kStride;
    for (int k = 0; k < kStride; ++k) mo[k] = src[k];
  } else {
    const int m = sample_size(est), ds = datum_size(est);
    const double* pd = data + (size_t)offsets[p] * ds;
    double subset[kMaxSampleDoubles];
    for (int i = 0; i < m; ++i) {
      const int idx = ev_samples[e * kMaxSample + i];
      for (int k = 0; k < ds; ++k) subset[i * ds + k] = pd[(size_t)idx * ds + k];
    }
    double mloc[kMaxCap * kStride];
    const int nm = estimate_models(est, subset, mloc, ep);
    if (ev_slot[e] < nm) for (int k = 0; k < kStride; ++k) mo[k] = mloc[ev_slot[e] * kStride + k];
  }
  for (int k = 0; k < kStride; ++k) ev_model[(size_t)e * kStride + k] = mo[k];
  if (est == THEIA_EST_FUNDAMENTAL_MATRIX) {   // nine doubles per event, row-major
    double* fc = ev_cam + (size_t)e * 9;
    for (int k = 0; k < 9; ++k) fc[k] = mo[k];
    return;
  }
  if (est == THEIA_EST_HOMOGRAPHY) {   // homography->data(): Eigen's column-major storage order, nine doubles per event
    double* hc = ev_cam + (size_t)e * 9;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) hc[i + 3 * j] = mo[3 * i + j];
    return;
  }
  double* c = ev_cam + (size_t)e * 6;
  if (est == THEIA_EST_RELATIVE_POSE || est == THEIA_EST_UNCALIBRATED_RELATIVE_POSE) {
    // TwoViewInfo{rotation_2, position_2} of RefineModel (estimate_relative_pose.cc:115-118, estimate_uncalibrated_relative_pose.cc:157-160)
    rsc::eigen_rot_to_rotvec(mo + 9, c);
    c[3] = mo[18]; c[4] = mo[19]; c[5] = mo[20];
    return;
  }
  // Camera::SetPosition / SetOrientationFromRotationMatrix
  c[0] = mo[9]; c[1] = mo[10]; c[2] = mo[11];
  rsc::rot_to_angle_axis(mo, c + 3);
}

// inliers of the event's model, in data order, compacted as (uv, X, 1) for the view batch (absolute pose) or as
// correspondences for the two-view batch (relative pose); one wave per event
__global__ __launch_bounds__(64) void k_lo_gather(int est, const int* __restrict__ ev_prob, const int64_t* __restrict__ offsets,
                                                  const double* __restrict__ data, const double* __restrict__ ev_model,
                                                  double thresh, const double* __restrict__ ev_sqt /* LMED: per-event bound on r^2, else null */,
                                                  const int64_t* __restrict__ ev_off, int* __restrict__ ev_count,
                                                  double2* __restrict__ uv, double4* __restrict__ X) {
  const int e = blockIdx.x, lane = threadIdx.x;
  const int p = ev_prob[e];
  const int n = (int)(offsets[p + 1] - offsets[p]);
  const int ds = datum_size(est);
  const double* pd = data + (size_t)offsets[p] * ds;
  double m[kStride];
  for (int k = 0; k < kStride; ++k) m[k] = ev_model[(size_t)e * kStride + k];
  int base = 0;
  for (int i0 = 0; i0 < n; i0 += 64) {
    const int i = i0 + lane;
    bool in = false;
    if (i < n) {
      const double r = model_error(est, m, pd + (size_t)i * ds);
      in = ev_sqt ? (r * r < ev_sqt[e]) : (r < thresh);
    }
    const unsigned long long b = __ballot(in);
    if (in) {
      const int pos = base + __popcll(b & ((1ull << lane) - 1ull));
      const double* d = pd + (size_t)i * ds;
      if (est == THEIA_EST_RELATIVE_POSE || est == THEIA_EST_HOMOGRAPHY || est == THEIA_EST_FUNDAMENTAL_MATRIX) {   // the correspondence itself
        X[ev_off[e] + pos] = make_double4(d[0], d[1], d[2], d[3]);
      } else if (est == THEIA_EST_UNCALIBRATED_RELATIVE_POSE) {   // normalised by the model's focal lengths (:146-155)
        X[ev_off[e] + pos] = make_double4(d[0] / m[21], d[1] / m[21], d[2] / m[22], d[3] / m[22]);
      } else {
        uv[ev_off[e] + pos] = make_double2(d[0], d[1]);
        X[ev_off[e] + pos] = make_double4(d[2], d[3], d[4], 1.0);
      }
    }
    base += __popcll(b);
  }
  if (lane == 0) ev_count[e] = base;
}

struct LoOut { int success, term, iters, nsucc; double c0, c1; };   // = ba_batch.hip ViewOut

// refined pose back into the problem's model (written even when RefineModel returns false)
__global__ void k_lo_finish(int est, int nev, const int* __restrict__ ev_prob, const double* __restrict__ ev_cam,
                            const double* __restrict__ ev_model, const LoOut* __restrict__ out, double* __restrict__ cur_models,
                            int* __restrict__ ev_success) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nev) return;
  const int p = ev_prob[e];
  double* mo = cur_models + (size_t)p * kStride;
  if (est == THEIA_EST_FUNDAMENTAL_MATRIX) {   // estimate_fundamental_matrix.cc:82-90
    const double* fc = ev_cam + (size_t)e * 9;
    for (int k = 0; k < 9; ++k) mo[k] = fc[k];
    for (int k = 9; k < kStride; ++k) mo[k] = 0.0;
    ev_success[e] = (out[e].c1 < out[e].c0 && out[e].success) ? 1 : 0;
    return;
  }
  if (est == THEIA_EST_HOMOGRAPHY) {   // estimate_homography.cc:100-103: H (already divided by H(2,2)) replaces the model
    const double* hc = ev_cam + (size_t)e * 9;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) mo[3 * i + j] = hc[i + 3 * j];
    for (int k = 9; k < kStride; ++k) mo[k] = 0.0;
    ev_success[e] = (out[e].c1 < out[e].c0) ? 1 : 0;
    return;
  }
  const double* c = ev_cam + (size_t)e * 6;
  if (est == THEIA_EST_RELATIVE_POSE || est == THEIA_EST_UNCALIBRATED_RELATIVE_POSE) {
    // estimate_relative_pose.cc:130-135 / estimate_uncalibrated_relative_pose.cc:170-175: rotation and position are
    // replaced, the essential (fundamental) matrix and the focal lengths of the model are NOT recomputed -- Error keeps
    // scoring the Sampson distance of the unrefined matrix.  Success = the cost went down (and, uncalibrated, no FAILURE).
    for (int k = 0; k < kStride; ++k) mo[k] = ev_model[(size_t)e * kStride + k];
    rsc::eigen_rotvec_to_rot(c, mo + 9);
    mo[18] = c[3]; mo[19] = c[4]; mo[20] = c[5];
    const bool ok = out[e].c1 < out[e].c0;
    ev_success[e] = (est == THEIA_EST_RELATIVE_POSE ? ok : (ok && out[e].success)) ? 1 : 0;
    return;
  }
  rsc::angle_axis_to_rot(c + 3, mo);
  mo[9] = c[0]; mo[10] = c[1]; mo[11] = c[2];
  for (int k = 12; k < kStride; ++k) mo[k] = 0.0;
  ev_success[e] = (out[e].c1 < out[e].c0 && out[e].success) ? 1 : 0;
}

__global__ void k_select_models(int nprob, const int* __restrict__ use_cur, const double* __restrict__ cur_models,
                                double* __restrict__ best_models) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= nprob || !use_cur[p]) return;
  for (int k = 0; k < kStride; ++k) best_models[(size_t)p * kStride + k] = cur_models[(size_t)p * kStride + k];
}

// SQPnP on problems of any size (the directly bound solver, sfm.cc:592): one thread per problem
__global__ void k_sqpnp(int num, const int64_t* __restrict__ offsets, const double* __restrict__ feat,
                        const double* __restrict__ world, double* __restrict__ quats, double* __restrict__ ts,
                        int* __restrict__ nsol) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= num) return;
  double q[72], t[54];
  const int64_t o = offsets[i];
  const int n = rsc::sqpnp((int)(offsets[i + 1] - o), feat + 2 * o, world + 3 * o, q, t);
  nsol[i] = n;
  for (int k = 0; k < 72; ++k) quats[(size_t)i * 72 + k] = (k < 4 * n) ? q[k] : 0.0;
  for (int k = 0; k < 54; ++k) ts[(size_t)i * 54 + k] = (k < 3 * n) ? t[k] : 0.0;
}


// ---- five-point hypotheses (relative pose / essential matrix) in three stages instead of k_fit: the null space, the
// constraint matrix and its elimination (5 KB of work arrays per hypothesis) and the 10 x 10 eigen-decomposition (27 k
// read-modify-writes of its two work matrices) were per-lane scratch, i.e. HBM traffic (~200 KB per hypothesis for the
// eigen-solver alone); both run on chip now, by teams of lanes per hypothesis with the arrays in LDS (fit5_team.h,
// eig_team.h: bit-identical to the one-thread routines).  After them one thread per hypothesis turns the real
// eigenvectors into essential matrices and poses exactly as estimate_models() does.
constexpr int kFpWs = 136;    // per hypothesis: null space N (9 x 4) | action matrix M (10 x 10)

// stage A by teams of 16 lanes with the work arrays in LDS (fit5_team.h): N and M bit-identical to five_point_pre()
constexpr int kFpPreTeam = 16;
__global__ __launch_bounds__(64, 4) void k_fit5_a_team(int nprob, int B, const int64_t* __restrict__ offsets, const double* __restrict__ data,
                                                    const int* __restrict__ samples, const int* __restrict__ active_iters,
                                                    double* __restrict__ ws, int* __restrict__ ok) {
  __shared__ double lds[64 / kFpPreTeam][rsc::kFit5TeamLds];
  const int team = threadIdx.x / kFpPreTeam, tl = threadIdx.x % kFpPreTeam;
  const size_t hyp = (size_t)blockIdx.x * (64 / kFpPreTeam) + team;
  if (hyp >= (size_t)nprob * B) return;
  const int p = (int)(hyp / B), b = (int)(hyp % B);
  if (b >= active_iters[p]) { if (tl == 0) ok[hyp] = 0; return; }
  double* w = ws + hyp * kFpWs;
  const bool good = rsc::five_point_pre_team<kFpPreTeam>(data + (size_t)offsets[p] * 4, samples + hyp * 5, lds[team], w, w + 36, tl);
  if (tl == 0) ok[hyp] = good ? 1 : 0;
}

// stage B: teams of 8 lanes (eig_team.h), H and V in LDS, the work array X of the back-substitution in the hypothesis' own
// action-matrix slot in HBM (free once H is on chip).  The kernel is bound by the number of matrices resident per CU --
// the eigen-iteration is one dependent chain per matrix -- i.e. by LDS per matrix: 1.8 KB here (87 matrices per CU), and
// the same time with 16 or 8 lanes per team (measured), so the narrower team only halves the LDS per wave.
constexpr int kFpTeam = 8, kFpTeamsPerWave = 64 / kFpTeam;
template <bool P4PF = false>   // P4PF: the same eigen stage for the P4Pf action matrix; sol = rows 0..4 of the real eigenvectors
__global__ __launch_bounds__(64) void k_fit5_b(size_t nhyp, const int* __restrict__ ok, double* __restrict__ ws,
                                               double* __restrict__ sol, int* __restrict__ solmask) {
  __shared__ double lds[kFpTeamsPerWave][230];   // H (100) | V (100) | wr | wi | ort
  const int team = threadIdx.x / kFpTeam, tl = threadIdx.x % kFpTeam;
  const size_t hyp = (size_t)blockIdx.x * kFpTeamsPerWave + team;
  if (hyp >= nhyp) return;
  if (!ok[hyp]) { if (tl == 0) solmask[hyp] = 0; return; }
  double* H = lds[team]; double* V = H + 100; double* wr = V + 100; double* wi = wr + 10; double* ort = wi + 10;
  double* M = ws + hyp * kFpWs + 36;
  double* X = M;
  for (int e = tl; e < 100; e += kFpTeam) H[e] = M[e];
  rsc::team_sync();
  const bool good = rsc::eig_team<kFpTeam, false>(10, H, V, X, wr, wi, ort, tl);
  int bit = 0;
  for (int c = tl; c < 10; c += kFpTeam)
    if (good && wi[c] == 0.0) {   // only real solutions (five_point_relative_pose.cc:281-284)
      if (P4PF) {
        for (int k = 0; k < 5; ++k) sol[(hyp * 10 + c) * 5 + k] = V[k * 10 + c];
      } else {
        double v4[4];
        rsc::five_point_v4(V, c, v4);
        for (int k = 0; k < 4; ++k) sol[(hyp * 10 + c) * 4 + k] = v4[k];
      }
      bit |= 1 << c;
    }
  for (int o = kFpTeam / 2; o >= 1; o >>= 1) bit |= __shfl_xor(bit, o, kFpTeam);
  if (tl == 0) solmask[hyp] = bit;
}

// ---- SQPnP fit in three stages (sqpnp_pre -> 9 x 9 SVD by teams of nine lanes in LDS -> sqpnp_post): the SVD was 77 % of
// the one-thread-per-hypothesis kernel, whose 12.6 KB of arrays per lane lived in scratch.
// ws per hypothesis: [Omega 81 | P 27 | centroid 3 | U 81 | S 9]
#ifndef THIP_SQ_TEAM
#define THIP_SQ_TEAM 5   // lanes per 9 x 9 SVD (svd_team.h): 5 -> twelve matrices per wave; 9 (seven per wave) was rounds 3 - 5
#endif
constexpr int kSqWs = 201, kSqTeam = THIP_SQ_TEAM, kSqTeamsPerWave = 64 / kSqTeam;
__global__ __launch_bounds__(64) void k_sqp_a(int nprob, int B, const int64_t* __restrict__ offsets, const double* __restrict__ data,
                                              const int* __restrict__ samples, const int* __restrict__ active_iters,
                                              double* __restrict__ ws, int* __restrict__ ok) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const int p = blockIdx.y;
  if (b >= B || p >= nprob) return;
  const size_t hyp = (size_t)p * B + b;
  if (b >= active_iters[p]) { ok[hyp] = 0; return; }
  const double* pd = data + (size_t)offsets[p] * 5;
  double feat[6], world[9];
  for (int i = 0; i < 3; ++i) {
    const int idx = samples[hyp * 3 + i];
    feat[2 * i] = pd[(size_t)idx * 5]; feat[2 * i + 1] = pd[(size_t)idx * 5 + 1];
    for (int k = 0; k < 3; ++k) world[3 * i + k] = pd[(size_t)idx * 5 + 2 + k];
  }
  double Om[81], P[27], mean[3];
  const bool good = rsc::sqpnp_pre(3, feat, world, Om, P, mean);
  ok[hyp] = good ? 1 : 0;
  if (!good) return;
  double* w = ws + hyp * kSqWs;
  for (int k = 0; k < 81; ++k) w[k] = Om[k];
  for (int k = 0; k < 27; ++k) w[81 + k] = P[k];
  for (int k = 0; k < 3; ++k) w[108 + k] = mean[k];
}

__global__ __launch_bounds__(64) void k_sqp_b(size_t nhyp, const int* __restrict__ ok, double* __restrict__ ws) {
  __shared__ double lds[kSqTeamsPerWave][81 + 81 + 9];   // W | U | S
  const int team = threadIdx.x / kSqTeam, tl = threadIdx.x % kSqTeam;
  if (team >= kSqTeamsPerWave) return;
  const size_t hyp = (size_t)blockIdx.x * kSqTeamsPerWave + team;
  if (hyp >= nhyp || !ok[hyp]) return;
  double* W = lds[team]; double* U = W + 81; double* S = U + 81;
  double* w = ws + hyp * kSqWs;
  rsc::svd9_team<kSqTeam>(w, W, U, S, tl);
  for (int e = tl; e < 9; e += kSqTeam) {
    for (int i = 0; i < 9; ++i) w[111 + i * 9 + e] = U[i * 9 + e];
    w[192 + e] = S[e];
  }
}

__global__ __launch_bounds__(64) void k_sqp_c(int nprob, int B, const int* __restrict__ active_iters, const int* __restrict__ ok,
                                              const double* __restrict__ ws, double* __restrict__ models, int* __restrict__ counts,
                                              int* __restrict__ dense_count, int* __restrict__ tags, int* __restrict__ hyp_base) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const int p = blockIdx.y;
  if (b >= B || p >= nprob) return;
  const size_t hyp = (size_t)p * B + b;
  if (b >= active_iters[p] || !ok[hyp]) { counts[hyp] = 0; return; }
  const double* w = ws + hyp * kSqWs;
  double quats[72], ts[54];
  // Omega, P, the centroid, U and S are read where they lie (201 doubles of the workspace row): copies in per-lane arrays
  // would double the scratch frame of this kernel
  const int nm = rsc::sqpnp_post_impl(w, w + 81, w + 108, w + 111, w + 192, quats, ts);
  counts[hyp] = nm;
  if (nm == 0) return;
  constexpr int mm = 18;   // max_models(THEIA_EST_ABSOLUTE_POSE_SQPNP)
  const int base = atomicAdd(&dense_count[p], nm);
  hyp_base[hyp] = base;
  double* mo = models + ((size_t)p * B * mm + base) * (size_t)kStride;
  int* tg = tags + (size_t)p * B * mm + base;
  for (int j = 0; j < nm; ++j) {   // quaternion -> matrix as the estimator does (estimate_calibrated_absolute_pose.cc:99-106)
    double R[9];
    rsc::quat_to_rot(quats + 4 * j, R);
    const double* t = ts + 3 * j;
    double* m = mo + (size_t)j * kStride;
    for (int k = 0; k < 9; ++k) m[k] = R[k];
    for (int c = 0; c < 3; ++c) m[9 + c] = -((R[c] * t[0] + R[3 + c] * t[1]) + R[6 + c] * t[2]);
    for (int k = 12; k < kStride; ++k) m[k] = 0.0;
    tg[j] = b * mm + j;
  }
}

template <int EST>
__global__ __launch_bounds__(64) void k_fit5_c(int nprob, int B, const int64_t* __restrict__ offsets, const double* __restrict__ data,
                                               const int* __restrict__ samples, const int* __restrict__ active_iters,
                                               const double* __restrict__ ws, const double* __restrict__ sol,
                                               const int* __restrict__ solmask, double* __restrict__ models,
                                               int* __restrict__ counts, int* __restrict__ dense_count, int* __restrict__ tags,
                                               int* __restrict__ hyp_base) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const int p = blockIdx.y;
  if (b >= B || p >= nprob) return;
  const size_t hyp = (size_t)p * B + b;
  if (b >= active_iters[p]) { counts[hyp] = 0; return; }
  const int mask = solmask[hyp];
  if (!mask) { counts[hyp] = 0; return; }
  const double* pd = data + (size_t)offsets[p] * 4;
  double subset[20];
  for (int i = 0; i < 5; ++i) {
    const int idx = samples[hyp * 5 + i];
    for (int k = 0; k < 4; ++k) subset[i * 4 + k] = pd[(size_t)idx * 4 + k];
  }
  double N[36];
  for (int k = 0; k < 36; ++k) N[k] = ws[hyp * kFpWs + k];
  double mloc[10 * kStride];
  int nm = 0;
  for (int i = 0; i < 10; ++i) {
    if (!((mask >> i) & 1)) continue;
    double v4[4];
    for (int k = 0; k < 4; ++k) v4[k] = sol[(hyp * 10 + i) * 4 + k];
    double* m = mloc + kStride * nm;
    rsc::five_point_E(N, v4, m);
    if (EST == THEIA_EST_ESSENTIAL_MATRIX) { for (int k = 9; k < kStride; ++k) m[k] = 0.0; nm++; continue; }
    const int nfront = rsc::best_pose_from_E(m, subset, 5, m + 9, m + 18);
    if (nfront >= 4) nm++;
  }
  counts[hyp] = nm;
  if (nm == 0) return;
  const int mm = 10;
  const int base = atomicAdd(&dense_count[p], nm);
  hyp_base[hyp] = base;
  double* mo = models + ((size_t)p * B * mm + base) * (size_t)kStride;
  int* tg = tags + (size_t)p * B * mm + base;
  for (int j = 0; j < nm; ++j) {
    for (int k = 0; k < kStride; ++k) mo[j * kStride + k] = mloc[j * kStride + k];
    tg[j] = b * mm + j;
  }
}

// ---- P4Pf hypotheses (uncalibrated absolute pose) in three stages (p4pf_device.h): the transposed elimination template by one
// workgroup per hypothesis in LDS -> the five-point kernel's 10 x 10 eigen stage -> one thread per hypothesis for the
// projection matrices.  ws per hypothesis: [normalisation 36 | action matrix 100] (the five-point layout: kFpWs).
static_assert(p4pfdev::kWs == kFpWs, "the P4Pf workspace row shares the five-point eigen stage");
__global__ __launch_bounds__(p4pfdev::kThreads) void k_p4pf_a(int nprob, int B, const int64_t* __restrict__ offsets, const double* __restrict__ data,
                                                            const int* __restrict__ samples, const int* __restrict__ active_iters,
                                                            double* __restrict__ ws, int* __restrict__ ok) {
  extern __shared__ __attribute__((aligned(16))) double sm_p4pf[];
  const size_t hyp = blockIdx.x;
  const int p = (int)(hyp / B), b = (int)(hyp % B);
  if (b >= active_iters[p]) { if (threadIdx.x == 0) ok[hyp] = 0; return; }
  const bool good = p4pfdev::p4pf_action_wg(data + (size_t)offsets[p] * 5, samples + hyp * 4, sm_p4pf, ws + hyp * kFpWs);
  if (threadIdx.x == 0) ok[hyp] = good ? 1 : 0;
}

__global__ __launch_bounds__(64) void k_p4pf_c(int nprob, int B, const int* __restrict__ active_iters, const double* __restrict__ ws,
                                               const double* __restrict__ sol, const int* __restrict__ solmask, double* __restrict__ models,
                                               int* __restrict__ counts, int* __restrict__ dense_count, int* __restrict__ tags,
                                               int* __restrict__ hyp_base) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const int p = blockIdx.y;
  if (b >= B || p >= nprob) return;
  const size_t hyp = (size_t)p * B + b;
  if (b >= active_iters[p]) { counts[hyp] = 0; return; }
  const int mask = solmask[hyp];
  if (!mask) { counts[hyp] = 0; return; }
  double N[p4pfdev::kNorm];
  for (int k = 0; k < p4pfdev::kNorm; ++k) N[k] = ws[hyp * kFpWs + k];
  double mloc[10 * 12];
  int nm = 0;
  for (int i = 0; i < 10; ++i) {
    if (!((mask >> i) & 1)) continue;
    const double* v = sol + (hyp * 10 + i) * 5;
    const double w = v[4] / v[0];
    if (!(w >= 0.0)) continue;   // negative or NaN focal length^2 (four_point_focal_length_helper.cc:917-921)
    p4pfdev::projection_from_solution(N, w, v[3] / v[0], v[2] / v[0], v[1] / v[0], mloc + 12 * nm);
    nm++;
  }
  counts[hyp] = nm;
  if (nm == 0) return;
  const int mm = 10;
  const int base = atomicAdd(&dense_count[p], nm);
  hyp_base[hyp] = base;
  double* mo = models + ((size_t)p * B * mm + base) * (size_t)kStride;
  int* tg = tags + (size_t)p * B * mm + base;
  for (int j = 0; j < nm; ++j) {
    for (int k = 0; k < 12; ++k) mo[j * kStride + k] = mloc[j * 12 + k];
    for (int k = 12; k < kStride; ++k) mo[j * kStride + k] = 0.0;
    tg[j] = b * mm + j;
  }
}

// ---- homography hypotheses in three stages (as SQPnP: the 9 x 9 Jacobi SVD of A^T A was 2.3 KB of dynamically indexed
// per-lane arrays in scratch, 49 ns per hypothesis): M, T1, T2 per thread -> svd9_team with V -> H per thread.
// ws per hypothesis: [M 81 | T1 9 | T2 9 | Hn 9]
constexpr int kHomWs = 108;
__global__ __launch_bounds__(64) void k_hom_a(int nprob, int B, const int64_t* __restrict__ offsets, const double* __restrict__ data,
                                              const int* __restrict__ samples, const int* __restrict__ active_iters,
                                              double* __restrict__ ws) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const int p = blockIdx.y;
  if (b >= B || p >= nprob || b >= active_iters[p]) return;
  const size_t hyp = (size_t)p * B + b;
  const double* pd = data + (size_t)offsets[p] * 4;
  double subset[16], M[81], T1[9], T2[9];
  for (int i = 0; i < 4; ++i) {
    const int idx = samples[hyp * 4 + i];
    for (int k = 0; k < 4; ++k) subset[i * 4 + k] = pd[(size_t)idx * 4 + k];
  }
  rsc::four_point_homography_pre(subset, M, T1, T2);
  double* w = ws + hyp * kHomWs;
  for (int k = 0; k < 81; ++k) w[k] = M[k];
  for (int k = 0; k < 9; ++k) { w[81 + k] = T1[k]; w[90 + k] = T2[k]; }
}
__global__ __launch_bounds__(64) void k_hom_b(size_t nhyp, int B, const int* __restrict__ active_iters, double* __restrict__ ws) {
  __shared__ double lds[kSqTeamsPerWave][81 + 81 + 81 + 9];   // W | U | V | S
  const int team = threadIdx.x / kSqTeam, tl = threadIdx.x % kSqTeam;
  if (team >= kSqTeamsPerWave) return;
  const size_t hyp = (size_t)blockIdx.x * kSqTeamsPerWave + team;
  if (hyp >= nhyp || (int)(hyp % B) >= active_iters[hyp / B]) return;
  double* W = lds[team]; double* U = W + 81; double* V = U + 81; double* S = V + 81;
  double* w = ws + hyp * kHomWs;
  rsc::svd9_team<kSqTeam>(w, W, U, S, tl, V);
  for (int e = tl; e < 9; e += kSqTeam) w[99 + e] = V[9 * e + 8];   // the last right singular vector
}
__global__ __launch_bounds__(64) void k_hom_c(int nprob, int B, const int* __restrict__ active_iters, const double* __restrict__ ws,
                                              double* __restrict__ models, int* __restrict__ counts, int* __restrict__ dense_count,
                                              int* __restrict__ tags, int* __restrict__ hyp_base) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const int p = blockIdx.y;
  if (b >= B || p >= nprob) return;
  const size_t hyp = (size_t)p * B + b;
  if (b >= active_iters[p]) { counts[hyp] = 0; return; }
  const double* w = ws + hyp * kHomWs;
  double Hn[9], T1[9], T2[9], H[9];
  for (int k = 0; k < 9; ++k) { T1[k] = w[81 + k]; T2[k] = w[90 + k]; Hn[k] = w[99 + k]; }
  rsc::four_point_homography_post(Hn, T1, T2, H);
  counts[hyp] = 1;
  const int base = atomicAdd(&dense_count[p], 1);
  hyp_base[hyp] = base;
  double* mo = models + ((size_t)p * B + base) * (size_t)kStride;   // max_models = 1
  for (int k = 0; k < 9; ++k) mo[k] = H[k];
  for (int k = 9; k < kStride; ++k) mo[k] = 0.0;
  tags[(size_t)p * B + base] = b;
}

// ---- DLS-PnP hypotheses (estimate_calibrated_absolute_pose.cc:89-97): stage A = dls_kernels.hip (launch_dls_stage_a), then the
// eigen stage below.
__global__ __launch_bounds__(64) void k_dls_b(int nprob, int B, const int64_t* __restrict__ offsets,
                                              const double* __restrict__ data, const int* __restrict__ samples,
                                              const int* __restrict__ active_iters, const double* __restrict__ action,
                                              const double* __restrict__ tfac, const int* __restrict__ okflag,
                                              double* __restrict__ models, int* __restrict__ counts,
                                              int* __restrict__ dense_count, int* __restrict__ tags, int* __restrict__ hyp_base) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const int p = blockIdx.y;
  if (b >= B || p >= nprob) return;
  const size_t hyp = (size_t)p * B + b;
  if (b >= active_iters[p]) { counts[hyp] = 0; return; }
  int nm = 0;
  double quats[4 * dlsdev::kMaxSolutions], ts[3 * dlsdev::kMaxSolutions];
  if (okflag[hyp]) {
    double H[729], V[729], tf[27];
    const double* a = action + hyp * 729;
    for (int k = 0; k < 729; ++k) H[k] = a[k];
    for (int k = 0; k < 27; ++k) tf[k] = tfac[hyp * 27 + k];
    const double* pd = data + (size_t)offsets[p] * 5;
    nm = dlsdev::stage_b(H, V, tf, 3, pd + 2, 5, samples + hyp * 3, quats, ts);
  }
  counts[hyp] = nm;
  if (nm == 0) return;
  const int mm = dlsdev::kMaxSolutions;
  const int base = atomicAdd(&dense_count[p], nm);
  hyp_base[hyp] = base;
  double* mo = models + ((size_t)p * B * mm + base) * (size_t)kStride;
  int* tg = tags + (size_t)p * B * mm + base;
  for (int j = 0; j < nm; ++j) {
    double R[9];
    rsc::quat_to_rot(quats + 4 * j, R);
    const double* t = ts + 3 * j;
    double* m = mo + (size_t)j * kStride;
    for (int k = 0; k < 9; ++k) m[k] = R[k];
    for (int c = 0; c < 3; ++c) m[9 + c] = -((R[c] * t[0] + R[3 + c] * t[1]) + R[6 + c] * t[2]);
    for (int k = 12; k < kStride; ++k) m[k] = 0.0;
    tg[j] = b * mm + j;
  }
}

// stage B on chip: a team of 32 lanes per hypothesis, H / V / X in LDS (eig_team.h), lanes 0..26 turn one eigenvector
// column each into a pose, kept in column order by a team prefix sum
#ifndef THIP_DLS_TEAM
#define THIP_DLS_TEAM 32
#endif
constexpr int kDlsTeam = THIP_DLS_TEAM, kDlsColRounds = (27 + kDlsTeam - 1) / kDlsTeam, kDlsTeamsPerWave = 64 / kDlsTeam, kDlsTeamLds = 729 + 27 * dlsdev::kKeptRows + 81;   // H | kept rows of V | wr, wi, ort
__global__ __launch_bounds__(64) void k_dls_b_team(size_t nhyp, int B, const int64_t* __restrict__ offsets,
                                                   const double* __restrict__ data, const int* __restrict__ samples,
                                                   const int* __restrict__ active_iters, double* __restrict__ action,
                                                   const double* __restrict__ tfac, const int* __restrict__ okflag,
                                                   double* __restrict__ models, int* __restrict__ counts,
                                                   int* __restrict__ dense_count, int* __restrict__ tags, int* __restrict__ hyp_base) {
  __shared__ double lds[kDlsTeamsPerWave][kDlsTeamLds];
  const int team = threadIdx.x / kDlsTeam, tl = threadIdx.x % kDlsTeam;
  const size_t hyp = (size_t)blockIdx.x * kDlsTeamsPerWave + team;
  if (hyp >= nhyp) return;
  const int p = (int)(hyp / B), b = (int)(hyp % B);
  if (b >= active_iters[p] || !okflag[hyp]) { if (tl == 0) counts[hyp] = 0; return; }
  // The kernel is bound by the matrices resident per CU (one dependent chain each), i.e. by LDS per matrix.  H stays in
  // LDS; the work array X of the back-substitution lives in the hypothesis' own action-matrix slot in HBM (free once H is
  // on chip); so does the full eigenvector matrix while the Householder reflectors are accumulated -- after that only the
  // four rows column_solution() reads are carried (eig_team.h, NR): 14.7 instead of 24.6 KB of LDS per wave = ten
  // instead of six resident waves per CU
  double* H = lds[team]; double* Vk = H + 729; double* wr = Vk + 27 * dlsdev::kKeptRows; double* wi = wr + 27; double* ort = wi + 27;
  double* a = action + hyp * 729;
  for (int e = tl; e < 729; e += kDlsTeam) H[e] = a[e];
  rsc::team_sync();
  int nn = 27;
  asm volatile("" : "+s"(nn));   // the order stays a run-time value: with the literal the loops unroll into 250 VGPRs (two waves per SIMD)
  const bool good = rsc::eig_team<kDlsTeam, true, dlsdev::kKeptRows>(nn, H, a, a, wr, wi, ort, tl, Vk, dlsdev::kKeptRow);
  // the 27 eigenvector columns by lane (kDlsColRounds rounds of the team), the poses kept in column order by team prefix sums
  double quat[kDlsColRounds][4], tr[kDlsColRounds][3];
  int rank[kDlsColRounds];
  int nm = 0;
#pragma unroll
  for (int rd = 0; rd < kDlsColRounds; ++rd) {
    const int col = tl + rd * kDlsTeam;
    bool keep = false;
    if (good && col < 27) {
      const double* pd = data + (size_t)offsets[p] * 5;
      keep = dlsdev::column_solution<true>(Vk, wi, col, tfac + hyp * 27, 3, pd + 2, 5, samples + hyp * 3, quat[rd], tr[rd]);
    }
    int incl = keep ? 1 : 0;
    for (int o = 1; o < kDlsTeam; o <<= 1) { const int v = __shfl_up(incl, o, kDlsTeam); if (tl >= o) incl += v; }
    rank[rd] = keep ? nm + incl - 1 : -1;
    nm += __shfl(incl, kDlsTeam - 1, kDlsTeam);
  }
  int base = 0;
  if (tl == 0) { counts[hyp] = nm; if (nm) { base = atomicAdd(&dense_count[p], nm); hyp_base[hyp] = base; } }
  base = __shfl(base, 0, kDlsTeam);
#pragma unroll
  for (int rd = 0; rd < kDlsColRounds; ++rd) {
    if (rank[rd] < 0) continue;
    const int mm = dlsdev::kMaxSolutions, j = rank[rd];
    double* m = models + ((size_t)p * B * mm + base + j) * (size_t)kStride;
    double R[9];
    rsc::quat_to_rot(quat[rd], R);
    for (int k = 0; k < 9; ++k) m[k] = R[k];
    for (int c = 0; c < 3; ++c) m[9 + c] = -((R[c] * tr[rd][0] + R[3 + c] * tr[rd][1]) + R[6 + c] * tr[rd][2]);
    for (int k = 12; k < kStride; ++k) m[k] = 0.0;
    tags[(size_t)p * B * mm + base + j] = b * mm + j;
  }
}

__global__ __launch_bounds__(64) void k_gdls_b_team(size_t nhyp, int B, const int64_t* __restrict__ offsets,
                                                    const double* __restrict__ data, const int* __restrict__ samples,
                                                    const int* __restrict__ active_iters, double* __restrict__ action,
                                                    const double* __restrict__ tfac, const int* __restrict__ okflag,
                                                    double* __restrict__ models, int* __restrict__ counts,
                                                    int* __restrict__ dense_count, int* __restrict__ tags, int* __restrict__ hyp_base) {
  __shared__ double lds[kDlsTeamsPerWave][kDlsTeamLds];
  const int team = threadIdx.x / kDlsTeam, tl = threadIdx.x % kDlsTeam;
  const size_t hyp = (size_t)blockIdx.x * kDlsTeamsPerWave + team;
  if (hyp >= nhyp) return;
  const int p = (int)(hyp / B), b = (int)(hyp % B);
  if (b >= active_iters[p] || !okflag[hyp]) { if (tl == 0) counts[hyp] = 0; return; }
  double* H = lds[team]; double* Vk = H + 729; double* wr = Vk + 27 * dlsdev::kKeptRows; double* wi = wr + 27; double* ort = wi + 27;
  double* a = action + hyp * 729;
  for (int e = tl; e < 729; e += kDlsTeam) H[e] = a[e];
  rsc::team_sync();
  int nn = 27;
  asm volatile("" : "+s"(nn));
  const bool good = rsc::eig_team<kDlsTeam, true, dlsdev::kKeptRows>(nn, H, a, a, wr, wi, ort, tl, Vk, dlsdev::kKeptRow);
  double quat[kDlsColRounds][4], tr[kDlsColRounds][3], sc[kDlsColRounds];
  int rank[kDlsColRounds];
  int nm = 0;
#pragma unroll
  for (int rd = 0; rd < kDlsColRounds; ++rd) {
    const int col = tl + rd * kDlsTeam;
    bool keep = false;
    sc[rd] = 0.0;
    if (good && col < 27)
      keep = dlsdev::column_solution_gdls(Vk, wi, col, tfac + hyp * 36, 4, data + (size_t)offsets[p] * kSimDatum, kSimDatum, 0, 9, 3,
                                          samples + hyp * 4, quat[rd], tr[rd], &sc[rd]);
    int incl = keep ? 1 : 0;
    for (int o = 1; o < kDlsTeam; o <<= 1) { const int v = __shfl_up(incl, o, kDlsTeam); if (tl >= o) incl += v; }
    rank[rd] = keep ? nm + incl - 1 : -1;
    nm += __shfl(incl, kDlsTeam - 1, kDlsTeam);
  }
  int base = 0;
  if (tl == 0) { counts[hyp] = nm; if (nm) { base = atomicAdd(&dense_count[p], nm); hyp_base[hyp] = base; } }
  base = __shfl(base, 0, kDlsTeam);
#pragma unroll
  for (int rd = 0; rd < kDlsColRounds; ++rd) {
    if (rank[rd] < 0) continue;
    const int mm = dlsdev::kMaxSolutions, j = rank[rd];
    double* m = models + ((size_t)p * B * mm + base + j) * (size_t)kStride;
    double Rs[9];
    rsc::quat_to_rot(quat[rd], Rs);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) m[3 * r + c] = Rs[3 * c + r];                              // rotation = R^T
    for (int r = 0; r < 3; ++r) m[9 + r] = (m[3 * r] * -tr[rd][0] + m[3 * r + 1] * -tr[rd][1]) + m[3 * r + 2] * -tr[rd][2];       // rotation * -t
    m[12] = sc[rd];
    for (int k = 13; k < kStride; ++k) m[k] = 0.0;
    tags[(size_t)p * B * mm + base + j] = b * mm + j;
  }
}

// DlsPnp on problems of any size (the directly bound solver, sfm.cc:577): stage A per workgroup (dls_kernels.hip), then a thread per problem
__global__ __launch_bounds__(64) void k_dls_solve_b(int num, const int64_t* __restrict__ offsets, const double* __restrict__ world,
                                                    const double* __restrict__ action, const double* __restrict__ tfac,
                                                    const int* __restrict__ okflag, double* __restrict__ quats,
                                                    double* __restrict__ ts, int* __restrict__ nsol) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= num) return;
  double q[4 * dlsdev::kMaxSolutions], t[3 * dlsdev::kMaxSolutions];
  int n = 0;
  if (okflag[i]) {
    double H[729], V[729], tf[27];
    for (int k = 0; k < 729; ++k) H[k] = action[(size_t)i * 729 + k];
    for (int k = 0; k < 27; ++k) tf[k] = tfac[(size_t)i * 27 + k];
    const int64_t o = offsets[i];
    n = dlsdev::stage_b(H, V, tf, (int)(offsets[i + 1] - o), world + 3 * o, 3, nullptr, q, t);
  }
  nsol[i] = n;
  for (int k = 0; k < 4 * dlsdev::kMaxSolutions; ++k) quats[(size_t)i * 4 * dlsdev::kMaxSolutions + k] = (k < 4 * n) ? q[k] : 0.0;
  for (int k = 0; k < 3 * dlsdev::kMaxSolutions; ++k) ts[(size_t)i * 3 * dlsdev::kMaxSolutions + k] = (k < 3 * n) ? t[k] : 0.0;
}

// the index tables of the polynomial system live in constant memory, built once per process
// k_p4pf_a needs 70 KB of dynamic LDS (above the 64 KB default)
int p4pf_kernel_ready() {
  static std::once_flag once;
  static int status = 0;
  std::call_once(once, [] {
    const hipError_t e = hipFuncSetAttribute((const void*)k_p4pf_a, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p4pfdev::kLdsBytes);
    if (e != hipSuccess) status = set_error(THEIA_HIP_ERR_NO_DEVICE, hipGetErrorString(e));
  });
  return status;
}

// Macaulay terms of DlsPnp calls [0, ncalls) of a process
void dls_terms(std::vector<double>& u, dls::GlibcRand& gen, size_t ncalls) {
  while (u.size() < 4 * ncalls) u.push_back(dls::macaulay_term_from_rand(gen.next()));
}

// ------------------------------------------------------------------ host side
// std::mt19937 + libstdc++ uniform_int_distribution<int> (Lemire), i.e. the
// stream RandomNumberGenerator::RandInt draws (util/random.cc:46-84).
struct Mt19937 {
  uint32_t mt[624];
  int idx;
  void seed(uint32_t s) {
    mt[0] = s;
    for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    idx = 624;
  }
  uint32_t next() {
    if (idx >= 624) {
      for (int i = 0; i < 624; ++i) {
        const uint32_t y = (mt[i] & 0x80000000u) | (mt[(i + 1) % 624] & 0x7fffffffu);
        mt[i] = mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
      }
      idx = 0;
    }
    uint32_t y = mt[idx++];
    y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
    return y;
  }
  // libstdc++ std::uniform_real_distribution<double>(lo, hi): generate_canonical<double, 53> = two 32-bit draws (g0 + g1 * 2^32) / 2^64
  // (nextafter(1, 0) should the quotient round to 1), then * (hi - lo) + lo -- RandomNumberGenerator::RandDouble (util/random.cc:68-72)
  double rand_double(double lo, double hi) {
    double sum = 0.0, tmp = 1.0;
    for (int k = 0; k < 2; ++k) { sum += (double)next() * tmp; tmp *= 4294967296.0; }
    double ret = sum / tmp;
    if (ret >= 1.0) ret = std::nextafter(1.0, 0.0);
    return ret * (hi - lo) + lo;
  }
  int rand_int(int lo, int hi) {
    const uint32_t urange = (uint32_t)hi - (uint32_t)lo;
    uint32_t ret;
    if (urange == 0xffffffffu) ret = next();
    else {
      const uint32_t range = urange + 1u;
      uint64_t product = (uint64_t)next() * (uint64_t)range;
      uint32_t low = (uint32_t)product;
      if (low < range) {
        const uint32_t threshold = (uint32_t)(-range) % range;
        while (low < threshold) { product = (uint64_t)next() * (uint64_t)range; low = (uint32_t)product; }
      }
      ret = (uint32_t)(product >> 32);
    }
    return (int)(ret + (uint32_t)lo);
  }
};

// sample_consensus_estimator.h:252-297
// The minimal-solver kernels need up to ~12 KB of scratch per lane; the runtime sizes a hardware queue's
// scratch arena for a full chip of such waves, and two queues asking for it at the same time end in
// HSA_STATUS_ERROR_OUT_OF_RESOURCES (queue abort).  Every RANSAC kernel of the process therefore goes to ONE stream
// (= one hardware queue, one arena), whichever host thread enqueues it: calls from a thread pool interleave their
// launches on it (each call owns its buffers, the stream keeps each call's own order) and wait for their OWN work
// through an event -- no host-side lock, nobody waits for another caller's synchronisation.
hipStream_t solver_stream() {
  static hipStream_t s = [] {
    hipStream_t x = nullptr;
    if (hipStreamCreateWithFlags(&x, hipStreamNonBlocking) != hipSuccess) x = nullptr;
    return x;
  }();
  return s;
}
// transfers that may run beside the solver stream's kernels (no kernel ever goes here: no scratch arena)
hipStream_t copy_stream() {
  static hipStream_t s = [] {
    hipStream_t x = nullptr;
    if (hipStreamCreateWithFlags(&x, hipStreamNonBlocking) != hipSuccess) x = nullptr;
    return x;
  }();
  return s;
}
struct CallSync {   // "my work on the shared stream is done"
  hipEvent_t e = nullptr;
  CallSync() { (void)hipEventCreateWithFlags(&e, hipEventDisableTiming); }
  ~CallSync() { if (e) (void)hipEventDestroy(e); }
  hipError_t wait(hipStream_t st) {
    hipError_t r = hipEventRecord(e, st);
    return r != hipSuccess ? r : hipEventSynchronize(e);
  }
};

// Host-side loops over independent problems (sample streams, acceptance replay) on the library's persistent team of host
// threads (host_team.h: starting and joining 16 threads per loop cost ~0.6 ms, a round has two such loops); a second caller
// inside the team's region falls back to threads of its own.  THEIA_HIP_HOST_THREADS caps the count (default min(hardware
// threads, 32); 1 = serial).
template <class F>
void host_parallel_for(int n, F&& fn) {
  static const unsigned cap = [] {
    const char* e = getenv("THEIA_HIP_HOST_THREADS");
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    return e ? (unsigned)std::max(1, atoi(e)) : std::min(hw, 32u);
  }();
  const unsigned nt = std::min<unsigned>(cap, (unsigned)std::max(1, n / 4));
  if (nt < 2) { for (int i = 0; i < n; ++i) fn(i); return; }
  const std::function<void(int)> job = [&fn](int i) { fn(i); };
  if (host_team().run(n, nt, job)) return;
  std::atomic<int> next{0};
  std::vector<std::thread> th;
  th.reserve(nt);
  for (unsigned t = 0; t < nt; ++t)
    th.emplace_back([&] { for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) fn(i); });
  for (auto& t : th) t.join();
}

int compute_max_iterations(const theia_ransac_params& P, double min_sample_size, double inlier_ratio,
                           double log_failure_prob, int total) {
  if (inlier_ratio == 1.0) return P.min_iterations;
  const int ninl = (int)(inlier_ratio * total);
  const double num_samples = P.use_Tdd_test ? min_sample_size + 1 : min_sample_size;
  double a = 1.0, b = 1.0;
  for (int i = 0; i < num_samples; ++i) { a *= ninl - i; b *= total - i; }
  const double prob_all_inliers = a / b;
  if (prob_all_inliers < std::numeric_limits<double>::epsilon()) return P.max_iterations;
  if (prob_all_inliers >= 1.0 - std::numeric_limits<double>::epsilon()) return P.min_iterations;
  const double num_iterations = log_failure_prob / std::log(1.0 - prob_all_inliers);
  return (int)std::max((double)P.min_iterations, std::min(num_iterations, (double)P.max_iterations));
}

// ProsacSampler::Sample (solvers/prosac_sampler.cc:62-128): data sorted by
// quality; the k-th sample draws m-1 points from the top n-1 and the n-th point
// (or m from the top n once T'_n < k).  The reference pushes index `n` itself,
// which is one past the end once n reaches N; that single case is clamped to N-1
// here (the reference reads out of bounds there).
void prosac_sample(Mt19937& rng, int N, int m, int kth, int* out) {
  double t_n = 20000.0;  // ransac_convergence_iterations_
  int n = m;
  for (int i = 0; i < m; ++i) t_n *= (double)(n - i) / (N - i);
  double t_n_prime = 1.0;
  for (int t = 1; t <= kth; ++t) {
    if (t > t_n_prime && n < N) {
      const double t_n_plus1 = (t_n * (n + 1.0)) / (n + 1.0 - m);
      t_n_prime += std::ceil(t_n_plus1 - t_n);
      t_n = t_n_plus1;
      n++;
    }
  }
  auto draw_unique = [&](int count, int hi) {
    for (int i = 0; i < count; ++i) {
      int r;
      bool dup;
      do {
        r = rng.rand_int(0, hi);
        dup = false;
        for (int q = 0; q < i; ++q) dup |= (out[q] == r);
      } while (dup);
      out[i] = r;
    }
  };
  if (t_n_prime < kth) draw_unique(m, n - 1);
  else { draw_unique(m - 1, n - 2); out[m - 1] = std::min(n, N - 1); }
}

struct ProblemState {
  Mt19937 rng;
  std::vector<int> idx;
  double best_cost;
  int max_iterations, it, n;
  bool done;
  int best_slot;
  int best_hyp = -1;          // hypothesis (chunk-local problem * B + iteration) of the best model, if set in this round
  int best_samples[kMaxSample];
  int round_iters;
  int kth;  // PROSAC sample counter
  int ex_i, ex_j;  // ExhaustiveSampler cursor (exhaustive_sampler.cc:48,61-79)
  // replay cursor inside the current round and LO-RANSAC state
  int base_it, rb, rj;
  bool round_done, best_refined;
  double pending_ratio;
  int num_lo;
};

#define HIP_TRYR(expr)                                                                               \
  do {                                                                                               \
    hipError_t e_ = (expr);                                                                          \
    if (e_ != hipSuccess)                                                                            \
      return set_error(e_ == hipErrorOutOfMemory ? THEIA_HIP_ERR_OUT_OF_MEMORY : THEIA_HIP_ERR_NO_DEVICE, \
                       "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);   \
  } while (0)

}  // namespace
}  // namespace thip

using namespace thip;

extern "C" {

void theia_ransac_params_default(theia_ransac_params* p) {
  // sample_consensus_estimator.h:59-68
  std::memset(p, 0, sizeof(*p));
  p->error_thresh = -1;
  p->failure_probability = 0.01;
  p->min_inlier_ratio = 0;
  p->min_iterations = 100;
  p->max_iterations = std::numeric_limits<int>::max();
  p->use_mle = 0; p->use_Tdd_test = 0; p->use_lo = 0; p->lo_start_iterations = 50;
  p->seed = 0;
}

int theia_hip_ransac_estimate_batch(const theia_ransac_batch* batch, const theia_ransac_params* params,
                                    theia_ransac_result* result) {
  if (!batch || !params || !result) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "null argument");
  const theia_ransac_params& P = *params;
  // SampleConsensusEstimator ctor CHECKs (sample_consensus_estimator.h:217-223)
  if (!(P.error_thresh > 0)) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "Error threshold must be set to greater than zero");
  if (!(P.min_inlier_ratio <= 1.0) || !(P.min_inlier_ratio >= 0.0)) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "min_inlier_ratio must be in [0, 1]");
  if (!(P.failure_probability < 1.0) || !(P.failure_probability > 0.0)) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "failure_probability must be in (0, 1)");
  if (P.max_iterations < P.min_iterations) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "max_iterations < min_iterations");
  const int est = batch->estimator;
  EstParams ep{0.0, 0.0};
  if (est == THEIA_EST_UNCALIBRATED_RELATIVE_POSE && !batch->estimator_params)
    return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "the uncalibrated relative-pose estimator needs estimator_params = {min, max focal length}");
  if (est == THEIA_EST_UNCALIBRATED_RELATIVE_POSE && batch->estimator_params) {
    ep.min_focal = batch->estimator_params[0];
    ep.max_focal = batch->estimator_params[1];
  }
  const bool gdls_est = est == THEIA_EST_SIMILARITY_2D3D;
  const bool dls_est = est == THEIA_EST_ABSOLUTE_POSE_DLS || gdls_est;   // the Macaulay pipeline: stage A -> eigen stage
  if (est < 0 || est > THEIA_EST_RADIAL_DIST_UNCALIBRATED_ABSOLUTE_POSE) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "unknown estimator id");
  const bool abs_pose = est == THEIA_EST_ABSOLUTE_POSE_KNEIP || est == THEIA_EST_ABSOLUTE_POSE_SQPNP || (dls_est && !gdls_est);
  // estimators that keep Estimator::RefineModel's default "return true" (solvers/estimator.h:86-88): LO only counts
  const bool trivial_refine = est == THEIA_EST_ESSENTIAL_MATRIX || est == THEIA_EST_DOMINANT_PLANE ||
                              est == THEIA_EST_RELATIVE_POSE_KNOWN_ORIENTATION || est == THEIA_EST_ABSOLUTE_POSE_KNOWN_ORIENTATION ||
                              est == THEIA_EST_TRIANGULATION || est == THEIA_EST_RADIAL_HOMOGRAPHY || est == THEIA_EST_SIMILARITY_2D3D ||
                              est == THEIA_EST_UNCALIBRATED_ABSOLUTE_POSE || est == THEIA_EST_RIGID_TRANSFORMATION_2D3D ||
                              est == THEIA_EST_RADIAL_DIST_UNCALIBRATED_ABSOLUTE_POSE;
  const bool rel_pose = est == THEIA_EST_RELATIVE_POSE, uncal_pose = est == THEIA_EST_UNCALIBRATED_RELATIVE_POSE;
  const bool homog = est == THEIA_EST_HOMOGRAPHY, fund = est == THEIA_EST_FUNDAMENTAL_MATRIX;
  // every estimator's RefineModel is built: BundleAdjustView (absolute pose), BundleAdjustTwoViewsAngular ((un)calibrated
  // relative pose), OptimizeHomography, OptimizeFundamentalMatrix, and the default "return true" of the rest
  // exhaustive_sampler.cc:49-51 CHECK
  if (P.ransac_type == THEIA_RANSAC_EXHAUSTIVE && sample_size(est) != 2)
    return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "ExhaustiveSampler makes a hard assumption that the number of samples needed is 2.");
  const bool lmed = P.ransac_type == THEIA_RANSAC_LMED;
  if (P.ransac_type != THEIA_RANSAC_RANSAC && P.ransac_type != THEIA_RANSAC_PROSAC && !lmed &&
      P.ransac_type != THEIA_RANSAC_EXHAUSTIVE)
    return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "unknown ransac_type");
  const int nprob = batch->num_problems;
  if (nprob < 0 || (nprob > 0 && (!batch->offsets || !batch->data))) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "bad batch");
  if (nprob > 0 && (!result->success || !result->models || !result->num_inliers || !result->inlier_mask ||
                    !result->num_iterations || !result->confidence))
    return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "null result array");
  result->hypotheses_evaluated = 0; result->models_scored = 0; result->time_fit_score_seconds = 0.0;
  result->time_fit_seconds = 0.0; result->time_score_seconds = 0.0;
  if (nprob == 0) return 0;
  int rc = ensure_device();
  if (rc) return rc;
  if (dls_est && (rc = dls_ensure_tables())) return rc;
  const bool upnp_est = est == THEIA_EST_RIGID_TRANSFORMATION_2D3D;
  if (upnp_est && (rc = upnp_ensure_tables())) return rc;
  // P4Pfr: estimator_params = RadialDistUncalibratedAbsolutePoseMetaData {max focal length, min focal length, max distortion, min
  // distortion} [, first call of the process (the solver's static generator re-seeds the shared stream with 42)]
  const bool p4pfr_est = est == THEIA_EST_RADIAL_DIST_UNCALIBRATED_ABSOLUTE_POSE;
  double p4pfr_limits[4] = {0.0, 0.0, 0.0, 0.0};
  bool p4pfr_first_call = false;
  if (p4pfr_est) {
    if (!batch->estimator_params)
      return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "the radial-distortion absolute-pose estimator needs estimator_params = {max focal length, min focal length, max distortion, min distortion, first call (0 / 1)}");
    for (int k = 0; k < 4; ++k) p4pfr_limits[k] = batch->estimator_params[k];
    p4pfr_first_call = batch->estimator_params[4] != 0.0;
    // the reference CHECKs these (four_point_focal_length_radial_distortion.cc:82-90)
    if (!(p4pfr_limits[1] >= 0.0 && p4pfr_limits[0] >= 0.0 && p4pfr_limits[0] >= p4pfr_limits[1] && p4pfr_limits[2] <= 0.0 && p4pfr_limits[3] <= 0.0 &&
          p4pfr_limits[2] <= p4pfr_limits[3]))
      return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "P4Pfr: needs 0 <= min focal length <= max focal length and max distortion <= min distortion <= 0");
    if (P.ransac_type == THEIA_RANSAC_EXHAUSTIVE) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "the exhaustive sampler draws pairs: sample size 4 does not fit");
    if ((rc = p4pfr_ensure_tables())) return rc;
  }
  const auto t_entry = std::chrono::steady_clock::now();
  const int m = sample_size(est), ds = datum_size(est);
  const int64_t total = batch->offsets[nprob];
  int nmax = 0;
  std::vector<uint8_t> undersized(nprob, 0);
  for (int p = 0; p < nprob; ++p) {
    const int64_t n = batch->offsets[p + 1] - batch->offsets[p];
    if (n <= 0) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "Cannot perform estimation with 0 data measurements!");
    // fewer data than the minimal sample: the reference's sampler cannot be initialised for this problem (CHECK).  In a
    // batch only that problem fails (success = 0, no inliers, zero model); the others run.
    if (n < m) undersized[p] = 1;
    if (n > (1 << 30)) return set_error(THEIA_HIP_ERR_UNSUPPORTED, "problem too large");
    nmax = std::max(nmax, (int)n);
  }
  hipStream_t st = solver_stream();
  PoolStreamScope pool_scope(st);   // blocks this call hands back to the caches are tagged with an event on this stream (pools.h)
  if (!st) return set_error(THEIA_HIP_ERR_NO_DEVICE, "could not create the solver stream");
  CallSync mine;
  hipEvent_t ev0, ev1, evm;
  HIP_TRYR(hipEventCreate(&ev0)); HIP_TRYR(hipEventCreate(&ev1)); HIP_TRYR(hipEventCreate(&evm));
  struct EvGuard { hipEvent_t a, b, c; ~EvGuard() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); (void)hipEventDestroy(c); } } eguard{ev0, ev1, evm};
  double fit_ms = 0.0, score_ms = 0.0;

  DBuf<double> d_data; DBuf<int64_t> d_off;
  if ((rc = d_data.ensure((size_t)total * ds)) || (rc = d_off.ensure(nprob + 1))) return rc;
  // The correspondences come from pageable memory: the copy occupies the calling thread for total * ds * 8 B / ~11 GB/s (6 ms for
  // 1000 pairs x 2000 matches).  It runs on a helper thread while this one draws the first chunk's sample streams; nothing is
  // enqueued behind it on the stream before upload.wait() (the first device work of the chunk loop, and every exit path).
  struct Upload {
    std::thread th; hipError_t err = hipSuccess;
    int wait() { if (th.joinable()) th.join(); return err == hipSuccess ? 0 : set_error(THEIA_HIP_ERR_INTERNAL, "upload of the correspondences: %s", hipGetErrorString(err)); }
    ~Upload() { if (th.joinable()) th.join(); }
  } upload;
  {
    const double* src = batch->data; double* dst = d_data.p; const size_t bytes = sizeof(double) * (size_t)total * ds;
    const int dev = [] { int d = 0; (void)hipGetDevice(&d); return d; }();
    upload.th = std::thread([&upload, src, dst, bytes, st, dev] { (void)hipSetDevice(dev); upload.err = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st); });
  }

  const double log_failure_prob = std::log(P.failure_probability);
  // round size: everything at once when the iteration count is fixed/small,
  // otherwise chunks that the adaptive bound usually ends within
  int first_round = std::max(128, std::min(P.max_iterations, std::max(P.min_iterations, 512)));
  first_round = std::min(first_round, 4096);
  const int next_round = 1024;
  // problems per chunk: bound the model workspace (3 GiB by default -- 1.5 GiB cost five-point 1.4 % and SQPnP 5 % in per-chunk
  // synchronisations; THEIA_HIP_RANSAC_WORKSPACE_MB overrides)
  const int kMaxModels = max_models(est);   // slot stride of this estimator
  const size_t per_hyp = (size_t)kMaxModels * kStride * sizeof(double);
  static const size_t ws_bytes = [] { const char* e = getenv("THEIA_HIP_RANSAC_WORKSPACE_MB"); return e && atol(e) > 0 ? (size_t)atol(e) << 20 : (size_t)3 << 30; }();
  int chunk = (int)std::max<size_t>(1, ws_bytes / (per_hyp * (size_t)first_round));
  chunk = std::min(chunk, nprob);

  DBuf<int> d_samples2[2], d_counts, d_ninl, d_active, d_best_samples, d_best_slot, d_dense, d_tags;   // d_samples2: by host sample buffer
  DBuf<int> d_hyp_base;   // [problem][iteration] first dense model of the hypothesis (k_fit)
  DBuf<int> d_save;       // {problem, hypothesis, slot} triples of k_save_best
  DBuf<double> d_models, d_cost, d_best_models;
  DBuf<double> d_pcost; DBuf<int> d_pninl, d_prefix;   // the round's scores packed for the download (k_pack_scores)
  DBuf<double> d_dls_action, d_dls_tfac, d_dls_u; DBuf<int> d_dls_ok, d_iter_base;   // DLS: stage A -> stage B
  DBuf<double> d_fp_ws, d_fp_sol; DBuf<int> d_fp_ok, d_fp_mask;   // five-point: stages a -> b -> c
  std::vector<double> h_dls_u; dls::GlibcRand dls_gen; std::vector<int> h_iter_base;
  DBuf<uint8_t> d_mask;
  HBuf<double> h_rot2[2];   // P4Pfr: the "random rotation" matrix of every hypothesis of a round (made where its draws are taken)
  DBuf<double> d_rot;
  HBuf<int> h_samples2[2], h_counts, h_ninl, h_active2[2];   // pinned: sources / destinations of the per-round transfers (samples / active
                                                             // counts twice: the next chunk's first round is drawn while the GPU works)
  HBuf<double> h_cost;
  HBuf<int> h_hyp_base, h_prefix;   // per hypothesis: first model in its problem's dense order; per problem: first packed score
  size_t lmed_lds = (size_t)nmax * sizeof(double);
  int lmed_in_lds = 1;
  if (lmed) {
    // the squared residuals of a model stay in LDS for the radix select: 160 KB per workgroup on gfx950, 8 KB kept for the rest;
    // beyond that (more than 19 456 data in a problem) the passes of the select re-evaluate them (lmed_sq)
    if (lmed_lds > 152 * 1024) { lmed_lds = 0; lmed_in_lds = 0; }
    if (lmed_lds > 48 * 1024) {
      HIP_TRYR(hipFuncSetAttribute((const void*)k_score_lmed, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lmed_lds));
      HIP_TRYR(hipFuncSetAttribute((const void*)k_inlier_mask_lmed, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lmed_lds));
      HIP_TRYR(hipFuncSetAttribute((const void*)k_lo_lmed_bound, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lmed_lds));
    }
  }
  const bool use_lds = (size_t)nmax * ds * sizeof(double) <= 96 * 1024;
  const size_t lds_bytes = use_lds ? (size_t)nmax * ds * sizeof(double) : 0;
  if (use_lds && lds_bytes > 48 * 1024) {
    // every instance that can be launched below opts in (the compile-time-estimator instances are kernels of their own)
    const void* fn = (const void*)k_score<true>;
    if (est == THEIA_EST_RELATIVE_POSE) fn = (const void*)k_score<true, THEIA_EST_RELATIVE_POSE>;
    else if (est == THEIA_EST_ESSENTIAL_MATRIX) fn = (const void*)k_score<true, THEIA_EST_ESSENTIAL_MATRIX>;
    else if (est == THEIA_EST_ABSOLUTE_POSE_KNEIP) fn = (const void*)k_score<true, THEIA_EST_ABSOLUTE_POSE_KNEIP>;
    else if (est == THEIA_EST_ABSOLUTE_POSE_DLS) fn = (const void*)k_score<true, THEIA_EST_ABSOLUTE_POSE_DLS>;
    else if (est == THEIA_EST_ABSOLUTE_POSE_SQPNP) fn = (const void*)k_score<true, THEIA_EST_ABSOLUTE_POSE_SQPNP>;
    HIP_TRYR(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  }

  DBuf<double> d_upnp_state;   // UPnP: the estimator's accumulating cost parameters per problem (upnp_kernels.hip)
  if (upnp_est) {
    if ((rc = d_upnp_state.ensure((size_t)nprob * upnp_state_doubles()))) return rc;
    HIP_TRYR(hipMemsetAsync(d_upnp_state.p, 0, sizeof(double) * (size_t)nprob * upnp_state_doubles(), st));
  }
  std::vector<int> best_samples_all((size_t)nprob * kMaxSample, 0), best_slot_all(nprob, -1);
  std::vector<ProblemState> S(nprob);
  host_parallel_for(nprob, [&](int p) {   // (the generator's 624-word seeding and the index permutation of every problem: ~1 us each)
    ProblemState& s = S[p];
    s.n = (int)(batch->offsets[p + 1] - batch->offsets[p]);
    s.rng.seed(batch->seeds ? batch->seeds[p] : P.seed + (uint32_t)p);
    s.idx.resize(s.n);
    for (int i = 0; i < s.n; ++i) s.idx[i] = i;
    s.best_cost = std::numeric_limits<double>::max();
    s.max_iterations = P.max_iterations;
    if (P.min_inlier_ratio > 0)
      s.max_iterations = std::min(compute_max_iterations(P, m, P.min_inlier_ratio, log_failure_prob, s.n), P.max_iterations);
    s.it = 0; s.done = s.max_iterations <= 0 || undersized[p]; s.best_slot = -1; s.kth = 1; s.ex_i = 0; s.ex_j = 1;
    s.base_it = 0; s.rb = 0; s.rj = 0; s.round_done = true; s.best_refined = false; s.pending_ratio = 0.0; s.num_lo = 0;
    for (int k = 0; k < kMaxSample; ++k) s.best_samples[k] = 0;
  });
  double fit_score_ms = 0.0;
  // ---- LO-RANSAC (absolute / relative pose): batched RefineModel over a list of events
  DBuf<int> d_ev_prob, d_ev_samples, d_ev_slot, d_ev_hyp, d_ev_count, d_ev_success, d_lo_model_id;
  DBuf<int64_t> d_ev_off;
  DBuf<double> d_ev_model, d_ev_cam, d_lo_uv, d_lo_X, d_cur_models, d_lo_intr, d_ev_sqt;
  DBuf<char> d_lo_out;
  theia_ba_options lo_opts;
  theia_ba_options_default(&lo_opts);
  lo_opts.max_num_iterations = 2;                        // estimate_calibrated_absolute_pose.cc:124-129
  lo_opts.use_homogeneous_point_parametrization = 0;
  lo_opts.intrinsics_to_optimize = THEIA_INTR_NONE;
  lo_opts.loss_function_type = THEIA_LOSS_HUBER;
  lo_opts.robust_loss_width = P.error_thresh * 1.5;
  lo_opts.use_inner_iterations = 0;
  if (rel_pose || homog) {                               // estimate_relative_pose.cc:120-126, estimate_homography.cc:94-98
    lo_opts.max_num_iterations = 15;
    lo_opts.loss_function_type = THEIA_LOSS_TRUNCATED;
    lo_opts.robust_loss_width = P.error_thresh;
  }
  if (fund) { lo_opts.max_num_iterations = 2; lo_opts.loss_function_type = THEIA_LOSS_TRIVIAL; }   // estimate_fundamental_matrix.cc:56-57
  if (uncal_pose) lo_opts.max_num_iterations = 10;      // estimate_uncalibrated_relative_pose.cc:162-165 (HUBER, 1.5 x thresh)
  if (P.use_lo && (rc = d_cur_models.ensure((size_t)nprob * kStride))) return rc;
  if ((rc = d_best_models.ensure((size_t)nprob * kStride))) return rc;
  HIP_TRYR(hipMemsetAsync(d_best_models.p, 0, sizeof(double) * nprob * kStride, st));
  struct LoEvent { int prob, slot, hyp; int samples[8]; };   // hyp: (problem in chunk) * B + iteration of the round, or -1
  int lo_round_B = 0;
  // refines every event's model on its inliers; writes the refined pose to d_cur_models[prob]
  auto run_lo = [&](const std::vector<LoEvent>& evs, std::vector<int>& success) -> int {
    const int nev = (int)evs.size();
    success.assign(nev, 0);
    if (nev == 0) return 0;
    std::vector<int> hp(nev), hs((size_t)nev * kMaxSample), hsl(nev), hmod(nev, THEIA_CAM_PINHOLE), hhyp(nev);
    std::vector<int64_t> hoff(nev + 1, 0);
    std::vector<double> hintr((size_t)nev * THEIA_MAX_INTRINSICS, 0.0);
    for (int e = 0; e < nev; ++e) {
      hp[e] = evs[e].prob; hsl[e] = evs[e].slot; hhyp[e] = evs[e].hyp;
      for (int k = 0; k < kMaxSample; ++k) hs[(size_t)e * kMaxSample + k] = evs[e].samples[k];
      hoff[e + 1] = hoff[e] + S[evs[e].prob].n;          // capacity: every datum could be an inlier
      hintr[(size_t)e * THEIA_MAX_INTRINSICS] = 1.0; hintr[(size_t)e * THEIA_MAX_INTRINSICS + 1] = 1.0;   // Camera(): f = 1, aspect 1
    }
    int rc2;
    if ((rc2 = d_ev_prob.ensure(nev)) || (rc2 = d_ev_samples.ensure((size_t)nev * kMaxSample)) || (rc2 = d_ev_slot.ensure(nev)) ||
        (rc2 = d_ev_count.ensure(nev)) || (rc2 = d_ev_success.ensure(nev)) || (rc2 = d_ev_off.ensure(nev + 1)) ||
        (rc2 = d_ev_model.ensure((size_t)nev * kStride)) || (rc2 = d_ev_cam.ensure((size_t)nev * 9)) ||
        (rc2 = d_lo_uv.ensure((size_t)hoff[nev] * 2)) || (rc2 = d_lo_X.ensure((size_t)hoff[nev] * 4)) ||
        (rc2 = d_lo_intr.ensure(hintr.size())) || (rc2 = d_lo_model_id.ensure(nev)) ||
        (rc2 = d_lo_out.ensure(views_batch_out_bytes() * nev)))
      return rc2;
    HIP_TRYR(hipMemcpyAsync(d_ev_prob.p, hp.data(), sizeof(int) * nev, hipMemcpyHostToDevice, st));
    HIP_TRYR(hipMemcpyAsync(d_ev_samples.p, hs.data(), sizeof(int) * nev * kMaxSample, hipMemcpyHostToDevice, st));
    HIP_TRYR(hipMemcpyAsync(d_ev_slot.p, hsl.data(), sizeof(int) * nev, hipMemcpyHostToDevice, st));
    if ((rc2 = d_ev_hyp.ensure(nev))) return rc2;
    HIP_TRYR(hipMemcpyAsync(d_ev_hyp.p, hhyp.data(), sizeof(int) * nev, hipMemcpyHostToDevice, st));
    HIP_TRYR(hipMemcpyAsync(d_ev_off.p, hoff.data(), sizeof(int64_t) * (nev + 1), hipMemcpyHostToDevice, st));
    HIP_TRYR(hipMemcpyAsync(d_lo_intr.p, hintr.data(), sizeof(double) * hintr.size(), hipMemcpyHostToDevice, st));
    HIP_TRYR(hipMemcpyAsync(d_lo_model_id.p, hmod.data(), sizeof(int) * nev, hipMemcpyHostToDevice, st));
    k_lo_prepare<<<(nev + 63) / 64, 64, 0, st>>>(est, nev, d_ev_prob.p, d_ev_samples.p, d_ev_slot.p, d_ev_hyp.p, lo_round_B,
                                                 d_models.p, d_hyp_base.p, d_off.p, d_data.p,
                                                 d_cur_models.p, d_ev_model.p, d_ev_cam.p, ep);
    if (lmed) {
      if ((rc2 = d_ev_sqt.ensure(nev))) return rc2;
      k_lo_lmed_bound<<<nev, 256, lmed_lds, st>>>(est, d_ev_prob.p, d_off.p, d_data.p, d_ev_model.p, d_ev_sqt.p, lmed_in_lds);
    }
    k_lo_gather<<<nev, 64, 0, st>>>(est, d_ev_prob.p, d_off.p, d_data.p, d_ev_model.p, P.error_thresh, lmed ? d_ev_sqt.p : nullptr, d_ev_off.p, d_ev_count.p,
                                    reinterpret_cast<double2*>(d_lo_uv.p), reinterpret_cast<double4*>(d_lo_X.p));
    if (fund)
      fundamental_batch_device(nev, d_ev_off.p, d_ev_count.p, d_lo_X.p, d_ev_cam.p, &lo_opts, d_lo_out.p, st);
    else if (homog)
      homography_batch_device(nev, d_ev_off.p, d_ev_count.p, d_lo_X.p, d_ev_cam.p, &lo_opts, d_lo_out.p, st);
    else if (rel_pose || uncal_pose)   // the relative-pose RefineModel asks for CGNR, the uncalibrated one keeps the direct default
      twoview_batch_device(nev, d_ev_off.p, d_ev_count.p, d_lo_X.p, d_ev_cam.p, &lo_opts, rel_pose ? 1 : 0, d_lo_out.p, st);
    else
      views_batch_device(nev, d_ev_off.p, d_ev_count.p, d_lo_uv.p, nullptr, d_lo_X.p, d_ev_cam.p, d_lo_intr.p, d_lo_model_id.p,
                         nullptr, &lo_opts, d_lo_out.p, st);
    k_lo_finish<<<(nev + 63) / 64, 64, 0, st>>>(est, nev, d_ev_prob.p, d_ev_cam.p, d_ev_model.p,
                                                reinterpret_cast<const LoOut*>(d_lo_out.p), d_cur_models.p, d_ev_success.p);
    HIP_TRYR(hipMemcpyAsync(success.data(), d_ev_success.p, sizeof(int) * nev, hipMemcpyDeviceToHost, st));
    HIP_TRYR(mine.wait(st));
    return 0;
  };
  const bool host_timing = getenv("THEIA_HIP_RANSAC_TIMING") != nullptr;
  // one round of a chunk on the host: iterations per problem and the sample stream (RandomSampler::Sample with its
  // persistent permutation, PROSAC, EXHAUSTIVE); problems are independent (own generator, own slice): host threads share them
  auto gen_round = [&](int c0, int cn, bool first, HBuf<int>& act, HBuf<int>& smp, HBuf<double>& rotb, int* B_out) -> int {
    int B = 0;
    if (!act.assign(cn, 0)) return set_error(THEIA_HIP_ERR_OUT_OF_MEMORY, "hipHostMalloc failed");
    for (int q = 0; q < cn; ++q) {
      ProblemState& s = S[c0 + q];
      s.round_iters = 0;
      if (s.done) continue;
      const int cap = first ? first_round : next_round;
      s.round_iters = std::min(cap, s.max_iterations - s.it);
      act[q] = s.round_iters;
      B = std::max(B, s.round_iters);
    }
    *B_out = B;
    if (B == 0) return 0;
    if (!smp.resize((size_t)cn * B * m)) return set_error(THEIA_HIP_ERR_OUT_OF_MEMORY, "hipHostMalloc failed");
    if (p4pfr_est && !rotb.resize((size_t)cn * B * 9)) return set_error(THEIA_HIP_ERR_OUT_OF_MEMORY, "hipHostMalloc failed");
    host_parallel_for(cn, [&](int q) {
      ProblemState& s = S[c0 + q];
      int* out = smp.data() + (size_t)q * B * m;
      // P4Pfr takes three RandDouble(-0.5, 0.5) from the SAME generator after every sample (every RandomNumberGenerator object
      // shares one std::mt19937, util/random.cc:46-66); the solver's static RandomNumberGenerator(42) re-seeds that generator the
      // first time it runs in a process (four_point_focal_length_radial_distortion.cc:134-138)
      auto p4pfr_draws = [&](int b) {
        if (!p4pfr_est) return;
        if (p4pfr_first_call && s.it + b == 0) s.rng.seed(42);
        double v[3];
        for (int k = 0; k < 3; ++k) v[k] = s.rng.rand_double(-0.5, 0.5);
        p4pfr_rotation_from_draws(v, rotb.data() + ((size_t)q * B + b) * 9);
      };
      for (int b = 0; b < s.round_iters; ++b) {
        if (P.ransac_type == THEIA_RANSAC_PROSAC) { prosac_sample(s.rng, s.n, m, s.kth++, out + (size_t)b * m); p4pfr_draws(b); continue; }
        if (P.ransac_type == THEIA_RANSAC_EXHAUSTIVE) {   // all pairs (i, j > i), wrapping around
          out[(size_t)b * 2] = s.ex_i; out[(size_t)b * 2 + 1] = s.ex_j;
          if (++s.ex_j >= s.n) {
            if (++s.ex_i >= s.n - 1) s.ex_i = 0;
            s.ex_j = s.ex_i + 1;
          }
          continue;
        }
        for (int i = 0; i < m; ++i) {
          std::swap(s.idx[i], s.idx[s.rng.rand_int(i, s.n - 1)]);
          out[(size_t)b * m + i] = s.idx[i];
        }
        p4pfr_draws(b);
      }
      for (size_t e = (size_t)s.round_iters * m; e < (size_t)B * m; ++e) out[e] = 0;   // iterations beyond this problem's round
    });
    return 0;
  };
  int bufi = 0, pre_c0 = -1, pre_B = 0;   // pre_*: the first round of chunk pre_c0 already sits in buffer 1 - bufi
  bool offsets_up = false;
  // The sample streams of the NEXT chunk's first round are drawn while this chunk's kernels run (pre_*); their 30 MB go up on
  // the copy stream at once, into the device buffer of their host buffer, and the solver stream waits for the event instead
  // of copying them between two chunks' kernels.
  hipStream_t copy_st = copy_stream();
  struct PreUpload {
    hipEvent_t ev = nullptr; bool pending = false;
    PreUpload() { (void)hipEventCreateWithFlags(&ev, hipEventDisableTiming); }
    ~PreUpload() { if (ev) { if (pending) (void)hipEventSynchronize(ev); (void)hipEventDestroy(ev); } }   // (the buffers outlive the copy)
  } pre_up;
  const auto t_loop = std::chrono::steady_clock::now();
  for (int c0 = 0; c0 < nprob; c0 += chunk) {
    const int cn = std::min(chunk, nprob - c0);
    bool first = true;
    while (true) {
      int B = 0;
      const auto tp0 = std::chrono::steady_clock::now();
      bool samples_up = false;
      if (first && pre_c0 == c0) { bufi ^= 1; B = pre_B; pre_c0 = -1; samples_up = pre_up.pending; }
      else if ((rc = gen_round(c0, cn, first, h_active2[bufi], h_samples2[bufi], h_rot2[bufi], &B))) return rc;
      HBuf<int>& h_active = h_active2[bufi];
      HBuf<int>& h_samples = h_samples2[bufi];
      DBuf<int>& d_samples = d_samples2[bufi];
      if (B == 0) break;
      first = false;
      const auto tp1 = std::chrono::steady_clock::now();
      const size_t nh = (size_t)cn * B;
      if ((rc = d_samples.ensure(nh * m)) || (rc = d_counts.ensure(nh)) || (rc = d_models.ensure(nh * kMaxModels * kStride)) ||
          (rc = d_cost.ensure(nh * kMaxModels)) || (rc = d_ninl.ensure(nh * kMaxModels)) || (rc = d_active.ensure(cn)) ||
          (rc = d_dense.ensure(cn)) || (rc = d_tags.ensure(nh * kMaxModels)) || (rc = d_hyp_base.ensure(nh)))
        return rc;
      if (!offsets_up) {
        if ((rc = upload.wait())) return rc;
        HIP_TRYR(hipMemcpyAsync(d_off.p, batch->offsets, sizeof(int64_t) * (nprob + 1), hipMemcpyHostToDevice, st));
        offsets_up = true;
      }
      HIP_TRYR(hipMemsetAsync(d_dense.p, 0, sizeof(int) * cn, st));
      if (samples_up) { HIP_TRYR(hipStreamWaitEvent(st, pre_up.ev, 0)); pre_up.pending = false; }
      else HIP_TRYR(hipMemcpyAsync(d_samples.p, h_samples.data(), sizeof(int) * nh * m, hipMemcpyHostToDevice, st));
      HIP_TRYR(hipMemcpyAsync(d_active.p, h_active.data(), sizeof(int) * cn, hipMemcpyHostToDevice, st));
      HIP_TRYR(hipEventRecord(ev0, st));
      if (dls_est) {
        h_iter_base.assign(cn, 0);
        int calls = 0;
        for (int q = 0; q < cn; ++q) { h_iter_base[q] = S[c0 + q].it; calls = std::max(calls, S[c0 + q].it + S[c0 + q].round_iters); }
        const size_t had = h_dls_u.size();
        dls_terms(h_dls_u, dls_gen, (size_t)calls);
        if ((rc = d_dls_action.ensure(nh * 729)) || (rc = d_dls_tfac.ensure(nh * 36)) || (rc = d_dls_ok.ensure(nh)) ||
            (rc = d_iter_base.ensure(cn)))
          return rc;
        if (h_dls_u.size() != had || d_dls_u.cap < h_dls_u.size()) {
          if ((rc = d_dls_u.ensure(std::max<size_t>(h_dls_u.size(), 4 * 4096)))) return rc;
          HIP_TRYR(hipMemcpyAsync(d_dls_u.p, h_dls_u.data(), sizeof(double) * h_dls_u.size(), hipMemcpyHostToDevice, st));
        }
        HIP_TRYR(hipMemcpyAsync(d_iter_base.p, h_iter_base.data(), sizeof(int) * cn, hipMemcpyHostToDevice, st));
        HIP_TRYR(hipEventRecord(ev0, st));   // (re-recorded: the uploads above are not part of the fit time)
        if (gdls_est) {
          launch_dls_stage_a(true, kSimDatum, cn, B, d_off.p + c0, d_data.p, d_samples.p, d_active.p, d_iter_base.p, d_dls_u.p,
                             d_dls_action.p, d_dls_tfac.p, d_dls_ok.p, st);
          k_gdls_b_team<<<(unsigned)((nh + kDlsTeamsPerWave - 1) / kDlsTeamsPerWave), 64, 0, st>>>(
              nh, B, d_off.p + c0, d_data.p, d_samples.p, d_active.p, d_dls_action.p, d_dls_tfac.p, d_dls_ok.p, d_models.p,
              d_counts.p, d_dense.p, d_tags.p, d_hyp_base.p);
        } else {
        launch_dls_stage_a(false, 5, cn, B, d_off.p + c0, d_data.p, d_samples.p, d_active.p, d_iter_base.p, d_dls_u.p,
                           d_dls_action.p, d_dls_tfac.p, d_dls_ok.p, st);
        if (getenv("THEIA_HIP_DLS_THREAD_EIG"))
          k_dls_b<<<dim3((B + 63) / 64, cn), 64, 0, st>>>(cn, B, d_off.p + c0, d_data.p, d_samples.p, d_active.p, d_dls_action.p,
                                                          d_dls_tfac.p, d_dls_ok.p, d_models.p, d_counts.p, d_dense.p, d_tags.p,
                                                          d_hyp_base.p);
        else
          k_dls_b_team<<<(unsigned)((nh + kDlsTeamsPerWave - 1) / kDlsTeamsPerWave), 64, 0, st>>>(
              nh, B, d_off.p + c0, d_data.p, d_samples.p, d_active.p, d_dls_action.p, d_dls_tfac.p, d_dls_ok.p, d_models.p,
              d_counts.p, d_dense.p, d_tags.p, d_hyp_base.p);
        }
      } else if ((est == THEIA_EST_RELATIVE_POSE || est == THEIA_EST_ESSENTIAL_MATRIX) && !getenv("THEIA_HIP_FIT_ONE_KERNEL")) {
        if ((rc = d_fp_ws.ensure(nh * kFpWs)) || (rc = d_fp_sol.ensure(nh * 40)) || (rc = d_fp_ok.ensure(nh)) || (rc = d_fp_mask.ensure(nh)))
          return rc;
        dim3 grid((B + 63) / 64, cn);
        k_fit5_a_team<<<(unsigned)((nh + 3) / 4), 64, 0, st>>>(cn, B, d_off.p + c0, d_data.p, d_samples.p, d_active.p, d_fp_ws.p, d_fp_ok.p);
        k_fit5_b<false><<<(unsigned)((nh + kFpTeamsPerWave - 1) / kFpTeamsPerWave), 64, 0, st>>>(nh, d_fp_ok.p, d_fp_ws.p, d_fp_sol.p, d_fp_mask.p);
        // (one LANE per root on teams of 16 -- a lane carries one model instead of ten -- measured 8 % slower on the leg: two or
        // three of a hypothesis' ten roots are real, so most lanes of such a team idle)
        if (est == THEIA_EST_RELATIVE_POSE)
          k_fit5_c<THEIA_EST_RELATIVE_POSE><<<grid, 64, 0, st>>>(cn, B, d_off.p + c0, d_data.p, d_samples.p, d_active.p, d_fp_ws.p, d_fp_sol.p,
                                                                d_fp_mask.p, d_models.p, d_counts.p, d_dense.p, d_tags.p, d_hyp_base.p);
        else
          k_fit5_c<THEIA_EST_ESSENTIAL_MATRIX><<<grid, 64, 0, st>>>(cn, B, d_off.p + c0, d_data.p, d_samples.p, d_active.p, d_fp_ws.p, d_fp_sol.p,
                                                                   d_fp_mask.p, d_models.p, d_counts.p, d_dense.p, d_tags.p, d_hyp_base.p);
      } else if (p4pfr_est) {
        if ((rc = d_fp_ws.ensure(nh * (size_t)p4pfr_workspace_doubles())) || (rc = d_rot.ensure(nh * 9))) return rc;
        HIP_TRYR(hipMemcpyAsync(d_rot.p, h_rot2[bufi].data(), sizeof(double) * nh * 9, hipMemcpyHostToDevice, st));
        HIP_TRYR(hipEventRecord(ev0, st));   // (re-recorded: the upload above is not part of the fit time)
        launch_p4pfr_fit(cn, B, d_off.p + c0, d_data.p, d_samples.p, d_active.p, d_rot.p, p4pfr_limits, d_fp_ws.p, d_models.p, d_counts.p, d_dense.p,
                         d_tags.p, d_hyp_base.p, st);
      } else if (upnp_est) {
        if ((rc = d_fp_ws.ensure(nh * (size_t)upnp_workspace_doubles()))) return rc;
        launch_upnp_fit(cn, B, d_off.p + c0, d_data.p, d_samples.p, d_active.p, d_upnp_state.p + (size_t)c0 * upnp_state_doubles(), d_fp_ws.p,
                        d_models.p, d_counts.p, d_dense.p, d_tags.p, d_hyp_base.p, st);
      } else if (est == THEIA_EST_UNCALIBRATED_ABSOLUTE_POSE) {
        if ((rc = d_fp_ws.ensure(nh * kFpWs)) || (rc = d_fp_sol.ensure(nh * 50)) || (rc = d_fp_ok.ensure(nh)) || (rc = d_fp_mask.ensure(nh)))
          return rc;
        if ((rc = p4pf_kernel_ready())) return rc;
        dim3 grid((B + 63) / 64, cn);
        k_p4pf_a<<<(unsigned)nh, p4pfdev::kThreads, p4pfdev::kLdsBytes, st>>>(cn, B, d_off.p + c0, d_data.p, d_samples.p, d_active.p, d_fp_ws.p, d_fp_ok.p);
        k_fit5_b<true><<<(unsigned)((nh + kFpTeamsPerWave - 1) / kFpTeamsPerWave), 64, 0, st>>>(nh, d_fp_ok.p, d_fp_ws.p, d_fp_sol.p, d_fp_mask.p);
        k_p4pf_c<<<grid, 64, 0, st>>>(cn, B, d_active.p, d_fp_ws.p, d_fp_sol.p, d_fp_mask.p, d_models.p, d_counts.p, d_dense.p, d_tags.p, d_hyp_base.p);
      } else if (est == THEIA_EST_HOMOGRAPHY && !getenv("THEIA_HIP_FIT_ONE_KERNEL")) {
        if ((rc = d_fp_ws.ensure(nh * kHomWs))) return rc;
        dim3 grid((B + 63) / 64, cn);
        k_hom_a<<<grid, 64, 0, st>>>(cn, B, d_off.p + c0, d_data.p, d_samples.p, d_active.p, d_fp_ws.p);
        k_hom_b<<<(unsigned)((nh + kSqTeamsPerWave - 1) / kSqTeamsPerWave), 64, 0, st>>>(nh, B, d_active.p, d_fp_ws.p);
        k_hom_c<<<grid, 64, 0, st>>>(cn, B, d_active.p, d_fp_ws.p, d_models.p, d_counts.p, d_dense.p, d_tags.p, d_hyp_base.p);
      } else if (est == THEIA_EST_ABSOLUTE_POSE_SQPNP && !getenv("THEIA_HIP_FIT_ONE_KERNEL")) {
        if ((rc = d_fp_ws.ensure(nh * kSqWs)) || (rc = d_fp_ok.ensure(nh))) return rc;
        dim3 grid((B + 63) / 64, cn);
        k_sqp_a<<<grid, 64, 0, st>>>(cn, B, d_off.p + c0, d_data.p, d_samples.p, d_active.p, d_fp_ws.p, d_fp_ok.p);
        k_sqp_b<<<(unsigned)((nh + kSqTeamsPerWave - 1) / kSqTeamsPerWave), 64, 0, st>>>(nh, d_fp_ok.p, d_fp_ws.p);
        k_sqp_c<<<grid, 64, 0, st>>>(cn, B, d_active.p, d_fp_ok.p, d_fp_ws.p, d_models.p, d_counts.p, d_dense.p, d_tags.p, d_hyp_base.p);
      } else {
        dim3 grid((B + 63) / 64, cn);
#define THIP_FIT(E) k_fit<E><<<grid, 64, 0, st>>>(est, cn, B, d_off.p + c0, d_data.p, d_samples.p, d_active.p, d_models.p, d_counts.p, d_dense.p, d_tags.p, d_hyp_base.p, ep)
        switch (est) {
          case THEIA_EST_RELATIVE_POSE: THIP_FIT(THEIA_EST_RELATIVE_POSE); break;
          case THEIA_EST_ESSENTIAL_MATRIX: THIP_FIT(THEIA_EST_ESSENTIAL_MATRIX); break;
          case THEIA_EST_ABSOLUTE_POSE_KNEIP: THIP_FIT(THEIA_EST_ABSOLUTE_POSE_KNEIP); break;
          case THEIA_EST_ABSOLUTE_POSE_SQPNP: THIP_FIT(THEIA_EST_ABSOLUTE_POSE_SQPNP); break;
          case THEIA_EST_FUNDAMENTAL_MATRIX: THIP_FIT(THEIA_EST_FUNDAMENTAL_MATRIX); break;
          case THEIA_EST_HOMOGRAPHY: THIP_FIT(THEIA_EST_HOMOGRAPHY); break;
          case THEIA_EST_DOMINANT_PLANE: THIP_FIT(THEIA_EST_DOMINANT_PLANE); break;
          case THEIA_EST_RELATIVE_POSE_KNOWN_ORIENTATION: THIP_FIT(THEIA_EST_RELATIVE_POSE_KNOWN_ORIENTATION); break;
          case THEIA_EST_UNCALIBRATED_RELATIVE_POSE: THIP_FIT(THEIA_EST_UNCALIBRATED_RELATIVE_POSE); break;
          case THEIA_EST_TRIANGULATION: THIP_FIT(THEIA_EST_TRIANGULATION); break;
          case THEIA_EST_RADIAL_HOMOGRAPHY: THIP_FIT(THEIA_EST_RADIAL_HOMOGRAPHY); break;
          default: THIP_FIT(THEIA_EST_ABSOLUTE_POSE_KNOWN_ORIENTATION); break;
        }
#undef THIP_FIT
      }
      HIP_TRYR(hipEventRecord(evm, st));
      if (lmed) {
        dim3 grid(B * kMaxModels, cn);
        k_score_lmed<<<grid, 256, lmed_lds, st>>>(est, cn, B, d_off.p + c0, d_data.p, d_models.p, d_dense.p, d_tags.p, d_cost.p, d_ninl.p, lmed_in_lds);
      } else {
#define THIP_SCORE(L, E, BYTES) k_score<L, E><<<dim3((B * kMaxModels + score_threads(E) - 1) / score_threads(E), cn), score_threads(E), BYTES, st>>>(est, cn, B, d_off.p + c0, d_data.p, d_models.p, d_dense.p, d_tags.p, P.error_thresh, P.use_mle, d_cost.p, d_ninl.p)
        if (use_lds) {
          if (est == THEIA_EST_RELATIVE_POSE) THIP_SCORE(true, THEIA_EST_RELATIVE_POSE, lds_bytes);
          else if (est == THEIA_EST_ESSENTIAL_MATRIX) THIP_SCORE(true, THEIA_EST_ESSENTIAL_MATRIX, lds_bytes);
          else if (est == THEIA_EST_ABSOLUTE_POSE_KNEIP) THIP_SCORE(true, THEIA_EST_ABSOLUTE_POSE_KNEIP, lds_bytes);
          else if (est == THEIA_EST_ABSOLUTE_POSE_DLS) THIP_SCORE(true, THEIA_EST_ABSOLUTE_POSE_DLS, lds_bytes);
          else if (est == THEIA_EST_ABSOLUTE_POSE_SQPNP) THIP_SCORE(true, THEIA_EST_ABSOLUTE_POSE_SQPNP, lds_bytes);
          else THIP_SCORE(true, -1, lds_bytes);
        } else {
          if (est == THEIA_EST_RELATIVE_POSE) THIP_SCORE(false, THEIA_EST_RELATIVE_POSE, 0);
          else if (est == THEIA_EST_ABSOLUTE_POSE_SQPNP) THIP_SCORE(false, THEIA_EST_ABSOLUTE_POSE_SQPNP, 0);
          else THIP_SCORE(false, -1, 0);
        }
#undef THIP_SCORE
      }
      HIP_TRYR(hipEventRecord(ev1, st));
      if (!h_counts.resize(nh) || !h_hyp_base.resize(nh) || !h_prefix.resize((size_t)cn + 1))
        return set_error(THEIA_HIP_ERR_OUT_OF_MEMORY, "hipHostMalloc failed");
      HIP_TRYR(hipMemcpyAsync(h_counts.data(), d_counts.p, sizeof(int) * nh, hipMemcpyDeviceToHost, st));
      HIP_TRYR(hipMemcpyAsync(h_hyp_base.data(), d_hyp_base.p, sizeof(int) * nh, hipMemcpyDeviceToHost, st));
      HIP_TRYR(hipMemcpyAsync(h_prefix.data() + 1, d_dense.p, sizeof(int) * cn, hipMemcpyDeviceToHost, st));
      HIP_TRYR(hipGetLastError());
      // while the GPU fits and scores this round: the first round of the next chunk (its problems are not touched before)
      if (pre_c0 < 0 && c0 + chunk < nprob) {
        const int nc0 = c0 + chunk;
        if ((rc = gen_round(nc0, std::min(chunk, nprob - nc0), true, h_active2[1 - bufi], h_samples2[1 - bufi], h_rot2[1 - bufi], &pre_B))) return rc;
        pre_c0 = nc0;
        const size_t pnh = (size_t)std::min(chunk, nprob - nc0) * pre_B;
        if (pre_B > 0 && copy_st && pre_up.ev && !pre_up.pending) {
          if ((rc = d_samples2[1 - bufi].ensure(pnh * m))) return rc;
          HIP_TRYR(hipMemcpyAsync(d_samples2[1 - bufi].p, h_samples2[1 - bufi].data(), sizeof(int) * pnh * m, hipMemcpyHostToDevice, copy_st));
          HIP_TRYR(hipEventRecord(pre_up.ev, copy_st));
          pre_up.pending = true;
        }
      }
      HIP_TRYR(mine.wait(st));
      {   // the scores, packed: only the models that exist travel (all max_models slots of every hypothesis were 100 - 270 MB a round)
        h_prefix.data()[0] = 0;
        int most = 0;
        for (int q = 0; q < cn; ++q) { most = std::max(most, h_prefix.data()[q + 1]); h_prefix.data()[q + 1] += h_prefix.data()[q]; }
        const size_t total_models = (size_t)h_prefix.data()[cn];
        if (!h_cost.resize(std::max<size_t>(1, total_models)) || !h_ninl.resize(std::max<size_t>(1, total_models)))
          return set_error(THEIA_HIP_ERR_OUT_OF_MEMORY, "hipHostMalloc failed");
        if (total_models > 0) {
          if ((rc = d_prefix.ensure((size_t)cn + 1)) || (rc = d_pcost.ensure(total_models)) || (rc = d_pninl.ensure(total_models))) return rc;
          HIP_TRYR(hipMemcpyAsync(d_prefix.p, h_prefix.data(), sizeof(int) * ((size_t)cn + 1), hipMemcpyHostToDevice, st));
          k_pack_scores<<<dim3((unsigned)std::min(64, (most + 255) / 256), cn), 256, 0, st>>>(B, kMaxModels, d_dense.p, d_prefix.p, d_cost.p, d_ninl.p,
                                                                                          d_pcost.p, d_pninl.p);
          HIP_TRYR(hipMemcpyAsync(h_cost.data(), d_pcost.p, sizeof(double) * total_models, hipMemcpyDeviceToHost, st));
          HIP_TRYR(hipMemcpyAsync(h_ninl.data(), d_pninl.p, sizeof(int) * total_models, hipMemcpyDeviceToHost, st));
          HIP_TRYR(hipGetLastError());
          HIP_TRYR(mine.wait(st));
        }
      }
      const auto tp2 = std::chrono::steady_clock::now();
      { float ms = 0.f; if (hipEventElapsedTime(&ms, ev0, ev1) == hipSuccess) fit_score_ms += ms;
        if (hipEventElapsedTime(&ms, ev0, evm) == hipSuccess) fit_ms += ms;
        if (hipEventElapsedTime(&ms, evm, ev1) == hipSuccess) score_ms += ms; }
      // sequential replay of the acceptance rules (sample_consensus_estimator.h:330-394).  With
      // use_lo a problem pauses at each RefineModel event; the events of all problems are refined
      // as one batch, then every replay resumes where it stopped.
      for (int q = 0; q < cn; ++q) {
        ProblemState& s = S[c0 + q];
        s.base_it = s.it; s.rb = 0; s.rj = 0; s.round_done = s.done; s.best_hyp = -1;
      }
      lo_round_B = B;
      std::vector<LoEvent> events;
      std::vector<int> ev_q, ev_ok;
      std::atomic<long long> n_hyp{0}, n_scored{0};
      while (true) {
        events.clear(); ev_q.clear();
        // without LO nothing is shared between the problems' replays (the counters are atomics): host threads
        auto replay_one = [&](int q) {
          ProblemState& s = S[c0 + q];
          if (s.round_done) return;
          long long my_hyp = 0, my_scored = 0;
          bool paused = false;
          // (a hypothesis interrupted by an LO event is finished even if max_iterations dropped meanwhile)
          while (!paused && s.rb < s.round_iters && (s.rj > 0 || s.base_it + s.rb < s.max_iterations)) {
            const size_t hyp = (size_t)q * B + s.rb;
            const int nm = h_counts[hyp];
            if (s.rj == 0) my_hyp++;
            while (s.rj < nm) {
              const int j = s.rj++;
              const size_t at = (size_t)h_prefix.data()[q] + (size_t)h_hyp_base[hyp] + (size_t)j;
              const double cost = h_cost[at];
              const int ninl = h_ninl[at];
              my_scored++;
              const double inlier_ratio = (double)ninl / (double)s.n;
              if (cost < s.best_cost) {
                s.best_cost = cost;
                s.best_slot = j;
                s.best_hyp = (int)hyp;
                s.best_refined = false;
                for (int i = 0; i < m; ++i) s.best_samples[i] = h_samples[hyp * m + i];
                if (inlier_ratio < m / (double)s.n) continue;
                if (P.use_lo && trivial_refine && s.base_it + s.rb >= P.lo_start_iterations) {
                  s.num_lo++;   // RefineModel = "return true": nothing changes but the counter
                } else if (P.use_lo && s.base_it + s.rb >= P.lo_start_iterations) {   // :373-381
                  LoEvent ev; ev.prob = c0 + q; ev.slot = j; ev.hyp = (int)hyp;
                  for (int i = 0; i < kMaxSample; ++i) ev.samples[i] = s.best_samples[i];
                  events.push_back(ev); ev_q.push_back(q);
                  s.pending_ratio = inlier_ratio;
                  paused = true;
                  break;
                }
                s.max_iterations = std::min(compute_max_iterations(P, m, inlier_ratio, log_failure_prob, s.n), s.max_iterations);
              }
            }
            if (!paused) { s.rb++; s.rj = 0; }
          }
          if (!paused) {
            s.round_done = true;
            s.it = s.base_it + s.rb;
            if (s.it >= s.max_iterations) s.done = true;
          }
          n_hyp += my_hyp; n_scored += my_scored;
        };
        if (P.use_lo) { for (int q = 0; q < cn; ++q) replay_one(q); }   // LO events are collected in problem order
        else host_parallel_for(cn, replay_one);
        if (events.empty()) break;
        if ((rc = run_lo(events, ev_ok))) return rc;
        for (size_t e = 0; e < events.size(); ++e) {
          ProblemState& s = S[events[e].prob];
          s.best_refined = true;                       // RefineModel overwrites the pose even when it fails
          if (getenv("THEIA_HIP_RANSAC_DEBUG"))
            std::fprintf(stderr, "[hip] prob %d it %d slot %d ratio %.17g lo %d\n", events[e].prob, s.base_it + s.rb, events[e].slot, s.pending_ratio, ev_ok[e]);
          if (!ev_ok[e]) continue;                     // "continue": no max_iterations update
          s.num_lo++;
          s.max_iterations = std::min(compute_max_iterations(P, m, s.pending_ratio, log_failure_prob, s.n), s.max_iterations);
        }
      }
      if (host_timing) {
        const auto tp3 = std::chrono::steady_clock::now();
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        std::fprintf(stderr, "[theia_hip ransac] chunk %d+%d round B=%d: samples %.1f ms, upload + kernels + download %.1f ms, replay %.1f ms\n",
                     c0, cn, B, ms(tp0, tp1), ms(tp1, tp2), ms(tp2, tp3));
      }
      {   // best models found in this round -> d_best_models
        std::vector<int> sp, sh, ss;
        for (int q = 0; q < cn; ++q) {
          const ProblemState& s = S[c0 + q];
          if (s.best_hyp >= 0) { sp.push_back(c0 + q); sh.push_back(s.best_hyp); ss.push_back(s.best_slot); }
        }
        if (!sp.empty()) {
          const int ns = (int)sp.size();
          if ((rc = d_save.ensure((size_t)3 * ns))) return rc;
          HIP_TRYR(hipMemcpyAsync(d_save.p, sp.data(), sizeof(int) * ns, hipMemcpyHostToDevice, st));
          HIP_TRYR(hipMemcpyAsync(d_save.p + ns, sh.data(), sizeof(int) * ns, hipMemcpyHostToDevice, st));
          HIP_TRYR(hipMemcpyAsync(d_save.p + 2 * ns, ss.data(), sizeof(int) * ns, hipMemcpyHostToDevice, st));
          k_save_best<<<(ns * kStride + 255) / 256, 256, 0, st>>>(est, ns, d_save.p, d_save.p + ns, d_save.p + 2 * ns, B, d_models.p,
                                                                  d_hyp_base.p, d_best_models.p);
          HIP_TRYR(mine.wait(st));   // the host vectors above are the sources of asynchronous uploads
        }
      }
      result->hypotheses_evaluated += n_hyp.load(); result->models_scored += n_scored.load();
    }
  }
  const auto t_final = std::chrono::steady_clock::now();
  if (!offsets_up) {   // (no round ran at all)
    if ((rc = upload.wait())) return rc;
    HIP_TRYR(hipMemcpyAsync(d_off.p, batch->offsets, sizeof(int64_t) * (nprob + 1), hipMemcpyHostToDevice, st));
  }
  // final models + inlier masks
  for (int p = 0; p < nprob; ++p) {
    best_slot_all[p] = S[p].best_slot;
    for (int k = 0; k < kMaxSample; ++k) best_samples_all[(size_t)p * kMaxSample + k] = S[p].best_samples[k];
  }
  if ((rc = d_best_samples.ensure((size_t)nprob * kMaxSample)) || (rc = d_best_slot.ensure(nprob)) ||
      (rc = d_best_models.ensure((size_t)nprob * kStride)) || (rc = d_mask.ensure((size_t)total)))
    return rc;
  HIP_TRYR(hipMemcpyAsync(d_best_samples.p, best_samples_all.data(), sizeof(int) * nprob * kMaxSample, hipMemcpyHostToDevice, st));
  HIP_TRYR(hipMemcpyAsync(d_best_slot.p, best_slot_all.data(), sizeof(int) * nprob, hipMemcpyHostToDevice, st));
  // d_best_models already holds every problem's best model (k_save_best); THEIA_HIP_RANSAC_REFIT=1 recomputes them from the
  // best samples instead (the same bits: the solver is deterministic)
  if (getenv("THEIA_HIP_RANSAC_REFIT") && !dls_est)
    k_refit<<<(nprob + 63) / 64, 64, 0, st>>>(est, nprob, d_off.p, d_data.p, d_best_samples.p, d_best_slot.p, d_best_models.p, ep);
  std::vector<int> use_cur;   // source of an asynchronous upload: lives until the final synchronisation
  if (P.use_lo && !trivial_refine) {   // the best model of a problem may be the refined pose of its last LO event
    use_cur.resize(nprob);
    for (int p = 0; p < nprob; ++p) use_cur[p] = (S[p].best_refined && S[p].best_slot >= 0) ? 1 : 0;
    if ((rc = d_ev_slot.ensure(nprob))) return rc;
    HIP_TRYR(hipMemcpyAsync(d_ev_slot.p, use_cur.data(), sizeof(int) * nprob, hipMemcpyHostToDevice, st));
    k_select_models<<<(nprob + 63) / 64, 64, 0, st>>>(nprob, d_ev_slot.p, d_cur_models.p, d_best_models.p);
  }
  if (lmed) {
    k_inlier_mask_lmed<<<nprob, 256, lmed_lds, st>>>(est, nprob, d_off.p, d_data.p, d_best_models.p, d_mask.p, lmed_in_lds);
  } else {
    dim3 grid((nmax + 255) / 256, nprob);
    k_inlier_mask<<<grid, 256, 0, st>>>(est, nprob, d_off.p, d_data.p, d_best_models.p, P.error_thresh, d_mask.p);
  }
  if (P.use_lo && trivial_refine) {   // :401-406 with RefineModel = "return true": the counter only
    for (int p = 0; p < nprob; ++p) S[p].num_lo++;   // "++summary->num_lo_iterations" is unconditional
  } else if (P.use_lo) {   // sample_consensus_estimator.h:401-406: one more RefineModel on the final inliers (result unused)
    HIP_TRYR(hipMemcpyAsync(d_cur_models.p, d_best_models.p, sizeof(double) * nprob * kStride, hipMemcpyDeviceToDevice, st));
    std::vector<LoEvent> evs;
    for (int p = 0; p < nprob; ++p)
      if (S[p].best_slot >= 0) { LoEvent ev; ev.prob = p; ev.slot = -1; ev.hyp = -1; for (int k = 0; k < kMaxSample; ++k) ev.samples[k] = 0; evs.push_back(ev); }
    std::vector<int> ok;
    if ((rc = run_lo(evs, ok))) return rc;
    for (int p = 0; p < nprob; ++p) S[p].num_lo++;   // unconditional in the reference, also when no model was found
    HIP_TRYR(hipMemcpyAsync(d_best_models.p, d_cur_models.p, sizeof(double) * nprob * kStride, hipMemcpyDeviceToDevice, st));
  }
  HIP_TRYR(hipGetLastError());
  HIP_TRYR(mine.wait(st));
  // caller-owned (pageable) destinations: blocking copies after the stream has drained
  HIP_TRYR(hipMemcpy(result->models, d_best_models.p, sizeof(double) * nprob * kStride, hipMemcpyDeviceToHost));
  HIP_TRYR(hipMemcpy(result->inlier_mask, d_mask.p, (size_t)total, hipMemcpyDeviceToHost));
  {   // the slots of a model row past the estimator's layout (theia_hip.h) are padding: handed back as zeros
    const int used = model_doubles(est);
    for (int p = 0; p < nprob; ++p)
      for (int k = used; k < kStride; ++k) result->models[(size_t)p * kStride + k] = 0.0;
  }
  for (int p = 0; p < nprob; ++p) {
    const ProblemState& s = S[p];
    int cnt = 0;
    for (int64_t i = batch->offsets[p]; i < batch->offsets[p + 1]; ++i) cnt += result->inlier_mask[i];
    result->num_inliers[p] = cnt;
    result->num_iterations[p] = s.it;
    if (result->num_lo_iterations) result->num_lo_iterations[p] = s.num_lo;
    result->success[p] = 1;
    const double inlier_ratio = (double)cnt / s.n;
    result->confidence[p] = 1.0 - std::pow(1.0 - std::pow(inlier_ratio, (double)m), (double)s.it);
    if (undersized[p]) {
      result->success[p] = 0; result->num_inliers[p] = 0; result->confidence[p] = 0.0;
      for (int64_t i = batch->offsets[p]; i < batch->offsets[p + 1]; ++i) result->inlier_mask[i] = 0;
    }
  }
  result->time_fit_score_seconds = fit_score_ms * 1e-3;
  result->time_fit_seconds = fit_ms * 1e-3; result->time_score_seconds = score_ms * 1e-3;
  if (host_timing) {
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    std::fprintf(stderr, "[theia_hip ransac] call: set-up %.1f ms, rounds %.1f ms, best models + inlier masks + results %.1f ms\n", ms(t_entry, t_loop),
                 ms(t_loop, t_final), ms(t_final, std::chrono::steady_clock::now()));
  }
  return 0;
}

int theia_hip_five_point_relative_pose(int32_t num, const double* corr, double* essential_matrices, int32_t* num_solutions) {
  if (num < 0 || (num > 0 && (!corr || !essential_matrices || !num_solutions))) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "bad argument");
  if (num == 0) return 0;
  int rc = ensure_device();
  if (rc) return rc;
  hipStream_t st = solver_stream();
  PoolStreamScope pool_scope(st);   // blocks this call hands back to the caches are tagged with an event on this stream (pools.h)
  if (!st) return set_error(THEIA_HIP_ERR_NO_DEVICE, "could not create the solver stream");
  CallSync mine;
  DBuf<double> dc, de; DBuf<int> dn;
  if ((rc = dc.ensure((size_t)num * 20)) || (rc = de.ensure((size_t)num * 90)) || (rc = dn.ensure(num))) return rc;
  HIP_TRYR(hipMemcpyAsync(dc.p, corr, sizeof(double) * num * 20, hipMemcpyHostToDevice, st));
  k_five_point<<<(num + 63) / 64, 64, 0, st>>>(num, dc.p, de.p, dn.p);
  HIP_TRYR(hipMemcpyAsync(essential_matrices, de.p, sizeof(double) * num * 90, hipMemcpyDeviceToHost, st));
  HIP_TRYR(hipMemcpyAsync(num_solutions, dn.p, sizeof(int) * num, hipMemcpyDeviceToHost, st));
  HIP_TRYR(hipGetLastError());
  HIP_TRYR(mine.wait(st));
  return 0;
}

int theia_hip_four_point_pose_and_focal_length(int32_t num, const double* corr2d3d, double* projection_matrices, int32_t* num_solutions) {
  if (num < 0 || (num > 0 && (!corr2d3d || !projection_matrices || !num_solutions))) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "bad argument");
  if (num == 0) return 0;
  int rc = ensure_device();
  if (rc || (rc = p4pf_kernel_ready())) return rc;
  hipStream_t st = solver_stream();
  PoolStreamScope pool_scope(st);   // blocks this call hands back to the caches are tagged with an event on this stream (pools.h)
  if (!st) return set_error(THEIA_HIP_ERR_NO_DEVICE, "could not create the solver stream");
  CallSync mine;
  // one problem of four data per call row, one hypothesis each with the identity sample: the RANSAC stages as they are
  DBuf<double> dc, dws, dsol, dmod; DBuf<int> dn, dok, dmask, dsamp, dact, ddense, dtags, dbase; DBuf<int64_t> doff;
  const size_t n = (size_t)num;
  if ((rc = dc.ensure(n * 20)) || (rc = dws.ensure(n * kFpWs)) || (rc = dsol.ensure(n * 50)) || (rc = dmod.ensure(n * 10 * kStride)) ||
      (rc = dn.ensure(n)) || (rc = dok.ensure(n)) || (rc = dmask.ensure(n)) || (rc = dsamp.ensure(n * 4)) || (rc = dact.ensure(n)) ||
      (rc = ddense.ensure(n)) || (rc = dtags.ensure(n * 10)) || (rc = dbase.ensure(n)) || (rc = doff.ensure(n + 1)))
    return rc;
  std::vector<int64_t> off(n + 1);
  std::vector<int> samp(n * 4), act(n, 1);
  for (size_t i = 0; i <= n; ++i) off[i] = (int64_t)(4 * i);
  for (size_t i = 0; i < n * 4; ++i) samp[i] = (int)(i % 4);
  HIP_TRYR(hipMemcpyAsync(dc.p, corr2d3d, sizeof(double) * n * 20, hipMemcpyHostToDevice, st));
  HIP_TRYR(hipMemcpyAsync(doff.p, off.data(), sizeof(int64_t) * (n + 1), hipMemcpyHostToDevice, st));
  HIP_TRYR(hipMemcpyAsync(dsamp.p, samp.data(), sizeof(int) * n * 4, hipMemcpyHostToDevice, st));
  HIP_TRYR(hipMemcpyAsync(dact.p, act.data(), sizeof(int) * n, hipMemcpyHostToDevice, st));
  HIP_TRYR(hipMemsetAsync(ddense.p, 0, sizeof(int) * n, st));
  HIP_TRYR(hipMemsetAsync(dmod.p, 0, sizeof(double) * n * 10 * kStride, st));
  k_p4pf_a<<<(unsigned)n, p4pfdev::kThreads, p4pfdev::kLdsBytes, st>>>(num, 1, doff.p, dc.p, dsamp.p, dact.p, dws.p, dok.p);
  k_fit5_b<true><<<(unsigned)((n + kFpTeamsPerWave - 1) / kFpTeamsPerWave), 64, 0, st>>>(n, dok.p, dws.p, dsol.p, dmask.p);
  k_p4pf_c<<<dim3(1, num), 64, 0, st>>>(num, 1, dact.p, dws.p, dsol.p, dmask.p, dmod.p, dn.p, ddense.p, dtags.p, dbase.p);
  std::vector<double> hm(n * 10 * kStride);
  HIP_TRYR(hipMemcpyAsync(hm.data(), dmod.p, sizeof(double) * hm.size(), hipMemcpyDeviceToHost, st));
  HIP_TRYR(hipMemcpyAsync(num_solutions, dn.p, sizeof(int) * n, hipMemcpyDeviceToHost, st));
  HIP_TRYR(hipGetLastError());
  HIP_TRYR(mine.wait(st));
  for (size_t i = 0; i < n; ++i)
    for (int j = 0; j < 10; ++j)
      for (int k = 0; k < 12; ++k) projection_matrices[(i * 10 + j) * 12 + k] = j < num_solutions[i] ? hm[(i * 10 + j) * kStride + k] : 0.0;
  return 0;
}

// n draws of RandomNumberGenerator(seed).RandInt(lo, hi) (util/random.cc:46-84: std::mt19937 + uniform_int_distribution<int>), for
// host code that has to follow the reference's generator outside the sampler (the random candidates of the guided matcher)
int theia_hip_randint_stream(uint32_t seed, int32_t n, int32_t lo, int32_t hi, int32_t* out) {
  if (n < 0 || hi < lo || (n > 0 && !out)) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "bad argument");
  Mt19937 g;
  g.seed(seed);
  for (int i = 0; i < n; ++i) out[i] = g.rand_int(lo, hi);
  return 0;
}

int theia_hip_four_point_focal_length_radial_distortion(int32_t num, const double* corr2d3d, const double* limits, const double* rotation_draws,
                                                        double* models, int32_t* num_solutions) {
  // (the six-argument form of rounds 1 - 4 keeps its symbol and its ABI; the solver's pre-filter count is the _ex form's)
  return theia_hip_four_point_focal_length_radial_distortion_ex(num, corr2d3d, limits, rotation_draws, models, num_solutions, nullptr);
}

int theia_hip_four_point_focal_length_radial_distortion_ex(int32_t num, const double* corr2d3d, const double* limits, const double* rotation_draws,
                                                           double* models, int32_t* num_solutions, int32_t* num_solver_solutions) {
  if (num < 0 || !limits || (num > 0 && (!corr2d3d || !models || !num_solutions))) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "bad argument");
  if (!(limits[1] >= 0.0 && limits[0] >= 0.0 && limits[0] >= limits[1] && limits[2] <= 0.0 && limits[3] <= 0.0 && limits[2] <= limits[3]))
    return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "P4Pfr: needs 0 <= min focal length <= max focal length and max distortion <= min distortion <= 0");
  if (num == 0) return 0;
  int rc = ensure_device();
  if (rc || (rc = p4pfr_ensure_tables())) return rc;
  hipStream_t st = solver_stream();
  PoolStreamScope pool_scope(st);   // blocks this call hands back to the caches are tagged with an event on this stream (pools.h)
  if (!st) return set_error(THEIA_HIP_ERR_NO_DEVICE, "could not create the solver stream");
  CallSync mine;
  // one problem of four data per call row, one hypothesis each with the identity sample: the RANSAC stages as they are
  constexpr int kMm = 13, kMd = 14;
  DBuf<double> dc, dws, dmod, drot; DBuf<int> dn, dsamp, dact, ddense, dtags, dbase, dsolver; DBuf<int64_t> doff;
  const size_t n = (size_t)num;
  if ((rc = dsolver.ensure(n))) return rc;
  if ((rc = dc.ensure(n * 20)) || (rc = dws.ensure(n * (size_t)p4pfr_workspace_doubles())) || (rc = dmod.ensure(n * kMm * kStride)) || (rc = drot.ensure(n * 9)) ||
      (rc = dn.ensure(n)) || (rc = dsamp.ensure(n * 4)) || (rc = dact.ensure(n)) || (rc = ddense.ensure(n)) || (rc = dtags.ensure(n * kMm)) ||
      (rc = dbase.ensure(n)) || (rc = doff.ensure(n + 1)))
    return rc;
  std::vector<int64_t> off(n + 1);
  std::vector<int> samp(n * 4), act(n, 1);
  std::vector<double> rot(n * 9);
  for (size_t i = 0; i <= n; ++i) off[i] = (int64_t)(4 * i);
  for (size_t i = 0; i < n * 4; ++i) samp[i] = (int)(i % 4);
  Mt19937 g;
  g.seed(42);   // rotation_draws == NULL: the calls of a fresh process, in order (the solver's static RandomNumberGenerator(42))
  for (size_t i = 0; i < n; ++i) {
    double v[3];
    for (int k = 0; k < 3; ++k) v[k] = rotation_draws ? rotation_draws[3 * i + k] : g.rand_double(-0.5, 0.5);
    p4pfr_rotation_from_draws(v, rot.data() + 9 * i);
  }
  HIP_TRYR(hipMemcpyAsync(dc.p, corr2d3d, sizeof(double) * n * 20, hipMemcpyHostToDevice, st));
  HIP_TRYR(hipMemcpyAsync(doff.p, off.data(), sizeof(int64_t) * (n + 1), hipMemcpyHostToDevice, st));
  HIP_TRYR(hipMemcpyAsync(dsamp.p, samp.data(), sizeof(int) * n * 4, hipMemcpyHostToDevice, st));
  HIP_TRYR(hipMemcpyAsync(dact.p, act.data(), sizeof(int) * n, hipMemcpyHostToDevice, st));
  HIP_TRYR(hipMemcpyAsync(drot.p, rot.data(), sizeof(double) * n * 9, hipMemcpyHostToDevice, st));
  HIP_TRYR(hipMemsetAsync(ddense.p, 0, sizeof(int) * n, st));
  HIP_TRYR(hipMemsetAsync(dmod.p, 0, sizeof(double) * n * kMm * kStride, st));
  launch_p4pfr_fit(num, 1, doff.p, dc.p, dsamp.p, dact.p, drot.p, limits, dws.p, dmod.p, dn.p, ddense.p, dtags.p, dbase.p, st, dsolver.p);
  if (num_solver_solutions) HIP_TRYR(hipMemcpyAsync(num_solver_solutions, dsolver.p, sizeof(int) * n, hipMemcpyDeviceToHost, st));
  std::vector<double> hm(n * kMm * kStride);
  HIP_TRYR(hipMemcpyAsync(hm.data(), dmod.p, sizeof(double) * hm.size(), hipMemcpyDeviceToHost, st));
  HIP_TRYR(hipMemcpyAsync(num_solutions, dn.p, sizeof(int) * n, hipMemcpyDeviceToHost, st));
  HIP_TRYR(hipGetLastError());
  HIP_TRYR(mine.wait(st));
  for (size_t i = 0; i < n; ++i)
    for (int j = 0; j < kMm; ++j)
      for (int k = 0; k < kMd; ++k) models[(i * kMm + j) * kMd + k] = j < num_solutions[i] ? hm[(i * kMm + j) * kStride + k] : 0.0;
  return 0;
}

int theia_hip_pose_from_three_points(int32_t num, const double* corr2d3d, double* rotations, double* translations, int32_t* num_solutions) {
  if (num < 0 || (num > 0 && (!corr2d3d || !rotations || !translations || !num_solutions))) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "bad argument");
  if (num == 0) return 0;
  int rc = ensure_device();
  if (rc) return rc;
  hipStream_t st = solver_stream();
  PoolStreamScope pool_scope(st);   // blocks this call hands back to the caches are tagged with an event on this stream (pools.h)
  if (!st) return set_error(THEIA_HIP_ERR_NO_DEVICE, "could not create the solver stream");
  CallSync mine;
  DBuf<double> dc, dr, dt; DBuf<int> dn;
  if ((rc = dc.ensure((size_t)num * 15)) || (rc = dr.ensure((size_t)num * 36)) || (rc = dt.ensure((size_t)num * 12)) || (rc = dn.ensure(num))) return rc;
  HIP_TRYR(hipMemcpyAsync(dc.p, corr2d3d, sizeof(double) * num * 15, hipMemcpyHostToDevice, st));
  k_p3p<<<(num + 63) / 64, 64, 0, st>>>(num, dc.p, dr.p, dt.p, dn.p);
  HIP_TRYR(hipMemcpyAsync(rotations, dr.p, sizeof(double) * num * 36, hipMemcpyDeviceToHost, st));
  HIP_TRYR(hipMemcpyAsync(translations, dt.p, sizeof(double) * num * 12, hipMemcpyDeviceToHost, st));
  HIP_TRYR(hipMemcpyAsync(num_solutions, dn.p, sizeof(int) * num, hipMemcpyDeviceToHost, st));
  HIP_TRYR(hipGetLastError());
  HIP_TRYR(mine.wait(st));
  return 0;
}

int theia_hip_sqpnp(int32_t num, const int64_t* offsets, const double* features, const double* world_points,
                    double* quaternions, double* translations, int32_t* num_solutions) {
  if (num < 0 || (num > 0 && (!offsets || !features || !world_points || !quaternions || !translations || !num_solutions)))
    return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "bad argument");
  if (num == 0) return 0;
  int rc = thip::ensure_device();
  if (rc) return rc;
  hipStream_t st = solver_stream();
  PoolStreamScope pool_scope(st);   // blocks this call hands back to the caches are tagged with an event on this stream (pools.h)
  if (!st) return set_error(THEIA_HIP_ERR_NO_DEVICE, "could not create the solver stream");
  CallSync mine;
  for (int i = 0; i < num; ++i)
    if (offsets[i + 1] < offsets[i]) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "offsets must be non-decreasing");
  const int64_t total = offsets[num] - offsets[0];
  if (offsets[0] != 0) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "offsets[0] must be 0");
  DBuf<double> df, dw, dq, dt; DBuf<int> dn; DBuf<int64_t> dof;
  if ((rc = df.ensure((size_t)std::max<int64_t>(1, total) * 2)) || (rc = dw.ensure((size_t)std::max<int64_t>(1, total) * 3)) ||
      (rc = dq.ensure((size_t)num * 72)) || (rc = dt.ensure((size_t)num * 54)) || (rc = dn.ensure(num)) || (rc = dof.ensure(num + 1)))
    return rc;
  if (total) {
    HIP_TRYR(hipMemcpyAsync(df.p, features, sizeof(double) * total * 2, hipMemcpyHostToDevice, st));
    HIP_TRYR(hipMemcpyAsync(dw.p, world_points, sizeof(double) * total * 3, hipMemcpyHostToDevice, st));
  }
  HIP_TRYR(hipMemcpyAsync(dof.p, offsets, sizeof(int64_t) * (num + 1), hipMemcpyHostToDevice, st));
  k_sqpnp<<<(num + 63) / 64, 64, 0, st>>>(num, dof.p, df.p, dw.p, dq.p, dt.p, dn.p);
  HIP_TRYR(hipMemcpyAsync(quaternions, dq.p, sizeof(double) * num * 72, hipMemcpyDeviceToHost, st));
  HIP_TRYR(hipMemcpyAsync(translations, dt.p, sizeof(double) * num * 54, hipMemcpyDeviceToHost, st));
  HIP_TRYR(hipMemcpyAsync(num_solutions, dn.p, sizeof(int) * num, hipMemcpyDeviceToHost, st));
  HIP_TRYR(hipGetLastError());
  HIP_TRYR(mine.wait(st));
  return 0;
}

void theia_hip_dls_macaulay_terms(int64_t first_call, int64_t num_calls, double* out) {
  if (first_call < 0 || num_calls <= 0 || !out) return;
  std::vector<double> u; dls::GlibcRand gen;
  dls_terms(u, gen, (size_t)(first_call + num_calls));
  std::memcpy(out, u.data() + 4 * first_call, sizeof(double) * 4 * num_calls);
}

int theia_hip_dls_pnp(int32_t num, const int64_t* offsets, const double* features, const double* world_points,
                      const int64_t* call_index, double* quaternions, double* translations, int32_t* num_solutions) {
  if (num < 0 || (num > 0 && (!offsets || !features || !world_points || !quaternions || !translations || !num_solutions)))
    return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "bad argument");
  if (num == 0) return 0;
  int rc = thip::ensure_device();
  if (rc || (rc = dls_ensure_tables())) return rc;
  hipStream_t st = solver_stream();
  PoolStreamScope pool_scope(st);   // blocks this call hands back to the caches are tagged with an event on this stream (pools.h)
  if (!st) return set_error(THEIA_HIP_ERR_NO_DEVICE, "could not create the solver stream");
  CallSync mine;
  if (offsets[0] != 0) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "offsets[0] must be 0");
  int64_t max_call = num - 1;
  for (int i = 0; i < num; ++i) {
    if (offsets[i + 1] < offsets[i]) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "offsets must be non-decreasing");
    if (call_index) {
      if (call_index[i] < 0 || call_index[i] > (1 << 26)) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "call_index out of range");
      max_call = std::max(max_call, call_index[i]);
    }
  }
  const int64_t total = offsets[num];
  std::vector<double> uall, u((size_t)num * 4); dls::GlibcRand gen;
  dls_terms(uall, gen, (size_t)max_call + 1);
  for (int i = 0; i < num; ++i) std::memcpy(&u[(size_t)4 * i], &uall[(size_t)4 * (call_index ? call_index[i] : i)], 4 * sizeof(double));
  constexpr int NS = dlsdev::kMaxSolutions;
  DBuf<double> df, dw, dq, dt, du, da, dtf; DBuf<int> dn, dok; DBuf<int64_t> dof;
  if ((rc = df.ensure((size_t)std::max<int64_t>(1, total) * 2)) || (rc = dw.ensure((size_t)std::max<int64_t>(1, total) * 3)) ||
      (rc = dq.ensure((size_t)num * 4 * NS)) || (rc = dt.ensure((size_t)num * 3 * NS)) || (rc = dn.ensure(num)) || (rc = dof.ensure(num + 1)) ||
      (rc = du.ensure((size_t)num * 4)) || (rc = da.ensure((size_t)num * 729)) || (rc = dtf.ensure((size_t)num * 27)) || (rc = dok.ensure(num)))
    return rc;
  if (total) {
    HIP_TRYR(hipMemcpyAsync(df.p, features, sizeof(double) * total * 2, hipMemcpyHostToDevice, st));
    HIP_TRYR(hipMemcpyAsync(dw.p, world_points, sizeof(double) * total * 3, hipMemcpyHostToDevice, st));
  }
  HIP_TRYR(hipMemcpyAsync(dof.p, offsets, sizeof(int64_t) * (num + 1), hipMemcpyHostToDevice, st));
  HIP_TRYR(hipMemcpyAsync(du.p, u.data(), sizeof(double) * num * 4, hipMemcpyHostToDevice, st));
  launch_dls_solve_a(num, dof.p, df.p, dw.p, du.p, da.p, dtf.p, dok.p, st);
  k_dls_solve_b<<<(num + 63) / 64, 64, 0, st>>>(num, dof.p, dw.p, da.p, dtf.p, dok.p, dq.p, dt.p, dn.p);
  HIP_TRYR(hipGetLastError());
  HIP_TRYR(hipMemcpyAsync(quaternions, dq.p, sizeof(double) * num * 4 * NS, hipMemcpyDeviceToHost, st));
  HIP_TRYR(hipMemcpyAsync(translations, dt.p, sizeof(double) * num * 3 * NS, hipMemcpyDeviceToHost, st));
  HIP_TRYR(hipMemcpyAsync(num_solutions, dn.p, sizeof(int) * num, hipMemcpyDeviceToHost, st));
  HIP_TRYR(hipGetLastError());
  HIP_TRYR(mine.wait(st));
  return 0;
}

void theia_hip_release_scratch(void) { dev_pool().release(); host_pool().release(); }

}  // extern "C"

#ifdef THIP_EIG_STAMPS
// development: the phase stamps of eig_team (eig_team.h) summed since the last call; out[8]
extern "C" int theia_hip_debug_eig_stamps(unsigned long long* out) {
  unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(thip::rsc::g_eig_stamps), sizeof(zero)) != hipSuccess) return -1;
  (void)hipMemcpyToSymbol(HIP_SYMBOL(thip::rsc::g_eig_stamps), zero, sizeof(zero));
  return 0;
}
#endif
