#include "bundle_adjuster_hip.h"

#include <chrono>
#include <cmath>
#include <cstring>

namespace theia_hip_shim {

void BundleAdjuster::AddIntrinsicsGroup(CameraIntrinsicsGroupId group, int camera_model, double* parameters, int num_parameters) {
  if (group_index_.count(group)) return;
  group_index_[group] = (int)group_ptr_.size();
  group_ptr_.push_back(parameters); group_model_.push_back(camera_model); group_k_.push_back(num_parameters);
}

void BundleAdjuster::AddCamera(ViewId view, double* extrinsics, CameraIntrinsicsGroupId group) {
  if (view_index_.count(view)) return;
  view_index_[view] = (int)cam_ptr_.size();
  cam_ptr_.push_back(extrinsics); cam_group_id_.push_back((int32_t)group); cam_const_.push_back(0);
}

void BundleAdjuster::AddTrack(TrackId track, double* point) {
  if (track_index_.count(track)) return;
  track_index_[track] = (int)pt_ptr_.size();
  pt_ptr_.push_back(point); point_const_.push_back(0);
}

void BundleAdjuster::AddObservation(ViewId view, TrackId track, const double feature[2], const double covariance_diag[2]) {
  obs_cam_.push_back(view_index_.at(view)); obs_pt_.push_back(track_index_.at(track));
  obs_uv_.push_back(feature[0]); obs_uv_.push_back(feature[1]);
  // reprojection_error.h:94-99: sqrt information = 1 / sqrt(covariance)
  obs_sqrt_info_.push_back(1.0 / std::sqrt(covariance_diag[0])); obs_sqrt_info_.push_back(1.0 / std::sqrt(covariance_diag[1]));
}

void BundleAdjuster::SetCameraExtrinsicsConstant(ViewId v) { cam_const_[view_index_.at(v)] |= THEIA_CAM_CONST_ALL; }
void BundleAdjuster::SetCameraPositionConstant(ViewId v) { cam_const_[view_index_.at(v)] |= THEIA_CAM_CONST_POSITION; }
void BundleAdjuster::SetCameraOrientationConstant(ViewId v) { cam_const_[view_index_.at(v)] |= THEIA_CAM_CONST_ORIENTATION; }
void BundleAdjuster::SetTrackConstant(TrackId t) { point_const_[track_index_.at(t)] = 1; }
void BundleAdjuster::SetTrackVariable(TrackId t) { point_const_[track_index_.at(t)] = 0; }

BundleAdjustmentSummary BundleAdjuster::Optimize() {
  const auto t0 = std::chrono::steady_clock::now();
  BundleAdjustmentSummary summary;
  const int nc = (int)cam_ptr_.size(), np = (int)pt_ptr_.size(), ng = (int)group_ptr_.size();
  // gather the blocks (the copies Ceres' Program::ParameterBlocksToStateVector would make)
  std::vector<double> cam(6 * (size_t)nc), pts(4 * (size_t)np), intr((size_t)THEIA_MAX_INTRINSICS * ng, 0.0);
  std::vector<int32_t> cam_group(nc);
  for (int c = 0; c < nc; ++c) {
    std::memcpy(&cam[6 * (size_t)c], cam_ptr_[c], 6 * sizeof(double));
    const auto it = group_index_.find((CameraIntrinsicsGroupId)cam_group_id_[c]);
    if (it == group_index_.end()) { error_ = "camera without a registered intrinsics group"; return summary; }
    cam_group[c] = it->second;
  }
  for (int p = 0; p < np; ++p) std::memcpy(&pts[4 * (size_t)p], pt_ptr_[p], 4 * sizeof(double));
  for (int g = 0; g < ng; ++g) std::memcpy(&intr[(size_t)THEIA_MAX_INTRINSICS * g], group_ptr_[g], group_k_[g] * sizeof(double));

  theia_ba_problem P;
  std::memset(&P, 0, sizeof(P));
  P.num_cameras = nc; P.num_groups = ng; P.num_points = np; P.num_obs = (int64_t)obs_cam_.size();
  P.cam_ext = cam.data(); P.intrinsics = intr.data(); P.group_model = group_model_.data(); P.cam_group = cam_group.data();
  P.cam_const = cam_const_.data(); P.points = pts.data(); P.point_const = point_const_.data();
  P.obs_uv = obs_uv_.data(); P.obs_sqrt_info = obs_sqrt_info_.data(); P.obs_cam = obs_cam_.data(); P.obs_pt = obs_pt_.data();

  theia_ba_options O;
  theia_ba_options_default(&O);
  O.loss_function_type = options_.loss_function_type; O.robust_loss_width = options_.robust_loss_width;
  O.max_num_iterations = options_.max_num_iterations; O.max_solver_time_in_seconds = options_.max_solver_time_in_seconds;
  O.use_inner_iterations = options_.use_inner_iterations; O.verbose = options_.verbose;
  O.use_homogeneous_point_parametrization = options_.use_homogeneous_point_parametrization;
  O.constant_camera_orientation = options_.constant_camera_orientation; O.constant_camera_position = options_.constant_camera_position;
  O.intrinsics_to_optimize = options_.intrinsics_to_optimize;
  O.function_tolerance = options_.function_tolerance; O.gradient_tolerance = options_.gradient_tolerance;
  O.parameter_tolerance = options_.parameter_tolerance; O.max_trust_region_radius = options_.max_trust_region_radius;

  theia_ba_summary S;
  std::memset(&S, 0, sizeof(S));
  const int rc = theia_hip_ba_solve(&P, &O, &S);
  if (rc != THEIA_HIP_OK) { error_ = theia_hip_last_error(); return summary; }   // the reference CHECKs / falls back to Ceres here
  // scatter back through the pointers Ceres would have written to
  for (int c = 0; c < nc; ++c) std::memcpy(cam_ptr_[c], &cam[6 * (size_t)c], 6 * sizeof(double));
  for (int p = 0; p < np; ++p) std::memcpy(pt_ptr_[p], &pts[4 * (size_t)p], 4 * sizeof(double));
  for (int g = 0; g < ng; ++g) std::memcpy(group_ptr_[g], &intr[(size_t)THEIA_MAX_INTRINSICS * g], group_k_[g] * sizeof(double));
  summary.success = S.success != 0;
  summary.initial_cost = S.initial_cost; summary.final_cost = S.final_cost;
  summary.solve_time_in_seconds = S.solve_time_in_seconds;
  summary.setup_time_in_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() - S.solve_time_in_seconds;
  return summary;
}

bool EstimateRelativePoseBatch(const RansacParameters& params, const std::vector<std::vector<double>>& correspondences,
                               std::vector<bool>* success, std::vector<RelativePose>* poses, std::vector<RansacSummary>* summaries,
                               std::string* error) {
  const int np = (int)correspondences.size();
  std::vector<int64_t> offsets(np + 1, 0);
  for (int p = 0; p < np; ++p) offsets[p + 1] = offsets[p] + (int64_t)correspondences[p].size() / 4;
  std::vector<double> data;
  data.reserve(4 * (size_t)offsets[np]);
  for (const auto& c : correspondences) data.insert(data.end(), c.begin(), c.end());
  theia_ransac_params prm;
  theia_ransac_params_default(&prm);
  prm.error_thresh = params.error_thresh; prm.failure_probability = params.failure_probability;
  prm.min_inlier_ratio = params.min_inlier_ratio; prm.min_iterations = params.min_iterations; prm.max_iterations = params.max_iterations;
  prm.use_mle = params.use_mle; prm.use_lo = params.use_lo; prm.lo_start_iterations = params.lo_start_iterations; prm.seed = params.seed;
  theia_ransac_batch batch;
  std::memset(&batch, 0, sizeof(batch));
  batch.estimator = THEIA_EST_RELATIVE_POSE; batch.num_problems = np; batch.offsets = offsets.data(); batch.data = data.data();
  std::vector<int32_t> ok(np), ninl(np), nit(np);
  std::vector<double> models((size_t)np * THEIA_RANSAC_MODEL_STRIDE), conf(np);
  std::vector<uint8_t> mask((size_t)offsets[np]);
  theia_ransac_result res;
  std::memset(&res, 0, sizeof(res));
  res.success = ok.data(); res.models = models.data(); res.num_inliers = ninl.data(); res.inlier_mask = mask.data();
  res.num_iterations = nit.data(); res.confidence = conf.data();
  if (theia_hip_ransac_estimate_batch(&batch, &prm, &res) != THEIA_HIP_OK) { if (error) *error = theia_hip_last_error(); return false; }
  success->assign(np, false); poses->resize(np); summaries->assign(np, RansacSummary());
  for (int p = 0; p < np; ++p) {
    (*success)[p] = ok[p] != 0;
    const double* m = &models[(size_t)p * THEIA_RANSAC_MODEL_STRIDE];   // E (9) | R (9) | position (3)
    std::memcpy((*poses)[p].essential_matrix, m, 9 * sizeof(double));
    std::memcpy((*poses)[p].rotation, m + 9, 9 * sizeof(double));
    std::memcpy((*poses)[p].position, m + 18, 3 * sizeof(double));
    RansacSummary& s = (*summaries)[p];
    s.num_iterations = nit[p]; s.confidence = conf[p];
    for (int64_t i = offsets[p]; i < offsets[p + 1]; ++i) if (mask[(size_t)i]) s.inliers.push_back((int)(i - offsets[p]));
  }
  return true;
}

}  // namespace theia_hip_shim
