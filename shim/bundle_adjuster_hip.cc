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

namespace {
void to_c_options(const BundleAdjustmentOptions& o, theia_ba_options* O) {
  theia_ba_options_default(O);
  O->loss_function_type = o.loss_function_type; O->robust_loss_width = o.robust_loss_width;
  O->max_num_iterations = o.max_num_iterations; O->max_solver_time_in_seconds = o.max_solver_time_in_seconds;
  O->use_inner_iterations = o.use_inner_iterations; O->verbose = o.verbose;
  O->use_homogeneous_point_parametrization = o.use_homogeneous_point_parametrization;
  O->constant_camera_orientation = o.constant_camera_orientation; O->constant_camera_position = o.constant_camera_position;
  O->intrinsics_to_optimize = o.intrinsics_to_optimize;
  O->function_tolerance = o.function_tolerance; O->gradient_tolerance = o.gradient_tolerance;
  O->parameter_tolerance = o.parameter_tolerance; O->max_trust_region_radius = o.max_trust_region_radius;
}
int datum_doubles(int estimator) {
  switch (estimator) {
    case THEIA_EST_ABSOLUTE_POSE_KNEIP: case THEIA_EST_ABSOLUTE_POSE_DLS: case THEIA_EST_ABSOLUTE_POSE_SQPNP:
    case THEIA_EST_ABSOLUTE_POSE_KNOWN_ORIENTATION: case THEIA_EST_UNCALIBRATED_ABSOLUTE_POSE:
    case THEIA_EST_RADIAL_DIST_UNCALIBRATED_ABSOLUTE_POSE: return 5;
    case THEIA_EST_DOMINANT_PLANE: return 3;
    case THEIA_EST_TRIANGULATION: return 33;
    case THEIA_EST_RADIAL_HOMOGRAPHY: return 12;
    case THEIA_EST_SIMILARITY_2D3D: case THEIA_EST_RIGID_TRANSFORMATION_2D3D: return 26;
    default: return 4;
  }
}
}  // namespace

// ceres::Covariance of the block-diagonal problems behind the *WithCov entry points: the state the registered pointers hold
// now (i.e. after Optimize()), through a device handle (theia_hip_ba_create + theia_hip_ba_covariance)
bool BundleAdjuster::covariances(std::vector<double>* point_cov, std::vector<double>* cam_cov) {
  const int nc = (int)cam_ptr_.size(), np = (int)pt_ptr_.size(), ng = (int)group_ptr_.size();
  std::vector<double> cam(6 * (size_t)nc), pts(4 * (size_t)np), intr((size_t)THEIA_MAX_INTRINSICS * ng, 0.0);
  std::vector<int32_t> cam_group(nc);
  for (int c = 0; c < nc; ++c) {
    std::memcpy(&cam[6 * (size_t)c], cam_ptr_[c], 6 * sizeof(double));
    const auto it = group_index_.find((CameraIntrinsicsGroupId)cam_group_id_[c]);
    if (it == group_index_.end()) { error_ = "camera without a registered intrinsics group"; return false; }
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
  to_c_options(options_, &O);
  theia_ba_handle h = nullptr;
  if (theia_hip_ba_create(&P, &O, &h) != THEIA_HIP_OK) { error_ = theia_hip_last_error(); return false; }
  const int d = options_.use_homogeneous_point_parametrization ? 3 : 4;
  if (point_cov) point_cov->assign((size_t)np * d * d, 0.0);
  if (cam_cov) cam_cov->assign((size_t)nc * 36, 0.0);
  const int rc = theia_hip_ba_covariance(h, point_cov ? point_cov->data() : nullptr, cam_cov ? cam_cov->data() : nullptr);
  if (rc != THEIA_HIP_OK) error_ = theia_hip_last_error();
  theia_hip_ba_destroy(h);
  return rc == THEIA_HIP_OK;
}

bool BundleAdjuster::GetCovarianceForTracks(const std::vector<TrackId>& tracks, std::vector<std::vector<double>>* covariances_out) {
  std::vector<double> pc;
  if (!covariances(&pc, nullptr)) return false;
  const int d = options_.use_homogeneous_point_parametrization ? 3 : 4;
  covariances_out->clear();
  for (TrackId t : tracks) {
    const auto it = track_index_.find(t);
    if (it == track_index_.end()) { error_ = "unknown track"; return false; }
    covariances_out->emplace_back(pc.begin() + (size_t)it->second * d * d, pc.begin() + (size_t)(it->second + 1) * d * d);
  }
  return true;
}

bool BundleAdjuster::GetCovarianceForViews(const std::vector<ViewId>& views, std::vector<std::vector<double>>* covariances_out) {
  std::vector<double> cc;
  if (!covariances(nullptr, &cc)) return false;
  covariances_out->clear();
  for (ViewId v : views) {
    const auto it = view_index_.find(v);
    if (it == view_index_.end()) { error_ = "unknown view"; return false; }
    covariances_out->emplace_back(cc.begin() + (size_t)it->second * 36, cc.begin() + (size_t)(it->second + 1) * 36);
  }
  return true;
}

bool BundleAdjustViews(const BundleAdjustmentOptions& options, std::vector<ViewProblem>* views,
                       std::vector<BundleAdjustmentSummary>* summaries, std::string* error) {
  const int n = (int)views->size();
  std::vector<int64_t> offsets(n + 1, 0);
  for (int i = 0; i < n; ++i) offsets[i + 1] = offsets[i] + (int64_t)(*views)[i].features.size() / 2;
  std::vector<double> uv, pts, cam(6 * (size_t)n), intr((size_t)THEIA_MAX_INTRINSICS * n, 0.0);
  std::vector<int32_t> model(n);
  for (int i = 0; i < n; ++i) {
    const ViewProblem& v = (*views)[i];
    uv.insert(uv.end(), v.features.begin(), v.features.end());
    pts.insert(pts.end(), v.points.begin(), v.points.end());
    std::memcpy(&cam[6 * (size_t)i], v.extrinsics, 6 * sizeof(double));
    std::memcpy(&intr[(size_t)THEIA_MAX_INTRINSICS * i], v.intrinsics, v.num_intrinsics * sizeof(double));
    model[i] = v.camera_model;
  }
  theia_ba_view_batch B;
  std::memset(&B, 0, sizeof(B));
  B.num_problems = n; B.offsets = offsets.data(); B.obs_uv = uv.data(); B.points = pts.data(); B.cam_ext = cam.data();
  B.intrinsics = intr.data(); B.model = model.data();
  theia_ba_options O;
  to_c_options(options, &O);
  O.use_inner_iterations = 0;      // bundle_adjustment.cc:225: BundleAdjustView switches them off
  std::vector<theia_ba_summary> S(n);
  std::memset(S.data(), 0, sizeof(theia_ba_summary) * n);
  if (theia_hip_ba_views_batch(&B, &O, S.data()) != THEIA_HIP_OK) { if (error) *error = theia_hip_last_error(); return false; }
  summaries->assign(n, BundleAdjustmentSummary());
  for (int i = 0; i < n; ++i) {
    std::memcpy((*views)[i].extrinsics, &cam[6 * (size_t)i], 6 * sizeof(double));
    (*summaries)[i].success = S[i].success != 0; (*summaries)[i].initial_cost = S[i].initial_cost; (*summaries)[i].final_cost = S[i].final_cost;
  }
  return true;
}

bool EstimateBatch(int estimator, int ransac_type, const RansacParameters& params, const std::vector<std::vector<double>>& data_in,
                   const double* estimator_params, EstimatorBatchResult* result, std::string* error) {
  const int np = (int)data_in.size(), ds = datum_doubles(estimator);
  std::vector<int64_t> offsets(np + 1, 0);
  for (int p = 0; p < np; ++p) {
    if (data_in[p].size() % ds) { if (error) *error = "datum size does not divide the data of a problem"; return false; }
    offsets[p + 1] = offsets[p] + (int64_t)data_in[p].size() / ds;
  }
  std::vector<double> data;
  data.reserve((size_t)ds * offsets[np]);
  for (const auto& c : data_in) data.insert(data.end(), c.begin(), c.end());
  theia_ransac_params prm;
  theia_ransac_params_default(&prm);
  prm.error_thresh = params.error_thresh; prm.failure_probability = params.failure_probability;
  prm.min_inlier_ratio = params.min_inlier_ratio; prm.min_iterations = params.min_iterations; prm.max_iterations = params.max_iterations;
  prm.use_mle = params.use_mle; prm.use_lo = params.use_lo; prm.lo_start_iterations = params.lo_start_iterations; prm.seed = params.seed;
  prm.ransac_type = ransac_type;
  theia_ransac_batch batch;
  std::memset(&batch, 0, sizeof(batch));
  batch.estimator = estimator; batch.num_problems = np; batch.offsets = offsets.data(); batch.data = data.data();
  batch.estimator_params = estimator_params;
  std::vector<int32_t> ok(np), ninl(np), nit(np);
  std::vector<double> models((size_t)np * THEIA_RANSAC_MODEL_STRIDE), conf(np);
  std::vector<uint8_t> mask((size_t)offsets[np]);
  theia_ransac_result res;
  std::memset(&res, 0, sizeof(res));
  res.success = ok.data(); res.models = models.data(); res.num_inliers = ninl.data(); res.inlier_mask = mask.data();
  res.num_iterations = nit.data(); res.confidence = conf.data();
  if (theia_hip_ransac_estimate_batch(&batch, &prm, &res) != THEIA_HIP_OK) { if (error) *error = theia_hip_last_error(); return false; }
  result->success.assign(np, false); result->models.assign(np, std::vector<double>()); result->summaries.assign(np, RansacSummary());
  for (int p = 0; p < np; ++p) {
    result->success[p] = ok[p] != 0;
    result->models[p].assign(models.begin() + (size_t)p * THEIA_RANSAC_MODEL_STRIDE, models.begin() + (size_t)(p + 1) * THEIA_RANSAC_MODEL_STRIDE);
    RansacSummary& s = result->summaries[p];
    s.num_iterations = nit[p]; s.confidence = conf[p];
    for (int64_t i = offsets[p]; i < offsets[p + 1]; ++i) if (mask[(size_t)i]) s.inliers.push_back((int)(i - offsets[p]));
  }
  return true;
}

bool EstimateCalibratedAbsolutePoseBatch(const RansacParameters& params, int ransac_type, PnPType pnp_type,
                                         const std::vector<std::vector<double>>& correspondences_2d_3d, std::vector<bool>* success,
                                         std::vector<CalibratedAbsolutePose>* poses, std::vector<RansacSummary>* summaries, std::string* error) {
  const int est = pnp_type == PnPType::KNEIP ? THEIA_EST_ABSOLUTE_POSE_KNEIP : (pnp_type == PnPType::DLS ? THEIA_EST_ABSOLUTE_POSE_DLS : THEIA_EST_ABSOLUTE_POSE_SQPNP);
  EstimatorBatchResult r;
  if (!EstimateBatch(est, ransac_type, params, correspondences_2d_3d, nullptr, &r, error)) return false;
  *success = r.success; *summaries = r.summaries;
  poses->resize(r.models.size());
  for (size_t p = 0; p < r.models.size(); ++p) {
    std::memcpy((*poses)[p].rotation, r.models[p].data(), 9 * sizeof(double));
    std::memcpy((*poses)[p].position, r.models[p].data() + 9, 3 * sizeof(double));
  }
  return true;
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
