// C++ host side of the drop-in boundary: what a maintainer puts behind theia::BundleAdjuster and the RANSAC estimators
// when the back-end is libtheia_hip.so (INTEGRATION.md sections 1 and 2, compiled and exercised by shim/shim_test.cc).
//
// BundleAdjuster keeps the reference's shape (sfm/bundle_adjustment/bundle_adjuster.h:60-200): blocks are registered by
// the POINTERS Ceres would have been given -- Camera::mutable_extrinsics() (6 doubles), the shared
// CameraIntrinsicsModel::mutable_parameters() of an intrinsics group, Track::MutablePoint() (4 doubles) -- residual
// blocks by (view, track, feature); constancy by the same calls (SetCameraExtrinsicsConstant, SetTrackConstant, ...).
// Optimize() flattens to the C-ABI arrays, calls theia_hip_ba_solve and scatters the result back through the registered
// pointers, so the caller's Reconstruction is updated exactly where Ceres would have updated it.  No Theia / Eigen /
// Ceres headers: ids are plain integers.
#ifndef THEIA_HIP_SHIM_BUNDLE_ADJUSTER_HIP_H_
#define THEIA_HIP_SHIM_BUNDLE_ADJUSTER_HIP_H_

#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>

#include "theia_hip.h"

namespace theia_hip_shim {

typedef uint32_t ViewId;
typedef uint32_t TrackId;
typedef uint32_t CameraIntrinsicsGroupId;

// bundle_adjustment.h:87-167 (the fields the C-ABI carries; the linear-solver selectors have no meaning here)
struct BundleAdjustmentOptions {
  int loss_function_type = THEIA_LOSS_TRIVIAL;
  double robust_loss_width = 2.0;
  int max_num_iterations = 100;
  double max_solver_time_in_seconds = 3600.0;
  bool use_inner_iterations = true;
  bool use_homogeneous_point_parametrization = true;
  bool constant_camera_orientation = false;
  bool constant_camera_position = false;
  int intrinsics_to_optimize = THEIA_INTR_NONE;
  double function_tolerance = 1e-6, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;
  double max_trust_region_radius = 1e12;
  bool verbose = false;
};

struct BundleAdjustmentSummary {   // bundle_adjustment.h:170-178
  bool success = false;
  double initial_cost = 0.0, final_cost = 0.0;
  double setup_time_in_seconds = 0.0, solve_time_in_seconds = 0.0;
};

class BundleAdjuster {
 public:
  explicit BundleAdjuster(const BundleAdjustmentOptions& options) : options_(options) {}

  // BundleAdjuster::AddView (bundle_adjuster.cc:116-173), split into its three registrations
  void AddCamera(ViewId view, double* extrinsics /* [6] position | angle axis */, CameraIntrinsicsGroupId group);
  void AddIntrinsicsGroup(CameraIntrinsicsGroupId group, int camera_model /* THEIA_CAM_* */, double* parameters /* [K] */, int num_parameters);
  // AddTrack (:175-221): the homogeneous point of the track
  void AddTrack(TrackId track, double* point /* [4] */);
  // AddReprojectionErrorResidual (:579-592): feature position and the diagonal of Feature::covariance_
  void AddObservation(ViewId view, TrackId track, const double feature[2], const double covariance_diag[2]);

  void SetCameraExtrinsicsConstant(ViewId view);   // :477-481
  void SetCameraPositionConstant(ViewId view);     // :483-493
  void SetCameraOrientationConstant(ViewId view);  // :495-505
  void SetTrackConstant(TrackId track);            // :520-527
  void SetTrackVariable(TrackId track);            // :529-536

  BundleAdjustmentSummary Optimize();              // :315-355
  const std::string& error() const { return error_; }

 private:
  BundleAdjustmentOptions options_;
  std::unordered_map<ViewId, int> view_index_;
  std::unordered_map<TrackId, int> track_index_;
  std::unordered_map<CameraIntrinsicsGroupId, int> group_index_;
  std::vector<double*> cam_ptr_, pt_ptr_, group_ptr_;
  std::vector<int32_t> cam_group_id_, group_model_, group_k_, obs_cam_, obs_pt_;
  std::vector<uint8_t> cam_const_, point_const_;
  std::vector<double> obs_uv_, obs_sqrt_info_;
  std::string error_;
};

// SampleConsensusEstimator front ends (sfm/estimators/estimate_relative_pose.h:49-70, estimate_calibrated_absolute_pose.h):
// one call per image pair list, all pairs as one device batch.
struct RansacParameters {   // solvers/sample_consensus_estimator.h:58-126
  double error_thresh = -1.0, failure_probability = 0.01, min_inlier_ratio = 0.0;
  int min_iterations = 100, max_iterations = 2147483647;
  bool use_mle = false, use_lo = false;
  int lo_start_iterations = 50;
  uint32_t seed = 0;
};
struct RansacSummary { std::vector<int> inliers; int num_iterations = 0; double confidence = 0.0; };
struct RelativePose { double essential_matrix[9], rotation[9], position[3]; };

// correspondences[p] = {x1, y1, x2, y2} x n_p (normalised image coordinates).  Returns false on an argument error.
bool EstimateRelativePoseBatch(const RansacParameters& params, const std::vector<std::vector<double>>& correspondences,
                               std::vector<bool>* success, std::vector<RelativePose>* poses, std::vector<RansacSummary>* summaries,
                               std::string* error);

}  // namespace theia_hip_shim
#endif
