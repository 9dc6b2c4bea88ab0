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
  // GetCovarianceForTracks (:705-741) after Optimize() with every camera constant: covariance[i] is the 3 x 3 tangent-space
  // block of tracks[i] (row-major; SphereManifold<4>), as ceres::Covariance returns it.  NOT multiplied by the empirical
  // variance factor (bundle_adjustment.cc:312-320 does that bookkeeping).  False + error() on failure.
  bool GetCovarianceForTracks(const std::vector<TrackId>& tracks, std::vector<std::vector<double>>* covariances);
  // GetCovarianceForViews (:743-773) with every track constant: the 6 x 6 block of each view's extrinsics.
  bool GetCovarianceForViews(const std::vector<ViewId>& views, std::vector<std::vector<double>>* covariances);
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
  bool covariances(std::vector<double>* point_cov, std::vector<double>* cam_cov);
};

// bundle_adjustment.cc:220-260 / :262-285: N x BundleAdjustView (one camera against constant points) and BundleAdjustTrack for
// every track of a flat problem, each as ONE device launch (theia_hip_ba_views_batch / theia_hip_ba_tracks_batch).
struct ViewProblem {                     // one BundleAdjustView call
  double* extrinsics;                    // [6] in / out
  const double* intrinsics; int num_intrinsics; int camera_model;
  std::vector<double> features;          // [n][2] pixels
  std::vector<double> points;            // [n][4] the tracks' homogeneous points (constant)
};
bool BundleAdjustViews(const BundleAdjustmentOptions& options, std::vector<ViewProblem>* views,
                       std::vector<BundleAdjustmentSummary>* summaries, std::string* error);

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


// Every other SampleConsensusEstimator front end of sfm/estimators/ goes through one routine: `estimator` = THEIA_EST_*,
// data[p] = the flattened data of problem p (datum layouts: include/theia_hip.h), models[p] = THEIA_RANSAC_MODEL_STRIDE
// doubles in the layout of that estimator.  ransac_type = THEIA_RANSAC_* (create_and_initialize_ransac_variant.h:52).
struct EstimatorBatchResult {
  std::vector<bool> success;
  std::vector<std::vector<double>> models;
  std::vector<RansacSummary> summaries;
};
bool EstimateBatch(int estimator, int ransac_type, const RansacParameters& params, const std::vector<std::vector<double>>& data,
                   const double* estimator_params, EstimatorBatchResult* result, std::string* error);

// The reference's names over it (one call = all problems as one device batch).  correspondences: [x1 y1 x2 y2] x n;
// correspondences_2d_3d: [u v X Y Z] x n.
enum class PnPType { KNEIP = 0, DLS = 1, SQPnP = 2 };            // estimate_calibrated_absolute_pose.h:52
struct CalibratedAbsolutePose { double rotation[9], position[3]; };
bool EstimateCalibratedAbsolutePoseBatch(const RansacParameters& params, int ransac_type, PnPType pnp_type,
                                         const std::vector<std::vector<double>>& correspondences_2d_3d, std::vector<bool>* success,
                                         std::vector<CalibratedAbsolutePose>* poses, std::vector<RansacSummary>* summaries, std::string* error);
inline bool EstimateEssentialMatrixBatch(const RansacParameters& p, int t, const std::vector<std::vector<double>>& c, EstimatorBatchResult* r, std::string* e) { return EstimateBatch(THEIA_EST_ESSENTIAL_MATRIX, t, p, c, nullptr, r, e); }
inline bool EstimateFundamentalMatrixBatch(const RansacParameters& p, int t, const std::vector<std::vector<double>>& c, EstimatorBatchResult* r, std::string* e) { return EstimateBatch(THEIA_EST_FUNDAMENTAL_MATRIX, t, p, c, nullptr, r, e); }
inline bool EstimateHomographyBatch(const RansacParameters& p, int t, const std::vector<std::vector<double>>& c, EstimatorBatchResult* r, std::string* e) { return EstimateBatch(THEIA_EST_HOMOGRAPHY, t, p, c, nullptr, r, e); }
inline bool EstimateDominantPlaneFromPointsBatch(const RansacParameters& p, int t, const std::vector<std::vector<double>>& pts, EstimatorBatchResult* r, std::string* e) { return EstimateBatch(THEIA_EST_DOMINANT_PLANE, t, p, pts, nullptr, r, e); }
inline bool EstimateRelativePoseWithKnownOrientationBatch(const RansacParameters& p, int t, const std::vector<std::vector<double>>& c, EstimatorBatchResult* r, std::string* e) { return EstimateBatch(THEIA_EST_RELATIVE_POSE_KNOWN_ORIENTATION, t, p, c, nullptr, r, e); }
inline bool EstimateAbsolutePoseWithKnownOrientationBatch(const RansacParameters& p, int t, const std::vector<std::vector<double>>& c, EstimatorBatchResult* r, std::string* e) { return EstimateBatch(THEIA_EST_ABSOLUTE_POSE_KNOWN_ORIENTATION, t, p, c, nullptr, r, e); }
inline bool EstimateUncalibratedRelativePoseBatch(const RansacParameters& p, int t, const double min_max_focal_length[2], const std::vector<std::vector<double>>& c, EstimatorBatchResult* r, std::string* e) { return EstimateBatch(THEIA_EST_UNCALIBRATED_RELATIVE_POSE, t, p, c, min_max_focal_length, r, e); }
inline bool EstimateUncalibratedAbsolutePoseBatch(const RansacParameters& p, int t, const std::vector<std::vector<double>>& c, EstimatorBatchResult* r, std::string* e) { return EstimateBatch(THEIA_EST_UNCALIBRATED_ABSOLUTE_POSE, t, p, c, nullptr, r, e); }
inline bool EstimateTriangulationBatch(const RansacParameters& p, int t, const std::vector<std::vector<double>>& obs33, EstimatorBatchResult* r, std::string* e) { return EstimateBatch(THEIA_EST_TRIANGULATION, t, p, obs33, nullptr, r, e); }
inline bool EstimateRadialHomographyMatrixBatch(const RansacParameters& p, int t, const std::vector<std::vector<double>>& c12, EstimatorBatchResult* r, std::string* e) { return EstimateBatch(THEIA_EST_RADIAL_HOMOGRAPHY, t, p, c12, nullptr, r, e); }
inline bool EstimateSimilarityTransformation2D3DBatch(const RansacParameters& p, int t, const std::vector<std::vector<double>>& c26, EstimatorBatchResult* r, std::string* e) { return EstimateBatch(THEIA_EST_SIMILARITY_2D3D, t, p, c26, nullptr, r, e); }

inline bool EstimateRigidTransformation2D3DBatch(const RansacParameters& p, int t, const std::vector<std::vector<double>>& c26, EstimatorBatchResult* r, std::string* e) { return EstimateBatch(THEIA_EST_RIGID_TRANSFORMATION_2D3D, t, p, c26, nullptr, r, e); }

// EstimateRadialDistUncalibratedAbsolutePose (estimate_radial_dist_uncalibrated_absolute_pose.h:63-83): meta = {max_focal_length,
// min_focal_length, max_radial_distortion, min_radial_distortion, first call of the process (0 / 1)}; rows [u v X Y Z]; model =
// rotation (9) | translation (3) | focal_length | radial_distortion
inline bool EstimateRadialDistUncalibratedAbsolutePoseBatch(const RansacParameters& p, int t, const double meta[5], const std::vector<std::vector<double>>& c, EstimatorBatchResult* r, std::string* e) { return EstimateBatch(THEIA_EST_RADIAL_DIST_UNCALIBRATED_ABSOLUTE_POSE, t, p, c, meta, r, e); }

}  // namespace theia_hip_shim
#endif
