/*
 * theia_hip.h -- C-ABI of the MI355X (gfx950) bundle-adjustment + RANSAC engine.
 *
 * This is the drop-in boundary: plain C structs of pointers and sizes, no
 * Eigen / STL / torch types.  Every entry point names the reference interface
 * (pyTheiaSfM, paths relative to the reference root) that it replaces.  The
 * reference-side shim (flatten Reconstruction -> call -> scatter back) is shown
 * in INTEGRATION.md.
 *
 * All floating point data is IEEE FP64.  All index data is int32 unless noted.
 * Return value of every function: 0 = THEIA_HIP_OK, negative = error code;
 * the library never throws and never aborts.  theia_hip_last_error() returns a
 * thread-local human-readable message for the last failing call.
 */
#ifndef THEIA_HIP_H_
#define THEIA_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ status */
enum {
  THEIA_HIP_OK = 0,
  THEIA_HIP_ERR_INVALID_ARGUMENT = -1, /* reference: glog CHECK -> abort        */
  THEIA_HIP_ERR_NO_DEVICE = -2,        /* no gfx950 device / HIP runtime error  */
  THEIA_HIP_ERR_UNSUPPORTED = -3,      /* option combination not built yet      */
  THEIA_HIP_ERR_OUT_OF_MEMORY = -4,
  THEIA_HIP_ERR_INTERNAL = -5
};

int theia_hip_init(int device_ordinal);
int theia_hip_shutdown(void);
int theia_hip_device_count(int* count);
const char* theia_hip_last_error(void);
const char* theia_hip_version(void);

/* ------------------------------------------------------- camera model enums */
/* src/theia/sfm/camera/camera_intrinsics_model_type.h:46-56 */
enum {
  THEIA_CAM_PINHOLE = 0,
  THEIA_CAM_PINHOLE_RADIAL_TANGENTIAL = 1,
  THEIA_CAM_FISHEYE = 2,
  THEIA_CAM_FOV = 3,
  THEIA_CAM_DIVISION_UNDISTORTION = 4,
  THEIA_CAM_DOUBLE_SPHERE = 5,
  THEIA_CAM_EXTENDED_UNIFIED = 6,
  THEIA_CAM_ORTHOGRAPHIC = 7
};
#define THEIA_MAX_INTRINSICS 10 /* largest kIntrinsicsSize (radial-tangential) */

/* src/theia/sfm/bundle_adjustment/create_loss_function.h:52-60 */
enum {
  THEIA_LOSS_TRIVIAL = 0,
  THEIA_LOSS_HUBER = 1,
  THEIA_LOSS_SOFTLONE = 2,
  THEIA_LOSS_CAUCHY = 3,
  THEIA_LOSS_ARCTAN = 4,
  THEIA_LOSS_TUKEY = 5,
  THEIA_LOSS_TRUNCATED = 6
};

/* src/theia/sfm/bundle_adjustment/bundle_adjustment.h:71-85 (bitmask) */
enum {
  THEIA_INTR_NONE = 0x00,
  THEIA_INTR_FOCAL_LENGTH = 0x01,
  THEIA_INTR_ASPECT_RATIO = 0x02,
  THEIA_INTR_SKEW = 0x04,
  THEIA_INTR_PRINCIPAL_POINTS = 0x08,
  THEIA_INTR_RADIAL_DISTORTION = 0x10,
  THEIA_INTR_TANGENTIAL_DISTORTION = 0x20,
  THEIA_INTR_ALL = 0x3f
};

/* per-camera constant mask bits (what BundleAdjuster freezes):
 *   bundle_adjuster.cc:477-518 SetCameraExtrinsicsConstant / PositionConstant /
 *   OrientationConstant / SetTzConstant. */
enum {
  THEIA_CAM_CONST_POSITION = 0x1,
  THEIA_CAM_CONST_ORIENTATION = 0x2,
  THEIA_CAM_CONST_ALL = 0x3,
  THEIA_CAM_CONST_TZ = 0x4
};

/* ------------------------------------------------------------ BA problem IR */
/*
 * Flattened form of what BundleAdjuster::AddView/AddTrack build inside Ceres
 * (bundle_adjuster.cc:116-221): one 2-residual block per (estimated view,
 * estimated track) observation; raw parameter blocks
 *   camera extrinsics  [position(3) | angle-axis(3)]   camera.h:202-204,251
 *   shared intrinsics  [K doubles per intrinsics group] camera_intrinsics_model.h
 *   homogeneous point  [x y z w]                        track.h:66,110
 * Parameters are updated IN PLACE (as Ceres writes through the borrowed
 * double* of the Reconstruction, bundle_adjuster.cc:585-591).
 * Caller owns every buffer; nothing is retained after a call returns unless a
 * handle was created, and a handle owns device copies only.
 */
/* problem flags: a track shard of a multi-GPU solve keeps every non-constant
 * camera in the reduced system (also those it does not observe) so that all
 * ranks index the reduced camera system identically. */
enum { THEIA_BA_FLAG_KEEP_UNOBSERVED_CAMERAS = 0x1, THEIA_BA_FLAG_INVERSE_DEPTH = 0x2 };

typedef struct theia_ba_problem {
  int32_t num_cameras;
  int32_t num_groups;
  int32_t num_points;
  int32_t flags;               /* THEIA_BA_FLAG_* */
  int64_t num_obs;

  double* cam_ext;             /* [num_cameras][6]  in/out */
  double* intrinsics;          /* [num_groups][THEIA_MAX_INTRINSICS] in/out */
  const int32_t* group_model;  /* [num_groups] THEIA_CAM_*                 */
  const int32_t* cam_group;    /* [num_cameras] intrinsics group of camera */
  const uint8_t* cam_const;    /* [num_cameras] THEIA_CAM_CONST_* bits, or NULL */
  const uint8_t* group_const;  /* [num_groups] 1 = whole block constant, or NULL
                                  (bundle_adjuster.cc:444-459)             */
  double* points;              /* [num_points][4]  in/out */
  const uint8_t* point_const;  /* [num_points] 1 = SetTrackConstant, or NULL */

  const double* obs_uv;        /* [num_obs][2] Feature::point_             */
  const double* obs_sqrt_info; /* [num_obs][2] 1/sqrt(cov_xx), 1/sqrt(cov_yy)
                                  (reprojection_error.h:94-99) or NULL = 1 */
  const int32_t* obs_cam;      /* [num_obs] camera index                   */
  const int32_t* obs_pt;       /* [num_obs] point index                    */
  /* Camera priors (View::{Position,Gravity,Orientation}Prior added by AddView / AddViewPriors,
   * bundle_adjuster.cc:159-172,291-313; functors position_error.h, gravity_error.h,
   * orientation_error.h: 3 residuals each, no loss).  All optional (NULL); a prior is used when its
   * bit is set both in cam_prior_mask[c] and in options->prior_mask.  Sharded solves: pass the
   * priors on exactly ONE rank (cameras are replicated). */
  const uint8_t* cam_prior_mask;                 /* [num_cameras] THEIA_PRIOR_* bits           */
  const double* cam_position_prior;              /* [num_cameras][3]                           */
  const double* cam_position_prior_sqrt_info;    /* [num_cameras][3][3] row-major              */
  const double* cam_gravity_prior;               /* [num_cameras][3] gravity in the camera frame */
  const double* cam_gravity_prior_sqrt_info;     /* [num_cameras][3][3]                        */
  const double* cam_orientation_prior;           /* [num_cameras][3] angle-axis                */
  const double* cam_orientation_prior_sqrt_info; /* [num_cameras][3][3]                        */
  /* Depth priors (BundleAdjuster::AddDepthPriorErrorResidual, bundle_adjuster.cc:154-156,643-650;
   * DepthPriorError, depth_prior_error.h): every feature with depth_prior != 0 of a solve with
   * use_depth_priors contributes ONE more residual block on (extrinsics, point),
   *   d = (R (X - w C))_z - depth_prior,  weighted by 1 / sqrt(depth_prior_variance),
   * under its own loss (options.robust_loss_width_depth_prior).  Here such a block is one more
   * OBSERVATION ROW of kind THEIA_OBS_DEPTH_PRIOR on the same camera and point:
   *   obs_uv = (depth_prior, 0), obs_sqrt_info = (1 / sqrt(variance), any) (obs_sqrt_info then required).
   * NULL = every row is a reprojection error. */
  const uint8_t* obs_kind;                       /* [num_obs] THEIA_OBS_* or NULL              */
  /* Inverse-depth parametrisation (BundleAdjustmentOptions::use_inverse_depth_parametrization; BundleAdjuster::AddInvTrack,
   * bundle_adjuster.cc:223-289; functors camera/reprojection_error.h:173-286): with THEIA_BA_FLAG_INVERSE_DEPTH the
   * variable of track t is point_inverse_depth[t] = Track::InverseDepth() along point_ref_bearing[t] =
   * Track::ReferenceBearingVector() in the frame of view point_ref_cam[t] = Track::ReferenceViewId(); `points` is not
   * read (the caller's UpdateHomogeneousPoint rebuilds it afterwards, bundle_adjustment.cc:47-65).  Every residual block
   * touches the reference camera, the observing camera, the intrinsics group of the observing camera
   * (bundle_adjuster.cc:594-622; optimised on the subset of options.intrinsics_to_optimize unless group_const) and the
   * inverse depth; the camera priors enter as AddViewPriors adds them (bundle_adjuster.cc:289-313: the views that observe
   * a track or are the reference view of one).  Through theia_hip_ba_solve and the handle API (create / reset_parameters /
   * set_options / run / download; covariance, evaluation and sharding: THEIA_HIP_ERR_UNSUPPORTED); no depth rows / inner
   * iterations in this mode; a track may be observed through at most 8 variable intrinsics groups (ERR_UNSUPPORTED beyond).
   * REFERENCE PARITY: the reference's BundleAdjust*Reconstruction entry points never optimise intrinsics in this mode
   * (only AddInvTrack runs, no view enters optimized_camera_intrinsics_groups_: bundle_adjuster.cc:441-459), so a caller
   * that mirrors them passes options.intrinsics_to_optimize = 0 (the Python mirror does). */
  const int32_t* point_ref_cam;                  /* [num_points] camera index of the reference view */
  const double* point_ref_bearing;               /* [num_points][3]                                 */
  double* point_inverse_depth;                   /* [num_points] in/out, > 0                        */
} theia_ba_problem;
enum { THEIA_PRIOR_POSITION = 1, THEIA_PRIOR_GRAVITY = 2, THEIA_PRIOR_ORIENTATION = 4 };
enum { THEIA_OBS_REPROJECTION = 0, THEIA_OBS_DEPTH_PRIOR = 1 };

/* Mirrors BundleAdjustmentOptions (bundle_adjustment.h:87-167), the fields the
 * HIP backend honours.  The linear-algebra selector fields of the reference
 * struct have no counterpart: this backend IS the linear solver. */
typedef struct theia_ba_options {
  int32_t loss_function_type;    /* THEIA_LOSS_*                  (:90) */
  int32_t intrinsics_to_optimize;/* THEIA_INTR_* bitmask          (:135) */
  int32_t max_num_iterations;    /*                               (:138) */
  int32_t use_homogeneous_point_parametrization; /* SphereManifold<4> (:127) */
  int32_t constant_camera_orientation;           /* (:122) */
  int32_t constant_camera_position;              /* (:123) */
  int32_t orthographic_camera;                   /* (:163) tz constant */
  int32_t use_inner_iterations;  /* (:144) accepted, see DESIGN.md deviation */
  int32_t verbose;               /* (:118) per-iteration table on stderr */
  int32_t prior_mask;            /* THEIA_PRIOR_* bits: use_position_priors (:154), use_gravity_priors (:166),
                                    use_orientation_priors (:157) */
  double robust_loss_width;      /* (:91) */
  double function_tolerance;     /* (:148) */
  double gradient_tolerance;     /* (:149) */
  double parameter_tolerance;    /* (:150) */
  double max_trust_region_radius;/* (:151) */
  double max_solver_time_in_seconds; /* (:141) */
  double robust_loss_width_depth_prior; /* (:94) width of the loss on THEIA_OBS_DEPTH_PRIOR rows */
} theia_ba_options;

void theia_ba_options_default(theia_ba_options* o);

/* termination_type values (ceres::TerminationType order) */
enum {
  THEIA_TERM_CONVERGENCE = 0,
  THEIA_TERM_NO_CONVERGENCE = 1,
  THEIA_TERM_FAILURE = 2
};

/* Mirrors BundleAdjustmentSummary (bundle_adjustment.h:170-178) plus the
 * per-iteration trace used for parity checking against the oracle. */
typedef struct theia_ba_summary {
  int32_t success;               /* IsSolutionUsable() (bundle_adjuster.cc:352) */
  int32_t termination_type;
  int32_t num_iterations;        /* LM iterations run (successful + rejected)   */
  int32_t num_successful_steps;
  double initial_cost;
  double final_cost;
  double setup_time_in_seconds;
  double solve_time_in_seconds;
  /* optional trace, caller-allocated arrays of trace_capacity entries each
   * (entry 0 = iteration 0), NULL to skip */
  int32_t trace_capacity;
  int32_t trace_size;
  double* trace_cost;
  double* trace_gradient_max_norm;
  double* trace_step_norm;
  double* trace_radius;
  int32_t* trace_accepted;
  /* time spent per phase (seconds, device-side events): linearize+Schur,
   * reduced solve, back-substitution+trial cost */
  double time_linearize;
  double time_solve_reduced;
  double time_backsub;
  /* the dominant kernel alone (ba_linearize_schur), HIP events on its stream */
  double time_kernel_linearize;
  int32_t num_linearize_launches;
  int32_t reserved1;
} theia_ba_summary;

/* One-shot solve: replaces BundleAdjuster::Optimize -> ceres::Solve
 * (bundle_adjuster.cc:315-355) for the problem built by
 * BundleAdjustReconstruction / BundleAdjustPartialReconstruction /
 * BundleAdjustPartialViewsConstant (bundle_adjustment.cc:111-217). */
int theia_hip_ba_solve(const theia_ba_problem* problem,
                       const theia_ba_options* options,
                       theia_ba_summary* summary);

/* A batch of INDEPENDENT single-view adjustments: problem i optimises the 6
 * extrinsics cam_ext[i] against its observations [offsets[i], offsets[i+1]) of
 * CONSTANT homogeneous world points, i.e. N calls of
 * BundleAdjustView(options, view_id, reconstruction) (bundle_adjustment.cc:220-237;
 * camera localisation, and RefineModel of the absolute-pose estimator,
 * estimate_calibrated_absolute_pose.cc:120-153) as one launch: one whole LM solve
 * per wavefront, no host round trips.  Honoured options: loss type / width,
 * max_num_iterations, the three tolerances, max_trust_region_radius,
 * constant_camera_{orientation,position}, orthographic_camera.  summaries[i]
 * receives success / termination / iterations / initial / final cost (no trace). */
typedef struct theia_ba_view_batch {
  int32_t num_problems;
  const int64_t* offsets;        /* [num_problems+1] observation offsets, offsets[0] = 0 */
  const double* obs_uv;          /* [total][2] pixels                          */
  const double* obs_sqrt_info;   /* [total][2] or NULL (= 1, 1)                */
  const double* points;          /* [total][4] homogeneous world point of each observation */
  double* cam_ext;               /* [num_problems][6] in/out                   */
  const double* intrinsics;      /* [num_problems][THEIA_MAX_INTRINSICS]       */
  const int32_t* model;          /* [num_problems] THEIA_CAM_*                 */
  const uint8_t* cam_const;      /* [num_problems] THEIA_CAM_CONST_* bits or NULL */
} theia_ba_view_batch;
int theia_hip_ba_views_batch(const theia_ba_view_batch* batch,
                             const theia_ba_options* options,
                             theia_ba_summary* summaries);

/* A batch of INDEPENDENT two-view bundle adjustments: N calls of BundleAdjustTwoViews(options, correspondences,
 * &camera1, &camera2, &points) (bundle_adjust_two_views.cc:110-185; the refinement step of
 * TwoViewMatchGeometricVerification::VerifyMatches, two_view_match_geometric_verification.cc:259-289) as one launch, one
 * LM solve per wavefront.  Camera 1 is held constant, camera 2 moves, the focal length of a camera moves unless
 * const_intrinsics says otherwise (lower bound 1), the points move as XYZW vectors without a manifold; reprojection
 * residuals in both views, trivial loss, Ceres' solver defaults as that function sets them (no inner iterations; pass
 * max_trust_region_radius = 1e16).  Honoured options: max_num_iterations, the three tolerances,
 * max_trust_region_radius.  The same arithmetic as theia_hip_ba_solve on the same flat problem. */
typedef struct theia_ba_two_view_full_batch {
  int32_t num_problems;
  const int64_t* offsets;          /* [num_problems+1] correspondence offsets, offsets[0] = 0            */
  const double* correspondences;   /* [total][4] = (x1, y1, x2, y2) pixels                                */
  double* cam_ext;                 /* [num_problems][2][6]: camera 1 (constant), camera 2 (in/out)       */
  double* intrinsics;              /* [num_problems][2][THEIA_MAX_INTRINSICS]; the focal lengths in/out  */
  const int32_t* model;            /* [num_problems][2] THEIA_CAM_*                                      */
  const uint8_t* const_intrinsics; /* [num_problems][2] 1 = the camera's focal length is held constant   */
  double* points;                  /* [total][4] XYZW of every correspondence, in/out                    */
} theia_ba_two_view_full_batch;
int theia_hip_ba_two_views_batch(const theia_ba_two_view_full_batch* batch, const theia_ba_options* options,
                                 theia_ba_summary* summaries);

/* A batch of INDEPENDENT two-view angular adjustments: problem i refines the relative
 * rotation (angle-axis) and the unit position of view 2 against the angular epipolar
 * error of its correspondences [offsets[i], offsets[i+1]), i.e. N calls of
 * BundleAdjustTwoViewsAngular(options, correspondences, &info)
 * (bundle_adjust_two_views.cc:189-246, residual angular_epipolar_error.h:54-91, bound at
 * src/pytheia/sfm/sfm.cc:1620) -- also RefineModel of the relative-pose estimator
 * (estimate_relative_pose.cc:111-138).  The position moves on SphereManifold<3>; the
 * linear solver is selected by linear_solver: EXACT = the solution of the 5 x 5 normal
 * equations (any direct BundleAdjustmentOptions::linear_solver_type, the default), CGNR =
 * conjugate gradients with the JACOBI preconditioner and Ceres' q-tolerance rule, an
 * inexact step (what the relative-pose RefineModel selects).
 * Honoured options: loss type / width, max_num_iterations, the three tolerances,
 * max_trust_region_radius.  Normalised image coordinates. */
enum { THEIA_TWO_VIEW_EXACT = 0, THEIA_TWO_VIEW_CGNR = 1 };
typedef struct theia_ba_two_view_batch {
  int32_t num_problems;
  int32_t linear_solver;           /* THEIA_TWO_VIEW_*                                       */
  const int64_t* offsets;          /* [num_problems+1], offsets[0] = 0                      */
  const double* correspondences;   /* [total][4] = (x1, y1, x2, y2)                          */
  double* rotation_position;       /* [num_problems][6] in/out: TwoViewInfo::rotation_2 | position_2 */
} theia_ba_two_view_batch;
int theia_hip_ba_two_views_angular_batch(const theia_ba_two_view_batch* batch,
                                         const theia_ba_options* options,
                                         theia_ba_summary* summaries);

/* A batch of INDEPENDENT homography refinements = N calls of OptimizeHomography(options,
 * correspondences, &homography) (bundle_adjust_two_views.cc:298-358; RefineModel of the homography
 * estimator, estimate_homography.cc:89-104): the nine entries of H against the symmetric geometric
 * distance (homography_error.h:45-95: H x1 - x2 and H^-1 x2 - x1, four residuals under one loss),
 * direct linear solver, result divided by H(2,2).  homographies [num_problems][9] row-major in/out;
 * correspondences [total][4] = (x1, y1, x2, y2).  Honoured options as for the two-view batch. */
int theia_hip_optimize_homography_batch(int32_t num_problems, const int64_t* offsets,
                                        const double* correspondences, double* homographies,
                                        const theia_ba_options* options,
                                        theia_ba_summary* summaries);

/* N calls of OptimizeRelativePositionWithKnownRotation(correspondences, rotation1, rotation2, &relative_position)
 * (bundle_adjustment/optimize_relative_position_with_known_rotation.cc:116-191; pybind sfm.cc:1621-1622; called once per
 * view pair from a thread pool by RefineRelativeTranslationsWithKnownRotations, reconstruction_estimator_utils.cc:263-291):
 * the IRLS null-vector solve on the epipolar constraints with both rotations known, at most 100 iterations, sign by the
 * cheirality majority.  One pair per wavefront (csrc/relpos_irls.hip).  correspondences [total][4] = normalised
 * (x1, y1, x2, y2); rotations [num_problems][6] = rotation1 | rotation2 as angle-axis (world -> camera, as
 * Camera::GetOrientationAsAngleAxis); relative_positions [num_problems][3] out (unit vectors; the reference returns true
 * always); num_iterations [num_problems] optional out.  An empty pair yields (0, 0, -1) like the reference's arithmetic. */
int theia_hip_optimize_relative_position_batch(int32_t num_problems, const int64_t* offsets,
                                               const double* correspondences, const double* rotations,
                                               double* relative_positions, int32_t* num_iterations);

/* N calls of OptimizeFundamentalMatrix(options, correspondences, &F) (bundle_adjust_two_views.cc:248-296;
 * RefineModel of the fundamental-matrix estimator, estimate_fundamental_matrix.cc:53-90): F moves on the
 * 7-dof manifold of fundamental_matrix_parameterization.h:15-75 (Plus through the SVD of the current F:
 * every accepted step re-normalises F to singular values (1, sigma, 0)) against one squared-Sampson
 * residual per correspondence (sampson_error.h:21-33); no loss function (options->loss_function_type
 * must be TRIVIAL), direct solver.  fundamental_matrices [num_problems][9] row-major in/out. */
int theia_hip_optimize_fundamental_matrix_batch(int32_t num_problems, const int64_t* offsets,
                                                const double* correspondences, double* fundamental_matrices,
                                                const theia_ba_options* options,
                                                theia_ba_summary* summaries);

/* Every point of `problem` as its OWN problem with all cameras constant: N calls of
 * BundleAdjustTrack(options, track_id, reconstruction) (bundle_adjustment.cc:262-285; the
 * per-track refinement after triangulation, estimate_track.cc:289) in one launch, one thread
 * per track running its whole LM.  problem->points is updated in place; points flagged in
 * point_const (or without observations) are left alone.  summaries[num_points]. */
int theia_hip_ba_tracks_batch(const theia_ba_problem* problem,
                              const theia_ba_options* options,
                              theia_ba_summary* summaries);

/* Per-track reprojection sweep = the loop body of SetOutlierTracksToUnestimated
 * (set_outlier_tracks_to_unestimated.cc:64-139; the same residual as the BA, without Jacobians): for
 * every point of `problem` over its observations: the mean squared (unweighted) reprojection error, the
 * number of views that see it at negative depth (Camera::ProjectPoint), and the smallest cosine between
 * two of its viewing rays (SufficientTriangulationAngle, triangulation.cc:236-250: a track passes when
 * that cosine is below cos(min_angle); 2.0 when it has fewer than two views).  One thread per track. */
int theia_hip_track_statistics(const theia_ba_problem* problem,
                               double* mean_sq_reprojection_error,
                               int32_t* num_behind_camera,
                               double* min_ray_cosine);

/* TrackEstimator::EstimateTrack (estimate_track.cc:206-319) for every point of `problem` that is
 * not flagged in point_const.  With triangulation_method = MIDPOINT (the default, estimate_track.h:82):
 *   1. fewer than two observations, or no pair of viewing rays further apart than
 *      min_triangulation_angle_degrees (SufficientTriangulationAngle, triangulation.cc:236-250) -> bad angle;
 *   2. TriangulateMidpoint (triangulation.cc:130-157): sum (I - d d^T) X = sum (I - d d^T) o, Cholesky;
 *      not positive definite -> failed triangulation; the point becomes (X, 1);
 *      SVD / L2_MINIMIZATION: TriangulateNViewSVD / TriangulateNView (triangulation.cc:178-214) on the pixels and
 *      Camera::GetProjectionMatrix of every observation (computed here from the extrinsics and the model's focal length,
 *      aspect ratio, skew and principal point); the homogeneous point is defined up to sign, as in the reference;
 *   3. bundle_adjustment: BundleAdjustTrack on that point (= theia_hip_ba_tracks_batch), !success -> rejected;
 *   4. AcceptableReprojectionError (estimate_track.cc:93-117): any view with Camera::ProjectPoint < 0, or a mean
 *      squared reprojection error >= max_acceptable_reprojection_error_pixels^2 -> bad reprojection.
 * One thread per track for every stage.  obs_ray_dir[num_obs][3] = Camera::PixelToUnitDepthRay(feature).normalized()
 * of each observation (the caller's camera classes undistort; estimate_track.cc:76-77), origins are the camera
 * positions.  problem->points is written for every track that reached stage 2 (as the reference writes
 * track->MutablePoint() before it knows the outcome).  estimated[num_points]: 1 = SetEstimated(true).
 * counters: {bad angles, failed triangulations, bad reprojections, rejected by the track BA}. */
typedef struct theia_track_estimate_options {
  double min_triangulation_angle_degrees;           /* estimate_track.h:69, default 3.0 */
  double max_acceptable_reprojection_error_pixels;  /* :64, default 5.0 */
  int32_t bundle_adjustment;                        /* :73, default 1 */
  int32_t triangulation_method;                     /* :82, TriangulationMethodType: 0 MIDPOINT (default), 1 SVD, 2 L2_MINIMIZATION */
} theia_track_estimate_options;
int theia_hip_estimate_tracks(const theia_ba_problem* problem, const double* obs_ray_dir,
                              const theia_ba_options* ba_options,
                              const theia_track_estimate_options* options,
                              uint8_t* estimated, int32_t counters[4]);

/* Handle API: problem resident in HBM across calls (bench, repeated solves).  Problems with THEIA_BA_FLAG_INVERSE_DEPTH
 * support create / reset_parameters / set_options / run / download / destroy (the other calls return ERR_UNSUPPORTED). */
typedef struct theia_ba_handle_s* theia_ba_handle;
int theia_hip_ba_create(const theia_ba_problem* problem,
                        const theia_ba_options* options, theia_ba_handle* out);
int theia_hip_ba_reset_parameters(theia_ba_handle h,
                                  const theia_ba_problem* problem);
/* Device-resident copy of the handle's current parameters (cameras, points,
 * intrinsics), and the way back: repeated solves from the same start without a
 * host round trip (bench.py: the timed region starts with its inputs in HBM). */
/* Optional, sharded solves: this rank's index and the number of ranks of the
 * all-reduce group.  With it the MAX-reduced scalar (gradient max-norm) rides in
 * the SUM all-reduce of the reduced camera system (one slot per rank), saving one
 * collective per LM iteration; without it a separate MAX all-reduce is issued. */
int theia_hip_ba_set_shard(theia_ba_handle h, int32_t rank, int32_t world_size);
int theia_hip_ba_snapshot_parameters(theia_ba_handle h);
int theia_hip_ba_restore_parameters(theia_ba_handle h);
/* Replace the solver-control options of a handle (iteration cap, tolerances,
 * loss, radius cap, verbosity).  Options that shape the problem
 * (parametrisation, constant-camera switches, intrinsics mask) must equal the
 * ones given to create(), else THEIA_HIP_ERR_INVALID_ARGUMENT. */
int theia_hip_ba_set_options(theia_ba_handle h, const theia_ba_options* options);
int theia_hip_ba_run(theia_ba_handle h, theia_ba_summary* summary);
int theia_hip_ba_download(theia_ba_handle h, theia_ba_problem* problem);
int theia_hip_ba_destroy(theia_ba_handle h);

/* Shape of the device plan of a handle, for roofline accounting (bench.py): reduced system size, elimination-tree
 * levels and FP64 flops of one reduced-camera solve (K3), workgroups ("runs") of the fused linearise + Schur kernel
 * (0 = the gather kernels are in use), tracks on the per-observation slow path.  Any pointer may be NULL. */
int theia_hip_ba_plan_info(theia_ba_handle h, int32_t* n, int32_t* k3_levels, double* k3_flops, int32_t* fused_runs,
                           int32_t* slow_path_tracks);


/* Covariance blocks at the handle's current state for the two block-diagonal problems behind the
 * *WithCov entry points (bundle_adjustment.cc:288-386,420-499; GetCovarianceFor{Track,Tracks,View,Views},
 * bundle_adjuster.cc:660-773 = ceres::Covariance (J'J)^-1 in tangent space, loss applied):
 *   point_cov[num_points][d][d], d = 3 (homogeneous manifold) or 4: needs every camera constant;
 *   cam_cov[num_cameras][6][6]: needs every point constant and no partially constant camera.  With optimised intrinsics
 *   the cameras of a group are coupled through the group's columns: the extrinsics blocks of the dense inverse of J'J are
 *   returned (reduced systems up to 2048 columns: a *WithCov call covers a handful of views).
 * Either pointer may be NULL.  Entries of constant / unobserved blocks are zero.  NOT yet multiplied
 * by the empirical variance factor 2 * final_cost / redundancy (the caller's bookkeeping). */
int theia_hip_ba_covariance(theia_ba_handle h, double* point_cov, double* cam_cov);


/* Introspection used by parity tests (device results copied to host):
 *  - per-observation residual r[2] and Jacobian blocks J_cam[2][6],
 *    J_pt[2][pt_dof] (tangent space, loss-corrected, NOT Jacobi-scaled) of
 *    ReprojectionError<Model> (reprojection_error.h:54-110); valid[i] = functor
 *    return value.
 *  - the dense reduced camera system  S (n x n, row-major, both triangles) and
 *    rhs (n) for trust-region radius `radius`,
 *    n = 10 * (#variable intrinsics groups) + 6 * (#variable cameras) with the
 *    intrinsics slots first, in Jacobi-scaled space exactly as handed to the
 *    Cholesky kernel. */
int theia_hip_ba_evaluate(theia_ba_handle h, double* cost, double* residuals,
                          double* jac_cam, double* jac_pt, uint8_t* valid);
/* as theia_hip_ba_evaluate plus J_intr[2][THEIA_MAX_INTRINSICS] per observation
 * (zero for constant groups / frozen parameters). */
int theia_hip_ba_evaluate_ex(theia_ba_handle h, double* cost, double* residuals,
                             double* jac_cam, double* jac_pt, double* jac_intr,
                             uint8_t* valid);
int theia_hip_ba_reduced_system(theia_ba_handle h, double radius, int32_t* n,
                                double* S, double* rhs, int64_t capacity);

/* K3 alone (introspection for the parity tests): solve the dense SPD system
 * A x = b with the reduced-camera Cholesky kernels.  A: n x n row-major, only
 * the lower triangle is read; x: n.  Returns THEIA_HIP_ERR_INTERNAL when a
 * pivot is not positive (what makes an LM step "invalid"). */
int theia_hip_dense_spd_solve(int32_t n, const double* A, const double* b, double* x);

/* Multi-GPU (one process per GPU): tracks are sharded by the caller; each
 * rank's handle holds its shard plus ALL cameras.  The reduced camera system
 * (and the scalar reductions) are summed across ranks through this callback,
 * which the host implements with RCCL (ncclAllReduce(sum, fp64)) -- see
 * pytheiasfm_amd/distributed.py and INTEGRATION.md.  The buffer is device
 * memory; `stream` is the hipStream_t the library enqueued its producers on. */
enum { THEIA_REDUCE_SUM = 0, THEIA_REDUCE_MAX = 1 };
typedef int (*theia_allreduce_fn)(void* ctx, void* device_buffer,
                                  size_t count_f64, int op, void* stream);
int theia_hip_ba_set_allreduce(theia_ba_handle h, theia_allreduce_fn fn,
                               void* ctx);

/* Inner iterations (options.use_inner_iterations, the reference's default) in a sharded solve.  A camera's residual blocks
 * are spread over the track shards, so each rank is given the FULL problem's observations once: after every trust-region
 * candidate the shards' candidate points are summed into the full point set (one all-reduce of 4 * num_points doubles),
 * every rank sweeps ALL cameras and intrinsics groups over the full observation set -- the same sums on every rank, nothing
 * to exchange afterwards --, then its own tracks, and the cost and the step norms of the sweep are all-reduced (4 doubles).
 * `full_problem`: the unsharded problem (same cameras and groups as the shard; every point, observation and camera prior);
 * point_global_index[shard points] = index of each of the shard's points in it.  The handle must have been created with
 * use_inner_iterations = 1.  Without this call a sharded run with inner iterations returns ERR_UNSUPPORTED.  When the
 * shard geometry is known (theia_hip_ba_set_shard / set_rccl) the cameras and the groups are dealt to the ranks by index and
 * their results summed (two more small all-reduces); otherwise every rank sweeps all of them. */
int theia_hip_ba_set_inner_global(theia_ba_handle h, const theia_ba_problem* full_problem, const int64_t* point_global_index);

/* The same all-reduce issued by the library itself: `ncclAllReduce` (RCCL) on the solve's own HIP stream, no host
 * callback inside the LM iteration.  librccl is looked up at run time (the copy the process already loaded, else
 * /opt/rocm/lib): no link-time dependency.  Rank 0 obtains the 128-byte id and hands it to the other ranks by any
 * channel the host has; every rank then creates its communicator and attaches it to its handle (which also declares
 * the shard geometry, as theia_hip_ba_set_shard).  The communicator outlives the handles using it and is destroyed
 * by the host. */
int theia_hip_rccl_unique_id(void* out128);
int theia_hip_rccl_comm_create(const void* id128, int32_t rank, int32_t world_size, void** comm_out);
int theia_hip_rccl_comm_destroy(void* comm);
/* ranks the communicator spans as RCCL itself reports them (ncclCommCount): what a scaling run quotes as its width */
int theia_hip_rccl_comm_count(void* comm, int32_t* count_out);
int theia_hip_ba_set_rccl(theia_ba_handle h, void* comm, int32_t rank, int32_t world_size);

/* ------------------------------------------------------------------ RANSAC */
/* src/theia/solvers/sample_consensus_estimator.h:58-126 */
typedef struct theia_ransac_params {
  double error_thresh;
  double failure_probability;
  double min_inlier_ratio;
  int32_t min_iterations;
  int32_t max_iterations;
  int32_t use_mle;
  int32_t use_lo;              /* LO-RANSAC: RefineModel of the absolute-pose (BundleAdjustView) and relative-pose
                                  (BundleAdjustTwoViewsAngular; also the uncalibrated one), homography
                                  (OptimizeHomography) and fundamental-matrix (OptimizeFundamentalMatrix)
                                  estimators as device batches, the default "return true" of the others */
  int32_t lo_start_iterations;
  int32_t use_Tdd_test;        /* reference: "Not currently implemented"  */
  uint32_t seed;               /* seeds the mt19937 stream (util/random.cc:60-66),
                                  i.e. RandomNumberGenerator(seed)        */
  int32_t ransac_type;         /* THEIA_RANSAC_* (create_and_initialize_ransac_variant.h:52) */
} theia_ransac_params;

/* RansacType: RANSAC = RandomSampler + InlierSupport/MLE; PROSAC = ProsacSampler
 * (prosac_sampler.cc:53-128, data sorted by quality, best first); LMED =
 * RandomSampler + LmedQualityMeasurement (lmed.h:64-70); EXHAUSTIVE =
 * ExhaustiveSampler (exhaustive_sampler.cc:45-79: every pair in order), which
 * CHECK-fails in the reference unless the estimator's sample size is 2, so it
 * is accepted for THEIA_EST_RELATIVE_POSE_KNOWN_ORIENTATION only and returns
 * THEIA_HIP_ERR_INVALID_ARGUMENT with the reference's message otherwise. */
enum { THEIA_RANSAC_RANSAC = 0, THEIA_RANSAC_PROSAC = 1, THEIA_RANSAC_LMED = 2, THEIA_RANSAC_EXHAUSTIVE = 3 };

void theia_ransac_params_default(theia_ransac_params* p);

/* estimator ids: which Estimate* front-end is replaced */
enum {
  /* EstimateRelativePose, estimators/estimate_relative_pose.cc:159-172 */
  THEIA_EST_RELATIVE_POSE = 0,
  /* EstimateEssentialMatrix, estimators/estimate_essential_matrix.cc:90-106 */
  THEIA_EST_ESSENTIAL_MATRIX = 1,
  /* EstimateCalibratedAbsolutePose, estimate_calibrated_absolute_pose.cc:176-190 */
  THEIA_EST_ABSOLUTE_POSE_KNEIP = 2,
  THEIA_EST_ABSOLUTE_POSE_DLS = 3,
  THEIA_EST_ABSOLUTE_POSE_SQPNP = 4,
  /* EstimateFundamentalMatrix, estimators/estimate_fundamental_matrix.cc:105-121
   * (normalised 8-point + squared Sampson distance; the error threshold is in
   * squared pixels as in the reference) */
  THEIA_EST_FUNDAMENTAL_MATRIX = 5,
  /* EstimateHomography, estimators/estimate_homography.cc:120-135 (4-point DLT,
   * asymmetric transfer error) */
  THEIA_EST_HOMOGRAPHY = 6,
  /* EstimateDominantPlaneFromPoints, estimate_dominant_plane_from_points.cc:95-107
   * (error = point-to-plane distance, not squared) */
  THEIA_EST_DOMINANT_PLANE = 7,
  /* EstimateRelativePoseWithKnownOrientation,
   * estimate_relative_pose_with_known_orientation.cc:66-81 (2 correspondences) */
  THEIA_EST_RELATIVE_POSE_KNOWN_ORIENTATION = 8,
  /* EstimateUncalibratedRelativePose, estimate_uncalibrated_relative_pose.cc:204-220:
   * 8-point F, focal lengths from F, E = K2 F K1 decomposed with the cheirality vote;
   * data in pixels with the principal point removed.  estimator_params = {min, max}
   * focal length (both >= 1 to be applied, as in the reference). */
  THEIA_EST_UNCALIBRATED_RELATIVE_POSE = 9,
  /* EstimateAbsolutePoseWithKnownOrientation, estimate_absolute_pose_with_known_orientation.cc:132-153:
   * 2 correspondences [u v X Y Z] whose features the caller has rotated into the world frame
   * (RotateCorrespondences, :53-72); model = camera position (3) */
  THEIA_EST_ABSOLUTE_POSE_KNOWN_ORIENTATION = 10,
  /* EstimateTriangulation, estimate_triangulation.cc:69-166: one problem = one track, datum = one observation WITH ITS
   * CAMERA (PointObservation, :53-58), 33 doubles:
   *   [0..11]  projection matrix [R | -R c], row-major 3 x 4 (:122-129)
   *   [12..13] normalised feature = hnormalized(PixelToNormalizedCoordinates(pixel)) (:132-135, the caller's camera class)
   *   [14..15] observed pixel
   *   [16..21] camera extrinsics: position (3), angle-axis (3)      [22] camera model (THEIA_CAM_*)     [23..32] intrinsics
   * Minimal sample = 2 observations -> Triangulate() (triangulation.cc:109-125: essential matrix of the two poses, the
   * optimal image-point correction, DLT by the 4 x 4 SVD), kept only if the point is in front of both cameras (:82-87);
   * Error = squared pixel reprojection error through Camera::ProjectPoint, DBL_MAX when the depth is not positive (:92-101).
   * model = homogeneous point (4).  The reference runs EXHAUSTIVE over all pairs for <= 15 observations with
   * min = max iterations = n (n - 1) / 2, RANSAC otherwise (:147-159): the caller passes that choice in the parameters. */
  THEIA_EST_TRIANGULATION = 11,
  /* EstimateRadialHomographyMatrix (estimate_radial_distortion_homography.cc:52-111): datum =
   * RadialDistortionFeatureCorrespondence (estimate_radial_distortion_homography.h:56-70), 12 doubles:
   *   [0,1] feature_left  [2,3] feature_right (pixels, principal point removed)  [4,5] normalized_feature_left
   *   [6,7] normalized_feature_right  [8] focal_length_estimate_left  [9] focal_length_estimate_right
   *   [10] min_radial_distortion  [11] max_radial_distortion (read from the first datum of a sample, as the reference does).
   * EstimateModel = SixPointRadialDistortionHomography (six_point_radial_distortion_homography.cc:62-148: null space of
   * the 6 x 8 constraint, a quadratic, least-squares direction of a 6 x 5 system), up to two models per sample; Error =
   * CheckRadialSymmetricError (:201-239).  model = RadialHomographyResult: H (9, row-major), l1, l2 (then H^-1, 9 doubles,
   * kept for the scoring kernels). */
  THEIA_EST_RADIAL_HOMOGRAPHY = 12,
  /* EstimateSimilarityTransformation2D3D (estimate_similarity_transformation_2d_3d.cc:72-178): datum =
   * CameraAndFeatureCorrespondence2D3D (sfm/similarity_transformation.h / estimators header), 26 doubles:
   *   [0..2] camera.PixelToUnitDepthRay(observation).normalized() (world frame; precomputed per datum: it does not depend on
   *          the model)   [3..6] point3d (homogeneous)   [7,8] observation pixel   [9..14] camera extrinsics: position,
   *          angle-axis   [15] camera model (THEIA_CAM_*)   [16..25] intrinsics.
   * EstimateModel = GdlsSimilarityTransform on four data (gdls_similarity_transform.cc:67-228: the DLS pipeline with the
   * generalised cost matrix; Macaulay terms of one Estimate() counted from 0 as for DLS), up to 27 models; Error = squared
   * pixel error of the camera moved by the transformation (TransformCamera), DBL_MAX behind it.
   * model = SimilarityTransformation: rotation (9, row-major), translation (3), scale. */
  THEIA_EST_SIMILARITY_2D3D = 13,
  /* EstimateUncalibratedAbsolutePose (estimate_uncalibrated_absolute_pose.cc:60-141): datum = FeatureCorrespondence2D3D
   * [u v X Y Z] with the principal point removed from the pixel; sample = 4; EstimateModel = FourPointPoseAndFocalLength
   * (P4Pf, four_point_focal_length.cc:100-222), up to 10 models; Error = squared reprojection error of the projection
   * matrix (:88-97, no cheirality test).  model = the 3 x 4 projection matrix K [R | t], row-major (12 doubles); the
   * reference's DecomposeProjectionMatrix step (:124-138: rotation, position, focal length) is left to the caller (the
   * Python mirror does it).  The elimination template is not the reference's generated one. */
  THEIA_EST_UNCALIBRATED_ABSOLUTE_POSE = 14,
  /* EstimateRigidTransformation2D3D (estimate_rigid_transformation_2d_3d.cc:62-182): datum = CameraAndFeatureCorrespondence2D3D in
   * the 26-double row of THEIA_EST_SIMILARITY_2D3D ([0..2] the observation's normalised world-frame ray, [3..6] point3d, [7,8]
   * pixel, [9..14] camera position | angle-axis, [15] camera model, [16..25] intrinsics); the overload for
   * FeatureCorrespondence2D3D (:159-182) is the same call with identity pinhole cameras (focal length 1).  Sample = 4;
   * EstimateModel = Upnp::EstimatePose (pose/upnp.cc:462-493: the 8-solution symmetric template, eliminated by the reference's own
   * Gauss-Jordan), up to 8 models; Error = squared pixel error of R X + t seen by the datum's camera, DBL_MAX behind it (:116-129).
   * As in the reference, the estimator's UPnP cost parameters ACCUMULATE over the samples of one Estimate() call (upnp.cc:191-200
   * adds to a member of the one Upnp object the estimator keeps): hypothesis k is solved from the samples 0 .. k.
   * model = RigidTransformation: rotation (9, row-major), translation (3). */
  THEIA_EST_RIGID_TRANSFORMATION_2D3D = 15,
  /* EstimateRadialDistUncalibratedAbsolutePose (estimate_radial_dist_uncalibrated_absolute_pose.cc:163-189): datum =
   * FeatureCorrespondence2D3D [u v X Y Z] (pixels with the principal point removed, as they were observed: distorted); sample = 4;
   * EstimateModel = FourPointsPoseFocalLengthRadialDistortion (pose/four_point_focal_length_radial_distortion.cc:68-288 and
   * _helper.cc: the 40 x 50 elimination template reduced through Eigen::FullPivLU of its transposed 37 x 40 block, 13 x 13 action
   * matrix), up to 13 models; Error = squared distance between the feature and the projection distorted by the division model,
   * 1e10 when the translation's z is negative (:130-147).  estimator_params (required) = RadialDistUncalibratedAbsolutePoseMetaData
   * {max_focal_length, min_focal_length, max_radial_distortion, min_radial_distortion} and a fifth entry "first call of the
   * process" (0 / 1).  The solver multiplies its null-space basis by a "random rotation" made from three RandDouble(-0.5, 0.5)
   * (:134-141), and every RandomNumberGenerator object of the reference shares ONE std::mt19937 (util/random.cc:46-66): the draws
   * come out of the SAMPLER's stream, three after every sample -- reproduced here; with the fifth entry set, the stream is re-seeded
   * with 42 after the first sample, as the solver's function-static RandomNumberGenerator(42) does the first time it runs in a
   * process.  model = rotation (9, row-major), translation (3), focal_length, radial_distortion. */
  THEIA_EST_RADIAL_DIST_UNCALIBRATED_ABSOLUTE_POSE = 16
};

/* A batch of independent estimation problems ("pairs").  Datum layout:
 *   relative pose / essential: FeatureCorrespondence = [x1 y1 x2 y2]
 *     (matching/feature_correspondence.h; only Feature::point_ is used)
 *     (also fundamental matrix, homography, known-orientation relative pose)
 *   absolute pose: FeatureCorrespondence2D3D = [u v X Y Z]
 *     (sfm/feature_correspondence_2d_3d.h:42-49)
 *   dominant plane: Eigen::Vector3d = [X Y Z]
 *   triangulation: one observation with its camera, 33 doubles (THEIA_EST_TRIANGULATION)
 *   radial-distortion homography: RadialDistortionFeatureCorrespondence, 12 doubles (THEIA_EST_RADIAL_HOMOGRAPHY)
 *   similarity 2D-3D, rigid transformation 2D-3D: CameraAndFeatureCorrespondence2D3D, 26 doubles (THEIA_EST_SIMILARITY_2D3D)
 *   uncalibrated absolute pose (with or without radial distortion): FeatureCorrespondence2D3D = [u v X Y Z], pixels with the
 *     principal point removed */
typedef struct theia_ransac_batch {
  int32_t estimator;           /* THEIA_EST_*                              */
  int32_t num_problems;
  const int64_t* offsets;      /* [num_problems+1] datum offsets           */
  const double* data;          /* [offsets[num_problems]][datum_size]      */
  const double* estimator_params; /* estimator constants (see THEIA_EST_*), or NULL; required (not NULL) for
                                   * THEIA_EST_UNCALIBRATED_RELATIVE_POSE and THEIA_EST_RADIAL_DIST_UNCALIBRATED_ABSOLUTE_POSE */
  const uint32_t* seeds;       /* [num_problems] RandomNumberGenerator seed of each problem, or NULL = params.seed + index.
                                  Lets a caller keep a pair's sample stream when it re-batches or shards the pairs. */
} theia_ransac_batch;

/* Result per problem.  model layout:
 *   RELATIVE_POSE:    E(9, row-major) R(9, row-major) position(3)  = 21
 *   ESSENTIAL_MATRIX: E(9)                                          = 9
 *   ABSOLUTE_POSE_*:  R(9, row-major) position(3)                   = 12
 *   FUNDAMENTAL_MATRIX / HOMOGRAPHY: 3x3 row-major                  = 9
 *   DOMINANT_PLANE:   point(3) unit_normal(3)                       = 6
 *   RELATIVE_POSE_KNOWN_ORIENTATION: unit position of camera 2      = 3
 *   UNCALIBRATED_RELATIVE_POSE: F(9) R(9) position(3) focal_length1 focal_length2 = 23
 *   UNCALIBRATED_ABSOLUTE_POSE: projection matrix 3 x 4, row-major    = 12
 *   RIGID_TRANSFORMATION_2D3D: R(9, row-major) translation(3)        = 12
 *   RADIAL_DIST_UNCALIBRATED_ABSOLUTE_POSE: R(9) translation(3) focal_length radial_distortion = 14 */
#define THEIA_RANSAC_MODEL_STRIDE 24
typedef struct theia_ransac_result {
  int32_t* success;            /* [num_problems] Estimate() return value; 0 (with no inliers and a zero
                                * model) for a problem with fewer data than the minimal sample, where the
                                * reference's sampler CHECK-fails -- the rest of the batch still runs */
  double* models;              /* [num_problems][THEIA_RANSAC_MODEL_STRIDE]*/
  int32_t* num_inliers;        /* [num_problems]                           */
  uint8_t* inlier_mask;        /* [offsets[num_problems]] 1 = inlier       */
  int32_t* num_iterations;     /* [num_problems] RansacSummary             */
  double* confidence;          /* [num_problems]                           */
  /* totals over the batch, for hypotheses/s accounting */
  int64_t hypotheses_evaluated;/* minimal samples fitted + fully scored    */
  int64_t models_scored;       /* models scored against all data           */
  double time_fit_score_seconds; /* device time in the fit+score kernels   */
  int32_t* num_lo_iterations;  /* [num_problems] RansacSummary::num_lo_iterations, or NULL */
  double time_fit_seconds;     /* device time of the model-fit kernels (HIP events on the library's stream) */
  double time_score_seconds;   /* device time of the scoring kernels                                        */
} theia_ransac_result;

/* Replaces SampleConsensusEstimator<E>::Estimate
 * (solvers/sample_consensus_estimator.h:300-415) with Ransac<E> +
 * RandomSampler (ransac.h:57-61, random_sampler.cc:53-72), for every problem
 * of the batch.  Problem i uses RandomNumberGenerator(seed + i). */
int theia_hip_ransac_estimate_batch(const theia_ransac_batch* batch,
                                    const theia_ransac_params* params,
                                    theia_ransac_result* result);

/* Directly bound minimal solvers (src/pytheia/sfm/sfm.cc:573-592,
 * pose_wrapper.cc:166-173).  Batched: `num` independent minimal problems.
 *  five point: in = [num][5][4] (x1 y1 x2 y2), out E = [num][10][9],
 *              num_solutions[num]  (five_point_relative_pose.cc:212-293)
 *  p3p:        in = [num][3][5] (u v X Y Z), out R=[num][4][9], t=[num][4][3]
 *              (perspective_three_point.cc PoseFromThreePoints)            */
int theia_hip_five_point_relative_pose(int32_t num, const double* corr,
                                       double* essential_matrices,
                                       int32_t* num_solutions);
int theia_hip_pose_from_three_points(int32_t num, const double* corr2d3d,
                                     double* rotations, double* translations,
                                     int32_t* num_solutions);
/* SQPnP (sfm/pose/sqpnp.h:58-77, bound in src/pytheia/sfm/sfm.cc:592), batched:
 * problem i uses points [offsets[i], offsets[i+1]) of features[.][2] (normalised
 * image coordinates) and world_points[.][3] (>= 3 points, else 0 solutions).
 * Outputs per problem up to 18 solutions: quaternions[num][18][4] = [w x y z]
 * (world -> camera rotation), translations[num][18][3], zero padded. */
int theia_hip_sqpnp(int32_t num, const int64_t* offsets, const double* features,
                    const double* world_points, double* quaternions,
                    double* translations, int32_t* num_solutions);

/* DlsPnp (sfm/pose/dls_pnp.h:56-67, bound in src/pytheia/sfm/sfm.cc:577), batched like theia_hip_sqpnp; up to 27
 * solutions per problem: quaternions[num][27][4] = [w x y z] (world -> camera), translations[num][27][3], zero padded.
 * The reference mixes a random linear form into the Macaulay matrix: 100 * Eigen::Vector4d::Random() =
 * four std::rand() draws per call, never seeded (dls_pnp.cc:134).  Problem i takes the draws of the
 * call_index[i]-th DlsPnp call of a process (call_index == NULL: i).  theia_hip_dls_macaulay_terms returns those
 * terms, out[num_calls][4], from the library's restatement of glibc's rand() (host only, needs no GPU). */
int theia_hip_dls_pnp(int32_t num, const int64_t* offsets, const double* features,
                      const double* world_points, const int64_t* call_index,
                      double* quaternions, double* translations, int32_t* num_solutions);
void theia_hip_dls_macaulay_terms(int64_t first_call, int64_t num_calls, double* out);
/* P4Pf (FourPointPoseAndFocalLength, sfm/pose/four_point_focal_length.cc:100-222; bound in pose_wrapper.cc), batched:
 * corr2d3d = [num][4][5] (u v X Y Z, principal point removed), projection_matrices = [num][10][12] (3 x 4 row-major,
 * zero padded), num_solutions[num] (0 for a degenerate sample; the reference returns -1 there). */
int theia_hip_four_point_pose_and_focal_length(int32_t num, const double* corr2d3d, double* projection_matrices,
                                               int32_t* num_solutions);
/* P4Pfr (FourPointsPoseFocalLengthRadialDistortion, sfm/pose/four_point_focal_length_radial_distortion.cc:68-288; bound in
 * pose_wrapper.cc:189-215 / src/pytheia/sfm/sfm.cc:581), batched: corr2d3d = [num][4][5] (u v X Y Z, distorted pixels with the
 * principal point removed); limits = {max_focal_length, min_focal_length, max_radial_distortion, min_radial_distortion}
 * (RadialDistUncalibratedAbsolutePoseMetaData; the reference's CHECKs -> THEIA_HIP_ERR_INVALID_ARGUMENT); rotation_draws =
 * [num][3] the three RandDouble(-0.5, 0.5) each call takes from the process-wide generator (:134-138), or NULL = the draws of the
 * first num calls of a fresh process (the solver's static RandomNumberGenerator(42)).  models = [num][13][14]: rotation (9,
 * row-major), translation (3), focal length, radial distortion, zero padded; num_solutions[num] = solutions that passed the
 * focal-length / distortion range tests; num_solver_solutions[num] (may be NULL) = the solver's valid_solutions BEFORE those
 * tests -- the reference's return value is `valid_solutions.size() > 0` (:287), so it can report success with empty outputs. */
int theia_hip_four_point_focal_length_radial_distortion_ex(int32_t num, const double* corr2d3d, const double* limits,
                                                           const double* rotation_draws, double* models, int32_t* num_solutions,
                                                           int32_t* num_solver_solutions);
/* the same without the trailing output: the symbol and ABI of the earlier library versions (a caller built against the
 * six-argument declaration must not find a seven-argument function behind it) */
int theia_hip_four_point_focal_length_radial_distortion(int32_t num, const double* corr2d3d, const double* limits,
                                                        const double* rotation_draws, double* models, int32_t* num_solutions);

/* FindKNearestNeighbors of GuidedEpipolarMatcher (matching/guided_epipolar_matcher.cc:356-412), the descriptor search behind
 * the guided_matching branch of TwoViewMatchGeometricVerification::VerifyMatches (two_view_match_geometric_verification.cc:
 * 157-170), for all epiline groups of an image pair in one launch: group g has the query features q_idx[q_off[g] ..
 * q_off[g+1]) of image 1 (rows of desc1 [n1][dim]) and the candidate features c_idx[c_off[g] .. c_off[g+1]) of image 2 (rows
 * of desc2 [n2][dim]); for every query, in q order: nn_dist[2] = the two smallest squared L2 distances (float, summed over
 * the dimensions in sequence), nn_index[2] = the candidates' feature indices (ties: the earlier candidate of the list, as the
 * reference's partial_sort of (distance, position) pairs; -1 / FLT_MAX where the group has fewer than two candidates). */
int theia_hip_guided_knn(int32_t num_groups, const int64_t* q_off, const int32_t* q_idx, const int64_t* c_off,
                         const int32_t* c_idx, int32_t n1, int32_t n2, int32_t dim, const float* desc1, const float* desc2,
                         float* nn_dist, int32_t* nn_index);
/* n draws of RandomNumberGenerator(seed).RandInt(lo, hi) (util/random.cc:46-84): host code that follows the reference's
 * generator outside the RANSAC sampler (the random candidates of GuidedEpipolarMatcher::FindFeaturesNearEpipolarLines). */
int theia_hip_randint_stream(uint32_t seed, int32_t n, int32_t lo, int32_t hi, int32_t* out);
/* Self-check of the cross-lane primitives the bit-exact kernels stand on (wave_reduce.h: the XOR-butterfly sums on permlane swaps
 * + DPP, the wave maximum of |a| as two unsigned reductions, the row broadcasts) against plain shuffle loops on `count` random
 * wavefronts; *mismatches = lanes whose bits differ (0 on a device / compiler the library is right for).  No reference
 * counterpart: a maintainer's first call on new hardware. */
int theia_hip_selftest_wave_primitives(int32_t count, int32_t* mismatches);

/* The batch entry points above keep their device workspace and the pinned host blocks of their per-round transfers in
 * process-wide caches between calls (up to 6 GiB of device memory and 2 GiB of pinned host memory); the buffers of a
 * destroyed BA handle and the host staging of theia_hip_ba_create go to the same caches, so that a pipeline that
 * solves problem after problem does not pay hipMalloc / hipFree and page faults each time.  This returns both caches
 * to the runtime.  All RANSAC kernels of
 * the process run on one library-owned stream; calls from several host threads are safe and overlap their host work. */
void theia_hip_release_scratch(void);

#ifdef __cplusplus
}
#endif
#endif /* THEIA_HIP_H_ */
