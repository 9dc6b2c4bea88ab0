// theia_hip_internal.h -- shared by the translation units of libtheia_hip.so
#pragma once
#include <string>

namespace thip {
extern thread_local std::string g_last_error;
int set_error(int code, const char* fmt, ...);
int ensure_device();  // lazily selects device 0 unless theia_hip_init chose one
}  // namespace thip

#include <hip/hip_runtime.h>
#include <cstdint>
#include "theia_hip.h"
namespace thip {
// ba_batch.hip: batched single-view LM on device-resident data (one wave per problem);
// problem p covers [offsets[p], offsets[p+1]) or, with counts, [offsets[p], offsets[p] + counts[p]).
// d_out receives views_batch_out_bytes() per problem: {int success, term, iters, nsucc; double c0, c1}.
int views_batch_device(int num, const int64_t* d_offsets, const int* d_counts, const double* d_uv, const double* d_si,
                       const double* d_X, double* d_cam, const double* d_intr, const int* d_model, const uint8_t* d_mask,
                       const theia_ba_options* o, void* d_out, hipStream_t st);
size_t views_batch_out_bytes();
// twoview_lm.hip: batched BundleAdjustTwoViewsAngular on device-resident (x1, y1, x2, y2) rows; d_pose = [num][6]
// (rotation_2 | position_2) in/out; cgnr = 1: CGNR + JACOBI steps, 0: exact normal-equation solve; d_out as for views_batch_device.
int twoview_batch_device(int num, const int64_t* d_offsets, const int* d_counts, const double* d_corr, double* d_pose,
                         const theia_ba_options* o, int cgnr, void* d_out, hipStream_t st);
// twoview_lm.hip: batched OptimizeFundamentalMatrix (no loss); d_F = [num][9] row-major in/out
int fundamental_batch_device(int num, const int64_t* d_offsets, const int* d_counts, const double* d_corr, double* d_F,
                             const theia_ba_options* o, void* d_out, hipStream_t st);
// twoview_lm.hip: batched OptimizeHomography; d_H = [num][9] in Eigen's column-major storage order, in/out (normalised by H(2,2))
int homography_batch_device(int num, const int64_t* d_offsets, const int* d_counts, const double* d_corr, double* d_H,
                            const theia_ba_options* o, void* d_out, hipStream_t st);
// dls_kernels.hip: stage A of the DLS / gDLS minimal solver (Macaulay elimination) for B hypotheses of nprob problems, and for
// whole problems (the directly bound DlsPnp); dls_ensure_tables uploads the index tables once per process.
int dls_ensure_tables();
void launch_dls_stage_a(bool gdls, int datum_stride, int nprob, int B, const int64_t* offsets, const double* data, const int* samples,
                        const int* active_iters, const int* iter_base, const double* uvals, double* action, double* tfac, int* okflag,
                        hipStream_t st);
void launch_dls_solve_a(int num, const int64_t* offsets, const double* feat, const double* world, const double* uvals, double* action,
                        double* tfac, int* okflag, hipStream_t st);
// upnp_kernels.hip: the UPnP hypotheses of THEIA_EST_RIGID_TRANSFORMATION_2D3D for B iterations of nprob problems.  state: the
// estimator's accumulating cost parameters, [nprob][upnp_state_doubles()], zero before the first round; ws: [nprob * B][upnp_workspace_doubles()]
int upnp_ensure_tables();
int upnp_workspace_doubles();
int upnp_state_doubles();
void launch_upnp_fit(int nprob, int B, const int64_t* offsets, const double* data, const int* samples, const int* active_iters,
                     double* state, double* ws, double* models, int* counts, int* dense_count, int* tags, int* hyp_base, hipStream_t st);
// p4pfr_kernels.hip: the P4Pfr hypotheses of THEIA_EST_RADIAL_DIST_UNCALIBRATED_ABSOLUTE_POSE for B iterations of nprob problems.
// rot: [nprob * B][9] the "random rotation" matrix of every hypothesis (p4pfr_rotation_from_draws of its three RandDouble draws, made
// on the host); limits: {max focal length, min focal length, max distortion, min distortion}; ws: [nprob * B][p4pfr_workspace_doubles()]
int p4pfr_ensure_tables();
int p4pfr_workspace_doubles();
void p4pfr_rotation_from_draws(const double* v, double* R);
void launch_p4pfr_fit(int nprob, int B, const int64_t* offsets, const double* data, const int* samples, const int* active_iters,
                      const double* rot, const double* limits, double* ws, double* models, int* counts, int* dense_count, int* tags,
                      int* hyp_base, hipStream_t st, int* solver_counts = nullptr);   // solver_counts[hyp]: solutions before the range tests
// ba_invdepth.hip: bundle adjustment with the inverse-depth track parametrisation (THEIA_BA_FLAG_INVERSE_DEPTH)
int ba_solve_inverse_depth(const theia_ba_problem* p, const theia_ba_options* o, theia_ba_summary* S);
// the same problem as a device-resident object behind the handle API (theia_hip_ba_create with THEIA_BA_FLAG_INVERSE_DEPTH)
int id_handle_create(const theia_ba_problem* p, const theia_ba_options* o, void** out);
int id_handle_reset(void* h, const theia_ba_problem* p);
int id_handle_run(void* h, const theia_ba_options* o, theia_ba_summary* S);
int id_handle_download(void* h, theia_ba_problem* p);
void id_handle_destroy(void* h);
int id_handle_snapshot(void* h);
int id_handle_restore(void* h);
void id_handle_plan_info(void* h, int32_t* n, int32_t* k3_levels, double* k3_flops);
}  // namespace thip
