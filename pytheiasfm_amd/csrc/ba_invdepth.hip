// Bundle adjustment with the inverse-depth track parametrisation
// (BundleAdjustmentOptions::use_inverse_depth_parametrization; the reference: BundleAdjuster::AddInvTrack,
// bundle_adjuster.cc:223-289, residual blocks :594-622, functors camera/reprojection_error.h:173-286).
//
// The variable of track t is rho_t = Track::InverseDepth() along b_t = Track::ReferenceBearingVector() in the frame
// of its reference view r(t).  An observation of t in view v sees the world point X = R_r^T (b / rho) + c_r through
// camera v: the residual touches the REFERENCE camera, the observing camera and rho (for v = r(t) it depends on rho
// alone: InvReprojectionPoseError composes T^-1 T).  So a row has two 2 x 6 camera blocks and one 2 x 1 point block;
// the point blocks are scalars, the Schur complement of a track is a rank-one update of the camera system,
//     S -= w w^T / (v + d),   w = [sum_obs F_c^T e]_c over the cameras of the track (its reference camera gets the sum
// over ALL its observations), and the cameras of one observation couple directly through F_ref^T F_oth.
//
// A first, correctness-first device path: per-observation records {F_ref | F_oth | e | r} in HBM, one thread per
// observation / per track, FP64 atomics into the dense reduced system, the library's Cholesky for the solve, the LM
// rules of ba_solver.hip run by the host between launches.  Intrinsics are constant in this mode; no camera priors,
// depth rows, inner iterations or sharding.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

#include "ba_device.h"
#include "ba_kernels.h"
#include "theia_hip.h"
#include "theia_hip_internal.h"

namespace thip {
namespace {

#define HIP_TRY(expr)                                                                                       \
  do {                                                                                                      \
    hipError_t e_ = (expr);                                                                                 \
    if (e_ != hipSuccess)                                                                                   \
      return set_error(e_ == hipErrorOutOfMemory ? THEIA_HIP_ERR_OUT_OF_MEMORY : THEIA_HIP_ERR_NO_DEVICE,   \
                       "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);         \
  } while (0)

constexpr int kRec = 28;   // record of one observation: F_ref (2 x 6) | F_oth (2 x 6) | e (2) | r (2)

struct IdProblem {
  int nc, np;
  int64_t nobs;
  int n;                          // reduced system size: 6 * (#variable cameras)
  const double* intr;             // [ng][10]
  const int* group_model;
  const int* cam_group;
  const int* cam_red;             // [nc] reduced index or -1
  const uint8_t* cam_mask;        // [nc] bit q = column frozen
  const uint8_t* pt_const;        // [np]
  const int* pt_ref;              // [np] reference camera
  const double* bearing;          // [np][3]
  const double2* obs_uv;
  const double2* obs_si;          // always valid (ones when the caller gave none)
  const int* obs_cam;
  const int* obs_pt;
  const int64_t* pt_off;          // [np + 1] observations by track
  const int* pt_obs;
  const double* scale_c;          // [nc][6] Jacobi scaling
  const double* scale_r;          // [np]
  int loss_type;
  double loss_width;
};

enum { ID_COST = 0, ID_INVALID = 1, ID_NOTPD = 2, ID_MCC = 3, ID_STEPSQ = 4, ID_XNORMSQ = 5, ID_GMAX = 6, ID_SCALARS = 8 };

__device__ inline void atomic_max_nonneg(double* addr, double v) {
  atomicMax(reinterpret_cast<unsigned long long*>(addr), (unsigned long long)__double_as_longlong(v));
}

// residual (and Jacobians) of one observation at (cam, rho): rec = {F_ref | F_oth | e | r}, unscaled / unweighted
template <bool WANT_JAC>
__device__ bool id_observe(const IdProblem& P, const double* __restrict__ cam, const double* __restrict__ rho, int64_t o,
                           double* rec) {
  const int c = P.obs_cam[o], p = P.obs_pt[o], cr = P.pt_ref[p];
  const int grp = P.cam_group[c];
  const double* b = P.bearing + 3 * (size_t)p;
  const double ir = 1.0 / rho[p];
  const double pr[3] = {b[0] * ir, b[1] * ir, b[2] * ir};
  const double2 uv = P.obs_uv[o], si = P.obs_si[o];
  ObsLin ol;
  if (c == cr) {
    // InvReprojectionPoseError: the camera-frame point is b / rho itself
    const double zero_ext[6] = {0, 0, 0, 0, 0, 0};
    const double X[4] = {pr[0], pr[1], pr[2], 1.0};
    observe<WANT_JAC, false>(P.group_model[grp], zero_ext, P.intr + (size_t)grp * THEIA_MAX_INTRINSICS, X, uv.x, uv.y, si.x, si.y, ol);
    rec[26] = ol.r[0]; rec[27] = ol.r[1];
    if (WANT_JAC) {
      for (int k = 0; k < 24; ++k) rec[k] = 0.0;
      for (int a = 0; a < 2; ++a) rec[24 + a] = -ir * ((ol.Jx[4 * a] * pr[0] + ol.Jx[4 * a + 1] * pr[1]) + ol.Jx[4 * a + 2] * pr[2]);
    }
    return ol.valid;
  }
  const double* er = cam + 6 * (size_t)cr;
  const double mw[3] = {-er[3], -er[4], -er[5]};
  RotTerms tr;
  rotation_terms(mw, tr);                                                  // R_ref^T = R(-omega)
  const double pw[3] = {(tr.R[0] * pr[0] + tr.R[1] * pr[1]) + tr.R[2] * pr[2], (tr.R[3] * pr[0] + tr.R[4] * pr[1]) + tr.R[5] * pr[2],
                        (tr.R[6] * pr[0] + tr.R[7] * pr[1]) + tr.R[8] * pr[2]};
  const double X[4] = {pw[0] + er[0], pw[1] + er[1], pw[2] + er[2], 1.0};
  observe<WANT_JAC, false>(P.group_model[grp], cam + 6 * (size_t)c, P.intr + (size_t)grp * THEIA_MAX_INTRINSICS, X, uv.x, uv.y, si.x, si.y, ol);
  rec[26] = ol.r[0]; rec[27] = ol.r[1];
  if (WANT_JAC) {
    double M[9];
    rotation_dq_dw(mw, pr, tr, M);                                          // d(R(w) p)/dw at w = -omega; d/d omega = -M
    for (int a = 0; a < 2; ++a) {
      const double* jx = ol.Jx + 4 * a;                                     // d residual / d X (world point)
      rec[6 * a + 0] = jx[0]; rec[6 * a + 1] = jx[1]; rec[6 * a + 2] = jx[2];               // position of the reference camera
      for (int k = 0; k < 3; ++k) rec[6 * a + 3 + k] = -((jx[0] * M[k] + jx[1] * M[3 + k]) + jx[2] * M[6 + k]);
      for (int q = 0; q < 6; ++q) rec[12 + 6 * a + q] = ol.Jc[6 * a + q];                   // observing camera
      rec[24 + a] = -ir * ((jx[0] * pw[0] + jx[1] * pw[1]) + jx[2] * pw[2]);                // d/d rho = dX . R^T b (-1 / rho^2)
    }
  }
  return ol.valid;
}

// Linearisation: records (scaled by the Jacobi scaling and the loss corrector), cost.  mode 0: cost only;
// mode 1: records + cost; mode 2: squared column norms of the UNSCALED Jacobian (once per solve)
__global__ __launch_bounds__(256) void k_id_obs(IdProblem P, const double* __restrict__ cam, const double* __restrict__ rho, int mode,
                                                double* __restrict__ recs, double* __restrict__ scal, double* __restrict__ colsq_c,
                                                double* __restrict__ colsq_r) {
  const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= P.nobs) return;
  double rec[kRec];
  const bool ok = mode == 0 ? id_observe<false>(P, cam, rho, o, rec) : id_observe<true>(P, cam, rho, o, rec);
  if (!ok) atomicAdd(&scal[ID_INVALID], 1.0);
  double rho1;
  const double lc = loss_eval(P.loss_type, P.loss_width, rec[26] * rec[26] + rec[27] * rec[27], &rho1);
  atomicAdd(&scal[ID_COST], 0.5 * lc);
  if (mode == 0) return;
  const int c = P.obs_cam[o], p = P.obs_pt[o], cr = P.pt_ref[p];
  const double sr = sqrt(rho1);
  if (mode == 2) {
    for (int q = 0; q < 6; ++q) {
      if (P.cam_red[cr] >= 0 && !((P.cam_mask[cr] >> q) & 1)) atomicAdd(&colsq_c[6 * (size_t)cr + q], rho1 * (rec[q] * rec[q] + rec[6 + q] * rec[6 + q]));
      if (c != cr && P.cam_red[c] >= 0 && !((P.cam_mask[c] >> q) & 1)) atomicAdd(&colsq_c[6 * (size_t)c + q], rho1 * (rec[12 + q] * rec[12 + q] + rec[18 + q] * rec[18 + q]));
    }
    if (!P.pt_const[p]) atomicAdd(&colsq_r[p], rho1 * (rec[24] * rec[24] + rec[25] * rec[25]));
    return;
  }
  double* out = recs + (size_t)kRec * o;
  for (int a = 0; a < 2; ++a) {
    for (int q = 0; q < 6; ++q) {
      const bool fr = P.cam_red[cr] < 0 || ((P.cam_mask[cr] >> q) & 1), fo = P.cam_red[c] < 0 || ((P.cam_mask[c] >> q) & 1);
      out[6 * a + q] = fr ? 0.0 : sr * rec[6 * a + q] * P.scale_c[6 * (size_t)cr + q];
      out[12 + 6 * a + q] = fo ? 0.0 : sr * rec[12 + 6 * a + q] * P.scale_c[6 * (size_t)c + q];
    }
    out[24 + a] = P.pt_const[p] ? 0.0 : sr * rec[24 + a] * P.scale_r[p];
    out[26 + a] = sr * rec[26 + a];
  }
}

__device__ inline void add_block(double* S, int n, int ri, int rj, const double* A /* 2 x 6 */, const double* B /* 2 x 6 */, double sgn) {
  // S(ri, rj) += sgn * A^T B, lower triangle only (ri >= rj block-wise; within a diagonal block j <= i)
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) {
      if (ri == rj && j > i) continue;
      atomicAdd(&S[(size_t)(6 * ri + i) * n + 6 * rj + j], sgn * (A[i] * B[j] + A[6 + i] * B[6 + j]));
    }
}

// One thread per track: the camera-side normal equations of its rows, v = e^T e, the LM-damped inverse, and the
// rank-one Schur update over the cameras of the track.  S lower triangle, rhs, gc, colsq (scaled), vinv / g_rho out.
__global__ __launch_bounds__(64) void k_id_track(IdProblem P, const double* __restrict__ recs, const double* __restrict__ radius_p,
                                                 double* __restrict__ S, double* __restrict__ rhs, double* __restrict__ gc,
                                                 double* __restrict__ colsq, double* __restrict__ vinv, double* __restrict__ grho,
                                                 double* __restrict__ scal) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P.np) return;
  const int64_t b0 = P.pt_off[p], b1 = P.pt_off[p + 1];
  if (b1 == b0) return;
  const int cr = P.pt_ref[p], rr = P.cam_red[cr], n = P.n;
  double v = 0.0, g = 0.0, wref[6] = {0, 0, 0, 0, 0, 0};
  for (int64_t k = b0; k < b1; ++k) {
    const int o = P.pt_obs[k];
    const double* R = recs + (size_t)kRec * o;
    const int c = P.obs_cam[o], rc = P.cam_red[c];
    v += R[24] * R[24] + R[25] * R[25];
    g += R[24] * R[26] + R[25] * R[27];
    for (int q = 0; q < 6; ++q) wref[q] += R[q] * R[24] + R[6 + q] * R[25];
    // direct camera terms of this row: F^T F, F^T r
    if (rr >= 0) {
      add_block(S, n, rr, rr, R, R, 1.0);
      for (int q = 0; q < 6; ++q) {
        const double gq = R[q] * R[26] + R[6 + q] * R[27];
        atomicAdd(&rhs[6 * rr + q], gq); atomicAdd(&gc[6 * rr + q], gq);
        atomicAdd(&colsq[6 * rr + q], R[q] * R[q] + R[6 + q] * R[6 + q]);
      }
    }
    if (c != cr && rc >= 0) {
      add_block(S, n, rc, rc, R + 12, R + 12, 1.0);
      for (int q = 0; q < 6; ++q) {
        const double gq = R[12 + q] * R[26] + R[18 + q] * R[27];
        atomicAdd(&rhs[6 * rc + q], gq); atomicAdd(&gc[6 * rc + q], gq);
        atomicAdd(&colsq[6 * rc + q], R[12 + q] * R[12 + q] + R[18 + q] * R[18 + q]);
      }
      if (rr >= 0) { if (rc > rr) add_block(S, n, rc, rr, R + 12, R, 1.0); else add_block(S, n, rr, rc, R, R + 12, 1.0); }
    }
  }
  if (P.pt_const[p]) return;
  const double d = fmin(fmax(v, 1e-6), 1e32) / *radius_p;
  const double vi = 1.0 / (v + d);
  vinv[p] = vi; grho[p] = g;
  atomic_max_nonneg(&scal[ID_GMAX], fabs(g / P.scale_r[p]));
  // Schur complement of the track: S -= w w^T vi, rhs -= w vi g; w_ref = sum over all rows, w_c = F_c^T e of the row of c
  if (rr >= 0) {
    for (int i = 0; i < 6; ++i) {
      for (int j = 0; j <= i; ++j) atomicAdd(&S[(size_t)(6 * rr + i) * n + 6 * rr + j], -wref[i] * wref[j] * vi);
      atomicAdd(&rhs[6 * rr + i], -wref[i] * vi * g);
    }
  }
  for (int64_t k = b0; k < b1; ++k) {
    const int o = P.pt_obs[k];
    const int c = P.obs_cam[o], rc = P.cam_red[c];
    if (c == cr || rc < 0) continue;
    const double* R = recs + (size_t)kRec * o;
    double wa[6];
    for (int q = 0; q < 6; ++q) wa[q] = R[12 + q] * R[24] + R[18 + q] * R[25];
    for (int i = 0; i < 6; ++i) {
      for (int j = 0; j <= i; ++j) atomicAdd(&S[(size_t)(6 * rc + i) * n + 6 * rc + j], -wa[i] * wa[j] * vi);
      atomicAdd(&rhs[6 * rc + i], -wa[i] * vi * g);
    }
    if (rr >= 0)
      for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
          if (rc > rr) atomicAdd(&S[(size_t)(6 * rc + i) * n + 6 * rr + j], -wa[i] * wref[j] * vi);
          else atomicAdd(&S[(size_t)(6 * rr + i) * n + 6 * rc + j], -wref[i] * wa[j] * vi);
        }
    for (int64_t k2 = b0; k2 < k; ++k2) {
      const int o2 = P.pt_obs[k2];
      const int c2 = P.obs_cam[o2], rc2 = P.cam_red[c2];
      if (c2 == cr || rc2 < 0) continue;
      const double* R2 = recs + (size_t)kRec * o2;
      double wb[6];
      for (int q = 0; q < 6; ++q) wb[q] = R2[12 + q] * R2[24] + R2[18 + q] * R2[25];
      for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
          if (rc > rc2) atomicAdd(&S[(size_t)(6 * rc + i) * n + 6 * rc2 + j], -wa[i] * wb[j] * vi);
          else atomicAdd(&S[(size_t)(6 * rc2 + i) * n + 6 * rc + j], -wb[i] * wa[j] * vi);
        }
    }
  }
}

// LM diagonal of the camera blocks, the camera gradient max
__global__ void k_id_finalize(int n, const double* __restrict__ radius_p, double* __restrict__ S, const double* __restrict__ colsq,
                              const double* __restrict__ gc, const double* __restrict__ scale_red, double* __restrict__ scal) {
  double gmax = 0.0;
  for (int d = threadIdx.x; d < n; d += blockDim.x) {
    S[(size_t)d * n + d] += fmin(fmax(colsq[d], 1e-6), 1e32) / *radius_p;
    gmax = fmax(gmax, fabs(gc[d] / scale_red[d]));
  }
  atomic_max_nonneg(&scal[ID_GMAX], gmax);
}

// candidate cameras: x - y * scale; |step|^2 and |x+|^2 of the variable cameras
__global__ void k_id_cam_update(IdProblem P, const double* __restrict__ cam, const double* __restrict__ y, double* __restrict__ cand,
                                double* __restrict__ scal) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= P.nc) return;
  const int rc = P.cam_red[c];
  double st = 0.0, xn = 0.0;
  for (int q = 0; q < 6; ++q) {
    const double x = cam[6 * (size_t)c + q];
    double xp = x;
    if (rc >= 0) {
      if (!((P.cam_mask[c] >> q) & 1)) xp = x - y[6 * rc + q] * P.scale_c[6 * (size_t)c + q];
      st += (x - xp) * (x - xp); xn += xp * xp;
    }
    cand[6 * (size_t)c + q] = xp;
  }
  if (rc >= 0) { atomicAdd(&scal[ID_STEPSQ], st); atomicAdd(&scal[ID_XNORMSQ], xn); }
}

// back-substitution: y_rho = (g - w^T y_c) / (v + d), candidate rho, model cost change of the track's rows
__global__ __launch_bounds__(64) void k_id_back(IdProblem P, const double* __restrict__ recs, const double* __restrict__ y,
                                                const double* __restrict__ vinv, const double* __restrict__ grho,
                                                const double* __restrict__ rho, double* __restrict__ cand_rho, double* __restrict__ scal) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P.np) return;
  cand_rho[p] = rho[p];
  const int64_t b0 = P.pt_off[p], b1 = P.pt_off[p + 1];
  if (b1 == b0) return;
  const int cr = P.pt_ref[p], rr = P.cam_red[cr];
  double wy = 0.0;
  for (int64_t k = b0; k < b1; ++k) {
    const int o = P.pt_obs[k];
    const double* R = recs + (size_t)kRec * o;
    const int c = P.obs_cam[o], rc = P.cam_red[c];
    for (int a = 0; a < 2; ++a) {
      double m = 0.0;
      if (rr >= 0) for (int q = 0; q < 6; ++q) m += R[6 * a + q] * y[6 * rr + q];
      if (c != cr && rc >= 0) for (int q = 0; q < 6; ++q) m += R[12 + 6 * a + q] * y[6 * rc + q];
      wy += R[24 + a] * m;
    }
  }
  double yr = 0.0;
  if (!P.pt_const[p]) {
    yr = vinv[p] * (grho[p] - wy);
    const double x = rho[p], xp = x - yr * P.scale_r[p];
    cand_rho[p] = xp;
    atomicAdd(&scal[ID_STEPSQ], (x - xp) * (x - xp)); atomicAdd(&scal[ID_XNORMSQ], xp * xp);
  }
  double mcc = 0.0;
  for (int64_t k = b0; k < b1; ++k) {
    const int o = P.pt_obs[k];
    const double* R = recs + (size_t)kRec * o;
    const int c = P.obs_cam[o], rc = P.cam_red[c];
    for (int a = 0; a < 2; ++a) {
      double m = -R[24 + a] * yr;
      if (rr >= 0) for (int q = 0; q < 6; ++q) m -= R[6 * a + q] * y[6 * rr + q];
      if (c != cr && rc >= 0) for (int q = 0; q < 6; ++q) m -= R[12 + 6 * a + q] * y[6 * rc + q];
      mcc -= m * (R[26 + a] + m / 2.0);
    }
  }
  atomicAdd(&scal[ID_MCC], mcc);
}

__global__ void k_id_make_scale(int count, const double* __restrict__ colsq, double* __restrict__ scale) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) scale[i] = 1.0 / (1.0 + sqrt(colsq[i]));
}
__global__ void k_id_scale_red(IdProblem P, double* __restrict__ scale_red) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= P.nc || P.cam_red[c] < 0) return;
  for (int q = 0; q < 6; ++q) scale_red[6 * P.cam_red[c] + q] = ((P.cam_mask[c] >> q) & 1) ? 1.0 : P.scale_c[6 * (size_t)c + q];
}
__global__ void k_id_xnorm(IdProblem P, const double* __restrict__ cam, const double* __restrict__ rho, double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  double s = 0.0;
  if (i < P.nc && P.cam_red[i] >= 0) for (int q = 0; q < 6; ++q) s += cam[6 * (size_t)i + q] * cam[6 * (size_t)i + q];
  if (i < P.np && !P.pt_const[i] && P.pt_off[i + 1] > P.pt_off[i]) s += rho[i] * rho[i];
  if (s != 0.0) atomicAdd(out, s);
}

template <typename T>
struct Buf {
  T* p = nullptr;
  size_t n = 0;
  ~Buf() { if (p) (void)hipFree(p); }
  int alloc(size_t count) {
    n = count;
    if (hipMalloc((void**)&p, std::max<size_t>(1, count) * sizeof(T)) != hipSuccess)
      return set_error(THEIA_HIP_ERR_OUT_OF_MEMORY, "hipMalloc(%zu) failed", count * sizeof(T));
    return 0;
  }
  int upload(const std::vector<T>& h) {
    int rc = alloc(h.size());
    if (rc) return rc;
    if (!h.empty() && hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess)
      return set_error(THEIA_HIP_ERR_NO_DEVICE, "hipMemcpy failed");
    return 0;
  }
};

void trace_push(theia_ba_summary* S, double cost, double g, double step, double radius, int acc) {
  if (!S->trace_cost || S->trace_size >= S->trace_capacity) return;
  const int k = S->trace_size++;
  S->trace_cost[k] = cost;
  if (S->trace_gradient_max_norm) S->trace_gradient_max_norm[k] = g;
  if (S->trace_step_norm) S->trace_step_norm[k] = step;
  if (S->trace_radius) S->trace_radius[k] = radius;
  if (S->trace_accepted) S->trace_accepted[k] = acc;
}

}  // namespace

int ba_solve_inverse_depth(const theia_ba_problem* p, const theia_ba_options* o, theia_ba_summary* S) {
  const auto t_start = std::chrono::steady_clock::now();
  if (!p->point_ref_cam || !p->point_ref_bearing || !p->point_inverse_depth)
    return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "inverse depth: point_ref_cam / point_ref_bearing / point_inverse_depth missing");
  if (o->intrinsics_to_optimize != THEIA_INTR_NONE)
    return set_error(THEIA_HIP_ERR_UNSUPPORTED, "inverse depth together with intrinsics optimisation is not built");
  if ((p->cam_prior_mask && o->prior_mask) || p->obs_kind)
    return set_error(THEIA_HIP_ERR_UNSUPPORTED, "inverse depth together with camera / depth priors is not built");
  int rc = ensure_device();
  if (rc) return rc;
  const int nc = p->num_cameras, np = p->num_points, ng = p->num_groups;
  const int64_t nobs = p->num_obs;
  S->trace_size = 0; S->success = 0; S->num_iterations = 0; S->num_successful_steps = 0;
  S->time_linearize = S->time_solve_reduced = S->time_backsub = 0.0; S->time_kernel_linearize = 0.0; S->num_linearize_launches = 0;
  // ---- structure: blocks, observation lists by track
  std::vector<uint8_t> cam_mask(nc, 0), cam_used(nc, 0), pt_const(np, 0);
  for (int c = 0; c < nc; ++c) {
    unsigned m = 0;
    const int cc = p->cam_const ? p->cam_const[c] : 0;
    if ((cc & THEIA_CAM_CONST_POSITION) || o->constant_camera_position) m |= 0x07;
    if ((cc & THEIA_CAM_CONST_ORIENTATION) || o->constant_camera_orientation) m |= 0x38;
    cam_mask[c] = (uint8_t)m;
  }
  std::vector<int64_t> pt_off(np + 1, 0);
  for (int64_t i = 0; i < nobs; ++i) {
    const int pt = p->obs_pt[i];
    if (p->point_ref_cam[pt] < 0 || p->point_ref_cam[pt] >= nc) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "track %d has no reference view", pt);
    if (!(p->point_inverse_depth[pt] > 0.0)) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "track %d: inverse depth must be positive", pt);
    cam_used[p->obs_cam[i]] = 1; cam_used[p->point_ref_cam[pt]] = 1;
    pt_off[pt + 1]++;
  }
  for (int q = 0; q < np; ++q) { pt_const[q] = (p->point_const && p->point_const[q]) ? 1 : 0; pt_off[q + 1] += pt_off[q]; }
  std::vector<int> pt_obs(nobs);
  { std::vector<int64_t> fill(pt_off.begin(), pt_off.end() - 1); for (int64_t i = 0; i < nobs; ++i) pt_obs[fill[p->obs_pt[i]]++] = (int)i; }
  std::vector<int> cam_red(nc, -1);
  int ncv = 0;
  for (int c = 0; c < nc; ++c) if (cam_used[c] && (cam_mask[c] & 0x3f) != 0x3f) cam_red[c] = ncv++;
  const int n = 6 * ncv;
  std::vector<double> si(2 * (size_t)nobs, 1.0);
  if (p->obs_sqrt_info) std::memcpy(si.data(), p->obs_sqrt_info, sizeof(double) * 2 * nobs);
  // ---- device buffers
  Buf<double> d_intr, d_bearing, d_uv, d_si, d_cam[2], d_rho[2], d_scale_c, d_scale_r, d_scale_red, d_recs, d_red, d_vinv, d_grho, d_scal, d_radius, d_work, d_colsq_c, d_colsq_r;
  Buf<int> d_gm, d_cg, d_cred, d_pref, d_ocam, d_opt, d_pobs;
  Buf<uint8_t> d_cmask, d_pconst;
  Buf<int64_t> d_poff;
  std::vector<double> hintr(p->intrinsics, p->intrinsics + (size_t)THEIA_MAX_INTRINSICS * ng), hb(p->point_ref_bearing, p->point_ref_bearing + 3 * (size_t)np);
  std::vector<double> huv(p->obs_uv, p->obs_uv + 2 * nobs), hcam(p->cam_ext, p->cam_ext + 6 * (size_t)nc), hrho(p->point_inverse_depth, p->point_inverse_depth + np);
  std::vector<int> hgm(p->group_model, p->group_model + ng), hcg(p->cam_group, p->cam_group + nc), hpref(p->point_ref_cam, p->point_ref_cam + np);
  std::vector<int> hoc(p->obs_cam, p->obs_cam + nobs), hop(p->obs_pt, p->obs_pt + nobs);
  const size_t red_count = (size_t)n * n + 3 * (size_t)n;   // S | rhs | colsq | gc
  if ((rc = d_intr.upload(hintr)) || (rc = d_bearing.upload(hb)) || (rc = d_uv.upload(huv)) || (rc = d_si.upload(si)) ||
      (rc = d_cam[0].upload(hcam)) || (rc = d_cam[1].upload(hcam)) || (rc = d_rho[0].upload(hrho)) || (rc = d_rho[1].upload(hrho)) ||
      (rc = d_gm.upload(hgm)) || (rc = d_cg.upload(hcg)) || (rc = d_cred.upload(cam_red)) || (rc = d_pref.upload(hpref)) ||
      (rc = d_ocam.upload(hoc)) || (rc = d_opt.upload(hop)) || (rc = d_pobs.upload(pt_obs)) || (rc = d_cmask.upload(cam_mask)) ||
      (rc = d_pconst.upload(pt_const)) || (rc = d_poff.upload(pt_off)) || (rc = d_scale_c.alloc(6 * (size_t)nc)) ||
      (rc = d_scale_r.alloc(np)) || (rc = d_scale_red.alloc(std::max(1, n))) || (rc = d_recs.alloc((size_t)kRec * nobs)) ||
      (rc = d_red.alloc(std::max<size_t>(1, red_count))) || (rc = d_vinv.alloc(np)) || (rc = d_grho.alloc(np)) ||
      (rc = d_scal.alloc(ID_SCALARS)) || (rc = d_radius.alloc(1)) || (rc = d_work.alloc(dense_cholesky_workspace(std::max(1, n)))) ||
      (rc = d_colsq_c.alloc(6 * (size_t)nc)) || (rc = d_colsq_r.alloc(np)))
    return rc;
  IdProblem P;
  P.nc = nc; P.np = np; P.nobs = nobs; P.n = n;
  P.intr = d_intr.p; P.group_model = d_gm.p; P.cam_group = d_cg.p; P.cam_red = d_cred.p; P.cam_mask = d_cmask.p; P.pt_const = d_pconst.p;
  P.pt_ref = d_pref.p; P.bearing = d_bearing.p; P.obs_uv = reinterpret_cast<const double2*>(d_uv.p); P.obs_si = reinterpret_cast<const double2*>(d_si.p);
  P.obs_cam = d_ocam.p; P.obs_pt = d_opt.p; P.pt_off = d_poff.p; P.pt_obs = d_pobs.p; P.scale_c = d_scale_c.p; P.scale_r = d_scale_r.p;
  P.loss_type = o->loss_function_type; P.loss_width = o->robust_loss_width;
  CholPlan* plan = chol_plan_create(n, nullptr);
  struct PlanGuard { CholPlan* pl; ~PlanGuard() { chol_plan_destroy(pl); } } guard{plan};
  hipStream_t st = nullptr;
  const int ob = (int)((nobs + 255) / 256), tb = (np + 63) / 64, cb = (nc + 63) / 64;
  double* dS = d_red.p; double* drhs = dS + (size_t)n * n; double* dcolsq = drhs + n; double* dgc = dcolsq + n;
  double hs[ID_SCALARS];
  auto read_scal = [&]() -> int { HIP_TRY(hipMemcpy(hs, d_scal.p, sizeof(hs), hipMemcpyDeviceToHost)); return 0; };
  auto cost_at = [&](const double* cam, const double* rho, double* cost, bool* ok) -> int {
    HIP_TRY(hipMemsetAsync(d_scal.p, 0, sizeof(hs), st));
    if (nobs) k_id_obs<<<ob, 256, 0, st>>>(P, cam, rho, 0, nullptr, d_scal.p, nullptr, nullptr);
    int r2 = read_scal();
    if (r2) return r2;
    *cost = hs[ID_COST]; *ok = hs[ID_INVALID] == 0.0 && std::isfinite(hs[ID_COST]);
    return 0;
  };
  // ---- Jacobi scaling (once): squared column norms of the unscaled Jacobian at the start
  HIP_TRY(hipMemsetAsync(d_colsq_c.p, 0, sizeof(double) * 6 * nc, st));
  HIP_TRY(hipMemsetAsync(d_colsq_r.p, 0, sizeof(double) * np, st));
  HIP_TRY(hipMemsetAsync(d_scal.p, 0, sizeof(hs), st));
  if (nobs) k_id_obs<<<ob, 256, 0, st>>>(P, d_cam[0].p, d_rho[0].p, 2, nullptr, d_scal.p, d_colsq_c.p, d_colsq_r.p);
  k_id_make_scale<<<(6 * nc + 255) / 256, 256, 0, st>>>(6 * nc, d_colsq_c.p, d_scale_c.p);
  k_id_make_scale<<<(np + 255) / 256, 256, 0, st>>>(np, d_colsq_r.p, d_scale_r.p);
  if (n) k_id_scale_red<<<cb, 64, 0, st>>>(P, d_scale_red.p);
  // |x| of the variable blocks
  auto xnorm_of = [&](const double* cam, const double* rho, double* out) -> int {
    HIP_TRY(hipMemsetAsync(d_scal.p, 0, sizeof(hs), st));
    k_id_xnorm<<<(std::max(nc, np) + 255) / 256, 256, 0, st>>>(P, cam, rho, d_scal.p);
    int r2 = read_scal();
    if (r2) return r2;
    *out = std::sqrt(hs[0]);
    return 0;
  };
  int cur = 0;
  double radius = 1e4, decrease_factor = 2.0, x_cost = 0.0, gmax = 0.0, x_norm = 0.0;
  auto linearize = [&]() -> int {   // records, reduced system with the damping of `radius`, gradient max
    HIP_TRY(hipMemcpyAsync(d_radius.p, &radius, sizeof(double), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(d_red.p, 0, sizeof(double) * std::max<size_t>(1, red_count), st));
    HIP_TRY(hipMemsetAsync(d_scal.p, 0, sizeof(hs), st));
    if (nobs) k_id_obs<<<ob, 256, 0, st>>>(P, d_cam[cur].p, d_rho[cur].p, 1, d_recs.p, d_scal.p, nullptr, nullptr);
    if (np) k_id_track<<<tb, 64, 0, st>>>(P, d_recs.p, d_radius.p, dS, drhs, dgc, dcolsq, d_vinv.p, d_grho.p, d_scal.p);
    if (n) k_id_finalize<<<1, 256, 0, st>>>(n, d_radius.p, dS, dcolsq, dgc, d_scale_red.p, d_scal.p);
    int r2 = read_scal();
    if (r2) return r2;
    x_cost = hs[ID_COST]; gmax = hs[ID_GMAX];
    S->num_linearize_launches++;
    return 0;
  };
  if ((rc = xnorm_of(d_cam[0].p, d_rho[0].p, &x_norm))) return rc;
  if ((rc = linearize())) return rc;
  S->initial_cost = x_cost;
  if (hs[ID_INVALID] > 0.0 || !std::isfinite(x_cost)) { S->termination_type = THEIA_TERM_FAILURE; S->final_cost = x_cost; return 0; }
  double minimum_cost = x_cost;
  bool step_successful = true;
  int iter = 0, invalid_steps = 0, term = THEIA_TERM_NO_CONVERGENCE;
  trace_push(S, x_cost, gmax, 0.0, radius, 1);
  bool fresh = true;   // the reduced system in d_red belongs to the current radius
  while (true) {
    const double elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    if (elapsed >= o->max_solver_time_in_seconds && iter > 0) { term = THEIA_TERM_NO_CONVERGENCE; break; }
    if (iter >= o->max_num_iterations) { term = THEIA_TERM_NO_CONVERGENCE; break; }
    if (step_successful && gmax <= o->gradient_tolerance) { term = THEIA_TERM_CONVERGENCE; break; }
    if (radius <= 1e-32) { term = THEIA_TERM_CONVERGENCE; break; }
    ++iter;
    if (!fresh && (rc = linearize())) return rc;   // same point, new radius: the damping sits inside the Schur complement
    fresh = false;
    const int nxt = 1 - cur;
    HIP_TRY(hipMemsetAsync(d_scal.p, 0, sizeof(hs), st));
    if (n) chol_plan_solve(plan, dS, n, drhs, d_work.p, d_scal.p + ID_NOTPD, st);
    if (nc) k_id_cam_update<<<cb, 64, 0, st>>>(P, d_cam[cur].p, drhs, d_cam[nxt].p, d_scal.p);
    if (np) k_id_back<<<tb, 64, 0, st>>>(P, d_recs.p, drhs, d_vinv.p, d_grho.p, d_rho[cur].p, d_rho[nxt].p, d_scal.p);
    if ((rc = read_scal())) return rc;
    const double mcc = hs[ID_MCC], stepsq = hs[ID_STEPSQ], xnormsq = hs[ID_XNORMSQ];
    const bool solved = hs[ID_NOTPD] == 0.0 && std::isfinite(mcc) && std::isfinite(stepsq);
    if (!(solved && mcc > 0.0)) {
      if (++invalid_steps >= 5) { term = THEIA_TERM_FAILURE; break; }
      radius /= decrease_factor; decrease_factor *= 2.0; step_successful = false;
      trace_push(S, x_cost, gmax, 0.0, radius, 0);
      continue;
    }
    invalid_steps = 0;
    double cand_cost; bool cok;
    if ((rc = cost_at(d_cam[nxt].p, d_rho[nxt].p, &cand_cost, &cok))) return rc;
    if (!cok) cand_cost = std::numeric_limits<double>::max();
    const double step_norm = std::sqrt(stepsq);
    if (step_norm <= o->parameter_tolerance * (x_norm + o->parameter_tolerance)) { trace_push(S, cand_cost, gmax, step_norm, radius, 0); term = THEIA_TERM_CONVERGENCE; break; }
    const double cost_change = x_cost - cand_cost;
    if (std::fabs(cost_change) <= o->function_tolerance * x_cost) { trace_push(S, cand_cost, gmax, step_norm, radius, 0); term = THEIA_TERM_CONVERGENCE; break; }
    const double rho_q = cost_change / mcc;
    if (rho_q > 1e-3) {
      cur = nxt;
      x_norm = std::sqrt(xnormsq);
      radius = std::min(o->max_trust_region_radius, radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * rho_q - 1.0, 3)));
      decrease_factor = 2.0; step_successful = true;
      S->num_successful_steps++;
      if ((rc = linearize())) return rc;
      fresh = true;
      if (x_cost < minimum_cost) minimum_cost = x_cost;
      trace_push(S, x_cost, gmax, step_norm, radius, 1);
    } else {
      radius /= decrease_factor; decrease_factor *= 2.0; step_successful = false;
      trace_push(S, cand_cost, gmax, step_norm, radius, 0);
    }
  }
  S->num_iterations = iter; S->termination_type = term; S->success = term != THEIA_TERM_FAILURE;
  S->final_cost = minimum_cost;
  HIP_TRY(hipMemcpy(p->cam_ext, d_cam[cur].p, sizeof(double) * 6 * nc, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(p->point_inverse_depth, d_rho[cur].p, sizeof(double) * np, hipMemcpyDeviceToHost));
  S->solve_time_in_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
  S->setup_time_in_seconds = 0.0;
  return 0;
}

}  // namespace thip
