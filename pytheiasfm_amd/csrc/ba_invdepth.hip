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
// rules of ba_solver.hip run by the host between launches.
// The intrinsics of the observing camera's group are the third camera-side block of a row (2 x 10 slots, the free subset of
// BundleAdjustmentOptions::intrinsics_to_optimize filled; the reduced system holds the groups after the cameras, 10 slots
// each, bounds by box projection of the step as in the main path), and AddViewPriors' position / gravity / orientation
// rows (bundle_adjuster.cc:289-313, no loss function) enter the camera blocks directly.  No depth rows, inner iterations
// or sharding in this mode.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

#include "ba_device.h"
#include "ba_kernels.h"
#include "ba_priors.h"
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

constexpr int kRec = 48;   // record of one observation: F_ref (2 x 6) | F_oth (2 x 6) | e (2) | r (2) | F_k (2 x 10)
constexpr int kRecK = 28;  // offset of the intrinsics block
constexpr int kKW = THEIA_MAX_INTRINSICS;
constexpr int kMaxTrackGroups = 8;   // variable intrinsics groups one track may be observed through

struct IdProblem {
  int nc, np;
  int64_t nobs;
  int n;                          // reduced system size: 6 * (#variable cameras) + 10 * (#variable groups)
  int ng, ncam6;                  // groups; 6 * (#variable cameras) = offset of the first group block
  const int* grp_red;             // [ng] reduced group index or -1
  const unsigned* grp_free;       // [ng] bit q = parameter q is free
  const int* grp_k;               // [ng] parameters of the group's model
  const double* scale_i;          // [ng][10] Jacobi scaling of the intrinsics columns
  int n_priors;                   // camera priors of variable cameras
  const int* prior_cam; const int* prior_kind; const double* prior_vec; const double* prior_info;
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

enum { ID_COST = 0, ID_INVALID = 1, ID_NOTPD = 2, ID_MCC = 3, ID_STEPSQ = 4, ID_XNORMSQ = 5, ID_GMAX = 6, ID_FIXED = 7, ID_SCALARS = 8 };

__device__ inline void atomic_max_nonneg(double* addr, double v) {
  atomicMax(reinterpret_cast<unsigned long long*>(addr), (unsigned long long)__double_as_longlong(v));
}

// residual (and Jacobians) of one observation at (cam, rho, intr): rec = {F_ref | F_oth | e | r | F_k}, unscaled / unweighted
template <bool WANT_JAC>
__device__ bool id_observe(const IdProblem& P, const double* __restrict__ cam, const double* __restrict__ rho,
                           const double* __restrict__ intr, int64_t o, double* rec) {
  const int c = P.obs_cam[o], p = P.obs_pt[o], cr = P.pt_ref[p];
  const int grp = P.cam_group[c];
  const double* b = P.bearing + 3 * (size_t)p;
  const double ir = 1.0 / rho[p];
  const double pr[3] = {b[0] * ir, b[1] * ir, b[2] * ir};
  const double2 uv = P.obs_uv[o], si = P.obs_si[o];
  ObsLinK ol;
  if (c == cr) {
    // InvReprojectionPoseError: the camera-frame point is b / rho itself
    const double zero_ext[6] = {0, 0, 0, 0, 0, 0};
    const double X[4] = {pr[0], pr[1], pr[2], 1.0};
    observe<WANT_JAC, WANT_JAC>(P.group_model[grp], zero_ext, intr + (size_t)grp * kKW, X, uv.x, uv.y, si.x, si.y, ol);
    rec[26] = ol.r[0]; rec[27] = ol.r[1];
    if (WANT_JAC) {
      for (int k = 0; k < 24; ++k) rec[k] = 0.0;
      for (int a = 0; a < 2; ++a) rec[24 + a] = -ir * ((ol.Jx[4 * a] * pr[0] + ol.Jx[4 * a + 1] * pr[1]) + ol.Jx[4 * a + 2] * pr[2]);
      for (int k = 0; k < 2 * kKW; ++k) rec[kRecK + k] = ol.Jk[k];
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
  observe<WANT_JAC, WANT_JAC>(P.group_model[grp], cam + 6 * (size_t)c, intr + (size_t)grp * kKW, X, uv.x, uv.y, si.x, si.y, ol);
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
    for (int k = 0; k < 2 * kKW; ++k) rec[kRecK + k] = ol.Jk[k];
  }
  return ol.valid;
}

// Linearisation: records (scaled by the Jacobi scaling and the loss corrector), cost.  mode 0: cost only;
// mode 1: records + cost; mode 2: squared column norms of the UNSCALED Jacobian (once per solve)
__global__ __launch_bounds__(256) void k_id_obs(IdProblem P, const double* __restrict__ cam, const double* __restrict__ rho,
                                                const double* __restrict__ intr, int mode,
                                                double* __restrict__ recs, double* __restrict__ scal, double* __restrict__ colsq_c,
                                                double* __restrict__ colsq_r, double* __restrict__ colsq_i) {
  const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= P.nobs) return;
  double rec[kRec];
  const bool ok = mode == 0 ? id_observe<false>(P, cam, rho, intr, o, rec) : id_observe<true>(P, cam, rho, intr, o, rec);
  if (!ok) atomicAdd(&scal[ID_INVALID], 1.0);
  double rho1;
  const double lc = loss_eval(P.loss_type, P.loss_width, rec[26] * rec[26] + rec[27] * rec[27], &rho1);
  atomicAdd(&scal[ID_COST], 0.5 * lc);
  if (mode == 0) return;
  const int c = P.obs_cam[o], p = P.obs_pt[o], cr = P.pt_ref[p], g = P.cam_group[c];
  const int gr = P.grp_red[g];
  const unsigned fm = gr >= 0 ? P.grp_free[g] : 0u;
  const double sr = sqrt(rho1);
  if (mode == 2) {
    for (int q = 0; q < 6; ++q) {
      if (P.cam_red[cr] >= 0 && !((P.cam_mask[cr] >> q) & 1)) atomicAdd(&colsq_c[6 * (size_t)cr + q], rho1 * (rec[q] * rec[q] + rec[6 + q] * rec[6 + q]));
      if (c != cr && P.cam_red[c] >= 0 && !((P.cam_mask[c] >> q) & 1)) atomicAdd(&colsq_c[6 * (size_t)c + q], rho1 * (rec[12 + q] * rec[12 + q] + rec[18 + q] * rec[18 + q]));
    }
    for (int q = 0; q < kKW; ++q)
      if ((fm >> q) & 1u) atomicAdd(&colsq_i[(size_t)g * kKW + q], rho1 * (rec[kRecK + q] * rec[kRecK + q] + rec[kRecK + kKW + q] * rec[kRecK + kKW + q]));
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
    for (int q = 0; q < kKW; ++q)
      out[kRecK + kKW * a + q] = ((fm >> q) & 1u) ? sr * rec[kRecK + kKW * a + q] * P.scale_i[(size_t)g * kKW + q] : 0.0;
    out[24 + a] = P.pt_const[p] ? 0.0 : sr * rec[24 + a] * P.scale_r[p];
    out[26 + a] = sr * rec[26 + a];
  }
}

// S(r0 .., c0 ..) += sgn * A^T B for a 2 x nr block A and a 2 x ncw block B (row strides sa / sb), r0 >= c0: the lower
// triangle of the reduced system; on the diagonal (r0 == c0) only j <= i
// (called by every lane of the track's wave: lane `ln` takes the entries e = ln, ln + 64, ... of the block)
__device__ inline void add_block(int ln, double* S, int n, int r0, int nr, int c0, int ncw, const double* A, int sa, const double* B, int sb, double sgn) {
  for (int e = ln; e < nr * ncw; e += 64) {
    const int i = e / ncw, j = e - i * ncw;
    if (r0 == c0 && j > i) continue;
    const double v = A[i] * B[j] + A[sa + i] * B[sb + j];
    if (v != 0.0) atomicAdd(&S[(size_t)(r0 + i) * n + c0 + j], sgn * v);
  }
}
// S(r0 .., c0 ..) += sgn * a b^T (vectors), r0 >= c0
__device__ inline void add_outer(int ln, double* S, int n, int r0, int nr, int c0, int ncw, const double* a, const double* b, double sgn) {
  for (int e = ln; e < nr * ncw; e += 64) {
    const int i = e / ncw, j = e - i * ncw;
    if (r0 == c0 && j > i) continue;
    const double v = a[i] * b[j];
    if (v != 0.0) atomicAdd(&S[(size_t)(r0 + i) * n + c0 + j], sgn * v);
  }
}
// the same with the two blocks in either order of their offsets
__device__ inline void add_outer_any(int ln, double* S, int n, int ra, int na, const double* a, int rb, int nb, const double* b, double sgn) {
  if (ra >= rb) add_outer(ln, S, n, ra, na, rb, nb, a, b, sgn); else add_outer(ln, S, n, rb, nb, ra, na, b, a, sgn);
}
__device__ inline void add_block_any(int ln, double* S, int n, int ra, int na, const double* A, int sa, int rb, int nb, const double* B, int sb, double sgn) {
  if (ra >= rb) add_block(ln, S, n, ra, na, rb, nb, A, sa, B, sb, sgn); else add_block(ln, S, n, rb, nb, ra, na, B, sb, A, sa, sgn);
}

// S(ra .., rb ..) (lower triangle: the block with the larger offset gives the rows) += A^T B - vi a b^T for two 2 x 6 blocks
// A, B (rows 6 apart) and their Schur vectors a, b: the direct and the Schur term of a camera pair in ONE atomic per entry
__device__ inline void add_pair(int ln, double* S, int n, int oa, const double* A, const double* a, int ob, const double* B, const double* b, double vi) {
  if (ln >= 36) return;
  const int i = ln / 6, j = ln - 6 * i;
  if (oa == ob) {
    if (j > i) return;
    const double v = (A[i] * B[j] + A[6 + i] * B[6 + j]) - vi * (a[i] * b[j]);
    if (v != 0.0) atomicAdd(&S[(size_t)(oa + i) * n + ob + j], v);
  } else if (oa > ob) {
    const double v = (A[i] * B[j] + A[6 + i] * B[6 + j]) - vi * (a[i] * b[j]);
    if (v != 0.0) atomicAdd(&S[(size_t)(oa + i) * n + ob + j], v);
  } else {
    const double v = (B[i] * A[j] + B[6 + i] * A[6 + j]) - vi * (b[i] * a[j]);
    if (v != 0.0) atomicAdd(&S[(size_t)(ob + i) * n + oa + j], v);
  }
}

// One WAVE per track (a thread per track left the device almost empty: 60 000 threads issuing ~1800 atomics each in a row;
// every lane carries the track's sums -- cheap, and each is needed for the outer products -- and the lanes share the entries
// of every block that goes to S): the camera-side normal equations of its rows (reference camera, observing camera, intrinsics
// group of the observing camera), v = e^T e, the LM-damped inverse, and the rank-one Schur update over the blocks of the
// track.  The direct term J^T J and the Schur term of a camera block / camera pair are folded into one atomic per entry
// (the reference camera's diagonal block: one instead of one per row).  S lower triangle, rhs, gc, colsq (scaled), vinv / g_rho out.
__global__ __launch_bounds__(64) void k_id_track(IdProblem P, const double* __restrict__ recs, const double* __restrict__ radius_p,
                                                 double* __restrict__ S, double* __restrict__ rhs, double* __restrict__ gc,
                                                 double* __restrict__ colsq, double* __restrict__ vinv, double* __restrict__ grho,
                                                 double* __restrict__ scal) {
  const int p = blockIdx.x, ln = threadIdx.x;
  if (p >= P.np) return;
  const int64_t b0 = P.pt_off[p], b1 = P.pt_off[p + 1];
  if (b1 == b0) return;
  const int cr = P.pt_ref[p], rr = P.cam_red[cr], n = P.n;
  const int oref = 6 * rr;
  double v = 0.0, g = 0.0, wref[6] = {0, 0, 0, 0, 0, 0};
  int gslot[kMaxTrackGroups], ngs = 0;           // reduced indices of the variable groups this track is seen through
  double wg[kMaxTrackGroups][kKW];               // sum over the rows of a group of F_k^T e
  // this lane's entry (i, j) of the reference camera's diagonal block (lower triangle) and, for lanes < 6, its rhs / colsq row
  const int ei = ln / 6, ej = ln - 6 * (ln / 6);
  const bool ref_entry = rr >= 0 && ln < 36 && ej <= ei;
  double dref = 0.0, gref = 0.0, cref = 0.0;
  for (int64_t k = b0; k < b1; ++k) {
    const int o = P.pt_obs[k];
    const double* R = recs + (size_t)kRec * o;
    const int c = P.obs_cam[o], rc = P.cam_red[c], gr = P.grp_red[P.cam_group[c]];
    const int ooth = 6 * rc, ogrp = P.ncam6 + kKW * gr;
    v += R[24] * R[24] + R[25] * R[25];
    g += R[24] * R[26] + R[25] * R[27];
    for (int q = 0; q < 6; ++q) wref[q] += R[q] * R[24] + R[6 + q] * R[25];
    if (ref_entry) dref += R[ei] * R[ej] + R[6 + ei] * R[6 + ej];
    if (rr >= 0 && ln < 6) { gref += R[ln] * R[26] + R[6 + ln] * R[27]; cref += R[ln] * R[ln] + R[6 + ln] * R[6 + ln]; }
    if (gr >= 0) {
      const double* K = R + kRecK;
      add_block(ln, S, n, ogrp, kKW, ogrp, kKW, K, kKW, K, kKW, 1.0);
      if (rr >= 0) add_block(ln, S, n, ogrp, kKW, oref, 6, K, kKW, R, 6, 1.0);               // group blocks lie below the cameras
      if (c != cr && rc >= 0) add_block(ln, S, n, ogrp, kKW, ooth, 6, K, kKW, R + 12, 6, 1.0);
      int s = 0;
      while (s < ngs && gslot[s] != gr) ++s;
      if (s == ngs) { gslot[ngs++] = gr; for (int q = 0; q < kKW; ++q) wg[s][q] = 0.0; }   // (the host rejects > kMaxTrackGroups)
      for (int q = 0; q < kKW; ++q) {
        if (q == ln) {
          const double gq = K[q] * R[26] + K[kKW + q] * R[27];
          if (gq != 0.0) { atomicAdd(&rhs[ogrp + q], gq); atomicAdd(&gc[ogrp + q], gq); }
          const double cq = K[q] * K[q] + K[kKW + q] * K[kKW + q];
          if (cq != 0.0) atomicAdd(&colsq[ogrp + q], cq);
        }
        wg[s][q] += K[q] * R[24] + K[kKW + q] * R[25];
      }
    }
  }
  const bool pconst = P.pt_const[p] != 0;
  double vi = 0.0;        // a constant track keeps the direct terms only
  if (!pconst) {
    const double d = fmin(fmax(v, 1e-6), 1e32) / *radius_p;
    vi = 1.0 / (v + d);
    if (ln == 0) { vinv[p] = vi; grho[p] = g; atomic_max_nonneg(&scal[ID_GMAX], fabs(g / P.scale_r[p])); }
  }
  // the reference camera: J^T J of all rows minus the Schur term, J^T r minus w vi g
  if (rr >= 0) {
    if (ref_entry) { const double val = dref - vi * (wref[ei] * wref[ej]); if (val != 0.0) atomicAdd(&S[(size_t)(oref + ei) * n + oref + ej], val); }
    if (ln < 6) {
      double wl = 0.0;
      for (int q = 0; q < 6; ++q) if (q == ln) wl = wref[q];
      atomicAdd(&rhs[oref + ln], gref - wl * vi * g); atomicAdd(&gc[oref + ln], gref); atomicAdd(&colsq[oref + ln], cref);
    }
  }
  if (!pconst)
    for (int s = 0; s < ngs; ++s) {
      const int og = P.ncam6 + kKW * gslot[s];
      add_outer(ln, S, n, og, kKW, og, kKW, wg[s], wg[s], -vi);
      for (int s2 = 0; s2 < s; ++s2) add_outer_any(ln, S, n, og, kKW, wg[s], P.ncam6 + kKW * gslot[s2], kKW, wg[s2], -vi);
      if (rr >= 0) add_outer(ln, S, n, og, kKW, oref, 6, wg[s], wref, -vi);
      for (int q = 0; q < kKW; ++q) if (q == ln && wg[s][q] != 0.0) atomicAdd(&rhs[og + q], -wg[s][q] * vi * g);
    }
  for (int64_t k = b0; k < b1; ++k) {
    const int o = P.pt_obs[k];
    const int c = P.obs_cam[o], rc = P.cam_red[c];
    if (c == cr || rc < 0) continue;
    const double* R = recs + (size_t)kRec * o;
    double wa[6];
    for (int q = 0; q < 6; ++q) wa[q] = R[12 + q] * R[24] + R[18 + q] * R[25];
    add_pair(ln, S, n, 6 * rc, R + 12, wa, 6 * rc, R + 12, wa, vi);                    // own diagonal block: direct + Schur
    if (rr >= 0) add_pair(ln, S, n, 6 * rc, R + 12, wa, oref, R, wref, vi);            // with the reference camera
    if (ln < 6) {
      double wl = 0.0;
      for (int q = 0; q < 6; ++q) if (q == ln) wl = wa[q];
      const double gq = R[12 + ln] * R[26] + R[18 + ln] * R[27];
      atomicAdd(&rhs[6 * rc + ln], gq - wl * vi * g); atomicAdd(&gc[6 * rc + ln], gq);
      atomicAdd(&colsq[6 * rc + ln], R[12 + ln] * R[12 + ln] + R[18 + ln] * R[18 + ln]);
    }
    if (pconst) continue;
    for (int s = 0; s < ngs; ++s) add_outer(ln, S, n, P.ncam6 + kKW * gslot[s], kKW, 6 * rc, 6, wg[s], wa, -vi);
    for (int64_t k2 = b0; k2 < k; ++k2) {
      const int o2 = P.pt_obs[k2];
      const int c2 = P.obs_cam[o2], rc2 = P.cam_red[c2];
      if (c2 == cr || rc2 < 0) continue;
      const double* R2 = recs + (size_t)kRec * o2;
      double wb[6];
      for (int q = 0; q < 6; ++q) wb[q] = R2[12 + q] * R2[24] + R2[18 + q] * R2[25];
      add_outer_any(ln, S, n, 6 * rc, 6, wa, 6 * rc2, 6, wb, -vi);
    }
  }
}

// Camera priors of the variable cameras (AddViewPriors: no loss function).  mode 0: cost; 1: cost + normal equations
// (J scaled by the Jacobi scaling, frozen columns masked); 2: squared column norms of the unscaled rows;
// 3: model cost change of the step y
__global__ void k_id_priors(IdProblem P, const double* __restrict__ cam, int mode, double* __restrict__ S, double* __restrict__ rhs,
                            double* __restrict__ gc, double* __restrict__ colsq, double* __restrict__ colsq_c,
                            const double* __restrict__ y, double* __restrict__ scal) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= P.n_priors) return;
  const int c = P.prior_cam[k], rc = P.cam_red[c];
  double r[3], J[18];
  camera_prior(P.prior_kind[k], cam + 6 * (size_t)c, P.prior_vec + 3 * (size_t)k, P.prior_info + 9 * (size_t)k, mode != 0 && rc >= 0, r, J);
  // a prior on a constant camera is a residual block without variable parameters: Ceres' fixed cost (reported, not minimised)
  if (mode == 0 || mode == 1) atomicAdd(&scal[rc >= 0 ? ID_COST : ID_FIXED], 0.5 * ((r[0] * r[0] + r[1] * r[1]) + r[2] * r[2]));
  if (mode == 0 || rc < 0) return;
  if (mode == 2) {
    for (int q = 0; q < 6; ++q)
      if (!((P.cam_mask[c] >> q) & 1)) atomicAdd(&colsq_c[6 * (size_t)c + q], (J[q] * J[q] + J[6 + q] * J[6 + q]) + J[12 + q] * J[12 + q]);
    return;
  }
  for (int a = 0; a < 3; ++a)
    for (int q = 0; q < 6; ++q) J[6 * a + q] = ((P.cam_mask[c] >> q) & 1) ? 0.0 : J[6 * a + q] * P.scale_c[6 * (size_t)c + q];
  if (mode == 3) {
    double mcc = 0.0;
    for (int a = 0; a < 3; ++a) {
      double m = 0.0;
      for (int q = 0; q < 6; ++q) m -= J[6 * a + q] * y[6 * rc + q];
      mcc -= m * (r[a] + m / 2.0);
    }
    atomicAdd(&scal[ID_MCC], mcc);
    return;
  }
  const int n = P.n, o = 6 * rc;
  for (int i = 0; i < 6; ++i) {
    for (int j = 0; j <= i; ++j) {
      const double v = (J[i] * J[j] + J[6 + i] * J[6 + j]) + J[12 + i] * J[12 + j];
      if (v != 0.0) atomicAdd(&S[(size_t)(o + i) * n + o + j], v);
    }
    const double gq = (J[i] * r[0] + J[6 + i] * r[1]) + J[12 + i] * r[2];
    atomicAdd(&rhs[o + i], gq); atomicAdd(&gc[o + i], gq);
    atomicAdd(&colsq[o + i], (J[i] * J[i] + J[6 + i] * J[6 + i]) + J[12 + i] * J[12 + i]);
  }
}

// LM diagonal of the camera-side blocks, the camera-side gradient max
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

// bundle_adjuster.cc:406-427 parameter bounds: the step is projected onto the box (as the main path)
__device__ inline void id_project_to_bounds(int model, double* k) {
  if (k[0] < 1.0) k[0] = 1.0;
  if (model == THEIA_CAM_DOUBLE_SPHERE) { k[5] = fmin(1.0, fmax(-1.0, k[5])); k[6] = fmin(1.0, fmax(0.0, k[6])); }
  if (model == THEIA_CAM_EXTENDED_UNIFIED) { k[5] = fmin(1.0, fmax(0.0, k[5])); k[6] = fmax(0.1, k[6]); }
}
// candidate intrinsics of the variable groups; |step|^2 and |x+|^2 over the parameters of their models
__global__ void k_id_intr_update(IdProblem P, const double* __restrict__ intr, const double* __restrict__ y, double* __restrict__ cand,
                                 double* __restrict__ scal) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= P.ng) return;
  const int gr = P.grp_red[g];
  double kk[kKW];
  for (int q = 0; q < kKW; ++q) {
    kk[q] = intr[(size_t)g * kKW + q];
    if (gr >= 0 && ((P.grp_free[g] >> q) & 1u)) kk[q] -= y[P.ncam6 + kKW * gr + q] * P.scale_i[(size_t)g * kKW + q];
  }
  if (gr >= 0) id_project_to_bounds(P.group_model[g], kk);
  double st = 0.0, xn = 0.0;
  for (int q = 0; q < kKW; ++q) {
    const double x = intr[(size_t)g * kKW + q];
    cand[(size_t)g * kKW + q] = kk[q];
    if (gr >= 0 && q < P.grp_k[g]) { st += (x - kk[q]) * (x - kk[q]); xn += kk[q] * kk[q]; }
  }
  if (gr >= 0) { atomicAdd(&scal[ID_STEPSQ], st); atomicAdd(&scal[ID_XNORMSQ], xn); }
}

// the camera-side part of a row's model residual: [F_ref | F_oth | F_k] y
__device__ inline double id_row_camside(const IdProblem& P, const double* R, int a, int rr, int rc, bool other, int gr, const double* y) {
  double m = 0.0;
  if (rr >= 0) for (int q = 0; q < 6; ++q) m += R[6 * a + q] * y[6 * rr + q];
  if (other && rc >= 0) for (int q = 0; q < 6; ++q) m += R[12 + 6 * a + q] * y[6 * rc + q];
  if (gr >= 0) for (int q = 0; q < kKW; ++q) m += R[kRecK + kKW * a + q] * y[P.ncam6 + kKW * gr + q];
  return m;
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
    const int c = P.obs_cam[o], rc = P.cam_red[c], gr = P.grp_red[P.cam_group[c]];
    for (int a = 0; a < 2; ++a) wy += R[24 + a] * id_row_camside(P, R, a, rr, rc, c != cr, gr, y);
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
    const int c = P.obs_cam[o], rc = P.cam_red[c], gr = P.grp_red[P.cam_group[c]];
    for (int a = 0; a < 2; ++a) {
      const double m = -R[24 + a] * yr - id_row_camside(P, R, a, rr, rc, c != cr, gr, y);
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
  if (c < P.nc && P.cam_red[c] >= 0)
    for (int q = 0; q < 6; ++q) scale_red[6 * P.cam_red[c] + q] = ((P.cam_mask[c] >> q) & 1) ? 1.0 : P.scale_c[6 * (size_t)c + q];
  if (c < P.ng && P.grp_red[c] >= 0)
    for (int q = 0; q < kKW; ++q) scale_red[P.ncam6 + kKW * P.grp_red[c] + q] = ((P.grp_free[c] >> q) & 1u) ? P.scale_i[(size_t)c * kKW + q] : 1.0;
}
__global__ void k_id_xnorm(IdProblem P, const double* __restrict__ cam, const double* __restrict__ rho, const double* __restrict__ intr,
                           double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  double s = 0.0;
  if (i < P.nc && P.cam_red[i] >= 0) for (int q = 0; q < 6; ++q) s += cam[6 * (size_t)i + q] * cam[6 * (size_t)i + q];
  if (i < P.ng && P.grp_red[i] >= 0) for (int q = 0; q < P.grp_k[i]; ++q) s += intr[(size_t)i * kKW + q] * intr[(size_t)i * kKW + q];
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
  int upload(const std::vector<T>& h, hipStream_t st) {   // (the vector outlives the copy: synchronised here)
    int rc = alloc(h.size());
    if (rc) return rc;
    if (!h.empty() && (hipMemcpyAsync(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, st) != hipSuccess ||
                       hipStreamSynchronize(st) != hipSuccess))
      return set_error(THEIA_HIP_ERR_NO_DEVICE, "hipMemcpy failed");
    return 0;
  }
};

// free parameters of a model under an OptimizeIntrinsicsType mask (GetSubsetFromOptimizeIntrinsicsType of every
// *_camera_model.cc), the sizes of the eight models, the bounds of bundle_adjuster.cc:406-427: as ba_solver.hip
unsigned id_free_mask(int model, int opt) {
  const bool noskew = (model == THEIA_CAM_FOV || model == THEIA_CAM_DIVISION_UNDISTORTION);
  unsigned m = 0;
  if (opt & THEIA_INTR_FOCAL_LENGTH) m |= 1u << 0;
  if (opt & THEIA_INTR_ASPECT_RATIO) m |= 1u << 1;
  if ((opt & THEIA_INTR_SKEW) && !noskew) m |= 1u << 2;
  if (opt & THEIA_INTR_PRINCIPAL_POINTS) m |= noskew ? (3u << 2) : (3u << 3);
  if (opt & THEIA_INTR_RADIAL_DISTORTION) {
    switch (model) {
      case THEIA_CAM_PINHOLE: case THEIA_CAM_DOUBLE_SPHERE: case THEIA_CAM_EXTENDED_UNIFIED: case THEIA_CAM_ORTHOGRAPHIC: m |= 3u << 5; break;
      case THEIA_CAM_PINHOLE_RADIAL_TANGENTIAL: m |= 7u << 5; break;
      case THEIA_CAM_FISHEYE: m |= 15u << 5; break;
      case THEIA_CAM_FOV: case THEIA_CAM_DIVISION_UNDISTORTION: m |= 1u << 4; break;
    }
  }
  if ((opt & THEIA_INTR_TANGENTIAL_DISTORTION) && model == THEIA_CAM_PINHOLE_RADIAL_TANGENTIAL) m |= 3u << 8;
  return m;
}
int id_intrinsics_size(int model) {
  static const int K[8] = {7, 10, 9, 5, 5, 7, 7, 7};
  return (model >= 0 && model < 8) ? K[model] : 0;
}
void id_project_to_bounds_host(int model, double* k) {
  if (k[0] < 1.0) k[0] = 1.0;
  if (model == THEIA_CAM_DOUBLE_SPHERE) { k[5] = std::min(1.0, std::max(-1.0, k[5])); k[6] = std::min(1.0, std::max(0.0, k[6])); }
  if (model == THEIA_CAM_EXTENDED_UNIFIED) { k[5] = std::min(1.0, std::max(0.0, k[5])); k[6] = std::max(0.1, k[6]); }
}

void trace_push(theia_ba_summary* S, double cost, double g, double step, double radius, int acc) {
  if (!S->trace_cost || S->trace_size >= S->trace_capacity) return;
  const int k = S->trace_size++;
  S->trace_cost[k] = cost;
  if (S->trace_gradient_max_norm) S->trace_gradient_max_norm[k] = g;
  if (S->trace_step_norm) S->trace_step_norm[k] = step;
  if (S->trace_radius) S->trace_radius[k] = radius;
  if (S->trace_accepted) S->trace_accepted[k] = acc;
}


// The inverse-depth problem resident on the device: structure, observation arrays, reduced-system plan (create), parameters
// (upload_parameters), the LM loop (run) and the way back (download).  theia_hip_ba_solve is create + run + download; the
// handle API (theia_hip_ba_create / reset_parameters / run / download / destroy) keeps the object between calls.
struct IdHandle {
  int nc = 0, np = 0, ng = 0, n = 0, ncam6 = 0, ngv = 0, npri = 0;
  int64_t nobs = 0;
  std::vector<int> grp_red, h_group_model;
  std::vector<uint8_t> h_pt_observed;   // tracks with at least one observation: their inverse depth must be positive
  struct StreamGuard { hipStream_t s = nullptr; ~StreamGuard() { if (s) (void)hipStreamDestroy(s); } } sg;
  hipStream_t st = nullptr;
  Buf<double> d_intr[2], d_scale_i, d_colsq_i, d_pvec, d_pinfo, d_bearing, d_uv, d_si, d_cam[2], d_rho[2], d_scale_c, d_scale_r, d_scale_red, d_recs, d_red, d_vinv, d_grho, d_scal, d_radius, d_work, d_colsq_c, d_colsq_r;
  Buf<int> d_gm, d_cg, d_cred, d_pref, d_ocam, d_opt, d_pobs, d_gred, d_gk, d_pcam, d_pkind;
  Buf<unsigned> d_gfree;
  Buf<uint8_t> d_cmask, d_pconst;
  Buf<int64_t> d_poff;
  IdProblem P;
  CholPlan* plan = nullptr;
  size_t red_count = 0;
  int cur = 0;
  theia_ba_options created_with;   // the structural options a later run must repeat
  ~IdHandle() { if (st) (void)hipStreamSynchronize(st); chol_plan_destroy(plan); }
  int create(const theia_ba_problem* p, const theia_ba_options* o);
  int upload_parameters(const theia_ba_problem* p);
  int run(const theia_ba_options* o, theia_ba_summary* S);
  int download(theia_ba_problem* p);
  // theia_hip_ba_snapshot_parameters / restore_parameters: the current parameters <-> a device-side copy
  Buf<double> snap_cam, snap_rho, snap_intr;
  bool has_snapshot = false;
  int snapshot();
  int restore();
};

int IdHandle::create(const theia_ba_problem* p, const theia_ba_options* o) {
  if (!p->point_ref_cam || !p->point_ref_bearing || !p->point_inverse_depth)
    return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "inverse depth: point_ref_cam / point_ref_bearing / point_inverse_depth missing");
  if (p->obs_kind)
    return set_error(THEIA_HIP_ERR_UNSUPPORTED, "inverse depth together with depth-prior rows is not built");
  int rc = ensure_device();
  if (rc) return rc;
  nc = p->num_cameras; np = p->num_points; ng = p->num_groups;
  nobs = p->num_obs;
  created_with = *o;
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
  h_pt_observed.assign(np, 0);
  for (int pt = 0; pt < np; ++pt) h_pt_observed[pt] = pt_off[pt + 1] > 0;
  for (int q = 0; q < np; ++q) { pt_const[q] = (p->point_const && p->point_const[q]) ? 1 : 0; pt_off[q + 1] += pt_off[q]; }
  std::vector<int> pt_obs(nobs);
  { std::vector<int64_t> fill(pt_off.begin(), pt_off.end() - 1); for (int64_t i = 0; i < nobs; ++i) pt_obs[fill[p->obs_pt[i]]++] = (int)i; }
  std::vector<int> cam_red(nc, -1);
  int ncv = 0;
  for (int c = 0; c < nc; ++c) if (cam_used[c] && (cam_mask[c] & 0x3f) != 0x3f) cam_red[c] = ncv++;
  // intrinsics groups (bundle_adjuster.cc:382-460): variable on the subset of intrinsics_to_optimize unless the caller marked
  // the group constant or none of its cameras observes a track
  grp_red.assign(ng, -1);
  std::vector<int> grp_k(ng, 0);
  std::vector<unsigned> grp_free(ng, 0u);
  std::vector<uint8_t> grp_used(ng, 0);
  for (int64_t i = 0; i < nobs; ++i) grp_used[p->cam_group[p->obs_cam[i]]] = 1;
  ngv = 0;
  for (int g = 0; g < ng; ++g) {
    grp_k[g] = id_intrinsics_size(p->group_model[g]);
    const unsigned fm = id_free_mask(p->group_model[g], o->intrinsics_to_optimize);
    if ((p->group_const && p->group_const[g]) || fm == 0 || !grp_used[g]) continue;
    grp_red[g] = ngv++; grp_free[g] = fm;
  }
  if (ngv) {   // k_id_track keeps the per-group sums of a track in a fixed table
    for (int q = 0; q < np; ++q) {
      int seen[kMaxTrackGroups], ns = 0;
      for (int64_t k = pt_off[q]; k < pt_off[q + 1]; ++k) {
        const int gr = grp_red[p->cam_group[p->obs_cam[pt_obs[k]]]];
        if (gr < 0) continue;
        int s2 = 0;
        while (s2 < ns && seen[s2] != gr) ++s2;
        if (s2 < ns) continue;
        if (ns == kMaxTrackGroups)
          return set_error(THEIA_HIP_ERR_UNSUPPORTED, "inverse depth: track %d is observed through more than %d variable intrinsics groups", q, kMaxTrackGroups);
        seen[ns++] = gr;
      }
    }
  }
  ncam6 = 6 * ncv; n = ncam6 + kKW * ngv;
  // camera priors of the views in the problem (AddViewPriors)
  std::vector<int> prior_cam, prior_kind;
  std::vector<double> prior_vec, prior_info;
  if (p->cam_prior_mask && o->prior_mask) {
    const double* vecs[3] = {p->cam_position_prior, p->cam_gravity_prior, p->cam_orientation_prior};
    const double* infos[3] = {p->cam_position_prior_sqrt_info, p->cam_gravity_prior_sqrt_info, p->cam_orientation_prior_sqrt_info};
    for (int c = 0; c < nc; ++c)
      for (int k = 0; k < 3; ++k) {
        const int bit = 1 << k;
        if (!cam_used[c] || !(p->cam_prior_mask[c] & bit) || !(o->prior_mask & bit) || !vecs[k] || !infos[k]) continue;
        prior_cam.push_back(c); prior_kind.push_back(bit);
        prior_vec.insert(prior_vec.end(), vecs[k] + 3 * (size_t)c, vecs[k] + 3 * (size_t)c + 3);
        prior_info.insert(prior_info.end(), infos[k] + 9 * (size_t)c, infos[k] + 9 * (size_t)c + 9);
      }
  }
  npri = (int)prior_cam.size();
  std::vector<double> si(2 * (size_t)nobs, 1.0);
  if (p->obs_sqrt_info) std::memcpy(si.data(), p->obs_sqrt_info, sizeof(double) * 2 * nobs);
  // ---- a stream of this object (non-blocking: the entry points stay callable from a thread pool; the legacy stream would
  // serialise against every other stream of the process)
  HIP_TRY(hipStreamCreateWithFlags(&sg.s, hipStreamNonBlocking));
  st = sg.s;
  h_group_model.assign(p->group_model, p->group_model + ng);
  std::vector<double> hb(p->point_ref_bearing, p->point_ref_bearing + 3 * (size_t)np);
  std::vector<double> huv(p->obs_uv, p->obs_uv + 2 * nobs);
  std::vector<int> hgm(p->group_model, p->group_model + ng), hcg(p->cam_group, p->cam_group + nc), hpref(p->point_ref_cam, p->point_ref_cam + np);
  std::vector<int> hoc(p->obs_cam, p->obs_cam + nobs), hop(p->obs_pt, p->obs_pt + nobs);
  red_count = (size_t)n * n + 3 * (size_t)n;   // S | rhs | colsq | gc
  if ((rc = d_intr[0].alloc((size_t)kKW * ng)) || (rc = d_intr[1].alloc((size_t)kKW * ng)) || (rc = d_scale_i.alloc((size_t)kKW * ng)) ||
      (rc = d_colsq_i.alloc((size_t)kKW * ng)) || (rc = d_gred.upload(grp_red, st)) || (rc = d_gk.upload(grp_k, st)) ||
      (rc = d_gfree.upload(grp_free, st)) || (rc = d_pcam.upload(prior_cam, st)) || (rc = d_pkind.upload(prior_kind, st)) ||
      (rc = d_pvec.upload(prior_vec, st)) || (rc = d_pinfo.upload(prior_info, st)) || (rc = d_bearing.upload(hb, st)) || (rc = d_uv.upload(huv, st)) || (rc = d_si.upload(si, st)) ||
      (rc = d_cam[0].alloc(6 * (size_t)nc)) || (rc = d_cam[1].alloc(6 * (size_t)nc)) || (rc = d_rho[0].alloc(np)) || (rc = d_rho[1].alloc(np)) ||
      (rc = d_gm.upload(hgm, st)) || (rc = d_cg.upload(hcg, st)) || (rc = d_cred.upload(cam_red, st)) || (rc = d_pref.upload(hpref, st)) ||
      (rc = d_ocam.upload(hoc, st)) || (rc = d_opt.upload(hop, st)) || (rc = d_pobs.upload(pt_obs, st)) || (rc = d_cmask.upload(cam_mask, st)) ||
      (rc = d_pconst.upload(pt_const, st)) || (rc = d_poff.upload(pt_off, st)) || (rc = d_scale_c.alloc(6 * (size_t)nc)) ||
      (rc = d_scale_r.alloc(np)) || (rc = d_scale_red.alloc(std::max(1, n))) || (rc = d_recs.alloc((size_t)kRec * nobs)) ||
      (rc = d_red.alloc(std::max<size_t>(1, red_count))) || (rc = d_vinv.alloc(np)) || (rc = d_grho.alloc(np)) ||
      (rc = d_scal.alloc(ID_SCALARS)) || (rc = d_radius.alloc(1)) || (rc = d_work.alloc(dense_cholesky_workspace(std::max(1, n)))) ||
      (rc = d_colsq_c.alloc(6 * (size_t)nc)) || (rc = d_colsq_r.alloc(np)))
    return rc;
  P.nc = nc; P.np = np; P.nobs = nobs; P.n = n;
  P.ng = ng; P.ncam6 = ncam6; P.grp_red = d_gred.p; P.grp_free = d_gfree.p; P.grp_k = d_gk.p; P.scale_i = d_scale_i.p;
  P.n_priors = npri; P.prior_cam = d_pcam.p; P.prior_kind = d_pkind.p; P.prior_vec = d_pvec.p; P.prior_info = d_pinfo.p;
  P.group_model = d_gm.p; P.cam_group = d_cg.p; P.cam_red = d_cred.p; P.cam_mask = d_cmask.p; P.pt_const = d_pconst.p;
  P.pt_ref = d_pref.p; P.bearing = d_bearing.p; P.obs_uv = reinterpret_cast<const double2*>(d_uv.p); P.obs_si = reinterpret_cast<const double2*>(d_si.p);
  P.obs_cam = d_ocam.p; P.obs_pt = d_opt.p; P.pt_off = d_poff.p; P.pt_obs = d_pobs.p; P.scale_c = d_scale_c.p; P.scale_r = d_scale_r.p;
  P.loss_type = o->loss_function_type; P.loss_width = o->robust_loss_width;
  {
    // tile co-visibility of the reduced system (64-wide tiles): every block a track writes couples the slots of its reference
    // camera, its observing cameras and their intrinsics groups pairwise; a camera's / group's own tiles always belong
    const int nt = (std::max(1, n) + 63) / 64;
    std::vector<uint8_t> adj((size_t)nt * nt, 0);
    std::vector<int> tl;
    auto push_slot = [&](int o, int len) { for (int t = o / 64; t <= (o + len - 1) / 64; ++t) tl.push_back(t); };
    for (int q = 0; q < np; ++q) {
      if (pt_off[q + 1] == pt_off[q]) continue;
      tl.clear();
      if (cam_red[p->point_ref_cam[q]] >= 0) push_slot(6 * cam_red[p->point_ref_cam[q]], 6);
      for (int64_t k = pt_off[q]; k < pt_off[q + 1]; ++k) {
        const int c = p->obs_cam[pt_obs[k]];
        if (cam_red[c] >= 0) push_slot(6 * cam_red[c], 6);
        if (grp_red[p->cam_group[c]] >= 0) push_slot(ncam6 + kKW * grp_red[p->cam_group[c]], kKW);
      }
      std::sort(tl.begin(), tl.end());
      tl.erase(std::unique(tl.begin(), tl.end()), tl.end());
      for (int a : tl) for (int b : tl) adj[(size_t)a * nt + b] = 1;
    }
    for (int c = 0; c < nc; ++c) if (cam_red[c] >= 0) { tl.clear(); push_slot(6 * cam_red[c], 6); for (int a : tl) for (int b : tl) adj[(size_t)a * nt + b] = 1; }
    for (int g = 0; g < ng; ++g) if (grp_red[g] >= 0) { tl.clear(); push_slot(ncam6 + kKW * grp_red[g], kKW); for (int a : tl) for (int b : tl) adj[(size_t)a * nt + b] = 1; }
    plan = chol_plan_create(n, (n > 0 && !getenv("THEIA_HIP_INVDEPTH_DENSE")) ? adj.data() : nullptr);
  }
  return upload_parameters(p);
}

// extrinsics, intrinsics (projected onto their bounds: the initial point, as Ceres does) and inverse depths of `p`
int IdHandle::upload_parameters(const theia_ba_problem* p) {
  if (p->num_cameras != nc || p->num_points != np || p->num_groups != ng)
    return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "problem shape differs from the handle's");
  if (!p->point_inverse_depth) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "inverse depth: point_inverse_depth missing");
  for (int pt = 0; pt < np; ++pt)   // what create() checks: a reset must not smuggle in what a fresh handle would refuse
    if (h_pt_observed[pt] && !(p->point_inverse_depth[pt] > 0.0)) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "track %d: inverse depth must be positive", pt);
  std::vector<double> hintr(p->intrinsics, p->intrinsics + (size_t)THEIA_MAX_INTRINSICS * ng);
  for (int g = 0; g < ng; ++g) if (grp_red[g] >= 0) id_project_to_bounds_host(h_group_model[g], &hintr[(size_t)g * kKW]);
  for (int k = 0; k < 2; ++k) {
    if (ng) HIP_TRY(hipMemcpyAsync(d_intr[k].p, hintr.data(), sizeof(double) * hintr.size(), hipMemcpyHostToDevice, st));
    if (nc) HIP_TRY(hipMemcpyAsync(d_cam[k].p, p->cam_ext, sizeof(double) * 6 * (size_t)nc, hipMemcpyHostToDevice, st));
    if (np) HIP_TRY(hipMemcpyAsync(d_rho[k].p, p->point_inverse_depth, sizeof(double) * (size_t)np, hipMemcpyHostToDevice, st));
  }
  HIP_TRY(hipStreamSynchronize(st));   // the sources are the caller's (and a local) arrays
  cur = 0;
  return 0;
}

int IdHandle::run(const theia_ba_options* o, theia_ba_summary* S) {
  const auto t_start = std::chrono::steady_clock::now();
  if (o->intrinsics_to_optimize != created_with.intrinsics_to_optimize || o->constant_camera_position != created_with.constant_camera_position ||
      o->constant_camera_orientation != created_with.constant_camera_orientation || o->prior_mask != created_with.prior_mask)
    return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "structural options differ from the ones the handle was created with");
  int rc = 0;
  S->trace_size = 0; S->success = 0; S->num_iterations = 0; S->num_successful_steps = 0;
  S->time_linearize = S->time_solve_reduced = S->time_backsub = 0.0; S->time_kernel_linearize = 0.0; S->num_linearize_launches = 0;
  P.loss_type = o->loss_function_type; P.loss_width = o->robust_loss_width;
  if (cur != 0) {   // the state of a previous run: continue from it in slot 0
    if (ng) HIP_TRY(hipMemcpyAsync(d_intr[0].p, d_intr[1].p, sizeof(double) * (size_t)kKW * ng, hipMemcpyDeviceToDevice, st));
    if (nc) HIP_TRY(hipMemcpyAsync(d_cam[0].p, d_cam[1].p, sizeof(double) * 6 * (size_t)nc, hipMemcpyDeviceToDevice, st));
    if (np) HIP_TRY(hipMemcpyAsync(d_rho[0].p, d_rho[1].p, sizeof(double) * (size_t)np, hipMemcpyDeviceToDevice, st));
    cur = 0;
  }
  const int ob = (int)((nobs + 255) / 256), tb = (np + 63) / 64, cb = (std::max(nc, ng) + 63) / 64, pb = (npri + 63) / 64;
  double* dS = d_red.p; double* drhs = dS + (size_t)n * n; double* dcolsq = drhs + n; double* dgc = dcolsq + n;
  double hs[ID_SCALARS];
  auto read_scal = [&]() -> int {
    HIP_TRY(hipMemcpyAsync(hs, d_scal.p, sizeof(hs), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return 0;
  };
  auto cost_at = [&](const double* cam, const double* rho, const double* intr, double* cost, bool* ok) -> int {
    HIP_TRY(hipMemsetAsync(d_scal.p, 0, sizeof(hs), st));
    if (nobs) k_id_obs<<<ob, 256, 0, st>>>(P, cam, rho, intr, 0, nullptr, d_scal.p, nullptr, nullptr, nullptr);
    if (npri) k_id_priors<<<pb, 64, 0, st>>>(P, cam, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, d_scal.p);
    int r2 = read_scal();
    if (r2) return r2;
    *cost = hs[ID_COST]; *ok = hs[ID_INVALID] == 0.0 && std::isfinite(hs[ID_COST]);
    return 0;
  };
  // ---- Jacobi scaling (once): squared column norms of the unscaled Jacobian at the start
  HIP_TRY(hipMemsetAsync(d_colsq_c.p, 0, sizeof(double) * 6 * nc, st));
  HIP_TRY(hipMemsetAsync(d_colsq_r.p, 0, sizeof(double) * np, st));
  HIP_TRY(hipMemsetAsync(d_scal.p, 0, sizeof(hs), st));
  HIP_TRY(hipMemsetAsync(d_colsq_i.p, 0, sizeof(double) * std::max<size_t>(1, (size_t)kKW * ng), st));
  if (nobs) k_id_obs<<<ob, 256, 0, st>>>(P, d_cam[0].p, d_rho[0].p, d_intr[0].p, 2, nullptr, d_scal.p, d_colsq_c.p, d_colsq_r.p, d_colsq_i.p);
  if (npri) k_id_priors<<<pb, 64, 0, st>>>(P, d_cam[0].p, 2, nullptr, nullptr, nullptr, nullptr, d_colsq_c.p, nullptr, d_scal.p);
  k_id_make_scale<<<(6 * nc + 255) / 256, 256, 0, st>>>(6 * nc, d_colsq_c.p, d_scale_c.p);
  if (ng) k_id_make_scale<<<(kKW * ng + 255) / 256, 256, 0, st>>>(kKW * ng, d_colsq_i.p, d_scale_i.p);
  k_id_make_scale<<<(np + 255) / 256, 256, 0, st>>>(np, d_colsq_r.p, d_scale_r.p);
  if (n) k_id_scale_red<<<cb, 64, 0, st>>>(P, d_scale_red.p);
  // |x| of the variable blocks
  auto xnorm_of = [&](const double* cam, const double* rho, const double* intr, double* out) -> int {
    HIP_TRY(hipMemsetAsync(d_scal.p, 0, sizeof(hs), st));
    k_id_xnorm<<<(std::max(std::max(nc, np), ng) + 255) / 256, 256, 0, st>>>(P, cam, rho, intr, d_scal.p);
    int r2 = read_scal();
    if (r2) return r2;
    *out = std::sqrt(hs[0]);
    return 0;
  };
  double radius = 1e4, decrease_factor = 2.0, x_cost = 0.0, gmax = 0.0, x_norm = 0.0, fixed_cost = 0.0;
  auto linearize = [&]() -> int {   // records, reduced system with the damping of `radius`, gradient max
    HIP_TRY(hipMemcpyAsync(d_radius.p, &radius, sizeof(double), hipMemcpyHostToDevice, st));
    // only the tiles the plan's assembly and factorisation touch are cleared (the whole n x n buffer with the dense plan)
    if (!(n > 0 && chol_plan_clear(plan, dS, n, st, drhs, 3 * (size_t)n)))
      HIP_TRY(hipMemsetAsync(d_red.p, 0, sizeof(double) * std::max<size_t>(1, red_count), st));
    HIP_TRY(hipMemsetAsync(d_scal.p, 0, sizeof(hs), st));
    if (nobs) k_id_obs<<<ob, 256, 0, st>>>(P, d_cam[cur].p, d_rho[cur].p, d_intr[cur].p, 1, d_recs.p, d_scal.p, nullptr, nullptr, nullptr);
    if (np) k_id_track<<<np, 64, 0, st>>>(P, d_recs.p, d_radius.p, dS, drhs, dgc, dcolsq, d_vinv.p, d_grho.p, d_scal.p);
    if (npri) k_id_priors<<<pb, 64, 0, st>>>(P, d_cam[cur].p, 1, dS, drhs, dgc, dcolsq, nullptr, nullptr, d_scal.p);
    if (n) k_id_finalize<<<1, 256, 0, st>>>(n, d_radius.p, dS, dcolsq, dgc, d_scale_red.p, d_scal.p);
    int r2 = read_scal();
    if (r2) return r2;
    x_cost = hs[ID_COST]; gmax = hs[ID_GMAX]; fixed_cost = hs[ID_FIXED];
    S->num_linearize_launches++;
    return 0;
  };
  if ((rc = xnorm_of(d_cam[0].p, d_rho[0].p, d_intr[0].p, &x_norm))) return rc;
  if ((rc = linearize())) return rc;
  S->initial_cost = x_cost + fixed_cost;
  if (hs[ID_INVALID] > 0.0 || !std::isfinite(x_cost)) { S->termination_type = THEIA_TERM_FAILURE; S->final_cost = x_cost + fixed_cost; return 0; }
  double minimum_cost = x_cost;
  bool step_successful = true;
  int iter = 0, invalid_steps = 0, term = THEIA_TERM_NO_CONVERGENCE;
  trace_push(S, x_cost + fixed_cost, gmax, 0.0, radius, 1);
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
    if (nc) k_id_cam_update<<<(nc + 63) / 64, 64, 0, st>>>(P, d_cam[cur].p, drhs, d_cam[nxt].p, d_scal.p);
    if (ng) k_id_intr_update<<<(ng + 63) / 64, 64, 0, st>>>(P, d_intr[cur].p, drhs, d_intr[nxt].p, d_scal.p);
    if (np) k_id_back<<<tb, 64, 0, st>>>(P, d_recs.p, drhs, d_vinv.p, d_grho.p, d_rho[cur].p, d_rho[nxt].p, d_scal.p);
    if (npri) k_id_priors<<<pb, 64, 0, st>>>(P, d_cam[cur].p, 3, nullptr, nullptr, nullptr, nullptr, nullptr, drhs, d_scal.p);
    if ((rc = read_scal())) return rc;
    const double mcc = hs[ID_MCC], stepsq = hs[ID_STEPSQ], xnormsq = hs[ID_XNORMSQ];
    const bool solved = hs[ID_NOTPD] == 0.0 && std::isfinite(mcc) && std::isfinite(stepsq);
    if (!(solved && mcc > 0.0)) {
      if (++invalid_steps >= 5) { term = THEIA_TERM_FAILURE; break; }
      radius /= decrease_factor; decrease_factor *= 2.0; step_successful = false;
      trace_push(S, x_cost + fixed_cost, gmax, 0.0, radius, 0);
      continue;
    }
    invalid_steps = 0;
    double cand_cost; bool cok;
    if ((rc = cost_at(d_cam[nxt].p, d_rho[nxt].p, d_intr[nxt].p, &cand_cost, &cok))) return rc;
    if (!cok) cand_cost = std::numeric_limits<double>::max();
    const double step_norm = std::sqrt(stepsq);
    if (step_norm <= o->parameter_tolerance * (x_norm + o->parameter_tolerance)) { trace_push(S, cand_cost + fixed_cost, gmax, step_norm, radius, 0); term = THEIA_TERM_CONVERGENCE; break; }
    const double cost_change = x_cost - cand_cost;
    if (std::fabs(cost_change) <= o->function_tolerance * x_cost) { trace_push(S, cand_cost + fixed_cost, gmax, step_norm, radius, 0); term = THEIA_TERM_CONVERGENCE; break; }
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
      trace_push(S, x_cost + fixed_cost, gmax, step_norm, radius, 1);
    } else {
      radius /= decrease_factor; decrease_factor *= 2.0; step_successful = false;
      trace_push(S, cand_cost + fixed_cost, gmax, step_norm, radius, 0);
    }
  }
  S->num_iterations = iter; S->termination_type = term; S->success = term != THEIA_TERM_FAILURE;
  S->final_cost = minimum_cost + fixed_cost;
  HIP_TRY(hipStreamSynchronize(st));
  S->solve_time_in_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
  S->setup_time_in_seconds = 0.0;
  return 0;
}

int IdHandle::snapshot() {
  int rc;
  if ((rc = snap_cam.alloc((size_t)6 * std::max(1, nc))) || (rc = snap_rho.alloc((size_t)std::max(1, np))) || (rc = snap_intr.alloc((size_t)kKW * std::max(1, ng)))) return rc;
  if (nc) HIP_TRY(hipMemcpyAsync(snap_cam.p, d_cam[cur].p, sizeof(double) * 6 * (size_t)nc, hipMemcpyDeviceToDevice, st));
  if (np) HIP_TRY(hipMemcpyAsync(snap_rho.p, d_rho[cur].p, sizeof(double) * (size_t)np, hipMemcpyDeviceToDevice, st));
  if (ng) HIP_TRY(hipMemcpyAsync(snap_intr.p, d_intr[cur].p, sizeof(double) * (size_t)kKW * ng, hipMemcpyDeviceToDevice, st));
  has_snapshot = true;
  return 0;
}
int IdHandle::restore() {
  if (!has_snapshot) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "restore_parameters without a snapshot");
  for (int k = 0; k < 2; ++k) {
    if (nc) HIP_TRY(hipMemcpyAsync(d_cam[k].p, snap_cam.p, sizeof(double) * 6 * (size_t)nc, hipMemcpyDeviceToDevice, st));
    if (np) HIP_TRY(hipMemcpyAsync(d_rho[k].p, snap_rho.p, sizeof(double) * (size_t)np, hipMemcpyDeviceToDevice, st));
    if (ng) HIP_TRY(hipMemcpyAsync(d_intr[k].p, snap_intr.p, sizeof(double) * (size_t)kKW * ng, hipMemcpyDeviceToDevice, st));
  }
  cur = 0;
  return 0;
}

int IdHandle::download(theia_ba_problem* p) {
  if (p->num_cameras != nc || p->num_points != np || p->num_groups != ng)
    return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "problem shape differs from the handle's");
  if (ngv) HIP_TRY(hipMemcpyAsync(p->intrinsics, d_intr[cur].p, sizeof(double) * (size_t)kKW * ng, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(p->cam_ext, d_cam[cur].p, sizeof(double) * 6 * nc, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(p->point_inverse_depth, d_rho[cur].p, sizeof(double) * np, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return 0;
}

}  // namespace

int ba_solve_inverse_depth(const theia_ba_problem* p, const theia_ba_options* o, theia_ba_summary* S) {
  const auto t_start = std::chrono::steady_clock::now();
  IdHandle h;
  int rc = h.create(p, o);
  if (rc || (rc = h.run(o, S))) return rc;
  if (S->termination_type == THEIA_TERM_FAILURE && S->num_iterations == 0) return 0;   // invalid start: the parameters stay as given
  rc = h.download(const_cast<theia_ba_problem*>(p));
  S->solve_time_in_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
  return rc;
}

// the handle API's view of the object (ba_solver.hip)
int id_handle_create(const theia_ba_problem* p, const theia_ba_options* o, void** out) {
  IdHandle* h = new (std::nothrow) IdHandle();
  if (!h) return set_error(THEIA_HIP_ERR_OUT_OF_MEMORY, "out of host memory");
  const int rc = h->create(p, o);
  if (rc) { delete h; return rc; }
  *out = h;
  return 0;
}
int id_handle_reset(void* h, const theia_ba_problem* p) { return static_cast<IdHandle*>(h)->upload_parameters(p); }
int id_handle_run(void* h, const theia_ba_options* o, theia_ba_summary* S) { return static_cast<IdHandle*>(h)->run(o, S); }
int id_handle_download(void* h, theia_ba_problem* p) { return static_cast<IdHandle*>(h)->download(p); }
void id_handle_destroy(void* h) { delete static_cast<IdHandle*>(h); }
int id_handle_snapshot(void* h) { return static_cast<IdHandle*>(h)->snapshot(); }
int id_handle_restore(void* h) { return static_cast<IdHandle*>(h)->restore(); }
void id_handle_plan_info(void* h, int32_t* n, int32_t* k3_levels, double* k3_flops) {
  IdHandle* H = static_cast<IdHandle*>(h);
  if (n) *n = H->n;
  if (k3_levels) *k3_levels = H->plan ? chol_plan_levels(H->plan) : 0;
  if (k3_flops) *k3_flops = H->plan ? chol_plan_flops(H->plan) : 0.0;
}

}  // namespace thip
