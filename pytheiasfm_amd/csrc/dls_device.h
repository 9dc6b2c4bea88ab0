// DLS-PnP on the device (the reference: sfm/pose/dls_pnp.cc:67-200).  Two stages per minimal problem:
//
//   stage A  one WORKGROUP of 192 threads per problem.  Cost matrix, the cost quartic, the three Jacobian cubics, then the
//            Schur complement of the Macaulay matrix by the REFERENCE'S route: a dense partial-pivot LU of the 93 x 93
//            block in the generated table's row / column order (dls_layout.h) with the 27 reduced columns as right-hand
//            sides (dls_pnp.cc:143-146), a column-oriented back-substitution and M00 - M01 X.  Every entry sees exactly
//            the operations of oracle/dls_oracle.h in the same order (first-maximum pivots, l = a / pivot, one fused
//            multiply-add per update, x = rhs * (1 / u_kk)), so the 27 x 27 result -- and with it every hypothesis, inlier set
//            and cost -- is BIT-IDENTICAL to the sequential restatement (round 3 eliminated degree by degree: a faster
//            but different route, 978 of 1000 pairs equal).
//            The augmented 93 x 120 matrix lives in REGISTERS: lane = (column group g = tid / 32, row group rg = tid % 32)
//            holds rows 3 rg + {0, 1, 2} at the columns 6 i + g -- 60 doubles.  Columns are dealt cyclically so that all
//            groups shrink together; after every six steps the lane's columns move down one register (folded into the
//            destination of that step's multiply-adds), so the pivot column is always register 0 of the half-wave g = k % 6
//            and every register index is a compile-time constant inside a rolled loop.  Per step: pivot search inside one
//            half-wave (butterfly), the factors l = a / pivot to LDS, barrier, the six lanes that hold the pivot row
//            broadcast it through LDS (and keep its part right of the diagonal for the back-substitution: 34 KB
//            triangular store), barrier, 3 x (20 - k / 6) multiply-adds per lane.  Rows are never moved: the oracle's
//            swap is bookkeeping of positions (ties go to the smallest POSITION, as the sequential scan takes them).
//            40 KB of LDS per workgroup: four problems per CU.
//   stage B  real Schur form + eigenvectors of the 27 x 27 matrix (orthes + hqr2, complex pairs included), root
//            extraction and the reference's solution filter.  In the RANSAC path a TEAM of 32 lanes per matrix with the
//            work arrays in LDS (eig_team.h); one thread per problem with the arrays in scratch for the
//            directly bound solver.
#ifndef THEIA_HIP_DLS_DEVICE_H_
#define THEIA_HIP_DLS_DEVICE_H_

#include <hip/hip_runtime.h>

#include "dls_tables.h"
#include "ransac_device.h"

namespace thip {
namespace dlsdev {

using dls::kReduced;
using dls::kBlock;
constexpr int kThreads = dls::kThreads;   // stage A workgroup (dls_stage_a.h, dls_kernels.hip)
constexpr int kMaxSolutions = 27;
// One eigenvector column -> (quaternion [w x y z], translation) if it is an admissible root (dls_pnp.cc:147-198):
// V (27 x 27, hqr2 column convention), wi: imaginary parts of the eigenvalues.
// COMPACT: V holds only the four rows read here, in the order {0, 9, 3, 1} (eig_team's kept rows).
constexpr int kKeptRows = 4;
__device__ __constant__ const int kKeptRow[kKeptRows] = {0, 9, 3, 1};
template <bool COMPACT = false>
__device__ inline bool column_solution(const double* V, const double* wi, int i, const double* __restrict__ tfac, int npts,
                                       const double* __restrict__ world, int wstride, const int* __restrict__ index,
                                       double* quat, double* tr) {
  const int re_col = wi[i] < 0 ? i - 1 : i;
  if (re_col < 0) return false;
  const bool cplx = wi[i] != 0.0;
  const double sg = wi[i] < 0 ? -1.0 : 1.0;
  const double d_re = V[re_col], d_im = cplx ? sg * V[re_col + 1] : 0.0;   // row 0
  if (d_re == 0.0 && d_im == 0.0) return false;
  double sr[3], si[3];
  const int rows[3] = {COMPACT ? 1 : 9, COMPACT ? 2 : 3, COMPACT ? 3 : 1};
  for (int k = 0; k < 3; ++k) {
    const double a = V[27 * rows[k] + re_col], b = cplx ? sg * V[27 * rows[k] + re_col + 1] : 0.0;
    rsc::eig_cdiv(a, b, d_re, d_im, &sr[k], &si[k]);
  }
  const double kEps = 1e-6;
  if (!(fabs(si[0]) < kEps && fabs(si[1]) < kEps && fabs(si[2]) < kEps)) return false;
  // Quaterniond(1, s1, s2, s3).inverse().normalized()
  const double n2 = ((1.0 + sr[0] * sr[0]) + sr[1] * sr[1]) + sr[2] * sr[2];
  const double qi[4] = {1.0 / n2, -sr[0] / n2, -sr[1] / n2, -sr[2] / n2};
  const double nq = sqrt(((qi[0] * qi[0] + qi[1] * qi[1]) + qi[2] * qi[2]) + qi[3] * qi[3]);
  const double qs[4] = {qi[0] / nq, qi[1] / nq, qi[2] / nq, qi[3] / nq};
  const double m2 = ((qs[0] * qs[0] + qs[1] * qs[1]) + qs[2] * qs[2]) + qs[3] * qs[3];
  const double qv[4] = {qs[0] / m2, -qs[1] / m2, -qs[2] / m2, -qs[3] / m2};
  double Rm[9], Rs[9], t[3];
  rsc::quat_to_rot(qv, Rm);
  for (int r = 0; r < 3; ++r) {
    double s = 0.0;
    for (int c = 0; c < 3; ++c) for (int k = 0; k < 3; ++k) s += tfac[9 * r + 3 * c + k] * Rm[3 * k + c];
    t[r] = s;
  }
  rsc::quat_to_rot(qs, Rs);
  for (int j = 0; j < npts; ++j) {
    const int id = index ? index[j] : j;
    const double* X = world + (size_t)id * wstride;
    const double z = ((Rs[6] * X[0] + Rs[7] * X[1]) + Rs[8] * X[2]) + t[2];
    if (z < 0) return false;
  }
  for (int k = 0; k < 4; ++k) quat[k] = qs[k];
  for (int k = 0; k < 3; ++k) tr[k] = t[k];
  return true;
}

// gDLS: one eigenvector column -> (quaternion [w x y z] = soln_rotation, translation, scale) if it is an admissible root
// (gdls_similarity_transform.cc:176-226).  V: the four kept rows {0, 9, 3, 1} (COMPACT layout of column_solution).
// tfac: translation factor (27) | scale factor (9).  Rays: dir / origin / homogeneous world point of the npts sampled data.
__device__ inline bool column_solution_gdls(const double* V, const double* wi, int i, const double* __restrict__ tfac, int npts,
                                            const double* __restrict__ data, int stride, int dir_off, int org_off, int wld_off,
                                            const int* __restrict__ index, double* quat, double* tr, double* scale) {
  const int re_col = wi[i] < 0 ? i - 1 : i;
  if (re_col < 0) return false;
  const bool cplx = wi[i] != 0.0;
  const double sg = wi[i] < 0 ? -1.0 : 1.0;
  const double d_re = V[re_col], d_im = cplx ? sg * V[re_col + 1] : 0.0;   // row 0
  if (d_re == 0.0 && d_im == 0.0) return false;
  double sr[3], si[3];
  for (int k = 0; k < 3; ++k) {
    const double a = V[27 * (k + 1) + re_col], b = cplx ? sg * V[27 * (k + 1) + re_col + 1] : 0.0;
    rsc::eig_cdiv(a, b, d_re, d_im, &sr[k], &si[k]);
  }
  const double kEps = 1e-6;
  if (!(fabs(si[0]) < kEps && fabs(si[1]) < kEps && fabs(si[2]) < kEps)) return false;
  const double n2 = ((1.0 + sr[0] * sr[0]) + sr[1] * sr[1]) + sr[2] * sr[2];
  const double qi[4] = {1.0 / n2, -sr[0] / n2, -sr[1] / n2, -sr[2] / n2};
  const double nq = sqrt(((qi[0] * qi[0] + qi[1] * qi[1]) + qi[2] * qi[2]) + qi[3] * qi[3]);
  const double qs[4] = {qi[0] / nq, qi[1] / nq, qi[2] / nq, qi[3] / nq};
  const double m2 = ((qs[0] * qs[0] + qs[1] * qs[1]) + qs[2] * qs[2]) + qs[3] * qs[3];
  const double qv[4] = {qs[0] / m2, -qs[1] / m2, -qs[2] / m2, -qs[3] / m2};
  double Rm[9], Rs[9], t[3], sc = 0.0;
  rsc::quat_to_rot(qv, Rm);
  for (int r = 0; r < 3; ++r) {
    double s = 0.0;
    for (int c = 0; c < 3; ++c) for (int k = 0; k < 3; ++k) s += tfac[9 * r + 3 * c + k] * Rm[3 * k + c];
    t[r] = s;
  }
  for (int c = 0; c < 3; ++c) for (int k = 0; k < 3; ++k) sc += tfac[27 + 3 * c + k] * Rm[3 * k + c];
  rsc::quat_to_rot(qs, Rs);
  for (int j = 0; j < npts; ++j) {   // every point in front of its ray: x . (R X + t - s c) >= 0
    const double* d = data + (size_t)(index ? index[j] : j) * stride;
    const double* x = d + dir_off; const double* c = d + org_off; const double* w = d + wld_off;
    const double X[3] = {w[0] / w[3], w[1] / w[3], w[2] / w[3]};
    double p[3];
    for (int r = 0; r < 3; ++r) p[r] = (((Rs[3 * r] * X[0] + Rs[3 * r + 1] * X[1]) + Rs[3 * r + 2] * X[2]) + t[r]) - sc * c[r];
    if ((x[0] * p[0] + x[1] * p[1]) + x[2] * p[2] < 0) return false;
  }
  for (int k = 0; k < 4; ++k) quat[k] = qs[k];
  for (int k = 0; k < 3; ++k) tr[k] = t[k];
  *scale = sc;
  return true;
}


// Stage B, one thread per problem (the directly bound solver): H, V = 729-double work arrays of the calling thread.
__device__ inline int stage_b(double* H, double* V, const double* __restrict__ tfac, int npts,
                              const double* __restrict__ world, int wstride, const int* __restrict__ index,
                              double* quats, double* ts) {
  double wr[27], wi[27];
  if (!rsc::eig_general_t<27, true>(27, H, wr, wi, V)) return 0;
  int ns = 0;
  for (int i = 0; i < 27; ++i)
    if (column_solution(V, wi, i, tfac, npts, world, wstride, index, quats + 4 * ns, ts + 3 * ns)) ns++;
  return ns;
}

}  // namespace dlsdev
}  // namespace thip
#endif
