// DLS-PnP on the device (the reference: sfm/pose/dls_pnp.cc:67-200).  Two stages per minimal problem:
//
//   stage A  one WAVE per problem, everything in LDS (31.8 KB, five problems per CU): cost matrix, the three Jacobian
//            cubics, then the Schur complement of the Macaulay matrix WITHOUT forming it.  With the non-reduced monomials
//            ordered by degree the 93 x 93 block is block upper triangular (dls_tables.h), so its partial-pivot LU is
//            five small LUs (3, 9, 18, 27, 36 rows, cubic coefficients only); the right-hand sides of a block are the
//            27 reduced columns minus the already solved lower-degree rows.  Lane = column of the augmented block
//            [B_dd | rhs] (at most 36 + 27 = 63 columns), one row operation per step; 115 k FMAs instead of the 0.5 M of
//            the dense 93 x 93 solve.  Output: the 27 x 27 multiplication matrix of f0 and the 3 x 9 translation factor.
//   stage B  real Schur form + eigenvectors of the 27 x 27 matrix (orthes + hqr2, complex pairs included), root
//            extraction and the reference's solution filter.  In the RANSAC path a TEAM of 32 lanes per matrix with the
//            three 27 x 27 work arrays in LDS (eig_team.h); one thread per problem with the arrays in scratch for the
//            directly bound solver (the scratch version moved ~2.7 MB of HBM traffic per matrix).
#ifndef THEIA_HIP_DLS_DEVICE_H_
#define THEIA_HIP_DLS_DEVICE_H_

#include <hip/hip_runtime.h>

#include "dls_tables.h"
#include "ransac_device.h"

namespace thip {
namespace dlsdev {

using dls::kReduced;
using dls::kMaxBlock;
constexpr int kAugCols = kMaxBlock + kReduced;   // 63
constexpr int kXRows = 60;   // solved rows kept: the 57 monomials of degree 3..6 and the 3 of degree 7 the result reads
constexpr int kMaxSolutions = 27;

__constant__ dls::Tables c_tab;

struct WaveLds {
  double aug[kMaxBlock * kAugCols];
  double X[kXRows * kReduced];
  double f[60];
  double T[27];
  double sf[9];   // gDLS: the scale factor row
  double u[4];
  int flag;
};

__device__ inline double wave_allsum(double v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Eigen's Matrix4d::inverse() restated as adjugate over determinant (oracle/dls_oracle.h: gdls_inverse4)
__device__ inline bool inverse4(const double* a, double* inv) {
  auto m3 = [&](int r0, int r1, int r2, int c0, int c1, int c2) {
    return a[4 * r0 + c0] * (a[4 * r1 + c1] * a[4 * r2 + c2] - a[4 * r1 + c2] * a[4 * r2 + c1]) -
           a[4 * r0 + c1] * (a[4 * r1 + c0] * a[4 * r2 + c2] - a[4 * r1 + c2] * a[4 * r2 + c0]) +
           a[4 * r0 + c2] * (a[4 * r1 + c0] * a[4 * r2 + c1] - a[4 * r1 + c1] * a[4 * r2 + c0]);
  };
  double cof[16];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int r0 = r == 0 ? 1 : 0, r1 = r <= 1 ? 2 : 1, r2 = r <= 2 ? 3 : 2;
      const int c0 = c == 0 ? 1 : 0, c1 = c <= 1 ? 2 : 1, c2 = c <= 2 ? 3 : 2;
      const double minor = m3(r0, r1, r2, c0, c1, c2);
      cof[4 * r + c] = ((r + c) & 1) ? -minor : minor;
    }
  const double det = ((a[0] * cof[0] + a[1] * cof[1]) + a[2] * cof[2]) + a[3] * cof[3];
  if (det == 0.0) return false;
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) inv[4 * r + c] = cof[4 * c + r] / det;
  return true;
}

// points: feat[i * fstride + {0,1}], world[i * wstride + {0,1,2}] for i = index ? index[k] : k, k < npts.
// Writes action[729] (row-major) and tfac[27]; returns false when a pivot vanished (degenerate sample).
// GDLS (GdlsSimilarityTransform, gdls_similarity_transform.cc:67-175): feat holds the UNIT ray direction (3), world the
// homogeneous point (4: hnormalized here), origin the ray origin (3); tfac = translation factor (27) | scale factor (9).
template <bool GDLS = false>
__device__ inline bool stage_a(WaveLds& L, int npts, const double* __restrict__ feat, int fstride,
                               const double* __restrict__ world, int wstride, const int* __restrict__ index,
                               const double* __restrict__ u4, double* __restrict__ action, double* __restrict__ tfac,
                               const double* __restrict__ origin = nullptr, int ostride = 0) {
  const int lane = threadIdx.x & 63;
  const dls::Tables& tb = c_tab;
  if (lane < 4) L.u[lane] = u4[lane];
  if (lane == 0) L.flag = 0;
  if constexpr (GDLS) {
    // ---- sums over the rays: the 4 x 4 matrix H^-1 (:80-96) and the 4 x 9 helper (:101-117)
    double hs[16], sv[36];
#pragma unroll
    for (int k = 0; k < 16; ++k) hs[k] = 0.0;
#pragma unroll
    for (int k = 0; k < 36; ++k) sv[k] = 0.0;
    for (int i = lane; i < npts; i += 64) {
      const int id = index ? index[i] : i;
      const double* xx = feat + (size_t)id * fstride; const double* cc = origin + (size_t)id * ostride; const double* ww = world + (size_t)id * wstride;
      const double x[3] = {xx[0], xx[1], xx[2]}, c[3] = {cc[0], cc[1], cc[2]}, X[3] = {ww[0] / ww[3], ww[1] / ww[3], ww[2] / ww[3]};
      const double cd = (c[0] * x[0] + c[1] * x[1]) + c[2] * x[2];
      hs[0] += ((c[0] * c[0] + c[1] * c[1]) + c[2] * c[2]) - cd * cd;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const double t = -c[r] + cd * x[r];
        hs[4 * (r + 1)] += t; hs[r + 1] += t;
#pragma unroll
        for (int k = 0; k < 3; ++k) hs[4 * (r + 1) + k + 1] += (r == k ? 1.0 : 0.0) - x[r] * x[k];
      }
      // L(X)[k][col] = X[col % 3] when col / 3 == k: the sums over k collapse to k = col / 3
#pragma unroll
      for (int col = 0; col < 9; ++col) {
        const int k = col / 3;
        const double lx = X[col % 3];
        sv[col] += (c[k] - cd * x[k]) * lx;
#pragma unroll
        for (int r = 0; r < 3; ++r) sv[9 * (r + 1) + col] += (x[r] * x[k] - (r == k ? 1.0 : 0.0)) * lx;
      }
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) hs[k] = wave_allsum(hs[k]);
#pragma unroll
    for (int k = 0; k < 36; ++k) sv[k] = wave_allsum(sv[k]);
    double Hm[16];
    if (!inverse4(hs, Hm)) { if (lane == 0) L.flag = 1; for (int k = 0; k < 16; ++k) Hm[k] = 0.0; }
    if (lane < 36) {
      const int r = lane / 9, col = lane % 9;
      double s2 = 0.0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        double pre = 0.0, hk = 0.0;   // sv[9 k + col], Hm[4 r + k]: compile-time register indices only
#pragma unroll
        for (int cc2 = 0; cc2 < 9; ++cc2) pre = (cc2 == col) ? sv[9 * k + cc2] : pre;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) hk = (rr == r) ? Hm[4 * rr + k] : hk;
        s2 += hk * pre;
      }
      if (r == 0) L.sf[col] = s2; else L.T[9 * (r - 1) + col] = s2;
    }
  } else {
  // ---- sums over the points: sum n n^T (6 unique) and sum (n n^T - I)_{ab} X_c (27)
  double acc[33];
#pragma unroll
  for (int k = 0; k < 33; ++k) acc[k] = 0.0;
  for (int i = lane; i < npts; i += 64) {
    const int id = index ? index[i] : i;
    const double fx = feat[(size_t)id * fstride], fy = feat[(size_t)id * fstride + 1];
    const double nrm = sqrt((fx * fx + fy * fy) + 1.0);
    const double n[3] = {fx / nrm, fy / nrm, 1.0 / nrm};
    const double X[3] = {world[(size_t)id * wstride], world[(size_t)id * wstride + 1], world[(size_t)id * wstride + 2]};
    acc[0] += n[0] * n[0]; acc[1] += n[0] * n[1]; acc[2] += n[0] * n[2];
    acc[3] += n[1] * n[1]; acc[4] += n[1] * n[2]; acc[5] += n[2] * n[2];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        const double m = n[a] * n[b] - (a == b ? 1.0 : 0.0);
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[6 + 9 * a + 3 * b + c] += m * X[c];
      }
  }
#pragma unroll
  for (int k = 0; k < 33; ++k) acc[k] = wave_allsum(acc[k]);
  // H = (n I - sum n n^T)^-1 (dls_pnp.cc:90-94), translation_factor = H * sum (n n^T - I) L(X) (dls_pnp.cc:98-105)
  {
    const double a0 = (double)npts - acc[0], a1 = -acc[1], a2 = -acc[2], a4 = (double)npts - acc[3], a5 = -acc[4], a8 = (double)npts - acc[5];
    const double c00 = a4 * a8 - a5 * a5, c01 = a5 * a2 - a1 * a8, c02 = a1 * a5 - a4 * a2;
    const double det = (a0 * c00 + a1 * c01) + a2 * c02;
    const double id = 1.0 / det;
    const double Hm[9] = {c00 * id, c01 * id, c02 * id,
                          c01 * id, (a0 * a8 - a2 * a2) * id, (a2 * a1 - a0 * a5) * id,
                          c02 * id, (a2 * a1 - a0 * a5) * id, (a0 * a4 - a1 * a1) * id};
    if (lane < 27) {
      const int r = lane / 9, c = lane % 9;
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        double pre = 0.0;   // acc[6 + 9 k + c], compile-time indices only
#pragma unroll
        for (int cc = 0; cc < 9; ++cc) pre = (cc == c) ? acc[6 + 9 * k + cc] : pre;
        s += Hm[3 * r + k] * pre;
      }
      L.T[lane] = s;
    }
  }
  }   // !GDLS
  __syncthreads();
  // ---- D = sum (L(X) + T)^T (I - n n^T) (L(X) + T)   (dls_pnp.cc:111-118): lane = entry (alpha, beta), two passes
  // (gDLS: W = L(X) - c scale_factor + T, gdls_similarity_transform.cc:123-133)
  double* Dm = L.aug;          // 81
  double* g = L.aug + 128;     // 90
  for (int e = lane; e < 81; e += 64) {
    const int al = e / 9, be = e % 9;
    double d = 0.0;
    for (int i = 0; i < npts; ++i) {
      const int id = index ? index[i] : i;
      double n[3], wa[3], wb[3];
      if constexpr (GDLS) {
        const double* xx = feat + (size_t)id * fstride; const double* cc = origin + (size_t)id * ostride; const double* ww = world + (size_t)id * wstride;
        n[0] = xx[0]; n[1] = xx[1]; n[2] = xx[2];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          wa[a] = (al / 3 == a ? ww[al % 3] / ww[3] : 0.0) + (L.T[9 * a + al] - cc[a] * L.sf[al]);
          wb[a] = (be / 3 == a ? ww[be % 3] / ww[3] : 0.0) + (L.T[9 * a + be] - cc[a] * L.sf[be]);
        }
      } else {
      const double fx = feat[(size_t)id * fstride], fy = feat[(size_t)id * fstride + 1];
      const double nrm = sqrt((fx * fx + fy * fy) + 1.0);
      n[0] = fx / nrm; n[1] = fy / nrm; n[2] = 1.0 / nrm;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        wa[a] = L.T[9 * a + al] + (al / 3 == a ? world[(size_t)id * wstride + al % 3] : 0.0);
        wb[a] = L.T[9 * a + be] + (be / 3 == a ? world[(size_t)id * wstride + be % 3] : 0.0);
      }
      }
      // wa^T (I - n n^T) wb = wa.wb - (n.wa)(n.wb)
      const double dab = (wa[0] * wb[0] + wa[1] * wb[1]) + wa[2] * wb[2];
      const double na = (n[0] * wa[0] + n[1] * wa[1]) + n[2] * wa[2];
      const double nb = (n[0] * wb[0] + n[1] * wb[1]) + n[2] * wb[2];
      d += dab - na * nb;
    }
    Dm[e] = d;
  }
  __syncthreads();
  // ---- g_k = ((D + D^T) rbar)_k over the 10 monomials of degree <= 2; f_i = sum_k (d rbar_k / d s_i) g_k
  for (int e = lane; e < 90; e += 64) {
    const int k = e / 10, m = e % 10;
    double s = 0.0;
    for (int l = 0; l < 9; ++l) { const int c = tb.R[l][m]; if (c) s += (double)c * (Dm[9 * k + l] + Dm[9 * l + k]); }
    g[e] = s;
  }
  __syncthreads();
  if (lane < 60) {
    const int i = lane / 20, m3 = lane % 20;
    double s = 0.0;
    for (int k = 0; k < 9; ++k)
      for (int q = 0; q < 4; ++q) {
        const int c = tb.dR[i][k][q], m2 = tb.div3[m3][q];
        if (c && m2 >= 0) s += (double)c * g[10 * k + m2];
      }
    L.f[lane] = s;
  }
  __syncthreads();
  // ---- the five diagonal blocks, ascending degree
  for (int bi = 0; bi < 5; ++bi) {
    const int r0 = tb.blk_off[bi], nd = tb.blk_off[bi + 1] - r0, ncol = nd + kReduced;
    for (int e = lane; e < nd * kAugCols; e += 64) L.aug[e] = 0.0;
    __syncthreads();
    for (int t = lane; t < nd * 20; t += 64) {
      const int r = t / 20, nu = t % 20;
      const int col = tb.col_of[r0 + r][nu];
      const double coef = L.f[20 * tb.row_poly[r0 + r] + nu];
      if (col < kReduced) L.aug[r * kAugCols + nd + col] = coef;
      else if (col - kReduced >= r0) L.aug[r * kAugCols + (col - kReduced - r0)] = coef;
    }
    __syncthreads();
    if (bi > 0) {
      for (int t = lane; t < nd * kReduced; t += 64) {
        const int r = t / kReduced, j = t % kReduced;
        const double* fr = L.f + 20 * tb.row_poly[r0 + r];
        double s = 0.0;
        for (int nu = 0; nu < 10; ++nu) {   // the terms of degree < 3 land on lower-degree columns
          const int col = tb.col_of[r0 + r][nu];
          if (col >= kReduced) s += fr[nu] * L.X[(col - kReduced) * kReduced + j];
        }
        L.aug[r * kAugCols + nd + j] -= s;
      }
      __syncthreads();
    }
    // partial-pivot elimination, lane = column
    for (int k = 0; k < nd; ++k) {
      double best = -1.0; int prow = k;
      if (lane >= k && lane < nd) { best = fabs(L.aug[lane * kAugCols + k]); prow = lane; }
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) {
        const double ob = __shfl_xor(best, o, 64);
        const int op = __shfl_xor(prow, o, 64);
        if (ob > best || (ob == best && op < prow)) { best = ob; prow = op; }
      }
      if (!(best > 0.0)) { if (lane == 0) L.flag = 1; best = 1.0; }
      double pk = 0.0;
      if (lane < ncol) {
        pk = L.aug[prow * kAugCols + lane];
        if (prow != k) { L.aug[prow * kAugCols + lane] = L.aug[k * kAugCols + lane]; L.aug[k * kAugCols + lane] = pk; }
      }
      __syncthreads();
      const double rp = 1.0 / L.aug[k * kAugCols + k];
      if (lane > k && lane < ncol) {
        // four rows per step: the LDS reads of a step are in flight together (the rolled loop paid one LDS round trip per row)
        int r = k + 1;
        for (; r + 4 <= nd; r += 4) {
          double lv[4], av[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) { lv[u] = L.aug[(r + u) * kAugCols + k]; av[u] = L.aug[(r + u) * kAugCols + lane]; }
#pragma unroll
          for (int u = 0; u < 4; ++u) L.aug[(r + u) * kAugCols + lane] = av[u] - (lv[u] * rp) * pk;
        }
        for (; r < nd; ++r) {
          const double l = L.aug[r * kAugCols + k] * rp;
          L.aug[r * kAugCols + lane] -= l * pk;
        }
      }
      __syncthreads();
    }
    // back-substitution, column oriented: solve x_c, then retire it from the rows above
    const int c_stop = (bi == 4) ? nd - 3 : 0;   // the degree-7 block: only its last three rows are read afterwards
    for (int c = nd - 1; c >= c_stop; --c) {
      if (lane < kReduced) {
        const double x = L.aug[c * kAugCols + nd + lane] / L.aug[c * kAugCols + c];
        L.aug[c * kAugCols + nd + lane] = x;
        const int xr = (bi == 4) ? (57 + c - (nd - 3)) : (r0 + c);
        L.X[xr * kReduced + lane] = x;
      }
      __syncthreads();
      for (int t = lane; t < (c - c_stop) * kReduced; t += 64) {
        const int i = c_stop + t / kReduced, j = t % kReduced;
        L.aug[i * kAugCols + nd + j] -= L.aug[i * kAugCols + c] * L.aug[c * kAugCols + nd + j];
      }
      __syncthreads();
    }
  }
  // ---- action matrix: row j = coefficients of f0 * mu_j reduced to the 27 reduced monomials
  for (int e = lane; e < kReduced * kReduced; e += 64) {
    const int j = e / kReduced, j2 = e % kReduced;
    double a = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int col = tb.mul[j][q];
      if (col < kReduced) a += (col == j2) ? L.u[q] : 0.0;
      else a -= L.u[q] * L.X[((col < kReduced + 57) ? col - kReduced : col - 60) * kReduced + j2];
    }
    action[e] = a;
  }
  if (lane < 27) tfac[lane] = L.T[lane];
  if (GDLS && lane < 9) tfac[27 + lane] = L.sf[lane];
  __syncthreads();
  return L.flag == 0;
}

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
