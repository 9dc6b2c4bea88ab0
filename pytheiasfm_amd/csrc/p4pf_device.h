// P4Pf on the device: absolute pose + focal length from four 2D-3D correspondences (FourPointPoseAndFocalLength,
// sfm/pose/four_point_focal_length.cc:100-222; UncalibratedAbsolutePoseEstimator, estimate_uncalibrated_absolute_pose.cc:60-103).
//
// The reference solves a generated 78 x 88 elimination template (four_point_focal_length_helper.cc); the template used here is
// derived from the four inner-product equations of the rigid point configuration by scripts/gen_p4pf_template.py
// (p4pf_tables.h: 77 multiples over 94 monomials) and reduced through the TRANSPOSED system -- see oracle/p4pf_oracle.h, whose
// operation order this file keeps so that the two agree bit for bit.
//
//   stage A  p4pf_action_wg     one WORKGROUP (256 threads) per hypothesis, the 94 x 82 transposed system in LDS (62 KB):
//                               partial-pivot elimination (wave 0 picks the pivot, one factor per row, four waves update
//                               rows k+1.. over the lanes' columns), column-oriented back-substitution, action matrix
//   stage B  eig_team           the 10 x 10 eigen-decomposition by 8-lane teams (the five-point kernel's)
//   stage C  p4pf_projection    one thread per hypothesis: real eigenvectors -> depths, focal length, rigid alignment
#ifndef THEIA_HIP_P4PF_DEVICE_H_
#define THEIA_HIP_P4PF_DEVICE_H_

#include "p4pf_tables.h"
#include "ransac_device.h"

namespace thip {
namespace p4pfdev {

using namespace thip::p4pf;

constexpr int kLd = kRows + kTargets;                  // row of the transposed system: [template rows | right-hand sides]
constexpr int kNorm = 31;                              // fn 8 | wn 12 | mean 3 | wvar | fvar | g 6
constexpr int kWs = 36 + 100;                          // per hypothesis in HBM: normalisation (kNorm of 36) | action matrix
constexpr int kThreads = 256;
constexpr size_t kLdsBytes = sizeof(double) * (kElim * kLd + kRows * kBasis + kElim + 4 * kMaxTerms) + sizeof(int) * (kElim + 4);

__constant__ uint8_t c_poly_terms[4] = THIP_P4PF_POLY_TERMS;
__constant__ uint8_t c_row_col[kRows][kMaxTerms] = THIP_P4PF_ROW_COL;
__constant__ uint8_t c_row_poly[kRows] = THIP_P4PF_ROW_POLY;

__device__ constexpr int kPair[6][2] = {{0, 1}, {0, 2}, {0, 3}, {1, 2}, {1, 3}, {2, 3}};

// four_point_focal_length.cc:108-144.  N: fn 8 | wn 12 | mean 3 | wvar | fvar | g 6
RDEV bool normalise(const double* subset /* 4 x [feature 2 | world 3] */, double* N) {
  double* fn = N; double* wn = N + 8; double* mean = N + 20; double* g = N + 25;
  for (int k = 0; k < 3; ++k) mean[k] = (((subset[2 + k] + subset[5 + 2 + k]) + subset[10 + 2 + k]) + subset[15 + 2 + k]) / 4.0;
  double nsum = 0.0;
  for (int i = 0; i < 4; ++i) {
    double s2 = 0.0;
    for (int k = 0; k < 3; ++k) { wn[3 * i + k] = subset[5 * i + 2 + k] - mean[k]; s2 += wn[3 * i + k] * wn[3 * i + k]; }
    nsum += sqrt(s2);
  }
  const double wvar = nsum / 4.0;
  for (int i = 0; i < 12; ++i) wn[i] /= wvar;
  double fsum = 0.0;
  for (int i = 0; i < 4; ++i) fsum += sqrt(subset[5 * i] * subset[5 * i] + subset[5 * i + 1] * subset[5 * i + 1]);
  const double fvar = fsum / 4.0;
  for (int i = 0; i < 4; ++i) { fn[2 * i] = subset[5 * i] / fvar; fn[2 * i + 1] = subset[5 * i + 1] / fvar; }
  N[23] = wvar; N[24] = fvar;
  for (int e = 0; e < 6; ++e) {
    double s2 = 0.0;
    for (int k = 0; k < 3; ++k) { const double d = wn[3 * kPair[e][0] + k] - wn[3 * kPair[e][1] + k]; s2 += d * d; }
    g[e] = s2;
  }
  const double prod = ((((g[0] * g[1]) * g[2]) * g[3]) * g[4]) * g[5];
  return !(prod < 1e-15);
}

// coefficients of the four polynomials in the generator's canonical term order; c: [4][kMaxTerms]
RDEV void coefficients(const double* N, double* c) {
  const double* a = N; const double* b = N + 2; const double* cp = N + 4; const double* d = N + 6; const double* g = N + 25;
  const double aa = a[0] * a[0] + a[1] * a[1], ab = a[0] * b[0] + a[1] * b[1], ac = a[0] * cp[0] + a[1] * cp[1],
               ad = a[0] * d[0] + a[1] * d[1], bc = b[0] * cp[0] + b[1] * cp[1], bd = b[0] * d[0] + b[1] * d[1],
               cd = cp[0] * d[0] + cp[1] * d[1], cc = cp[0] * cp[0] + cp[1] * cp[1], dd = d[0] * d[0] + d[1] * d[1];
  const double gab = g[0], gac = g[1], gad = g[2], gbc = g[3], gbd = g[4], gcd = g[5];
  const double k1 = ((gab + gac) - gbc) / (2.0 * gad), k2 = gac / gad, k3 = ((gab + gad) - gbd) / (2.0 * gad),
               k4 = ((gac + gad) - gcd) / (2.0 * gad);
  const double p1[12] = {1.0, -k1, -1.0, -1.0, bc, 2.0 * k1, -(k1 * dd), 1.0 - k1, -ab, -ac, (2.0 * k1) * ad, aa - k1 * aa};
  const double p2[12] = {1.0, -k2, -2.0, cc, 2.0 * k2, -(k2 * dd), 1.0 - k2, -(2.0 * ac), (2.0 * k2) * ad, aa - k2 * aa, 0.0, 0.0};
  const double p3[12] = {1.0, -k3, -1.0, 2.0 * k3 - 1.0, bd, -(k3 * dd), 1.0 - k3, -ab, (2.0 * k3) * ad - ad, aa - k3 * aa, 0.0, 0.0};
  const double p4[12] = {1.0, -k4, -1.0, 2.0 * k4 - 1.0, cd, -(k4 * dd), 1.0 - k4, -ac, (2.0 * k4) * ad - ad, aa - k4 * aa, 0.0, 0.0};
  for (int t = 0; t < 12; ++t) { c[t] = p1[t]; c[12 + t] = p2[t]; c[24 + t] = p3[t]; c[36 + t] = p4[t]; }
}

// Stage A, called by every thread of a 256-thread workgroup.  sm: kLdsBytes of LDS.  Writes the normalisation (kNorm doubles)
// and the 10 x 10 action matrix to ws; returns false (uniformly) for a degenerate sample or a vanishing pivot.
__device__ inline bool p4pf_action_wg(const double* __restrict__ pd, const int* __restrict__ sample, double* sm, double* __restrict__ ws) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double* G = sm;                          // [kElim][kLd]
  double* Bm = G + kElim * kLd;            // [kRows][kBasis]
  double* fac = Bm + kRows * kBasis;       // [kElim]
  double* coef = fac + kElim;              // [4][kMaxTerms]
  int* perm = (int*)(coef + 4 * kMaxTerms);   // [kElim]
  int* flag = perm + kElim;                // [0] ok, [1] pivot row of the step
  if (tid == 0) {
    double subset[20];
    for (int i = 0; i < 4; ++i) for (int k = 0; k < 5; ++k) subset[5 * i + k] = pd[(size_t)sample[i] * 5 + k];
    double N[kNorm];
    const bool good = normalise(subset, N);
    flag[0] = good ? 1 : 0;
    if (good) {
      double c[4 * kMaxTerms];
      coefficients(N, c);
      for (int t = 0; t < 4 * kMaxTerms; ++t) coef[t] = c[t];
      for (int t = 0; t < kNorm; ++t) ws[t] = N[t];
    }
  }
  for (int e = tid; e < kElim * kLd + kRows * kBasis; e += kThreads) G[e] = 0.0;
  if (tid < kElim) perm[tid] = tid;
  __syncthreads();
  if (!flag[0]) return false;
  if (tid < kRows) {   // template row r = column r of the transposed system
    const int k = c_row_poly[tid];
    for (int t = 0; t < c_poly_terms[k]; ++t) {
      const int col = c_row_col[tid][t];
      if (col < kElim) G[col * kLd + tid] = coef[k * kMaxTerms + t]; else Bm[tid * kBasis + (col - kElim)] = coef[k * kMaxTerms + t];
    }
  }
  if (tid < kTargets) G[(kOthers + tid) * kLd + kRows + tid] = 1.0;
  __syncthreads();
  for (int k = 0; k < kRows; ++k) {
    if (wave == 0) {   // pivot: the largest |G[perm[i]][k]| over i >= k, the first one on ties
      double best = -1.0; int bi = kElim;
      for (int i = k + lane; i < kElim; i += 64) {
        const double v = fabs(G[perm[i] * kLd + k]);
        if (v > best) { best = v; bi = i; }
      }
      for (int o = 32; o >= 1; o >>= 1) {
        const double ov = __shfl_xor(best, o); const int oi = __shfl_xor(bi, o);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
      }
      if (lane == 0) {
        if (!(best > 0.0)) flag[0] = 0;
        if (bi < kElim) { const int t = perm[k]; perm[k] = perm[bi]; perm[bi] = t; }
      }
    }
    __syncthreads();
    if (!flag[0]) return false;
    const int prow = perm[k];
    const double piv = G[prow * kLd + k];
    if (tid > k && tid < kElim) {
      const double v = G[perm[tid] * kLd + k];
      fac[tid] = (v == 0.0) ? 0.0 : v / piv;
    }
    __syncthreads();
    for (int i = k + 1 + wave; i < kElim; i += 4) {
      const double f = fac[i];
      if (f == 0.0) continue;
      double* row = G + perm[i] * kLd;
      for (int j = k + 1 + lane; j < kLd; j += 64) row[j] = row[j] - f * G[prow * kLd + j];
    }
    __syncthreads();
  }
  // back-substitution, column-oriented: Y[k][q] overwrites the right-hand side of row perm[k]
  for (int k = kRows - 1; k >= 0; --k) {
    double* pr = G + perm[k] * kLd;
    if (tid < kTargets) pr[kRows + tid] = pr[kRows + tid] / pr[k];
    __syncthreads();
    const int q = tid & 7;
    if (q < kTargets)
      for (int i = tid >> 3; i < k; i += kThreads / 8) {
        double* row = G + perm[i] * kLd;
        const double a = row[k];
        if (a != 0.0) row[kRows + q] = row[kRows + q] - a * pr[kRows + q];
      }
    __syncthreads();
  }
  double* T = ws + 36;
  if (tid < 50) {
    T[tid] = (tid == 1 || tid == 15 || tid == 26 || tid == 37 || tid == 48) ? 1.0 : 0.0;   // z * {1, z, y, x, w}: rows 0..4
    const int i = tid / 10, j = tid % 10;
    double acc = 0.0;
    for (int r = 0; r < kRows; ++r) {
      const double b = Bm[r * kBasis + j];
      if (b != 0.0) acc += G[perm[r] * kLd + kRows + i] * b;
    }
    T[(5 + i) * 10 + j] = -acc;
  }
  return true;
}

// one solution -> projection matrix (3 x 4 row-major): four_point_focal_length.cc:166-219 + GetRigidTransform :62-96
RDEV void projection_from_solution(const double* N, double w, double x, double y, double z, double* Pm) {
  const double* fn = N; const double* wn = N + 8; const double* mean = N + 20; const double* g = N + 25;
  const double wvar = N[23], fvar = N[24];
  const double f = sqrt(w);
  const double dep[4] = {1.0, x, y, z};
  double A[12];
  for (int i = 0; i < 4; ++i) { A[3 * i] = fn[2 * i] * dep[i]; A[3 * i + 1] = fn[2 * i + 1] * dep[i]; A[3 * i + 2] = f * dep[i]; }
  double dsum = 0.0;
  for (int e = 0; e < 6; ++e) {
    double s2 = 0.0;
    for (int k = 0; k < 3; ++k) { const double d = A[3 * kPair[e][0] + k] - A[3 * kPair[e][1] + k]; s2 += d * d; }
    dsum += sqrt(g[e] / s2);
  }
  const double gta = dsum / 6.0;
  for (int i = 0; i < 12; ++i) A[i] *= gta;
  double m1[3], m2[3];
  for (int k = 0; k < 3; ++k) {
    m1[k] = (((wn[k] + wn[3 + k]) + wn[6 + k]) + wn[9 + k]) / 4.0;
    m2[k] = (((A[k] + A[3 + k]) + A[6 + k]) + A[9 + k]) / 4.0;
  }
  double p1[12], p2[12];
  for (int i = 0; i < 4; ++i) {
    double n1 = 0.0, n2 = 0.0;
    for (int k = 0; k < 3; ++k) {
      p1[3 * i + k] = wn[3 * i + k] - m1[k]; p2[3 * i + k] = A[3 * i + k] - m2[k];
      n1 += p1[3 * i + k] * p1[3 * i + k]; n2 += p2[3 * i + k] * p2[3 * i + k];
    }
    n1 = sqrt(n1); n2 = sqrt(n2);
    for (int k = 0; k < 3; ++k) { p1[3 * i + k] /= n1; p2[3 * i + k] /= n2; }
  }
  double D[9], U[9], S[3], V[9];
  for (int r = 0; r < 3; ++r)
    for (int cI = 0; cI < 3; ++cI) {
      double acc = 0.0;
      for (int i = 0; i < 4; ++i) acc += p2[3 * i + r] * p1[3 * i + cI];
      D[3 * r + cI] = acc;
    }
  rsc::svd3(D, U, S, V);
  double UVt[9];
  for (int r = 0; r < 3; ++r) for (int cI = 0; cI < 3; ++cI) UVt[3 * r + cI] = (U[3 * r] * V[3 * cI] + U[3 * r + 1] * V[3 * cI + 1]) + U[3 * r + 2] * V[3 * cI + 2];
  const double sgn = rsc::det3(UVt) < 0 ? -1.0 : 1.0;
  double R[9];
  for (int r = 0; r < 3; ++r) for (int cI = 0; cI < 3; ++cI) R[3 * r + cI] = (U[3 * r] * V[3 * cI] + U[3 * r + 1] * V[3 * cI + 1]) + (U[3 * r + 2] * sgn) * V[3 * cI + 2];
  double t[3];
  for (int r = 0; r < 3; ++r) {
    const double tr = -((R[3 * r] * m1[0] + R[3 * r + 1] * m1[1]) + R[3 * r + 2] * m1[2]) + m2[r];
    t[r] = wvar * tr - ((R[3 * r] * mean[0] + R[3 * r + 1] * mean[1]) + R[3 * r + 2] * mean[2]);
  }
  const double fo = f * fvar;
  for (int cI = 0; cI < 3; ++cI) { Pm[cI] = fo * R[cI]; Pm[4 + cI] = fo * R[3 + cI]; Pm[8 + cI] = R[6 + cI]; }
  Pm[3] = fo * t[0]; Pm[7] = fo * t[1]; Pm[11] = t[2];
}

// UncalibratedAbsolutePoseEstimator::Error (estimate_uncalibrated_absolute_pose.cc:88-97)
RDEV double reprojection_error(const double* Pm, const double* d /* feature 2 | world 3 */) {
  const double px = ((Pm[0] * d[2] + Pm[1] * d[3]) + Pm[2] * d[4]) + Pm[3];
  const double py = ((Pm[4] * d[2] + Pm[5] * d[3]) + Pm[6] * d[4]) + Pm[7];
  const double pz = ((Pm[8] * d[2] + Pm[9] * d[3]) + Pm[10] * d[4]) + Pm[11];
  const double ex = px / pz - d[0], ey = py / pz - d[1];
  return ex * ex + ey * ey;
}

}  // namespace p4pfdev
}  // namespace thip

#endif
