// CPU ORACLE (test infrastructure, never shipped): P4Pf -- absolute pose and focal length from four 2D-3D correspondences
// (Bujnak, Kukelova, Pajdla, CVPR 2008).  Follows the reference's FourPointPoseAndFocalLength
// (sfm/pose/four_point_focal_length.cc:100-222: normalisation, distance ratios, scale fix, GetRigidTransform :62-96) and the
// UncalibratedAbsolutePoseEstimator around it (estimators/estimate_uncalibrated_absolute_pose.cc:60-103).
//
// Stated deviation: the reference's helper (four_point_focal_length_helper.cc:48-931) is a GENERATED 78 x 88 elimination
// template solved by partialPivLu; nothing of it is transcribed.  The template used here is derived from the four
// inner-product equations of the rigid point configuration by scripts/gen_p4pf_template.py (p4pf_tables.h): 77 multiples
// over 94 monomials that reduce z * {z^2, yz, xz, wz, wy} onto the basis [1, z, y, x, w, z^2, yz, xz, wz, wy].  The rows do
// not determine every monomial, so the reduction is obtained from the TRANSPOSED system  A^T Y = E  (94 equations, 77
// unknowns per target, consistent by construction) by Gaussian elimination with partial pivoting; the action matrix of
// multiplication by z, its real eigenvectors and everything after them are the reference's.  Same solutions to rounding;
// pinned by the reference's own test vectors (four_point_focal_length_test.cc:119-146) in tests/test_oracle_ransac.py.
#ifndef ORACLE_P4PF_ORACLE_H_
#define ORACLE_P4PF_ORACLE_H_

#include "p4pf_tables.h"

namespace p4pf_oracle {

using namespace thip::p4pf;

constexpr int kLd = kRows + kTargets;   // row of the transposed system: [77 template rows | 5 right-hand sides]

struct Normalised {
  double fn[8];     // image points / mean norm, (x, y) per point
  double wn[12];    // world points, centred, / mean norm
  double mean[3], wvar, fvar;
  double g[6];      // squared distances ab, ac, ad, bc, bd, cd
};

// four_point_focal_length.cc:108-140
inline bool normalise(const double* subset /* 4 x [feature 2 | world 3] */, Normalised* N) {
  for (int k = 0; k < 3; ++k)
    N->mean[k] = (((subset[2 + k] + subset[5 + 2 + k]) + subset[10 + 2 + k]) + subset[15 + 2 + k]) / 4.0;
  double nsum = 0.0;
  for (int i = 0; i < 4; ++i) {
    double s2 = 0.0;
    for (int k = 0; k < 3; ++k) { N->wn[3 * i + k] = subset[5 * i + 2 + k] - N->mean[k]; s2 += N->wn[3 * i + k] * N->wn[3 * i + k]; }
    nsum += std::sqrt(s2);
  }
  N->wvar = nsum / 4.0;
  for (int i = 0; i < 12; ++i) N->wn[i] /= N->wvar;
  double fsum = 0.0;
  for (int i = 0; i < 4; ++i) fsum += std::sqrt(subset[5 * i] * subset[5 * i] + subset[5 * i + 1] * subset[5 * i + 1]);
  N->fvar = fsum / 4.0;
  for (int i = 0; i < 4; ++i) { N->fn[2 * i] = subset[5 * i] / N->fvar; N->fn[2 * i + 1] = subset[5 * i + 1] / N->fvar; }
  static const int pq[6][2] = {{0, 1}, {0, 2}, {0, 3}, {1, 2}, {1, 3}, {2, 3}};
  for (int e = 0; e < 6; ++e) {
    double s2 = 0.0;
    for (int k = 0; k < 3; ++k) { const double d = N->wn[3 * pq[e][0] + k] - N->wn[3 * pq[e][1] + k]; s2 += d * d; }
    N->g[e] = s2;
  }
  const double prod = ((((N->g[0] * N->g[1]) * N->g[2]) * N->g[3]) * N->g[4]) * N->g[5];
  return !(prod < 1e-15);   // :142-144
}

// coefficients of the four polynomials in the generator's canonical term order (scripts/gen_p4pf_template.py POLYS)
inline void coefficients(const Normalised& N, double c[4][kMaxTerms]) {
  const double* a = N.fn; const double* b = N.fn + 2; const double* cc_ = N.fn + 4; const double* d = N.fn + 6;
  auto dot = [](const double* u, const double* v) { return u[0] * v[0] + u[1] * v[1]; };
  const double aa = dot(a, a), ab = dot(a, b), ac = dot(a, cc_), ad = dot(a, d), bc = dot(b, cc_), bd = dot(b, d), cd = dot(cc_, d),
               cc = dot(cc_, cc_), dd = dot(d, d);
  const double gab = N.g[0], gac = N.g[1], gad = N.g[2], gbc = N.g[3], gbd = N.g[4], gcd = N.g[5];
  const double k1 = ((gab + gac) - gbc) / (2.0 * gad), k2 = gac / gad, k3 = ((gab + gad) - gbd) / (2.0 * gad),
               k4 = ((gac + gad) - gcd) / (2.0 * gad);
  const double p1[12] = {1.0, -k1, -1.0, -1.0, bc, 2.0 * k1, -(k1 * dd), 1.0 - k1, -ab, -ac, (2.0 * k1) * ad, aa - k1 * aa};
  const double p2[10] = {1.0, -k2, -2.0, cc, 2.0 * k2, -(k2 * dd), 1.0 - k2, -(2.0 * ac), (2.0 * k2) * ad, aa - k2 * aa};
  const double p3[10] = {1.0, -k3, -1.0, 2.0 * k3 - 1.0, bd, -(k3 * dd), 1.0 - k3, -ab, (2.0 * k3) * ad - ad, aa - k3 * aa};
  const double p4[10] = {1.0, -k4, -1.0, 2.0 * k4 - 1.0, cd, -(k4 * dd), 1.0 - k4, -ac, (2.0 * k4) * ad - ad, aa - k4 * aa};
  for (int t = 0; t < 12; ++t) { c[0][t] = p1[t]; c[1][t] = t < 10 ? p2[t] : 0.0; c[2][t] = t < 10 ? p3[t] : 0.0; c[3][t] = t < 10 ? p4[t] : 0.0; }
}

// The action matrix of multiplication by z on the basis (10 x 10, row-major).  False when a pivot vanishes.
inline bool action_matrix(const double c[4][kMaxTerms], double* T) {
  static thread_local double G[kElim][kLd];     // transposed template: G[monomial][template row | rhs]
  static thread_local double Bm[kRows][kBasis];
  for (int i = 0; i < kElim; ++i) for (int j = 0; j < kLd; ++j) G[i][j] = 0.0;
  for (int r = 0; r < kRows; ++r) for (int j = 0; j < kBasis; ++j) Bm[r][j] = 0.0;
  for (int r = 0; r < kRows; ++r) {
    const int k = kRowPoly[r];
    for (int t = 0; t < kPolyTerms[k]; ++t) {
      const int col = kRowCol[r][t];
      if (col < kElim) G[col][r] = c[k][t]; else Bm[r][col - kElim] = c[k][t];
    }
  }
  for (int i = 0; i < kTargets; ++i) G[kOthers + i][kRows + i] = 1.0;
  int perm[kElim];
  for (int i = 0; i < kElim; ++i) perm[i] = i;
  for (int k = 0; k < kRows; ++k) {
    int p = k; double best = std::fabs(G[perm[k]][k]);
    for (int i = k + 1; i < kElim; ++i) { const double v = std::fabs(G[perm[i]][k]); if (v > best) { best = v; p = i; } }
    if (!(best > 0.0)) return false;
    std::swap(perm[k], perm[p]);
    const double* pr = G[perm[k]];
    const double piv = pr[k];
    for (int i = k + 1; i < kElim; ++i) {
      double* row = G[perm[i]];
      if (row[k] == 0.0) continue;
      const double f = row[k] / piv;
      for (int j = k + 1; j < kLd; ++j) row[j] = row[j] - f * pr[j];
    }
  }
  // back-substitution, column-oriented: Y[k][q] overwrites the right-hand side of row perm[k]
  for (int k = kRows - 1; k >= 0; --k) {
    double* pr = G[perm[k]];
    for (int q = 0; q < kTargets; ++q) pr[kRows + q] = pr[kRows + q] / pr[k];
    for (int i = 0; i < k; ++i) {
      double* row = G[perm[i]];
      if (row[k] == 0.0) continue;
      for (int q = 0; q < kTargets; ++q) row[kRows + q] = row[kRows + q] - row[k] * pr[kRows + q];
    }
  }
  for (int i = 0; i < 100; ++i) T[i] = 0.0;
  T[0 * 10 + 1] = 1.0; T[1 * 10 + 5] = 1.0; T[2 * 10 + 6] = 1.0; T[3 * 10 + 7] = 1.0; T[4 * 10 + 8] = 1.0;   // z * {1, z, y, x, w}
  for (int i = 0; i < kTargets; ++i)
    for (int j = 0; j < kBasis; ++j) {
      double acc = 0.0;
      for (int r = 0; r < kRows; ++r) if (Bm[r][j] != 0.0) acc += G[perm[r]][kRows + i] * Bm[r][j];
      T[(5 + i) * 10 + j] = -acc;
    }
  return true;
}

// one solution (focal length^2 = w, depths x, y, z of points b, c, d relative to a) -> projection matrix (3 x 4 row-major)
// four_point_focal_length.cc:166-219 + GetRigidTransform :62-96
inline void projection_from_solution(const Normalised& N, double w, double x, double y, double z, double* Pm) {
  const double f = std::sqrt(w);
  const double dep[4] = {1.0, x, y, z};
  double A[12];
  for (int i = 0; i < 4; ++i) { A[3 * i] = N.fn[2 * i] * dep[i]; A[3 * i + 1] = N.fn[2 * i + 1] * dep[i]; A[3 * i + 2] = f * dep[i]; }
  static const int pq[6][2] = {{0, 1}, {0, 2}, {0, 3}, {1, 2}, {1, 3}, {2, 3}};
  double dsum = 0.0;
  for (int e = 0; e < 6; ++e) {
    double s2 = 0.0;
    for (int k = 0; k < 3; ++k) { const double d = A[3 * pq[e][0] + k] - A[3 * pq[e][1] + k]; s2 += d * d; }
    dsum += std::sqrt(N.g[e] / s2);
  }
  const double gta = dsum / 6.0;
  for (int i = 0; i < 12; ++i) A[i] *= gta;
  double m1[3], m2[3];
  for (int k = 0; k < 3; ++k) {
    m1[k] = (((N.wn[k] + N.wn[3 + k]) + N.wn[6 + k]) + N.wn[9 + k]) / 4.0;
    m2[k] = (((A[k] + A[3 + k]) + A[6 + k]) + A[9 + k]) / 4.0;
  }
  double p1[12], p2[12];
  for (int i = 0; i < 4; ++i) {
    double n1 = 0.0, n2 = 0.0;
    for (int k = 0; k < 3; ++k) {
      p1[3 * i + k] = N.wn[3 * i + k] - m1[k]; p2[3 * i + k] = A[3 * i + k] - m2[k];
      n1 += p1[3 * i + k] * p1[3 * i + k]; n2 += p2[3 * i + k] * p2[3 * i + k];
    }
    n1 = std::sqrt(n1); n2 = std::sqrt(n2);
    for (int k = 0; k < 3; ++k) { p1[3 * i + k] /= n1; p2[3 * i + k] /= n2; }
  }
  double D[9], U[9], S[3], V[9];
  for (int r = 0; r < 3; ++r)
    for (int cI = 0; cI < 3; ++cI) {
      double acc = 0.0;
      for (int i = 0; i < 4; ++i) acc += p2[3 * i + r] * p1[3 * i + cI];
      D[3 * r + cI] = acc;
    }
  svd3(D, U, S, V);
  double UVt[9];
  for (int r = 0; r < 3; ++r) for (int cI = 0; cI < 3; ++cI) UVt[3 * r + cI] = (U[3 * r] * V[3 * cI] + U[3 * r + 1] * V[3 * cI + 1]) + U[3 * r + 2] * V[3 * cI + 2];
  const double sgn = det3(UVt) < 0 ? -1.0 : 1.0;
  double R[9];
  for (int r = 0; r < 3; ++r) for (int cI = 0; cI < 3; ++cI) R[3 * r + cI] = (U[3 * r] * V[3 * cI] + U[3 * r + 1] * V[3 * cI + 1]) + (U[3 * r + 2] * sgn) * V[3 * cI + 2];
  double t[3];
  for (int r = 0; r < 3; ++r) {
    const double tr = -((R[3 * r] * m1[0] + R[3 * r + 1] * m1[1]) + R[3 * r + 2] * m1[2]) + m2[r];
    t[r] = N.wvar * tr - ((R[3 * r] * N.mean[0] + R[3 * r + 1] * N.mean[1]) + R[3 * r + 2] * N.mean[2]);
  }
  const double fo = f * N.fvar;
  for (int cI = 0; cI < 3; ++cI) { Pm[cI] = fo * R[cI]; Pm[4 + cI] = fo * R[3 + cI]; Pm[8 + cI] = R[6 + cI]; }
  Pm[3] = fo * t[0]; Pm[7] = fo * t[1]; Pm[11] = t[2];
}

// FourPointPoseAndFocalLength: up to 10 projection matrices, `stride` doubles apart
inline int four_point_focal_length(const double* subset, double* models, int stride) {
  Normalised N;
  if (!normalise(subset, &N)) return 0;
  double c[4][kMaxTerms], T[100], wr[10], wi[10], V[100];
  coefficients(N, c);
  if (!action_matrix(c, T)) return 0;
  if (!eig_real_general(10, T, wr, wi, V)) return 0;
  int n = 0;
  for (int i = 0; i < 10; ++i) {
    if (wi[i] != 0.0) continue;   // complex eigenvalue: the eigenvector's ratios are complex (helper :917-921)
    const double v0 = V[0 * 10 + i];
    const double w = V[4 * 10 + i] / v0;
    if (!(w >= 0.0)) continue;    // negative or NaN
    projection_from_solution(N, w, V[3 * 10 + i] / v0, V[2 * 10 + i] / v0, V[1 * 10 + i] / v0, models + (size_t)stride * n);
    for (int k = 12; k < stride; ++k) models[(size_t)stride * n + k] = 0.0;
    ++n;
  }
  return n;
}

// UncalibratedAbsolutePoseEstimator::Error (estimate_uncalibrated_absolute_pose.cc:88-97)
inline double reprojection_error(const double* Pm, const double* d /* feature 2 | world 3 */) {
  const double px = ((Pm[0] * d[2] + Pm[1] * d[3]) + Pm[2] * d[4]) + Pm[3];
  const double py = ((Pm[4] * d[2] + Pm[5] * d[3]) + Pm[6] * d[4]) + Pm[7];
  const double pz = ((Pm[8] * d[2] + Pm[9] * d[3]) + Pm[10] * d[4]) + Pm[11];
  const double ex = px / pz - d[0], ey = py / pz - d[1];
  return ex * ex + ey * ey;
}

}  // namespace p4pf_oracle

#endif
