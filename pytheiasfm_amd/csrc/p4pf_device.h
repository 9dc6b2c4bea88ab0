// P4Pf on the device: absolute pose + focal length from four 2D-3D correspondences (FourPointPoseAndFocalLength,
// sfm/pose/four_point_focal_length.cc:100-222; UncalibratedAbsolutePoseEstimator, estimate_uncalibrated_absolute_pose.cc:60-103).
//
// The reference solves a generated 78 x 88 elimination template (four_point_focal_length_helper.cc); the template used here is
// derived from the four inner-product equations of the rigid point configuration by scripts/gen_p4pf_template.py
// (p4pf_tables.h: 77 multiples over 94 monomials) and reduced through the TRANSPOSED system -- see oracle/p4pf_oracle.h, whose
// operation order this file keeps so that the two agree bit for bit.
//
//   stage A  p4pf_action_wg     one WORKGROUP (256 threads) per hypothesis, the 94 x 82 transposed system in LDS (62 KB):
//                               partial-pivot elimination (two barriers per step), back-substitution by one wave per
//                               right-hand side without barriers, action matrix
//   stage B  eig_team           the 10 x 10 eigen-decomposition by 8-lane teams (the five-point kernel's)
//   stage C  p4pf_projection    one thread per hypothesis: real eigenvectors -> depths, focal length, rigid alignment
#ifndef THEIA_HIP_P4PF_DEVICE_H_
#define THEIA_HIP_P4PF_DEVICE_H_

#include "p4pf_tables.h"
#include "wave_reduce.h"
#include "ransac_device.h"

namespace thip {
namespace p4pfdev {

using namespace thip::p4pf;

constexpr int kLd = kRows + kTargets;                  // row of the transposed system: [template rows | right-hand sides]
constexpr int kNorm = 31;                              // fn 8 | wn 12 | mean 3 | wvar | fvar | g 6
constexpr int kWs = 36 + 100;                          // per hypothesis in HBM: normalisation (kNorm of 36) | action matrix
constexpr int kThreads = 256;
constexpr int kLdP = kLd + 1;                         // LDS row stride: odd, so that the lanes of a column access spread over the banks
constexpr size_t kLdsBytes = sizeof(double) * ((kElim + 1) * kLdP + kRows * kBasis + kElim + 1 + kRows + 4 * kMaxTerms) + sizeof(int) * (4 + 128 + 4 * kLd);

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
//
// Same arithmetic as oracle/p4pf_oracle.h action_matrix(), arranged for the machine: rows are swapped physically instead of
// through a permutation (no dependent index load in front of every access; pure data movement), every wave finds the pivot
// of a step by itself (no barrier between the search and the factors), the factors, the row swap and the pivot's value are
// written in the same phase (the swap leaves column k alone, which is all that phase reads), the update runs over the
// compacted lists of the rows with a non-zero factor and of the non-zero entries of the pivot row (the template is sparse:
// a sixth of the dense element updates; skipping is exact, |f| <= 1), four independent rows per pass, and the back-substitution runs
// without workgroup barriers, one wave per right-hand side with the right-hand side in registers (column-oriented, the
// oracle's order: e_i -= U[i][k] y_k for descending k).  Two barriers per elimination step.
__device__ inline bool p4pf_action_wg(const double* __restrict__ pd, const int* __restrict__ sample, double* sm, double* __restrict__ ws) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double* G = sm;                          // [kElim][kLdP]
  double* Bm = G + (kElim + 1) * kLdP;     // [kRows][kBasis]   (row kElim of G: spare, written by out-of-range update slots)
  double* fac = Bm + kRows * kBasis;       // [kElim + 1], fac[kElim] = 0 (spare row)
  double* pivv = fac + kElim + 1;          // [kRows] pivots
  double* coef = pivv + kRows;             // [4][kMaxTerms]
  int* flag = (int*)(coef + 4 * kMaxTerms);   // [4]
  int* cnt = flag + 2;                     // [2] non-zero factors found by waves 0 and 1
  int* rowlist = flag + 4;                 // [128] their rows
  int* collist = rowlist + 128;            // [4][kLd] per wave: columns with a non-zero pivot-row entry
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
  for (int e = tid; e < (kElim + 1) * kLdP + kRows * kBasis + kElim + 1; e += kThreads) G[e] = 0.0;   // G, Bm, fac
  __syncthreads();
  if (!flag[0]) return false;
  if (tid < kRows) {   // template row r = column r of the transposed system
    const int k = c_row_poly[tid];
    for (int t = 0; t < c_poly_terms[k]; ++t) {
      const int col = c_row_col[tid][t];
      if (col < kElim) G[col * kLdP + tid] = coef[k * kMaxTerms + t]; else Bm[tid * kBasis + (col - kElim)] = coef[k * kMaxTerms + t];
    }
  }
  if (tid < kTargets) G[(kOthers + tid) * kLdP + kRows + tid] = 1.0;
  __syncthreads();
  for (int k = 0; k < kRows; ++k) {
    // pivot: the largest |G[i][k]| over i >= k, the first one on ties (every wave for itself)
    const int ia = k + lane, ib = ia + 64;
    const double va = ia < kElim ? fabs(G[ia * kLdP + k]) : -1.0, vb = ib < kElim ? fabs(G[ib * kLdP + k]) : -1.0;
    const double best = wave_max_abs(fmax(va, vb));   // (wave_reduce.h: fmax over the lanes' bits, on the DPP network)
    const unsigned long long ma = __ballot(va == best), mb = __ballot(vb == best);
    const int bi = ma ? k + __ffsll((long long)ma) - 1 : (mb ? k + 64 + __ffsll((long long)mb) - 1 : kElim);
    if (!(best > 0.0)) return false;   // the same verdict in every wave
    const double piv = G[bi * kLdP + k];
    if (wave < 2) {                    // factor of row i (rows k and bi change places: row bi will hold the old row k),
      const int i = k + 1 + tid;       // and the list of the rows with a non-zero factor: the template is sparse
      double f = 0.0;
      if (i < kElim) {
        const double v = G[(i == bi ? k : i) * kLdP + k];
        f = (v == 0.0) ? 0.0 : v / piv;
        fac[i] = f;
      }
      const unsigned long long nz = __ballot(f != 0.0);
      if (f != 0.0) rowlist[wave * 64 + __popcll(nz & ((1ull << lane) - 1ull))] = i;
      if (lane == 0) cnt[wave] = __popcll(nz);
    }
    {
      const int j = k + 1 + tid;       // swap of the columns right of k
      if (j < kLd && bi != k) {
        const double a = G[k * kLdP + j], b = G[bi * kLdP + j];
        G[k * kLdP + j] = b; G[bi * kLdP + j] = a;
      }
      if (tid == 0) pivv[k] = piv;
    }
    __syncthreads();
    // update: lanes over the NON-ZERO entries of the pivot row (x - f * 0 = x: |f| <= 1 by the pivoting), the wave's share of
    // the rows with a non-zero factor four at a time (slots past the end go to the spare row kElim, factor 0)
    const int n0r = cnt[0], nr = n0r + cnt[1];
    const int j0 = k + 1 + lane, j1 = j0 + 64;
    const bool nz0 = j0 < kLd && G[k * kLdP + j0] != 0.0, nz1 = j1 < kLd && G[k * kLdP + j1] != 0.0;
    const unsigned long long m0 = __ballot(nz0), m1 = __ballot(nz1);
    const unsigned long long lt = (1ull << lane) - 1ull;
    int* cl = collist + wave * kLd;
    const int n0c = __popcll(m0), nc = n0c + __popcll(m1);
    if (nz0) cl[__popcll(m0 & lt)] = j0;
    if (nz1) cl[n0c + __popcll(m1 & lt)] = j1;
    rsc::team_sync();
    for (int c = lane; c < nc; c += 64) {
      const int j = cl[c];
      const double pr = G[k * kLdP + j];
      for (int t = wave; t < nr; t += 16) {
        const int t1 = t + 4, t2 = t + 8, t3 = t + 12;
        const int i0 = t < n0r ? rowlist[t] : rowlist[64 + t - n0r];
        const int i1 = t1 < nr ? (t1 < n0r ? rowlist[t1] : rowlist[64 + t1 - n0r]) : kElim;
        const int i2 = t2 < nr ? (t2 < n0r ? rowlist[t2] : rowlist[64 + t2 - n0r]) : kElim;
        const int i3 = t3 < nr ? (t3 < n0r ? rowlist[t3] : rowlist[64 + t3 - n0r]) : kElim;
        const double f0 = fac[i0], f1 = fac[i1], f2 = fac[i2], f3 = fac[i3];
        double* r0 = G + i0 * kLdP + j; double* r1 = G + i1 * kLdP + j; double* r2 = G + i2 * kLdP + j; double* r3 = G + i3 * kLdP + j;
        const double x0 = *r0, x1 = *r1, x2 = *r2, x3 = *r3;
        *r0 = x0 - f0 * pr; *r1 = x1 - f1 * pr; *r2 = x2 - f2 * pr; *r3 = x3 - f3 * pr;
      }
    }
    __syncthreads();
  }
  // back-substitution, column-oriented, one wave per right-hand side with e in registers; wave 0 carries the fifth one
  // through the same loop (two independent chains)
  {
    const int q = wave, q2 = kTargets - 1;
    const bool two = wave == 0;
    double e0 = lane < kRows ? G[lane * kLdP + kRows + q] : 0.0;
    double e1 = lane + 64 < kRows ? G[(lane + 64) * kLdP + kRows + q] : 0.0;
    double h0 = (two && lane < kRows) ? G[lane * kLdP + kRows + q2] : 0.0;
    double h1 = (two && lane + 64 < kRows) ? G[(lane + 64) * kLdP + kRows + q2] : 0.0;
    for (int k = kRows - 1; k >= 0; --k) {
      const double pk = pivv[k];
      const double ek = k >= 64 ? lane_value(e1, k - 64) : lane_value(e0, k);
      const double hk = k >= 64 ? lane_value(h1, k - 64) : lane_value(h0, k);
      const double y = ek / pk, y2 = hk / pk;
      if (lane == 0) { G[k * kLdP + kRows + q] = y; if (two) G[k * kLdP + kRows + q2] = y2; }
      if (lane < k) { const double a = G[lane * kLdP + k]; if (a != 0.0) { e0 = e0 - a * y; h0 = h0 - a * y2; } }
      if (lane + 64 < k) { const double a = G[(lane + 64) * kLdP + k]; if (a != 0.0) { e1 = e1 - a * y; h1 = h1 - a * y2; } }
    }
  }
  __syncthreads();
  double* T = ws + 36;
  if (tid < 50) {
    T[tid] = (tid == 1 || tid == 15 || tid == 26 || tid == 37 || tid == 48) ? 1.0 : 0.0;   // z * {1, z, y, x, w}: rows 0..4
    const int i = tid / 10, j = tid % 10;
    double acc = 0.0;
    for (int r = 0; r < kRows; ++r) {
      const double b = Bm[r * kBasis + j];
      if (b != 0.0) acc += G[r * kLdP + kRows + i] * b;
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
