// TEST INFRASTRUCTURE (CPU oracle) -- never linked into or loaded by the product library.
//
// P4Pfr: absolute pose, focal length and one radial-distortion coefficient from four 2D-3D correspondences.  Restates
//   sfm/pose/four_point_focal_length_radial_distortion.cc:68-288  FourPointsPoseFocalLengthRadialDistortion: normalisation (centroid,
//        JacobiSVD rotation, scales), the 5 x 8 linear system, its null space by HouseholderQR, the "random rotation" of the null-space
//        basis drawn from the process-wide generator, the particular solution by FullPivLU, D by ColPivHouseholderQR, and :219-285 the
//        solutions -> projection matrix -> (R, t, focal length, distortion) with the metadata's range tests
//   sfm/pose/four_point_focal_length_radial_distortion_helper.cc:50-1436  the 40 x 50 elimination template, alpha = FullPivLU of
//        C0^T (37 x 40, underdetermined: Eigen's solve() leaves the free unknowns at zero), the 13 x 13 action matrix of a3,
//        EigenSolver, solutions = eigenvector rows over the row of 1, kept when |Im a1| <= 1e-6
//   sfm/estimators/estimate_radial_dist_uncalibrated_absolute_pose.cc:60-160  the estimator: EstimateModel, Error (division-model
//        distortion of the projection, 1e10 when the translation's z is negative)
// The 327 generated coefficient formulas of the helper are NOT restated: p4pfr_layout.h (scripts/gen_p4pfr_layout.py) holds the ten
// polynomial equations they are the coefficients of -- derived from the geometry, every formula and every template entry checked
// symbolically against the reference in the build container -- and the coefficients come from polynomial arithmetic here.
// Eigen itself is not in the image: its decompositions are restated from their published algorithms in Eigen's operation order where
// that is documented in its sources' structure (pivot rules, Householder conventions, rank thresholds); bit equality with an
// Eigen-built reference is not claimed, the reference's known-answer scenes are the pin.
//
// Included by ransac_oracle.cpp after its small linear algebra (svd3_sweeps, eig_general_t, eig_cdiv, Mt19937).
#pragma once
#include <cfloat>
#include <cmath>
#include <vector>

#include "p4pfr_layout.h"

namespace p4pfr {
using namespace thip::p4pfr_layout;

constexpr int kModel = 14;       // rotation (9, row-major) | translation (3) | focal length | radial distortion
constexpr int kMaxModels = 13;

// libstdc++ std::uniform_real_distribution<double>(lo, hi)(std::mt19937&): generate_canonical<double, 53> takes two 32-bit draws
// (random.tcc: sum = g0 + g1 * 2^32, / 2^64, nextafter(1, 0) if the quotient rounds to 1), then * (hi - lo) + lo.
// Pinned by tests/golden/mt19937_randdouble.json (made by the real library).
template <class G>
inline double rand_double(G& g, double lo, double hi) {
  const double r = 4294967296.0;
  double sum = 0.0, tmp = 1.0;
  for (int k = 0; k < 2; ++k) { sum += (double)g.next() * tmp; tmp *= r; }
  double ret = sum / tmp;
  if (ret >= 1.0) ret = std::nextafter(1.0, 0.0);
  return ret * (hi - lo) + lo;
}

// Eigen::AngleAxisd(angle, axis).toRotationMatrix() for an axis that is NOT normalised (the reference passes the raw vector,
// :137-141; the result is then no rotation, but any invertible 3 x 3 matrix only changes the basis of the null space).  Row-major.
inline void angle_axis_matrix(double angle, const double* ax, double* R) {
  // sincos(), not sin() and cos(): a GCC build of the reference merges Eigen's two calls into one sincos (glibc's results differ
  // from the separate functions' by an ulp for ~0.1 % of the arguments), and the library must take the same one
  double s, c;
  sincos(angle, &s, &c);
  const double sa[3] = {s * ax[0], s * ax[1], s * ax[2]};
  const double ca[3] = {(1.0 - c) * ax[0], (1.0 - c) * ax[1], (1.0 - c) * ax[2]};
  double t;
  t = ca[0] * ax[1]; R[1] = t - sa[2]; R[3] = t + sa[2];
  t = ca[0] * ax[2]; R[2] = t + sa[1]; R[6] = t - sa[1];
  t = ca[1] * ax[2]; R[5] = t - sa[0]; R[7] = t + sa[0];
  R[0] = ca[0] * ax[0] + c; R[4] = ca[1] * ax[1] + c; R[8] = ca[2] * ax[2] + c;
}

// Householder QR, optionally column-pivoted (Eigen HouseholderQR / ColPivHouseholderQR::computeInPlace), of a rows x cols
// row-major matrix, in place: R in the upper triangle, the essential parts of the reflectors below it, hcoef their factors.
struct Qr {
  int rows, cols, size, nonzero_pivots;
  double hcoef[8];
  int transp[8];
};
inline void make_householder(double* A, int rows, int cols, int k, double* tau, double* beta) {
  const double c0 = A[k * cols + k];
  double tail = 0.0;
  for (int r = k + 1; r < rows; ++r) tail += A[r * cols + k] * A[r * cols + k];
  if (tail <= DBL_MIN) {
    *tau = 0.0; *beta = c0;
    for (int r = k + 1; r < rows; ++r) A[r * cols + k] = 0.0;
  } else {
    double b = std::sqrt(c0 * c0 + tail);
    if (c0 >= 0.0) b = -b;
    for (int r = k + 1; r < rows; ++r) A[r * cols + k] /= (c0 - b);
    *tau = (b - c0) / b; *beta = b;
  }
}
// applyHouseholderOnTheLeft of reflector k (essential part in column k of A below the diagonal) on column j of the
// ldb-strided matrix B, rows k .. rows - 1: tmp = essential^T bottom, tmp += top, top -= tau tmp, bottom -= tau essential tmp
inline void apply_householder(const double* A, int rows, int cols, int k, double tau, double* B, int ldb, int j) {
  if (tau == 0.0) return;
  double tmp = 0.0;
  for (int r = k + 1; r < rows; ++r) tmp += A[r * cols + k] * B[r * ldb + j];
  tmp += B[k * ldb + j];
  B[k * ldb + j] -= tau * tmp;
  for (int r = k + 1; r < rows; ++r) B[r * ldb + j] -= (tau * A[r * cols + k]) * tmp;
}
inline void qr_factor(double* A, int rows, int cols, bool pivot, Qr& f) {
  f.rows = rows; f.cols = cols; f.size = rows < cols ? rows : cols; f.nonzero_pivots = f.size;
  double norm_upd[8], norm_dir[8];
  double threshold_helper = 0.0;
  const double norm_downdate_threshold = std::sqrt(DBL_EPSILON);
  if (pivot) {
    double maxn = 0.0;
    for (int k = 0; k < cols; ++k) {
      double s2 = 0.0;
      for (int r = 0; r < rows; ++r) s2 += A[r * cols + k] * A[r * cols + k];
      norm_upd[k] = norm_dir[k] = std::sqrt(s2);
      if (norm_upd[k] > maxn) maxn = norm_upd[k];
    }
    threshold_helper = (maxn * DBL_EPSILON) * (maxn * DBL_EPSILON) / (double)rows;
  }
  for (int k = 0; k < f.size; ++k) {
    f.transp[k] = k;
    if (pivot) {
      int big = k;
      for (int j = k + 1; j < cols; ++j) if (norm_upd[j] > norm_upd[big]) big = j;
      const double big_sq = norm_upd[big] * norm_upd[big];
      if (f.nonzero_pivots == f.size && big_sq < threshold_helper * (double)(rows - k)) f.nonzero_pivots = k;
      f.transp[k] = big;
      if (k != big) {
        for (int r = 0; r < rows; ++r) std::swap(A[r * cols + k], A[r * cols + big]);
        std::swap(norm_upd[k], norm_upd[big]);
        std::swap(norm_dir[k], norm_dir[big]);
      }
    }
    double tau, beta;
    make_householder(A, rows, cols, k, &tau, &beta);
    f.hcoef[k] = tau;
    A[k * cols + k] = beta;
    for (int j = k + 1; j < cols; ++j) apply_householder(A, rows, cols, k, tau, A, cols, j);
    if (pivot)
      for (int j = k + 1; j < cols; ++j) {
        if (norm_upd[j] == 0.0) continue;
        double temp = std::fabs(A[k * cols + j]) / norm_upd[j];
        temp = (1.0 + temp) * (1.0 - temp);
        temp = temp < 0.0 ? 0.0 : temp;
        const double q = norm_upd[j] / norm_dir[j];
        if (temp * (q * q) <= norm_downdate_threshold) {
          double s2 = 0.0;
          for (int r = k + 1; r < rows; ++r) s2 += A[r * cols + j] * A[r * cols + j];
          norm_dir[j] = std::sqrt(s2);
          norm_upd[j] = norm_dir[j];
        } else {
          norm_upd[j] *= std::sqrt(temp);
        }
      }
  }
}
// householderQ() as a dense rows x rows matrix (HouseholderSequence::evalTo: the reflectors applied to the identity from the last
// to the first, each on the bottom-right corner it can touch)
inline void qr_q(const double* A, const Qr& f, double* Q) {
  const int n = f.rows;
  for (int i = 0; i < n * n; ++i) Q[i] = (i % (n + 1) == 0) ? 1.0 : 0.0;
  for (int k = f.size - 1; k >= 0; --k)
    for (int j = k; j < n; ++j) apply_householder(A, f.rows, f.cols, k, f.hcoef[k], Q, n, j);
}
// ColPivHouseholderQR::solve for nrhs right-hand sides (B: rows x nrhs, destroyed; X: cols x nrhs)
inline void qr_solve(const double* A, const Qr& f, double* B, int nrhs, double* X) {
  const int rows = f.rows, cols = f.cols, np = f.nonzero_pivots;
  for (int j = 0; j < nrhs; ++j) {
    for (int k = 0; k < np; ++k) apply_householder(A, rows, cols, k, f.hcoef[k], B, nrhs, j);
    double y[8];
    for (int i = np - 1; i >= 0; --i) {
      double s = B[i * nrhs + j];
      for (int c = i + 1; c < np; ++c) s -= A[i * cols + c] * y[c];
      y[i] = s / A[i * cols + i];
    }
    int perm[8];
    for (int i = 0; i < cols; ++i) perm[i] = i;
    for (int k = 0; k < f.size; ++k) std::swap(perm[k], perm[f.transp[k]]);
    for (int i = 0; i < cols; ++i) X[i * nrhs + j] = 0.0;
    for (int i = 0; i < np; ++i) X[perm[i] * nrhs + j] = y[i];
  }
}

// Eigen::FullPivLU of a rows x cols row-major matrix (compute: the pivot is the first strict maximum of the remaining corner
// scanned column by column) and solve() for nrhs right-hand sides (rank() with the default threshold eps * diagonalSize; the
// unknowns beyond the rank stay zero).  A is destroyed; B: rows x nrhs (destroyed); X: cols x nrhs.
inline int fullpiv_solve(double* A, int rows, int cols, double* B, int nrhs, double* X) {
  const int size = rows < cols ? rows : cols;
  std::vector<int> rowt(size), colt(size);
  int nonzero = size;
  double maxpivot = 0.0;
  for (int k = 0; k < size; ++k) {
    double best = -1.0; int br = k, bc = k;
    for (int j = k; j < cols; ++j)
      for (int i = k; i < rows; ++i) {
        const double a = std::fabs(A[i * cols + j]);
        if (a > best) { best = a; br = i; bc = j; }
      }
    if (best == 0.0) {
      nonzero = k;
      for (int i = k; i < size; ++i) { rowt[i] = i; colt[i] = i; }
      break;
    }
    if (best > maxpivot) maxpivot = best;
    rowt[k] = br; colt[k] = bc;
    if (br != k) for (int j = 0; j < cols; ++j) std::swap(A[k * cols + j], A[br * cols + j]);
    if (bc != k) for (int i = 0; i < rows; ++i) std::swap(A[i * cols + k], A[i * cols + bc]);
    if (k < rows - 1) for (int i = k + 1; i < rows; ++i) A[i * cols + k] /= A[k * cols + k];
    if (k < size - 1)
      for (int i = k + 1; i < rows; ++i)
        for (int j = k + 1; j < cols; ++j) A[i * cols + j] -= A[i * cols + k] * A[k * cols + j];
  }
  const double premult = maxpivot * (DBL_EPSILON * (double)size);
  int rank = 0;
  for (int i = 0; i < nonzero; ++i) rank += std::fabs(A[i * cols + i]) > premult;
  for (int i = 0; i < cols * nrhs; ++i) X[i] = 0.0;
  if (rank == 0) return 0;
  // c = P rhs: the row transpositions in the order they were made
  for (int k = 0; k < size; ++k)
    if (rowt[k] != k) for (int j = 0; j < nrhs; ++j) std::swap(B[k * nrhs + j], B[rowt[k] * nrhs + j]);
  // unit-lower solve on the leading size x size block, column-oriented; rows beyond cols take the remaining product
  for (int k = 0; k < size; ++k)
    for (int i = k + 1; i < size; ++i)
      for (int j = 0; j < nrhs; ++j) B[i * nrhs + j] -= A[i * cols + k] * B[k * nrhs + j];
  // upper solve on the leading rank x rank block, column-oriented
  for (int k = rank - 1; k >= 0; --k)
    for (int j = 0; j < nrhs; ++j) {
      B[k * nrhs + j] /= A[k * cols + k];
      for (int i = 0; i < k; ++i) B[i * nrhs + j] -= A[i * cols + k] * B[k * nrhs + j];
    }
  // dst.row(Q.indices(i)) = c.row(i): Q = the column transpositions applied on the right of the identity in order
  std::vector<int> perm(cols);
  for (int i = 0; i < cols; ++i) perm[i] = i;
  for (int k = 0; k < size; ++k) std::swap(perm[k], perm[colt[k]]);
  for (int i = 0; i < rank; ++i)
    for (int j = 0; j < nrhs; ++j) X[perm[i] * nrhs + j] = B[i * nrhs + j];
  return rank;
}

// U of Eigen::JacobiSVD<MatrixXd>(A, ComputeFullU | ComputeFullV) for a 3 x 4 matrix (row-major): the matrix over its largest
// |entry|, column-pivoted QR of the adjoint (the more-columns-than-rows preconditioner), two-sided Jacobi sweeps on R^T with U
// started at the column permutation.
inline void svd_u_3x4(const double* A, double* U) {
  double scale = 0.0;
  for (int i = 0; i < 12; ++i) scale = std::fmax(scale, std::fabs(A[i]));
  if (scale == 0.0) scale = 1.0;
  double At[12];   // 4 x 3
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) At[c * 3 + r] = A[r * 4 + c] / scale;
  Qr f;
  qr_factor(At, 4, 3, true, f);
  double W[9], V[9], S[3];
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) W[r * 3 + c] = (c <= r) ? At[c * 3 + r] : 0.0;   // R^T
  int perm[3] = {0, 1, 2};
  for (int k = 0; k < 3; ++k) std::swap(perm[k], perm[f.transp[k]]);
  for (int i = 0; i < 9; ++i) { U[i] = 0.0; V[i] = (i % 4 == 0) ? 1.0 : 0.0; }
  for (int j = 0; j < 3; ++j) U[perm[j] * 3 + j] = 1.0;
  svd3_sweeps(W, U, S, V);
}

// ---- the polynomial system (p4pfr_layout.h).  Nn: 8 x 4 row-major; D: 3 x 9 row-major; U0: first world point (normalised frame).
// C: the 40 x 50 template, row-major.
inline void build_template(const double* Nn, const double* D, double d0, const double* U0, double* C) {
  double eq[10][kCols];
  for (int e = 0; e < 10; ++e) for (int c = 0; c < kCols; ++c) eq[e][c] = 0.0;
  // rows of [p1; p2] over (a1 a2 a3 1): q[i] = Nn row (i < 3 ? i : i + 1); p3x p3y over tmp: D rows 0, 1; p3z = w = tmp[7]
  const double* q[6] = {Nn, Nn + 4, Nn + 8, Nn + 16, Nn + 20, Nn + 24};
  double p3[3][9];
  for (int t = 0; t < 9; ++t) { p3[0][t] = D[t]; p3[1][t] = D[9 + t]; p3[2][t] = (t == 7) ? 1.0 : 0.0; }
  for (int i = 0; i < 3; ++i)
    for (int s = 0; s < 4; ++s)
      for (int t = 0; t < 9; ++t) {
        if (i == 2 && t != 7) continue;
        eq[0][kMulATmp[s][t]] += q[3 + i][s] * p3[i][t];
        eq[1][kMulATmp[s][t]] += q[i][s] * p3[i][t];
      }
  double qq[6][6][10];   // products q_i q_j (i <= j) over the quadratic monomials
  for (int i = 0; i < 6; ++i)
    for (int j = i; j < 6; ++j) {
      for (int u = 0; u < 10; ++u) qq[i][j][u] = 0.0;
      for (int s = 0; s < 4; ++s) for (int t = 0; t < 4; ++t) qq[i][j][kMulAA[s][t]] += q[i][s] * q[j][t];
    }
  for (int i = 0; i < 3; ++i)
    for (int u = 0; u < 10; ++u) {
      eq[2][kA2Col[u]] += qq[i][3 + i][u];
      eq[3][kA2Col[u]] += qq[i][i][u];
    }
  for (int i = 0; i < 3; ++i) for (int u = 0; u < 10; ++u) eq[3][kA2Col[u]] -= qq[3 + i][3 + i][u];
  for (int f = 0; f < 5; ++f)
    for (int n = 0; n < kCubicTerms[f]; ++n) {
      const int i = kCubic[f][n][0], j = kCubic[f][n][1], l = kCubic[f][n][2];
      const double c = (double)kCubic[f][n][3];
      for (int u = 0; u < 10; ++u)
        for (int t = 0; t < 9; ++t) {
          if (l == 2 && t != 7) continue;
          eq[4 + f][kMulA2Tmp[u][t]] += c * (qq[i][j][u] * p3[l][t]);
        }
    }
  // (1 + k d0) - (U0x p3x + U0y p3y + U0z w + p3w)
  for (int t = 0; t < 9; ++t) eq[9][kTmpCol[t]] = -(((U0[0] * D[t] + U0[1] * D[9 + t]) + U0[2] * p3[2][t]) + D[18 + t]);
  eq[9][kTmpCol[8]] += 1.0;
  eq[9][kTmpCol[6]] += d0;
  for (int r = 0; r < kRows; ++r)
    for (int c = 0; c < kCols; ++c) C[r * kCols + c] = kRowSrc[r][c] >= 0 ? eq[kRowEq[r]][kRowSrc[r][c]] : 0.0;
}

// helper.cc:1361-1383: alpha = (C0^T).fullPivLu().solve(b^T), RR = [alpha^T C1 ; I], AM = RR rows kAmRow.  action: 13 x 13 row-major.
inline void action_matrix(const double* C, double* action) {
  std::vector<double> M(kElim * kRows), B(kElim * kReduced, 0.0), X(kRows * kReduced);
  for (int i = 0; i < kElim; ++i) for (int r = 0; r < kRows; ++r) M[i * kRows + r] = C[r * kCols + i];
  for (int i = 0; i < kReduced; ++i) B[(kFirstReduced + i) * kReduced + i] = -1.0;
  fullpiv_solve(M.data(), kElim, kRows, B.data(), kReduced, X.data());
  double RR[kReduced][kBasis];
  for (int i = 0; i < kReduced; ++i)
    for (int c = 0; c < kBasis; ++c) {
      double s = 0.0;
      for (int r = 0; r < kRows; ++r) s += X[r * kReduced + i] * C[r * kCols + kElim + c];
      RR[i][c] = s;
    }
  for (int i = 0; i < kBasis; ++i)
    for (int c = 0; c < kBasis; ++c)
      action[i * kBasis + c] = kAmRow[i] < kReduced ? RR[kAmRow[i]][c] : (kAmRow[i] - kReduced == c ? 1.0 : 0.0);
}

// helper.cc:1384-1408: the eigenvectors over their first row; kept when |Im a1| <= 1e-6; real parts.  sols: up to 13 x (a1 a2 a3 k w).
inline int solutions_from_action(const double* action, double* sols) {
  double H[kBasis * kBasis], wr[kBasis], wi[kBasis], V[kBasis * kBasis];
  for (int i = 0; i < kBasis * kBasis; ++i) H[i] = action[i];
  if (!eig_general_t<27, true>(kBasis, H, wr, wi, V)) return 0;
  const int rows[4] = {kRowA1, kRowA2, kRowK, kRowW};
  int n = 0;
  for (int j = 0; j < kBasis; ++j) {
    // EigenSolver::eigenvectors(): a real column when |Im lambda| <= |Re lambda| * 1e-12 or for the last column, else the pair
    // (V_j + i V_j+1, its conjugate); every column normalised
    const bool real = std::fabs(wi[j]) <= std::fabs(wr[j]) * 1e-12 || j + 1 == kBasis;
    double nrm2 = 0.0;
    for (int i = 0; i < kBasis; ++i)
      nrm2 += real ? V[kBasis * i + j] * V[kBasis * i + j] : V[kBasis * i + j] * V[kBasis * i + j] + V[kBasis * i + j + 1] * V[kBasis * i + j + 1];
    const double nrm = std::sqrt(nrm2);
    for (int c = 0; c < (real ? 1 : 2); ++c) {
      const double sg = c ? -1.0 : 1.0;
      const double v0r = V[j] / nrm, v0i = real ? 0.0 : sg * V[j + 1] / nrm;
      double re[4], im[4];
      for (int k = 0; k < 4; ++k) {
        const double xr = V[kBasis * rows[k] + j] / nrm, xi = real ? 0.0 : sg * V[kBasis * rows[k] + j + 1] / nrm;
        if (real) { re[k] = xr / v0r; im[k] = 0.0; }
        else eig_cdiv(xr, xi, v0r, v0i, &re[k], &im[k]);
      }
      if (im[0] < -1e-6 || im[0] > 1e-6) continue;
      double* s = sols + 5 * n++;
      s[0] = re[0]; s[1] = re[1]; s[2] = wr[j]; s[3] = re[2]; s[4] = re[3];
    }
    if (!real) ++j;
  }
  return n;
}

struct Normalisation {
  double R0[9], t0[3], scale, f0, k0;   // R0: world -> normalised frame (already transposed, :103-109)
  double Nn[32], D[27], d0, U0[3];      // null-space basis (8 x 4, last column the particular solution), D (3 x 9), d(0), U.col(0)
};

// four_point_focal_length_radial_distortion.cc:87-215.  feat: 4 x 2, world: 4 x 3, rot_vec: the three RandDouble(-0.5, 0.5) draws.
inline void normalise(const double* feat, const double* world, const double* rot_vec, Normalisation& n) {
  double d[4], u[2][4], Um[4][4];   // Um: rows x y z 1, one column per point
  for (int i = 0; i < 4; ++i) d[i] = feat[2 * i] * feat[2 * i] + feat[2 * i + 1] * feat[2 * i + 1];
  for (int r = 0; r < 3; ++r) n.t0[r] = ((world[r] + world[3 + r]) + (world[6 + r] + world[9 + r])) / 4.0;
  double A[12];
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) A[r * 4 + c] = world[3 * c + r] - n.t0[r];
  double Us[9];
  svd_u_3x4(A, Us);
  if (det3(Us) < 0.0) for (int r = 0; r < 3; ++r) Us[3 * r] *= -1.0;
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) n.R0[r * 3 + c] = Us[c * 3 + r];
  for (int c = 0; c < 4; ++c) {
    for (int r = 0; r < 3; ++r) Um[r][c] = (n.R0[3 * r] * A[c] + n.R0[3 * r + 1] * A[4 + c]) + n.R0[3 * r + 2] * A[8 + c];
    Um[3][c] = 1.0;
  }
  double cn[4], fn[4];
  for (int c = 0; c < 4; ++c) {
    cn[c] = std::sqrt((Um[0][c] * Um[0][c] + Um[1][c] * Um[1][c]) + Um[2][c] * Um[2][c]);
    fn[c] = std::sqrt(feat[2 * c] * feat[2 * c] + feat[2 * c + 1] * feat[2 * c + 1]);
  }
  n.scale = ((cn[0] + cn[1]) + (cn[2] + cn[3])) / 4.0;
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) Um[r][c] /= n.scale;
  n.f0 = ((fn[0] + fn[1]) + (fn[2] + fn[3])) / 4.0;
  for (int c = 0; c < 4; ++c) { u[0][c] = feat[2 * c] / n.f0; u[1][c] = feat[2 * c + 1] / n.f0; }
  n.k0 = ((d[0] + d[1]) + (d[2] + d[3])) / 4.0;
  for (int c = 0; c < 4; ++c) d[c] /= n.k0;
  // M (5 x 8) -> its transpose Mt (8 x 5) for the QR
  double Mt[40];
  for (int i = 0; i < 40; ++i) Mt[i] = 0.0;
  for (int c = 0; c < 4; ++c) { Mt[c * 5 + 0] = Um[c][0]; Mt[(4 + c) * 5 + 1] = Um[c][0]; }
  for (int k = 1; k < 4; ++k)
    for (int c = 0; c < 4; ++c) { Mt[c * 5 + k + 1] = u[1][k] * Um[c][k]; Mt[(4 + c) * 5 + k + 1] = -u[0][k] * Um[c][k]; }
  Qr f;
  qr_factor(Mt, 8, 5, false, f);
  double Q[64];
  qr_q(Mt, f, Q);
  double Rr[9];
  angle_axis_matrix(std::sqrt((rot_vec[0] * rot_vec[0] + rot_vec[1] * rot_vec[1]) + rot_vec[2] * rot_vec[2]), rot_vec, Rr);
  for (int r = 0; r < 8; ++r)
    for (int c = 0; c < 3; ++c)
      n.Nn[r * 4 + c] = (Q[r * 8 + 5] * Rr[c] + Q[r * 8 + 6] * Rr[3 + c]) + Q[r * 8 + 7] * Rr[6 + c];
  // x0 = Q[:, :5] (R[:5, :5]^T).fullPivLu().solve(b)
  double Rt[25], b[5] = {u[0][0], u[1][0], 0.0, 0.0, 0.0}, y[5];
  for (int r = 0; r < 5; ++r) for (int c = 0; c < 5; ++c) Rt[r * 5 + c] = (c <= r) ? Mt[c * 5 + r] : 0.0;
  fullpiv_solve(Rt, 5, 5, b, 1, y);
  for (int r = 0; r < 8; ++r) {
    double s = 0.0;
    for (int c = 0; c < 5; ++c) s += Q[r * 8 + c] * y[c];
    n.Nn[r * 4 + 3] = s;
  }
  // UN1 = U[:, 1:]^T N[:4], UN2 = U[:, 1:]^T N[4:]  (3 x 4 each), B (6 x 9), C (6 x 3), D = C.colPivHouseholderQr().solve(B)
  double UN[2][3][4];
  for (int h = 0; h < 2; ++h)
    for (int k = 0; k < 3; ++k)
      for (int c = 0; c < 4; ++c) {
        double s = 0.0;
        for (int r = 0; r < 4; ++r) s += Um[r][k + 1] * n.Nn[(4 * h + r) * 4 + c];
        UN[h][k][c] = s;
      }
  double B[54], Cm[18];
  for (int h = 0; h < 2; ++h)
    for (int k = 0; k < 3; ++k) {
      double* row = B + (3 * h + k) * 9;
      for (int c = 0; c < 3; ++c) row[c] = UN[h][k][c];
      for (int c = 0; c < 4; ++c) row[3 + c] = d[k + 1] * UN[h][k][c];
      row[7] = -u[h][k + 1] * Um[2][k + 1];
      row[8] = UN[h][k][3];
      Cm[(3 * h + k) * 3 + 0] = Um[0][k + 1] * u[h][k + 1];
      Cm[(3 * h + k) * 3 + 1] = Um[1][k + 1] * u[h][k + 1];
      Cm[(3 * h + k) * 3 + 2] = Um[3][k + 1] * u[h][k + 1];
    }
  Qr g;
  qr_factor(Cm, 6, 3, true, g);
  qr_solve(Cm, g, B, 9, n.D);
  n.d0 = d[0];
  for (int r = 0; r < 3; ++r) n.U0[r] = Um[r][0];
}

// :219-285.  limits: max_focal_length, min_focal_length, max_distortion, min_distortion (RadialDistUncalibratedAbsolutePoseMetaData).
inline int models_from_solutions(const Normalisation& n, const double* sols, int nsol, const double* limits, double* models) {
  int kept = 0;
  for (int s = 0; s < nsol; ++s) {
    const double* v = sols + 5 * s;
    const double k = v[3], P33 = v[4];
    const double alpha[4] = {v[0], v[1], v[2], 1.0};
    double P[12];
    for (int r = 0; r < 8; ++r) {
      double a = 0.0;
      for (int c = 0; c < 4; ++c) a += n.Nn[r * 4 + c] * alpha[c];
      P[r] = a;
    }
    const double tmp[9] = {alpha[0], alpha[1], alpha[2], k * alpha[0], k * alpha[1], k * alpha[2], k, P33, 1.0};
    double p3[3];
    for (int r = 0; r < 3; ++r) {
      double a = 0.0;
      for (int c = 0; c < 9; ++c) a += n.D[r * 9 + c] * tmp[c];
      p3[r] = a;
    }
    P[8] = p3[0]; P[9] = p3[1]; P[10] = P33; P[11] = p3[2];
    const double n3 = std::sqrt((P[8] * P[8] + P[9] * P[9]) + P[10] * P[10]);
    for (int i = 0; i < 12; ++i) P[i] /= n3;
    const double f = std::sqrt((P[0] * P[0] + P[1] * P[1]) + P[2] * P[2]);
    const double focal = f * n.f0;
    if (focal < limits[1] || focal > limits[0]) continue;
    const double rd = k / n.k0;
    if (rd < limits[2] || rd > limits[3] || rd > 0.0) continue;
    double Rt[12];
    const double kf = 1.0 / f;
    for (int c = 0; c < 4; ++c) { Rt[c] = kf * P[c]; Rt[4 + c] = kf * P[4 + c]; Rt[8 + c] = 1.0 * P[8 + c]; }
    const double R3[9] = {Rt[0], Rt[1], Rt[2], Rt[4], Rt[5], Rt[6], Rt[8], Rt[9], Rt[10]};
    if (det3(R3) < 0.0) for (int i = 0; i < 12; ++i) Rt[i] *= -1.0;
    double RR0[9];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) RR0[r * 3 + c] = (Rt[4 * r] * n.R0[c] + Rt[4 * r + 1] * n.R0[3 + c]) + Rt[4 * r + 2] * n.R0[6 + c];
    double* m = models + kModel * kept++;
    for (int r = 0; r < 3; ++r)
      m[9 + r] = Rt[4 * r + 3] * n.scale - ((RR0[3 * r] * n.t0[0] + RR0[3 * r + 1] * n.t0[1]) + RR0[3 * r + 2] * n.t0[2]);
    for (int i = 0; i < 9; ++i) m[i] = RR0[i];
    m[12] = focal; m[13] = rd;
  }
  return kept;
}

// FourPointsPoseFocalLengthRadialDistortion (:68-288) for one sample.  Returns the number of models (the reference returns
// valid_solutions.size() > 0 and the estimator then asks for rotations.size() > 0).
inline int solve(const double* feat, const double* world, const double* rot_vec, const double* limits, double* models,
                 double* template_out = nullptr, double* action_out = nullptr) {
  Normalisation n;
  normalise(feat, world, rot_vec, n);
  std::vector<double> C(kRows * kCols);
  build_template(n.Nn, n.D, n.d0, n.U0, C.data());
  double action[kBasis * kBasis], sols[5 * kMaxModels];
  action_matrix(C.data(), action);
  if (template_out) for (int i = 0; i < kRows * kCols; ++i) template_out[i] = C[i];
  if (action_out) for (int i = 0; i < kBasis * kBasis; ++i) action_out[i] = action[i];
  const int nsol = solutions_from_action(action, sols);
  return models_from_solutions(n, sols, nsol, limits, models);
}

// RadialDistUncalibratedAbsolutePoseEstimator::Error (estimate_radial_dist_uncalibrated_absolute_pose.cc:130-147) + DistortPoint
// (:56-74).  d: [u v X Y Z]
inline double reprojection_error(const double* m, const double* d) {
  if (m[11] < 0.0) return 1.0e10;
  double p[3];
  for (int r = 0; r < 3; ++r) p[r] = ((m[3 * r] * d[2] + m[3 * r + 1] * d[3]) + m[3 * r + 2] * d[4]) + m[9 + r];
  const double kp[3] = {m[12] * p[0], m[12] * p[1], 1.0 * p[2]};
  const double x = kp[0] / kp[2], y = kp[1] / kp[2];
  const double r2 = x * x + y * y;
  const double denom = 2.0 * m[13] * r2, inner = 1.0 - 4.0 * m[13] * r2;
  double dx = x, dy = y;
  if (!(std::fabs(denom) < 1e-15 || inner < 0.0)) {
    const double sc = (1.0 - std::sqrt(inner)) / denom;
    dx = x * sc; dy = y * sc;
  }
  const double ex = dx - d[0], ey = dy - d[1];
  return ex * ex + ey * ey;
}

}  // namespace p4pfr
