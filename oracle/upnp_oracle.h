// TEST INFRASTRUCTURE (CPU oracle) -- UPnP (Kneip, Li, Seo: "UPnP: an optimal O(n) solution to the absolute pose problem with
// universal applicability"), restated from the reference's live path:
//   sfm/pose/upnp.cc:80-217   cost parameters (H, V_i, G, J; M = sum A_i^T A_i, b = sum A_i^T b_i)
//   sfm/pose/upnp.cc:355-366  ComputeRotations: the 16-solution minimal template's result is OVERWRITTEN by the 8-solution
//                             symmetric one (:362-365), so only BuildActionMatrixUsingSymmetry is live
//   sfm/pose/build_upnp_action_matrix_using_symmetry.cc:2344-2502  input matrix (8 x 24), its reduction, the 141 x 149
//                             template, GaussJordan(140, 121), the 8 x 8 action matrix
//   math/matrix/gauss_jordan.h:46-193  the elimination itself (the reference's own: first-maximum pivots over rows
//                             k .. 139 -- the last row is never searched --, rows swapped, the pivot row DIVIDED by the pivot,
//                             multipliers below 1e-9 skipped, whole rows updated, bottom-up over rows 140 .. 121 only)
//   sfm/pose/upnp.cc:411-437  EigenSolver<8 x 8>, rows 4 .. 7 of the (normalised) eigenvectors' REAL parts as quaternions
//   sfm/pose/upnp.cc:318-351  RemoveDuplicateRotations (0.1 degrees, against the kept ones, last first)
//   sfm/pose/upnp.cc:218-240, 286-343  translations, DiscardBadSolutions (every point in front of its ray)
// A quirk that is REPRODUCED: Upnp::cost_params_ is a member that ComputeCostParameters only ever adds to (upnp.cc:191-200),
// and the RANSAC estimator keeps ONE Upnp object (estimate_rigid_transformation_2d_3d.cc:62, 103), so hypothesis k is solved
// from the SUM of the cost matrices of the samples 0 .. k of that Estimate() call.  UpnpCost carries that state.
// The template's layout (which multiple of which equation a row is, which monomial a column stands for) is upnp_layout.h,
// recovered from the structure of the reference's 2109 assignments by scripts/gen_upnp_layout.py; the input-matrix formulas are
// derived here from the polynomial (the script checks the derivation rule against the reference's text).
// Arithmetic: plain FP64, every sum in the order of the reference's expressions, no fused multiply-adds (the elimination is
// written with Eigen row expressions, not with its pmadd kernels); Eigen's 7 x 7 inverse is restated as a partial-pivot LU +
// substitutions.  PARITY: pinned to 1e-3 per entry by the reference's own golden action matrix
// (build_upnp_action_matrix_using_symmetry_test.cc:49-84 -> tests/golden/upnp_action_matrix.json) and by the known-answer
// scenes of upnp_test.cc; UNPINNED at rounding level (the reference cannot be built here).
// Included by ransac_oracle.cpp inside its anonymous namespace (needs eig_general_t, quat_to_rot).
#include "upnp_layout.h"

namespace upnp {

struct UpnpCost {   // Upnp::CostParameters without gamma (gamma only enters EvaluateCost, which Estimate() never calls)
  double A[100];    // quadratic_penalty_mat, row-major (bitwise symmetric)
  double b[10];
  UpnpCost() { std::memset(A, 0, sizeof(A)); std::memset(b, 0, sizeof(b)); }
};

// UPnP rotation vector s = (w^2 x^2 y^2 z^2 wx wy wz xy xz yz), exponents of (q0 .. q3) = (w x y z)
constexpr int kS[10][4] = {{2, 0, 0, 0}, {0, 2, 0, 0}, {0, 0, 2, 0}, {0, 0, 0, 2}, {1, 1, 0, 0}, {1, 0, 1, 0}, {1, 0, 0, 1}, {0, 1, 1, 0}, {0, 1, 0, 1}, {0, 0, 1, 1}};

struct Tables {
  int mono[24][4];                 // monomials of the 24 input columns: 20 cubics by (e3, e2, e1) ascending, then q0 .. q3
  struct Term { int coef, q, p; };
  Term term[4][24][3]; int nterm[4][24];      // cubic columns: coef * A(q, p); linear columns: coef * b[p] (q unused)
  short src[141][149];             // template entry = input[src / 24][src % 24], -1 = structural zero
  Tables() {
    int n = 0;
    for (int e3 = 0; e3 <= 3; ++e3) for (int e2 = 0; e2 + e3 <= 3; ++e2) for (int e1 = 0; e1 + e2 + e3 <= 3; ++e1) {
      mono[n][0] = 3 - e1 - e2 - e3; mono[n][1] = e1; mono[n][2] = e2; mono[n][3] = e3; ++n;
    }
    for (int k = 0; k < 4; ++k) { for (int t = 0; t < 4; ++t) mono[20 + k][t] = t == k; }
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 24; ++j) {
        int cnt = 0;
        if (j < 20) {
          for (int q = 9; q >= 0; --q)            // terms ordered by q descending
            for (int p = 0; p < 10; ++p) {
              if (kS[p][i] == 0) continue;
              bool hit = true;
              for (int t = 0; t < 4; ++t) hit = hit && (kS[p][t] - (t == i) + kS[q][t] == mono[j][t]);
              if (hit) term[i][j][cnt++] = Term{2 * kS[p][i], q, p};
            }
        } else {
          for (int p = 0; p < 10; ++p) {
            if (kS[p][i] == 0) continue;
            bool hit = true;
            for (int t = 0; t < 4; ++t) hit = hit && (kS[p][t] - (t == i) == mono[j][t]);
            if (hit) term[i][j][cnt++] = Term{2 * kS[p][i], 0, p};
          }
        }
        nterm[i][j] = cnt;
      }
    for (int r = 0; r < 141; ++r) for (int c = 0; c < 149; ++c) src[r][c] = -1;
    for (int r = 0; r < 141; ++r) {
      const int eq = thip::upnp_layout::kRowEq[r];
      for (int j = 0; j < 24; ++j) {
        const bool in = eq == 7 ? (j == 10 || j == 12 || j == 15 || j == 19 || j == 23) : (j == eq || (j >= 7 && j <= 9) || j >= 11);
        if (!in) continue;
        int e[4];
        for (int t = 0; t < 4; ++t) e[t] = thip::upnp_layout::kRowMul[r][t] + mono[j][t];
        int col = -1;
        for (int c = 0; c < 149 && col < 0; ++c) {
          bool same = true;
          for (int t = 0; t < 4; ++t) same = same && thip::upnp_layout::kColMono[c][t] == e[t];
          if (same) col = c;
        }
        src[r][col] = (short)(eq * 24 + j);
      }
    }
  }
};
inline const Tables& tables() { static const Tables t; return t; }

// ---- upnp.cc:80-217.  origin / dir / world: n x 3.  V (n x 9, row-major 3 x 3 each) is what ComputeTranslation reads.
inline void add_cost_parameters(int n, const double* origin, const double* dir, const double* world, UpnpCost* cost, double* V) {
  double Hinv[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  std::vector<double> outer((size_t)9 * n);
  for (int i = 0; i < n; ++i) {
    const double* f = dir + 3 * i;
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { outer[9 * i + 3 * r + c] = f[r] * f[c]; Hinv[3 * r + c] -= outer[9 * i + 3 * r + c]; }
  }
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Hinv[3 * r + c] += (double)n * (r == c ? 1.0 : 0.0);
  double H[9];
  {   // Eigen's 3 x 3 inverse: cofactors over the determinant (Eigen/src/LU/InverseImpl.h compute_inverse_size3_helper)
    const double* m = Hinv;
    auto cof = [&](int i, int j) {
      const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      return m[3 * i1 + j1] * m[3 * i2 + j2] - m[3 * i1 + j2] * m[3 * i2 + j1];
    };
    const double c0 = cof(0, 0), c1 = cof(1, 0), c2 = cof(2, 0);
    const double det = (c0 * m[0] + c1 * m[3]) + c2 * m[6];
    const double invdet = 1.0 / det;
    H[0] = c0 * invdet; H[1] = c1 * invdet; H[2] = c2 * invdet;
    H[3] = cof(0, 1) * invdet; H[4] = cof(1, 1) * invdet; H[5] = cof(2, 1) * invdet;
    H[6] = cof(0, 2) * invdet; H[7] = cof(1, 2) * invdet; H[8] = cof(2, 2) * invdet;
  }
  auto phi = [](const double* X, double* P) {   // LeftMultiply, 3 x 10 row-major (upnp.cc:98-136)
    const double x = X[0], y = X[1], z = X[2];
    const double r0[10] = {x, x, -x, -x, 0.0, 2 * z, -2 * y, 2 * y, 2 * z, 0.0};
    const double r1[10] = {y, -y, y, -y, -2.0 * z, 0.0, 2 * x, 2 * x, 0.0, 2 * z};
    const double r2[10] = {z, -z, -z, z, 2.0 * y, -2.0 * x, 0.0, 0.0, 2.0 * x, 2.0 * y};
    for (int k = 0; k < 10; ++k) { P[k] = r0[k]; P[10 + k] = r1[k]; P[20 + k] = r2[k]; }
  };
  double G[30], J[3] = {0, 0, 0};
  for (int k = 0; k < 30; ++k) G[k] = 0.0;
  for (int i = 0; i < n; ++i) {
    double D[9], P[30];
    for (int k = 0; k < 9; ++k) D[k] = outer[9 * i + k] - ((k % 4 == 0) ? 1.0 : 0.0);
    double* Vi = V + 9 * i;
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Vi[3 * r + c] = (H[3 * r] * D[c] + H[3 * r + 1] * D[3 + c]) + H[3 * r + 2] * D[6 + c];
    phi(world + 3 * i, P);
    const double* o = origin + 3 * i;
    for (int r = 0; r < 3; ++r) J[r] += (Vi[3 * r] * o[0] + Vi[3 * r + 1] * o[1]) + Vi[3 * r + 2] * o[2];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 10; ++c) G[10 * r + c] += (Vi[3 * r] * P[c] + Vi[3 * r + 1] * P[10 + c]) + Vi[3 * r + 2] * P[20 + c];
  }
  for (int i = 0; i < n; ++i) {
    double D[9], P[30], T[30], tb[3], s[3];
    for (int k = 0; k < 9; ++k) D[k] = outer[9 * i + k] - ((k % 4 == 0) ? 1.0 : 0.0);
    phi(world + 3 * i, P);
    for (int k = 0; k < 30; ++k) P[k] = P[k] + G[k];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 10; ++c) T[10 * r + c] = (D[3 * r] * P[c] + D[3 * r + 1] * P[10 + c]) + D[3 * r + 2] * P[20 + c];
    for (int r = 0; r < 3; ++r) s[r] = origin[3 * i + r] + J[r];
    for (int r = 0; r < 3; ++r) tb[r] = ((-D[3 * r]) * s[0] + (-D[3 * r + 1]) * s[1]) + (-D[3 * r + 2]) * s[2];
    for (int p = 0; p < 10; ++p) for (int q = 0; q < 10; ++q) cost->A[10 * p + q] += (T[p] * T[q] + T[10 + p] * T[10 + q]) + T[20 + p] * T[20 + q];
    for (int p = 0; p < 10; ++p) cost->b[p] += (T[p] * tb[0] + T[10 + p] * tb[1]) + T[20 + p] * tb[2];
  }
}

// ---- build_upnp_action_matrix_using_symmetry.cc:2344-2502.  input: 8 x 24 row-major.
inline void input_matrix(const double* A, const double* b, double* M1) {
  const Tables& T = tables();
  for (int k = 0; k < 192; ++k) M1[k] = 0.0;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 24; ++j) {
      double acc = 0.0;
      for (int t = 0; t < T.nterm[i][j]; ++t) {
        const Tables::Term& tm = T.term[i][j][t];
        const double v = (double)tm.coef * (j < 20 ? A[10 * tm.q + tm.p] : b[tm.p]);
        acc = t == 0 ? v : acc + v;
      }
      M1[24 * i + j] = acc;
    }
  for (int i = 0; i < 4; ++i) {   // q_i (|q|^2 - 1)
    for (int k = 0; k < 4; ++k) {
      int e[4] = {0, 0, 0, 0}; e[k] += 2; e[i] += 1;
      for (int j = 0; j < 20; ++j) if (T.mono[j][0] == e[0] && T.mono[j][1] == e[1] && T.mono[j][2] == e[2] && T.mono[j][3] == e[3]) M1[24 * (4 + i) + j] = 1.0;
    }
    M1[24 * (4 + i) + 20 + i] = -1.0;
  }
  // block<7, 24> = block<7, 7>^-1 * block<7, 24>: partial-pivot LU (first maximum), the inverse by substitution, the product
  double LU[49], inv[49];
  int perm[7];
  for (int r = 0; r < 7; ++r) { perm[r] = r; for (int c = 0; c < 7; ++c) LU[7 * r + c] = M1[24 * r + c]; }
  for (int k = 0; k < 7; ++k) {
    int best = k;
    for (int r = k + 1; r < 7; ++r) if (std::fabs(LU[7 * r + k]) > std::fabs(LU[7 * best + k])) best = r;
    if (best != k) { for (int c = 0; c < 7; ++c) std::swap(LU[7 * k + c], LU[7 * best + c]); std::swap(perm[k], perm[best]); }
    for (int r = k + 1; r < 7; ++r) {
      LU[7 * r + k] = LU[7 * r + k] / LU[7 * k + k];
      for (int c = k + 1; c < 7; ++c) LU[7 * r + c] -= LU[7 * r + k] * LU[7 * k + c];
    }
  }
  for (int c = 0; c < 7; ++c) {   // column c of the inverse: L y = P e_c, U x = y
    double y[7];
    for (int r = 0; r < 7; ++r) {
      double v = perm[r] == c ? 1.0 : 0.0;
      for (int k = 0; k < r; ++k) v -= LU[7 * r + k] * y[k];
      y[r] = v;
    }
    for (int r = 6; r >= 0; --r) {
      double v = y[r];
      for (int k = r + 1; k < 7; ++k) v -= LU[7 * r + k] * inv[7 * k + c];
      inv[7 * r + c] = v / LU[7 * r + r];
    }
  }
  double out[7 * 24];
  for (int r = 0; r < 7; ++r)
    for (int c = 0; c < 24; ++c) {
      double acc = inv[7 * r] * M1[c];
      for (int k = 1; k < 7; ++k) acc += inv[7 * r + k] * M1[24 * k + c];
      out[24 * r + c] = acc;
    }
  for (int k = 0; k < 7 * 24; ++k) M1[k] = out[k];
  for (int i = 6; i >= 0; --i) {   // "some more cancellation in column 10"
    const double f = M1[24 * i + 10] / M1[24 * 7 + 10];
    for (int c = 0; c < 24; ++c) M1[24 * i + c] -= f * M1[24 * 7 + c];
  }
}

// math/matrix/gauss_jordan.h:94-193 on the 141 x 149 template (row-major T, pitch 149): GaussJordan(140, 121, T)
inline void gauss_jordan_template(double* T) {
  constexpr int R = 141, C = 149;
  const double kPrecisionThreshold = 1e-9;
  for (int cur = 0; cur < R; ++cur) {
    double maxv = T[C * cur + cur];
    int at = cur;
    for (int r = cur; r < R - 1; ++r) {   // FindLargestAbsoluteValueInColumn(..., starting_row, ending_row = last row): `row < ending_row`
      const double cand = T[C * r + cur];
      if (std::fabs(maxv) < std::fabs(cand)) { maxv = cand; at = r; }
    }
    if (at != cur) for (int c = 0; c < C; ++c) std::swap(T[C * cur + c], T[C * at + c]);
    for (int c = 0; c < C; ++c) T[C * cur + c] /= maxv;
    T[C * cur + cur] = 1.0;
    for (int r = cur + 1; r < R; ++r) {
      const double l = T[C * r + cur];
      if (std::fabs(l) < kPrecisionThreshold) continue;
      for (int c = 0; c < C; ++c) T[C * r + c] -= l * T[C * cur + c];
    }
  }
  for (int cur = R - 1; cur >= 121; --cur)
    for (int r = cur - 1; r >= 121; --r) {
      const double l = T[C * r + cur];
      if (std::fabs(l) < kPrecisionThreshold) continue;
      for (int c = 0; c < C; ++c) T[C * r + c] -= l * T[C * cur + c];
    }
}

// A, b -> 8 x 8 action matrix (row-major).  template_out: the 141 x 149 matrix BEFORE the elimination, or null.
inline void action_matrix(const double* A, const double* b, double* action, double* template_out = nullptr, double* input_out = nullptr) {
  const Tables& Tb = tables();
  double M1[192];
  input_matrix(A, b, M1);
  if (input_out) for (int k = 0; k < 192; ++k) input_out[k] = M1[k];
  std::vector<double> T((size_t)141 * 149, 0.0);
  for (int r = 0; r < 141; ++r) for (int c = 0; c < 149; ++c) if (Tb.src[r][c] >= 0) T[(size_t)149 * r + c] = M1[Tb.src[r][c]];
  if (template_out) for (size_t k = 0; k < T.size(); ++k) template_out[k] = T[k];
  gauss_jordan_template(T.data());
  for (int k = 0; k < 64; ++k) action[k] = 0.0;
  for (int s = 0; s < 8; ++s) for (int r = 0; r < 4; ++r) action[8 * r + s] = 0.0 - T[(size_t)149 * (121 + r) + 141 + s];
  for (int r = 0; r < 4; ++r) action[8 * (4 + r) + r] = 1.0;
}

inline void quat_mul(const double* a, const double* b, double* o) {   // (w x y z); Eigen's scalar quaternion product
  o[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  o[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  o[2] = a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3];
  o[3] = a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1];
}
inline void quat_rotate(const double* q, const double* v, double* o) {   // QuaternionBase::_transformVector
  double uv[3] = {q[2] * v[2] - q[3] * v[1], q[3] * v[0] - q[1] * v[2], q[1] * v[1] - q[2] * v[0]};
  for (int k = 0; k < 3; ++k) uv[k] += uv[k];
  const double c[3] = {q[2] * uv[2] - q[3] * uv[1], q[3] * uv[0] - q[1] * uv[2], q[1] * uv[1] - q[2] * uv[0]};
  for (int k = 0; k < 3; ++k) o[k] = (v[k] + q[0] * uv[k]) + c[k];
}
inline double quat_angular_distance(const double* a, const double* b) {   // QuaternionBase::angularDistance
  const double bc[4] = {b[0], -b[1], -b[2], -b[3]};
  double d[4];
  quat_mul(a, bc, d);
  return 2.0 * std::atan2(std::sqrt((d[1] * d[1] + d[2] * d[2]) + d[3] * d[3]), std::fabs(d[0]));
}

// upnp.cc:411-437 + :318-351: candidate rotations from the action matrix, duplicates removed.  quats: up to 8 x (w x y z).
inline int rotations_from_action(const double* action, double* quats) {
  double H[64], wr[8], wi[8], V[64];
  for (int k = 0; k < 64; ++k) H[k] = action[k];
  if (!eig_general_t<8, true>(8, H, wr, wi, V)) return 0;
  double cand[32];
  for (int j = 0; j < 8; ++j) {
    // EigenSolver::eigenvectors(): "real" when |imag| <= |real| * 1e-12 or for the last column
    const bool real = std::fabs(wi[j]) <= std::fabs(wr[j]) * 1e-12 || j + 1 == 8;
    double nrm2 = 0.0;
    for (int i = 0; i < 8; ++i) nrm2 += real ? V[8 * i + j] * V[8 * i + j] : V[8 * i + j] * V[8 * i + j] + V[8 * i + j + 1] * V[8 * i + j + 1];
    const double nrm = std::sqrt(nrm2);
    const int cols = real ? 1 : 2;
    for (int c = 0; c < cols; ++c) {
      double q[4];
      for (int k = 0; k < 4; ++k) q[k] = V[8 * (4 + k) + j] / nrm;   // the real part is the same for both columns of a pair
      const double qn = std::sqrt(((q[1] * q[1] + q[2] * q[2]) + q[3] * q[3]) + q[0] * q[0]);   // coeffs (x y z w)
      for (int k = 0; k < 4; ++k) cand[4 * (j + c) + k] = q[k] / qn;
    }
    if (!real) ++j;
  }
  const double kAngleThreshold = 0.1 * (M_PI / 180.0);
  int n = 0;
  for (int i = 0; i < 8; ++i) {
    bool dup = false;
    for (int j = n - 1; j >= 0 && !dup; --j) dup = quat_angular_distance(cand + 4 * i, quats + 4 * j) < kAngleThreshold;
    if (!dup) { for (int k = 0; k < 4; ++k) quats[4 * n + k] = cand[4 * i + k]; ++n; }
  }
  return n;
}

// Upnp::EstimatePose (upnp.cc:462-493) with solution_costs = nullptr.  cost: the estimator's running state (see the header).
inline int estimate_pose(int n, const double* origin, const double* dir, const double* world, UpnpCost* cost, double* quats, double* ts) {
  std::vector<double> V((size_t)9 * n);
  add_cost_parameters(n, origin, dir, world, cost, V.data());
  double action[64], cand[32];
  action_matrix(cost->A, cost->b, action);
  const int nc = rotations_from_action(action, cand);
  int kept = 0;
  for (int s = 0; s < nc; ++s) {
    const double* q = cand + 4 * s;
    double t[3] = {0, 0, 0};
    for (int i = 0; i < n; ++i) {
      double rx[3], d[3];
      quat_rotate(q, world + 3 * i, rx);
      for (int k = 0; k < 3; ++k) d[k] = rx[k] - origin[3 * i + k];
      const double* Vi = V.data() + 9 * i;
      for (int r = 0; r < 3; ++r) t[r] += (Vi[3 * r] * d[0] + Vi[3 * r + 1] * d[1]) + Vi[3 * r + 2] * d[2];
    }
    bool front = true;
    for (int i = 0; i < n && front; ++i) {
      double rx[3], p[3];
      quat_rotate(q, world + 3 * i, rx);
      for (int k = 0; k < 3; ++k) p[k] = (rx[k] + t[k]) - origin[3 * i + k];
      // Quaterniond::FromTwoVectors(ray, UnitZ): v0 = ray.normalized(), c = v0.z; axis = v0 x z = (v0y, -v0x, 0)
      const double* f = dir + 3 * i;
      const double fn = std::sqrt((f[0] * f[0] + f[1] * f[1]) + f[2] * f[2]);
      const double v0[3] = {f[0] / fn, f[1] / fn, f[2] / fn};
      const double c = v0[2];
      double u[4];
      if (c < -1.0 + 1e-12) {   // (Eigen's SVD branch for opposite vectors: a half turn about an axis orthogonal to the ray)
        const double w2 = (1.0 + c) * 0.5;
        u[0] = std::sqrt(w2 > 0 ? w2 : 0.0);
        const double s2 = std::sqrt(1.0 - w2);
        u[1] = s2; u[2] = 0.0; u[3] = 0.0;
      } else {
        const double sq = std::sqrt((1.0 + c) * 2.0), invs = 1.0 / sq;
        u[1] = v0[1] * invs; u[2] = -v0[0] * invs; u[3] = 0.0 * invs; u[0] = sq * 0.5;
      }
      double rp[3];
      quat_rotate(u, p, rp);
      if (rp[2] < 0) front = false;
    }
    if (front) { for (int k = 0; k < 4; ++k) quats[4 * kept + k] = q[k]; for (int k = 0; k < 3; ++k) ts[3 * kept + k] = t[k]; ++kept; }
  }
  return kept;
}

}  // namespace upnp
