// ransac_device.h -- device numerics of the RANSAC minimal solvers (gfx950, FP64).
//
// One thread runs one minimal problem.  The linear algebra restates the
// published algorithms behind the Eigen calls of the reference:
//   Eigen::FullPivLU   (five_point_relative_pose.cc:242-247,261-263)
//   Eigen::EigenSolver (five_point_relative_pose.cc:275; companion-matrix
//                       roots, math/find_polynomial_roots_companion_matrix.cc)
//                      = EISPACK orthes + hqr2
//   Eigen::JacobiSVD   (essential_matrix_utils.cc:66-67; 9x9 in sqpnp.cc:234) = two-sided Jacobi
// Built with -ffp-contract=off: the operation order below is kept identical to
// the CPU oracle's transcription of the same algorithms, which is what makes
// RANSAC inlier sets bit-identical between the two (DESIGN.md "RANSAC parity").
#pragma once
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cstdint>

namespace thip {
namespace rsc {

#define RDEV __device__ inline

template <typename T>
RDEV void dswap(T& a, T& b) { T t = a; a = b; b = t; }

// ------------------------------------------------------- small linear algebra
// Full-pivot LU of a rows x cols row-major matrix (Eigen::FullPivLU::compute).
// The pivot search scans the remaining corner column by column and keeps the
// first strict maximum.
struct FullPivLU {
  int rows, cols, size, nonzero_pivots;
  double maxpivot;
  int rowt[10], colt[20];
};

RDEV void fullpiv_lu(double* A, int rows, int cols, FullPivLU& f) {
  f.rows = rows; f.cols = cols; f.size = rows < cols ? rows : cols;
  f.nonzero_pivots = f.size; f.maxpivot = 0.0;
  for (int k = 0; k < f.size; ++k) {
    double best = -1.0; int br = k, bc = k;
    for (int j = k; j < cols; ++j)
      for (int i = k; i < rows; ++i) {
        const double a = fabs(A[i * cols + j]);
        if (a > best) { best = a; br = i; bc = j; }
      }
    if (best == 0.0) {
      f.nonzero_pivots = k;
      for (int i = k; i < f.size; ++i) { f.rowt[i] = i; f.colt[i] = i; }
      break;
    }
    if (best > f.maxpivot) f.maxpivot = best;
    f.rowt[k] = br; f.colt[k] = bc;
    if (br != k) for (int j = 0; j < cols; ++j) dswap(A[k * cols + j], A[br * cols + j]);
    if (bc != k) for (int i = 0; i < rows; ++i) dswap(A[i * cols + k], A[i * cols + bc]);
    if (k < rows - 1) for (int i = k + 1; i < rows; ++i) A[i * cols + k] /= A[k * cols + k];
    if (k < f.size - 1)
      for (int i = k + 1; i < rows; ++i)
        for (int j = k + 1; j < cols; ++j) A[i * cols + j] -= A[i * cols + k] * A[k * cols + j];
  }
}

// Real general eigen-decomposition (EISPACK orthes + hqr2, as in JAMA) of an
// n x n row-major matrix, n <= 10.  H is destroyed.  wr/wi: eigenvalues; if V
// is non-null it receives the (un-normalised) eigenvectors of the REAL
// eigenvalues in the corresponding columns (row-major n x n).
// complex division (xr + i xi) / (yr + i yi), Smith's scaling (EISPACK cdiv)
RDEV inline void eig_cdiv(double xr, double xi, double yr, double yi, double* cr, double* ci) {
  if (fabs(yr) > fabs(yi)) {
    const double r = yi / yr, d = yr + r * yi;
    *cr = (xr + r * xi) / d; *ci = (xi - r * xr) / d;
  } else {
    const double r = yr / yi, d = yi + r * yr;
    *cr = (r * xr + xi) / d; *ci = (r * xi - xr) / d;
  }
}
const int EIG_MAXN = 10;
// MAXN: largest n of the instantiation (sizes the one local array).  CPLX: also back-substitute the complex
// conjugate pairs (EISPACK hqr2 convention: for wi[j] > 0 the columns j, j + 1 of V hold the real and the
// imaginary part of the eigenvector of wr[j] + i wi[j]; the DLS solver reads them, dls_pnp.cc:147-161).
template <int MAXN, bool CPLX>
RDEV bool eig_general_t(int nn, double* H, double* wr, double* wi, double* V) {
#define HH(i, j) H[(i) * nn + (j)]
#define VV(i, j) V[(i) * nn + (j)]
  const int low = 0, high = nn - 1;
  double ort[MAXN];
  // ---- orthes: reduce to Hessenberg form
  for (int m = low + 1; m <= high - 1; ++m) {
    double scale = 0.0;
    for (int i = m; i <= high; ++i) scale += fabs(HH(i, m - 1));
    if (scale != 0.0) {
      double h = 0.0;
      for (int i = high; i >= m; --i) { ort[i] = HH(i, m - 1) / scale; h += ort[i] * ort[i]; }
      double g = sqrt(h);
      if (ort[m] > 0) g = -g;
      h = h - ort[m] * g;
      ort[m] = ort[m] - g;
      for (int j = m; j < nn; ++j) {
        double f = 0.0;
        for (int i = high; i >= m; --i) f += ort[i] * HH(i, j);
        f = f / h;
        for (int i = m; i <= high; ++i) HH(i, j) -= f * ort[i];
      }
      for (int i = 0; i <= high; ++i) {
        double f = 0.0;
        for (int j = high; j >= m; --j) f += ort[j] * HH(i, j);
        f = f / h;
        for (int j = m; j <= high; ++j) HH(i, j) -= f * ort[j];
      }
      ort[m] = scale * ort[m];
      HH(m, m - 1) = scale * g;
    } else {
      ort[m] = 0.0;
    }
  }
  if (V) {
    for (int i = 0; i < nn; ++i) for (int j = 0; j < nn; ++j) VV(i, j) = (i == j) ? 1.0 : 0.0;
    for (int m = high - 1; m >= low + 1; --m) {
      if (HH(m, m - 1) != 0.0) {
        for (int i = m + 1; i <= high; ++i) ort[i] = HH(i, m - 1);
        for (int j = m; j <= high; ++j) {
          double g = 0.0;
          for (int i = m; i <= high; ++i) g += ort[i] * VV(i, j);
          g = (g / ort[m]) / HH(m, m - 1);
          for (int i = m; i <= high; ++i) VV(i, j) += g * ort[i];
        }
      }
    }
  }
  // ---- hqr2
  int n = nn - 1;
  const double eps = 2.220446049250313e-16;
  double exshift = 0.0;
  double p = 0, q = 0, r = 0, s = 0, z = 0, t, w, x, y;
  double norm = 0.0;
  for (int i = 0; i < nn; ++i)
    for (int j = (i - 1 > 0 ? i - 1 : 0); j < nn; ++j) norm += fabs(HH(i, j));
  int iter = 0, total_iter = 0;
  while (n >= low) {
    int l = n;
    while (l > low) {
      s = fabs(HH(l - 1, l - 1)) + fabs(HH(l, l));
      if (s == 0.0) s = norm;
      if (fabs(HH(l, l - 1)) < eps * s) break;
      l--;
    }
    if (l == n) {  // one root
      HH(n, n) = HH(n, n) + exshift;
      wr[n] = HH(n, n); wi[n] = 0.0;
      n--; iter = 0;
    } else if (l == n - 1) {  // two roots
      w = HH(n, n - 1) * HH(n - 1, n);
      p = (HH(n - 1, n - 1) - HH(n, n)) / 2.0;
      q = p * p + w;
      z = sqrt(fabs(q));
      HH(n, n) = HH(n, n) + exshift;
      HH(n - 1, n - 1) = HH(n - 1, n - 1) + exshift;
      x = HH(n, n);
      if (q >= 0) {  // real pair
        z = (p >= 0) ? p + z : p - z;
        wr[n - 1] = x + z;
        wr[n] = wr[n - 1];
        if (z != 0.0) wr[n] = x - w / z;
        wi[n - 1] = 0.0; wi[n] = 0.0;
        x = HH(n, n - 1);
        s = fabs(x) + fabs(z);
        p = x / s; q = z / s;
        r = sqrt(p * p + q * q);
        p = p / r; q = q / r;
        for (int j = n - 1; j < nn; ++j) { z = HH(n - 1, j); HH(n - 1, j) = q * z + p * HH(n, j); HH(n, j) = q * HH(n, j) - p * z; }
        for (int i = 0; i <= n; ++i) { z = HH(i, n - 1); HH(i, n - 1) = q * z + p * HH(i, n); HH(i, n) = q * HH(i, n) - p * z; }
        if (V) for (int i = low; i <= high; ++i) { z = VV(i, n - 1); VV(i, n - 1) = q * z + p * VV(i, n); VV(i, n) = q * VV(i, n) - p * z; }
      } else {  // complex pair
        wr[n - 1] = x + p; wr[n] = x + p; wi[n - 1] = z; wi[n] = -z;
      }
      n = n - 2; iter = 0;
    } else {
      x = HH(n, n); y = 0.0; w = 0.0;
      if (l < n) { y = HH(n - 1, n - 1); w = HH(n, n - 1) * HH(n - 1, n); }
      if (iter == 10) {  // Wilkinson's original ad hoc shift
        exshift += x;
        for (int i = low; i <= n; ++i) HH(i, i) -= x;
        s = fabs(HH(n, n - 1)) + fabs(HH(n - 1, n - 2));
        x = y = 0.75 * s;
        w = -0.4375 * s * s;
      }
      if (iter == 30) {  // MATLAB's new ad hoc shift
        s = (y - x) / 2.0;
        s = s * s + w;
        if (s > 0) {
          s = sqrt(s);
          if (y < x) s = -s;
          s = x - w / ((y - x) / 2.0 + s);
          for (int i = low; i <= n; ++i) HH(i, i) -= s;
          exshift += s;
          x = y = w = 0.964;
        }
      }
      iter = iter + 1;
      if (++total_iter > 40 * nn) return false;  // Eigen: m_maxIterationsPerRow * size
      int m = n - 2;
      while (m >= l) {
        z = HH(m, m);
        r = x - z; s = y - z;
        p = (r * s - w) / HH(m + 1, m) + HH(m, m + 1);
        q = HH(m + 1, m + 1) - z - r - s;
        r = HH(m + 2, m + 1);
        s = fabs(p) + fabs(q) + fabs(r);
        p = p / s; q = q / s; r = r / s;
        if (m == l) break;
        if (fabs(HH(m, m - 1)) * (fabs(q) + fabs(r)) <
            eps * (fabs(p) * (fabs(HH(m - 1, m - 1)) + fabs(z) + fabs(HH(m + 1, m + 1))))) break;
        m--;
      }
      for (int i = m + 2; i <= n; ++i) { HH(i, i - 2) = 0.0; if (i > m + 2) HH(i, i - 3) = 0.0; }
      for (int k = m; k <= n - 1; ++k) {
        const bool notlast = (k != n - 1);
        if (k != m) {
          p = HH(k, k - 1); q = HH(k + 1, k - 1);
          r = notlast ? HH(k + 2, k - 1) : 0.0;
          x = fabs(p) + fabs(q) + fabs(r);
          if (x == 0.0) continue;
          p = p / x; q = q / x; r = r / x;
        }
        s = sqrt(p * p + q * q + r * r);
        if (p < 0) s = -s;
        if (s != 0) {
          if (k != m) HH(k, k - 1) = -s * x;
          else if (l != m) HH(k, k - 1) = -HH(k, k - 1);
          p = p + s; x = p / s; y = q / s; z = r / s; q = q / p; r = r / p;
          for (int j = k; j < nn; ++j) {
            p = HH(k, j) + q * HH(k + 1, j);
            if (notlast) { p = p + r * HH(k + 2, j); HH(k + 2, j) = HH(k + 2, j) - p * z; }
            HH(k, j) = HH(k, j) - p * x;
            HH(k + 1, j) = HH(k + 1, j) - p * y;
          }
          const int imax = (n < k + 3) ? n : k + 3;
          for (int i = 0; i <= imax; ++i) {
            p = x * HH(i, k) + y * HH(i, k + 1);
            if (notlast) { p = p + z * HH(i, k + 2); HH(i, k + 2) = HH(i, k + 2) - p * r; }
            HH(i, k) = HH(i, k) - p;
            HH(i, k + 1) = HH(i, k + 1) - p * q;
          }
          if (V) for (int i = low; i <= high; ++i) {
            p = x * VV(i, k) + y * VV(i, k + 1);
            if (notlast) { p = p + z * VV(i, k + 2); VV(i, k + 2) = VV(i, k + 2) - p * r; }
            VV(i, k) = VV(i, k) - p;
            VV(i, k + 1) = VV(i, k + 1) - p * q;
          }
        }
      }
    }
  }
  if (!V) return true;
  if (norm == 0.0) return true;
  // back-substitution for the REAL eigenvectors of the quasi-triangular form
  for (n = nn - 1; n >= 0; --n) {
    p = wr[n]; q = wi[n];
    if (CPLX && q < 0 && n > 0) {   // second member of a conjugate pair: columns n - 1 (real part) and n (imaginary part)
      int l = n - 1;
      if (fabs(HH(n, n - 1)) > fabs(HH(n - 1, n))) {
        HH(n - 1, n - 1) = q / HH(n, n - 1);
        HH(n - 1, n) = -(HH(n, n) - p) / HH(n, n - 1);
      } else {
        double cr, ci;
        eig_cdiv(0.0, -HH(n - 1, n), HH(n - 1, n - 1) - p, q, &cr, &ci);
        HH(n - 1, n - 1) = cr; HH(n - 1, n) = ci;
      }
      HH(n, n - 1) = 0.0; HH(n, n) = 1.0;
      double lastra = 0.0, lastsa = 0.0, lastw = 0.0;
      for (int i = n - 2; i >= 0; --i) {
        double ra = 0.0, sa = 0.0;
        for (int j = l; j <= n; ++j) { ra = ra + HH(i, j) * HH(j, n - 1); sa = sa + HH(i, j) * HH(j, n); }
        w = HH(i, i) - p;
        if (wi[i] < 0.0) { lastw = w; lastra = ra; lastsa = sa; continue; }
        l = i;
        double cr, ci;
        if (wi[i] == 0.0) {
          eig_cdiv(-ra, -sa, w, q, &cr, &ci);
          HH(i, n - 1) = cr; HH(i, n) = ci;
        } else {
          x = HH(i, i + 1); y = HH(i + 1, i);
          double vr = (wr[i] - p) * (wr[i] - p) + wi[i] * wi[i] - q * q;
          const double vi = (wr[i] - p) * 2.0 * q;
          if (vr == 0.0 && vi == 0.0) vr = eps * norm * (fabs(w) + fabs(q) + fabs(x) + fabs(y) + fabs(lastw));
          eig_cdiv(x * lastra - lastw * ra + q * sa, x * lastsa - lastw * sa - q * ra, vr, vi, &cr, &ci);
          HH(i, n - 1) = cr; HH(i, n) = ci;
          if (fabs(x) > (fabs(lastw) + fabs(q))) {
            HH(i + 1, n - 1) = (-ra - w * HH(i, n - 1) + q * HH(i, n)) / x;
            HH(i + 1, n) = (-sa - w * HH(i, n) - q * HH(i, n - 1)) / x;
          } else {
            eig_cdiv(-lastra - y * HH(i, n - 1), -lastsa - y * HH(i, n), lastw, q, &cr, &ci);
            HH(i + 1, n - 1) = cr; HH(i + 1, n) = ci;
          }
        }
        t = fmax(fabs(HH(i, n - 1)), fabs(HH(i, n)));
        if ((eps * t) * t > 1) for (int j = i; j <= n; ++j) { HH(j, n - 1) = HH(j, n - 1) / t; HH(j, n) = HH(j, n) / t; }
      }
      continue;
    }
    if (q != 0) continue;
    int l = n;
    HH(n, n) = 1.0;
    for (int i = n - 1; i >= 0; --i) {
      w = HH(i, i) - p;
      r = 0.0;
      for (int j = l; j <= n; ++j) r = r + HH(i, j) * HH(j, n);
      if (wi[i] < 0.0) { z = w; s = r; }
      else {
        l = i;
        if (wi[i] == 0.0) {
          if (w != 0.0) HH(i, n) = -r / w;
          else HH(i, n) = -r / (eps * norm);
        } else {
          x = HH(i, i + 1); y = HH(i + 1, i);
          q = (wr[i] - p) * (wr[i] - p) + wi[i] * wi[i];
          t = (x * s - z * r) / q;
          HH(i, n) = t;
          if (fabs(x) > fabs(z)) HH(i + 1, n) = (-r - w * t) / x;
          else HH(i + 1, n) = (-s - y * t) / z;
        }
        t = fabs(HH(i, n));
        if ((eps * t) * t > 1) for (int j = i; j <= n; ++j) HH(j, n) = HH(j, n) / t;
      }
    }
  }
  // back transformation (only the real-eigenvalue columns are meaningful)
  for (int j = nn - 1; j >= low; --j) {
    if (!CPLX && wi[j] != 0) continue;
    for (int i = low; i <= high; ++i) {
      z = 0.0;
      for (int k = low; k <= j; ++k) z = z + VV(i, k) * HH(k, j);
      VV(i, j) = z;
    }
  }
  return true;
#undef HH
#undef VV
}
RDEV inline bool eig_real_general(int nn, double* H, double* wr, double* wi, double* V) {
  return eig_general_t<EIG_MAXN, false>(nn, H, wr, wi, V);
}

// Two-sided Jacobi SVD of a 3x3 row-major matrix (the algorithm of
// Eigen::JacobiSVD for square real matrices): A = U diag(s) V^T, s sorted
// descending, s >= 0.
RDEV void jacobi_rot_sym(double x, double y, double z, double* c, double* s) {
  // JacobiRotation::makeJacobi for the symmetric 2x2 [[x, y], [y, z]]
  const double deno = 2.0 * fabs(y);
  if (deno < DBL_MIN) { *c = 1.0; *s = 0.0; return; }
  const double tau = (x - z) / deno;
  const double w = sqrt(tau * tau + 1.0);
  const double t = (tau > 0) ? 1.0 / (tau + w) : 1.0 / (tau - w);
  const double sign_t = t > 0 ? 1.0 : -1.0;
  const double n = 1.0 / sqrt(t * t + 1.0);
  *s = -sign_t * (y / fabs(y)) * fabs(t) * n;
  *c = n;
}

// the sweeps, the sign fix and the sort on a work matrix W (destroyed); U, V hold their starting values (identity, or the
// preconditioner's: p4pfr_kernels.hip starts U at a column permutation)
RDEV void svd3_sweeps(double* W, double* U, double* S, double* V) {
  const double precision = 2.0 * DBL_EPSILON;
  double maxdiag = fmax(fabs(W[0]), fmax(fabs(W[4]), fabs(W[8])));
  bool finished = false;
  int sweeps = 0;
  while (!finished && sweeps++ < 64) {
    finished = true;
    for (int p = 1; p < 3; ++p)
      for (int q = 0; q < p; ++q) {
        const double threshold = fmax(DBL_MIN, precision * maxdiag);
        if (fabs(W[p * 3 + q]) > threshold || fabs(W[q * 3 + p]) > threshold) {
          finished = false;
          // real_2x2_jacobi_svd on [[W_pp, W_pq], [W_qp, W_qq]]
          double m00 = W[p * 3 + p], m01 = W[p * 3 + q], m10 = W[q * 3 + p], m11 = W[q * 3 + q];
          double r1c, r1s;
          const double tt = m00 + m11, dd = m10 - m01;
          if (fabs(dd) < DBL_MIN) { r1c = 1.0; r1s = 0.0; }
          else { const double u = tt / dd; const double tmp = sqrt(1.0 + u * u); r1s = 1.0 / tmp; r1c = u / tmp; }
          // m <- rot1 * m  (rows: r0' = c r0 + s r1 ; r1' = -s r0 + c r1)
          const double n00 = r1c * m00 + r1s * m10, n01 = r1c * m01 + r1s * m11;
          const double n11 = -r1s * m01 + r1c * m11;
          double jc, js;
          jacobi_rot_sym(n00, n01, n11, &jc, &js);
          // left rotation L = rot1 * jr^T   (jr = [[jc, js], [-js, jc]])
          const double lc = r1c * jc + r1s * js, ls = r1s * jc - r1c * js;
          // W <- L on rows (p, q): row_p' = lc row_p + ls row_q ; row_q' = -ls row_p + lc row_q
          for (int k = 0; k < 3; ++k) {
            const double a = W[p * 3 + k], b = W[q * 3 + k];
            W[p * 3 + k] = lc * a + ls * b; W[q * 3 + k] = -ls * a + lc * b;
          }
          // U <- U L^T on columns (p, q)
          for (int k = 0; k < 3; ++k) {
            const double a = U[k * 3 + p], b = U[k * 3 + q];
            U[k * 3 + p] = lc * a + ls * b; U[k * 3 + q] = -ls * a + lc * b;
          }
          // W <- W jr on columns (p, q): col_p' = jc col_p - js col_q ; col_q' = js col_p + jc col_q
          for (int k = 0; k < 3; ++k) {
            const double a = W[k * 3 + p], b = W[k * 3 + q];
            W[k * 3 + p] = jc * a - js * b; W[k * 3 + q] = js * a + jc * b;
          }
          for (int k = 0; k < 3; ++k) {
            const double a = V[k * 3 + p], b = V[k * 3 + q];
            V[k * 3 + p] = jc * a - js * b; V[k * 3 + q] = js * a + jc * b;
          }
          maxdiag = fmax(maxdiag, fmax(fabs(W[p * 3 + p]), fabs(W[q * 3 + q])));
        }
      }
  }
  for (int i = 0; i < 3; ++i) {
    const double a = W[i * 3 + i];
    S[i] = fabs(a);
    if (a < 0) for (int k = 0; k < 3; ++k) U[k * 3 + i] = -U[k * 3 + i];
  }
  for (int i = 0; i < 3; ++i) {  // selection sort, descending
    int best = i;
    for (int j = i + 1; j < 3; ++j) if (S[j] > S[best]) best = j;
    if (best != i) {
      dswap(S[i], S[best]);
      for (int k = 0; k < 3; ++k) { dswap(U[k * 3 + i], U[k * 3 + best]); dswap(V[k * 3 + i], V[k * 3 + best]); }
    }
  }
}

RDEV void svd3(const double* Ain, double* U, double* S, double* V) {
  double W[9];
  double scale = 0.0;
  for (int i = 0; i < 9; ++i) scale = fmax(scale, fabs(Ain[i]));
  if (scale == 0.0) scale = 1.0;
  for (int i = 0; i < 9; ++i) W[i] = Ain[i] / scale;
  for (int i = 0; i < 9; ++i) { U[i] = (i % 4 == 0) ? 1.0 : 0.0; V[i] = U[i]; }
  svd3_sweeps(W, U, S, V);
  for (int i = 0; i < 3; ++i) S[i] *= scale;
}

RDEV double det3(const double* M) {
  return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}

// ------------------------------------------------------------------ SQPnP
// sfm/pose/sqpnp.cc:23-353 + sqpnp_helper.{h,cc}.  All 3x3 matrices stored as
// 9-vectors are ROW-MAJOR (as in the reference).
//
// Two-sided Jacobi SVD of an N x N row-major matrix (Eigen::JacobiSVD for square
// real input: no QR preconditioner), generalisation of svd3 above.
template <int N>
RDEV void svd_sq(const double* Ain, double* U, double* S, double* V) {
  double W[N * N];
  double scale = 0.0;
  for (int i = 0; i < N * N; ++i) scale = fmax(scale, fabs(Ain[i]));
  if (scale == 0.0) scale = 1.0;
  for (int i = 0; i < N * N; ++i) W[i] = Ain[i] / scale;
  for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) { U[i * N + j] = (i == j) ? 1.0 : 0.0; V[i * N + j] = U[i * N + j]; }
  const double precision = 2.0 * DBL_EPSILON;
  double maxdiag = 0.0;
  for (int i = 0; i < N; ++i) maxdiag = fmax(maxdiag, fabs(W[i * N + i]));
  bool finished = false;
  int sweeps = 0;
  while (!finished && sweeps++ < 64) {
    finished = true;
    for (int p = 1; p < N; ++p)
      for (int q = 0; q < p; ++q) {
        const double threshold = fmax(DBL_MIN, precision * maxdiag);
        if (fabs(W[p * N + q]) > threshold || fabs(W[q * N + p]) > threshold) {
          finished = false;
          const double m00 = W[p * N + p], m01 = W[p * N + q], m10 = W[q * N + p], m11 = W[q * N + q];
          double r1c, r1s;
          const double tt = m00 + m11, dd = m10 - m01;
          if (fabs(dd) < DBL_MIN) { r1c = 1.0; r1s = 0.0; }
          else { const double u = tt / dd; const double tmp = sqrt(1.0 + u * u); r1s = 1.0 / tmp; r1c = u / tmp; }
          const double n00 = r1c * m00 + r1s * m10, n01 = r1c * m01 + r1s * m11;
          const double n11 = -r1s * m01 + r1c * m11;
          double jc, js;
          jacobi_rot_sym(n00, n01, n11, &jc, &js);
          const double lc = r1c * jc + r1s * js, ls = r1s * jc - r1c * js;
          for (int k = 0; k < N; ++k) {
            const double a = W[p * N + k], b = W[q * N + k];
            W[p * N + k] = lc * a + ls * b; W[q * N + k] = -ls * a + lc * b;
          }
          for (int k = 0; k < N; ++k) {
            const double a = U[k * N + p], b = U[k * N + q];
            U[k * N + p] = lc * a + ls * b; U[k * N + q] = -ls * a + lc * b;
          }
          for (int k = 0; k < N; ++k) {
            const double a = W[k * N + p], b = W[k * N + q];
            W[k * N + p] = jc * a - js * b; W[k * N + q] = js * a + jc * b;
          }
          for (int k = 0; k < N; ++k) {
            const double a = V[k * N + p], b = V[k * N + q];
            V[k * N + p] = jc * a - js * b; V[k * N + q] = js * a + jc * b;
          }
          maxdiag = fmax(maxdiag, fmax(fabs(W[p * N + p]), fabs(W[q * N + q])));
        }
      }
  }
  for (int i = 0; i < N; ++i) {
    const double a = W[i * N + i];
    S[i] = fabs(a);
    if (a < 0) for (int k = 0; k < N; ++k) U[k * N + i] = -U[k * N + i];
  }
  for (int i = 0; i < N; ++i) {  // selection sort, descending
    int best = i;
    for (int j = i + 1; j < N; ++j) if (S[j] > S[best]) best = j;
    if (best != i) {
      dswap(S[i], S[best]);
      for (int k = 0; k < N; ++k) { dswap(U[k * N + i], U[k * N + best]); dswap(V[k * N + i], V[k * N + best]); }
    }
  }
  for (int i = 0; i < N; ++i) S[i] *= scale;
}

// Eigen::Quaternion(Matrix3) and Quaternion::toRotationMatrix (Geometry/Quaternion.h).
// M row-major; q = [w, x, y, z].
RDEV void rot_to_quat(const double* M, double* q) {
  double t = (M[0] + M[4]) + M[8];
  if (t > 0.0) {
    t = sqrt(t + 1.0);
    q[0] = 0.5 * t;
    t = 0.5 / t;
    q[1] = (M[7] - M[5]) * t; q[2] = (M[2] - M[6]) * t; q[3] = (M[3] - M[1]) * t;
  } else {
    int i = 0;
    if (M[4] > M[0]) i = 1;
    if (M[8] > M[i * 3 + i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = sqrt(M[i * 3 + i] - M[j * 3 + j] - M[k * 3 + k] + 1.0);
    q[1 + i] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (M[k * 3 + j] - M[j * 3 + k]) * t;
    q[1 + j] = (M[j * 3 + i] + M[i * 3 + j]) * t;
    q[1 + k] = (M[k * 3 + i] + M[i * 3 + k]) * t;
  }
}
RDEV void quat_to_rot(const double* q, double* R) {
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1.0 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = 1.0 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1.0 - (txx + tyy);
}

namespace sqp {
#define SQP_RANK_TOL 1e-7
#define SQP_DET_THRESHOLD 1.001
#define SQP_ORTH_SQ_ERR 1e-8
#define SQP_EQUAL_VEC_SQ 1e-10
#define SQP_EQUAL_SQ_ERR 1e-6

struct Sol { double r[9], r_hat[9], t[3], sq_error; };

RDEV double det9(const double* r) {  // sqpnp_helper.h:58-62
  return r[0] * r[4] * r[8] + r[1] * r[5] * r[6] + r[2] * r[3] * r[7] - r[6] * r[4] * r[2] - r[7] * r[5] * r[0] - r[8] * r[3] * r[1];
}
RDEV double orthogonality_error(const double* a) {  // sqpnp_helper.cc:43-56
  const double n1 = a[0] * a[0] + a[1] * a[1] + a[2] * a[2], n2 = a[3] * a[3] + a[4] * a[4] + a[5] * a[5],
               n3 = a[6] * a[6] + a[7] * a[7] + a[8] * a[8];
  const double d12 = a[0] * a[3] + a[1] * a[4] + a[2] * a[5], d13 = a[0] * a[6] + a[1] * a[7] + a[2] * a[8],
               d23 = a[3] * a[6] + a[4] * a[7] + a[5] * a[8];
  return (n1 - 1) * (n1 - 1) + (n2 - 1) * (n2 - 1) + (n3 - 1) * (n3 - 1) + 2 * (d12 * d12 + d13 * d13 + d23 * d23);
}
// sqpnp_helper.cc:204-238 (uses the lower triangle of the row-major Q)
RDEV bool invert_symmetric3(const double* Q, double* Qi) {
  const double a = Q[0], b = Q[3], d = Q[4], c = Q[6], e = Q[7], f = Q[8];
  const double t2 = e * e, t4 = a * d, t7 = b * b, t9 = b * c, t12 = c * c;
  const double det = -t4 * f + a * t2 + t7 * f - 2.0 * t9 * e + t12 * d;
  if (fabs(det) < 1e-8) return false;
  const double t15 = 1.0 / det;
  const double t20 = (-b * f + c * e) * t15, t24 = (b * e - c * d) * t15, t30 = (a * e - t9) * t15;
  Qi[0] = (-d * f + t2) * t15;
  Qi[1] = Qi[3] = -t20;
  Qi[2] = Qi[6] = -t24;
  Qi[4] = -(a * f - t12) * t15;
  Qi[5] = Qi[7] = t30;
  Qi[8] = -(t4 - t7) * t15;
  return true;
}
// sqpnp_helper.cc:191-202: nearest rotation of the row-major 9-vector e
RDEV void nearest_rotation_svd(const double* e, double* r) {
  double E[9], U[9], S[3], V[9];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) E[i * 3 + j] = e[i + 3 * j];  // column-major Map
  svd3(E, U, S, V);
  const double duv = det3(U) * det3(V);
  const double dd[3] = {1.0, 1.0, duv};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double acc = 0.0;
      for (int k = 0; k < 3; ++k) acc += (U[i * 3 + k] * dd[k]) * V[j * 3 + k];
      r[i + 3 * j] = acc;
    }
}
RDEV double norm9(const double* v) {
  double s2 = 0.0;
#pragma unroll
  for (int i = 0; i < 9; ++i) s2 += v[i] * v[i];
  return sqrt(s2);
}
RDEV double dot9(const double* a, const double* b) {
  double s2 = 0.0;
#pragma unroll
  for (int i = 0; i < 9; ++i) s2 += a[i] * b[i];
  return s2;
}

// sqpnp_helper.cc:317-493.  H: 9x6, N: 9x3, K: 6x6 (row-major)
RDEV void row_and_null_space(const double* r, double* H, double* Nn, double* K) {
  const double norm_threshold = 0.1;
#define HH(i, j) H[(i) * 6 + (j)]
#define KK(i, j) K[(i) * 6 + (j)]
#define NN(i, j) Nn[(i) * 3 + (j)]
  for (int i = 0; i < 54; ++i) H[i] = 0.0;
  for (int i = 0; i < 36; ++i) K[i] = 0.0;
  const double norm_r1 = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
  const double inv_norm_r1 = norm_r1 > 1e-5 ? 1.0 / norm_r1 : 0.0;
  HH(0, 0) = r[0] * inv_norm_r1; HH(1, 0) = r[1] * inv_norm_r1; HH(2, 0) = r[2] * inv_norm_r1;
  KK(0, 0) = 2 * norm_r1;
  const double norm_r2 = sqrt(r[3] * r[3] + r[4] * r[4] + r[5] * r[5]);
  const double inv_norm_r2 = 1.0 / norm_r2;
  HH(3, 1) = r[3] * inv_norm_r2; HH(4, 1) = r[4] * inv_norm_r2; HH(5, 1) = r[5] * inv_norm_r2;
  KK(1, 1) = 2 * norm_r2;
  const double norm_r3 = sqrt(r[6] * r[6] + r[7] * r[7] + r[8] * r[8]);
  const double inv_norm_r3 = 1.0 / norm_r3;
  HH(6, 2) = r[6] * inv_norm_r3; HH(7, 2) = r[7] * inv_norm_r3; HH(8, 2) = r[8] * inv_norm_r3;
  KK(2, 2) = 2 * norm_r3;
  // q4
  const double dot_j4q1 = r[3] * HH(0, 0) + r[4] * HH(1, 0) + r[5] * HH(2, 0),
               dot_j4q2 = r[0] * HH(3, 1) + r[1] * HH(4, 1) + r[2] * HH(5, 1);
  HH(0, 3) = r[3] - dot_j4q1 * HH(0, 0); HH(1, 3) = r[4] - dot_j4q1 * HH(1, 0); HH(2, 3) = r[5] - dot_j4q1 * HH(2, 0);
  HH(3, 3) = r[0] - dot_j4q2 * HH(3, 1); HH(4, 3) = r[1] - dot_j4q2 * HH(4, 1); HH(5, 3) = r[2] - dot_j4q2 * HH(5, 1);
  const double inv_norm_j4 = 1.0 / sqrt(HH(0, 3) * HH(0, 3) + HH(1, 3) * HH(1, 3) + HH(2, 3) * HH(2, 3) +
                                        HH(3, 3) * HH(3, 3) + HH(4, 3) * HH(4, 3) + HH(5, 3) * HH(5, 3));
  for (int i = 0; i < 6; ++i) HH(i, 3) *= inv_norm_j4;
  KK(3, 0) = r[3] * HH(0, 0) + r[4] * HH(1, 0) + r[5] * HH(2, 0);
  KK(3, 1) = r[0] * HH(3, 1) + r[1] * HH(4, 1) + r[2] * HH(5, 1);
  KK(3, 3) = r[3] * HH(0, 3) + r[4] * HH(1, 3) + r[5] * HH(2, 3) + r[0] * HH(3, 3) + r[1] * HH(4, 3) + r[2] * HH(5, 3);
  // q5
  const double dot_j5q2 = r[6] * HH(3, 1) + r[7] * HH(4, 1) + r[8] * HH(5, 1),
               dot_j5q3 = r[3] * HH(6, 2) + r[4] * HH(7, 2) + r[5] * HH(8, 2),
               dot_j5q4 = r[6] * HH(3, 3) + r[7] * HH(4, 3) + r[8] * HH(5, 3);
  HH(0, 4) = -dot_j5q4 * HH(0, 3); HH(1, 4) = -dot_j5q4 * HH(1, 3); HH(2, 4) = -dot_j5q4 * HH(2, 3);
  HH(3, 4) = r[6] - dot_j5q2 * HH(3, 1) - dot_j5q4 * HH(3, 3);
  HH(4, 4) = r[7] - dot_j5q2 * HH(4, 1) - dot_j5q4 * HH(4, 3);
  HH(5, 4) = r[8] - dot_j5q2 * HH(5, 1) - dot_j5q4 * HH(5, 3);
  HH(6, 4) = r[3] - dot_j5q3 * HH(6, 2); HH(7, 4) = r[4] - dot_j5q3 * HH(7, 2); HH(8, 4) = r[5] - dot_j5q3 * HH(8, 2);
  {
    double s2 = 0.0;
    for (int i = 0; i < 9; ++i) s2 += HH(i, 4) * HH(i, 4);
    const double nrm = sqrt(s2);
    for (int i = 0; i < 9; ++i) HH(i, 4) /= nrm;
  }
  KK(4, 1) = r[6] * HH(3, 1) + r[7] * HH(4, 1) + r[8] * HH(5, 1);
  KK(4, 2) = r[3] * HH(6, 2) + r[4] * HH(7, 2) + r[5] * HH(8, 2);
  KK(4, 3) = r[6] * HH(3, 3) + r[7] * HH(4, 3) + r[8] * HH(5, 3);
  KK(4, 4) = r[6] * HH(3, 4) + r[7] * HH(4, 4) + r[8] * HH(5, 4) + r[3] * HH(6, 4) + r[4] * HH(7, 4) + r[5] * HH(8, 4);
  // q6
  const double dot_j6q1 = r[6] * HH(0, 0) + r[7] * HH(1, 0) + r[8] * HH(2, 0),
               dot_j6q3 = r[0] * HH(6, 2) + r[1] * HH(7, 2) + r[2] * HH(8, 2),
               dot_j6q4 = r[6] * HH(0, 3) + r[7] * HH(1, 3) + r[8] * HH(2, 3),
               dot_j6q5 = r[0] * HH(6, 4) + r[1] * HH(7, 4) + r[2] * HH(8, 4) + r[6] * HH(0, 4) + r[7] * HH(1, 4) + r[8] * HH(2, 4);
  HH(0, 5) = r[6] - dot_j6q1 * HH(0, 0) - dot_j6q4 * HH(0, 3) - dot_j6q5 * HH(0, 4);
  HH(1, 5) = r[7] - dot_j6q1 * HH(1, 0) - dot_j6q4 * HH(1, 3) - dot_j6q5 * HH(1, 4);
  HH(2, 5) = r[8] - dot_j6q1 * HH(2, 0) - dot_j6q4 * HH(2, 3) - dot_j6q5 * HH(2, 4);
  HH(3, 5) = -dot_j6q5 * HH(3, 4) - dot_j6q4 * HH(3, 3);
  HH(4, 5) = -dot_j6q5 * HH(4, 4) - dot_j6q4 * HH(4, 3);
  HH(5, 5) = -dot_j6q5 * HH(5, 4) - dot_j6q4 * HH(5, 3);
  HH(6, 5) = r[0] - dot_j6q3 * HH(6, 2) - dot_j6q5 * HH(6, 4);
  HH(7, 5) = r[1] - dot_j6q3 * HH(7, 2) - dot_j6q5 * HH(7, 4);
  HH(8, 5) = r[2] - dot_j6q3 * HH(8, 2) - dot_j6q5 * HH(8, 4);
  {
    double s2 = 0.0;
    for (int i = 0; i < 9; ++i) s2 += HH(i, 5) * HH(i, 5);
    const double nrm = sqrt(s2);
    for (int i = 0; i < 9; ++i) HH(i, 5) /= nrm;
  }
  KK(5, 0) = r[6] * HH(0, 0) + r[7] * HH(1, 0) + r[8] * HH(2, 0);
  KK(5, 2) = r[0] * HH(6, 2) + r[1] * HH(7, 2) + r[2] * HH(8, 2);
  KK(5, 3) = r[6] * HH(0, 3) + r[7] * HH(1, 3) + r[8] * HH(2, 3);
  KK(5, 4) = r[6] * HH(0, 4) + r[7] * HH(1, 4) + r[8] * HH(2, 4) + r[0] * HH(6, 4) + r[1] * HH(7, 4) + r[2] * HH(8, 4);
  KK(5, 5) = r[6] * HH(0, 5) + r[7] * HH(1, 5) + r[8] * HH(2, 5) + r[0] * HH(6, 5) + r[1] * HH(7, 5) + r[2] * HH(8, 5);
  // projector onto the null space of H, Pn = I - H H^T (symmetric; column j: Pn(i, j) = [i == j] - sum_k H(i, k) H(j, k)).
  // The 9 x 9 array of the reference is not kept: three of its columns are picked by data-dependent indices, which forces a
  // per-lane array into scratch memory (the candidate loop of k_sqp_c was bound by exactly that traffic).  Columns are
  // recomputed where they are used -- the same expression, so the same bits: column i of a loop over i with compile-time
  // indices, a picked column from the picked row of H (selected by compares, not by address).
#define PN_COLUMN_STATIC(i, out)                                              \
  _Pragma("unroll") for (int r_ = 0; r_ < 9; ++r_) {                           \
    double acc_ = 0.0;                                                         \
    _Pragma("unroll") for (int k_ = 0; k_ < 6; ++k_) acc_ += HH(r_, k_) * HH(i, k_); \
    out[r_] = ((r_ == (i)) ? 1.0 : 0.0) - acc_;                                \
  }
  auto pn_column_picked = [&](int idx, double* out) {
    double hsel[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      double v = 0.0;
#pragma unroll
      for (int rr = 0; rr < 9; ++rr) v = (rr == idx) ? HH(rr, k) : v;
      hsel[k] = v;
    }
#pragma unroll
    for (int rr = 0; rr < 9; ++rr) {
      double acc = 0.0;
#pragma unroll
      for (int k = 0; k < 6; ++k) acc += HH(rr, k) * hsel[k];
      out[rr] = ((rr == idx) ? 1.0 : 0.0) - acc;
    }
  };
  int index1 = -1, index2 = -1, index3 = -1;
  double max_norm1 = DBL_MIN, min_dot12 = DBL_MAX, min_dot1323 = DBL_MAX;
  double col_norms[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    double col[9];
    PN_COLUMN_STATIC(i, col)
    col_norms[i] = norm9(col);
    if (col_norms[i] >= norm_threshold && max_norm1 < col_norms[i]) { max_norm1 = col_norms[i]; index1 = i; }
  }
  if (index1 < 0) index1 = 0;   // the reference indexes with -1 here (undefined); never seen on rank-3 input
  double v1[9], v2[9], v3[9];
  pn_column_picked(index1, v1);
#pragma unroll
  for (int i = 0; i < 9; ++i) NN(i, 0) = v1[i] * (1.0 / max_norm1);
  double c1v[9];   // |Pn_i . v1| / |Pn_i|: the second search reads the same values again
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    c1v[i] = 0.0;
    if (i == index1) continue;
    if (col_norms[i] >= norm_threshold) {
      double col[9];
      PN_COLUMN_STATIC(i, col)
      const double c1 = fabs(dot9(col, v1) / col_norms[i]);
      c1v[i] = c1;
      if (c1 <= min_dot12) { index2 = i; min_dot12 = c1; }
    }
  }
  if (index2 < 0) index2 = (index1 + 1) % 9;
  pn_column_picked(index2, v2);
  {
    double n0[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) n0[i] = NN(i, 0);
    const double d = dot9(v2, n0);
    double t[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) t[i] = v2[i] - d * n0[i];
    const double nrm = norm9(t);
#pragma unroll
    for (int i = 0; i < 9; ++i) NN(i, 1) = t[i] / nrm;
  }
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    if (i == index2 || i == index1) continue;
    if (col_norms[i] >= norm_threshold) {
      double col[9];
      PN_COLUMN_STATIC(i, col)
      const double c1 = c1v[i];
      const double c2 = fabs(dot9(col, v2) / col_norms[i]);
      if (c1 + c2 <= min_dot1323) { index3 = i; min_dot1323 = c2 + c2; }   // sic (sqpnp_helper.cc:480)
    }
  }
  if (index3 < 0) index3 = (index2 + 1) % 9;
  pn_column_picked(index3, v3);
  {
    double n0[9], n1[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) { n0[i] = NN(i, 0); n1[i] = NN(i, 1); }
    const double d1 = dot9(v3, n1), d0 = dot9(v3, n0);
    double t[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) t[i] = v3[i] - (d1 * n1[i]) - (d0 * n0[i]);
    const double nrm = norm9(t);
#pragma unroll
    for (int i = 0; i < 9; ++i) NN(i, 2) = t[i] / nrm;
  }
#undef PN_COLUMN_STATIC
#undef HH
#undef KK
#undef NN
}

// sqpnp_helper.cc:241-310
RDEV void solve_sqp_system(const double* r, const double* Omega, double* delta) {
  const double sq1 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2], sq2 = r[3] * r[3] + r[4] * r[4] + r[5] * r[5],
               sq3 = r[6] * r[6] + r[7] * r[7] + r[8] * r[8];
  const double d12 = r[0] * r[3] + r[1] * r[4] + r[2] * r[5], d13 = r[0] * r[6] + r[1] * r[7] + r[2] * r[8],
               d23 = r[3] * r[6] + r[4] * r[7] + r[5] * r[8];
  double H[54], Nn[27], JH[36];
  row_and_null_space(r, H, Nn, JH);
  const double g[6] = {1 - sq1, 1 - sq2, 1 - sq3, -d12, -d23, -d13};
  double x[6];
#define J(i, j) JH[(i) * 6 + (j)]
  x[0] = g[0] / J(0, 0);
  x[1] = g[1] / J(1, 1);
  x[2] = g[2] / J(2, 2);
  x[3] = (g[3] - J(3, 0) * x[0] - J(3, 1) * x[1]) / J(3, 3);
  x[4] = (g[4] - J(4, 1) * x[1] - J(4, 2) * x[2] - J(4, 3) * x[3]) / J(4, 4);
  x[5] = (g[5] - J(5, 0) * x[0] - J(5, 2) * x[2] - J(5, 3) * x[3] - J(5, 4) * x[4]) / J(5, 5);
#undef J
  for (int i = 0; i < 9; ++i) {
    double acc = 0.0;
    for (int k = 0; k < 6; ++k) acc += H[i * 6 + k] * x[k];
    delta[i] = acc;
  }
  double NtO[27];   // 3 x 9
  for (int a = 0; a < 3; ++a)
    for (int j = 0; j < 9; ++j) {
      double acc = 0.0;
      for (int k = 0; k < 9; ++k) acc += Nn[k * 3 + a] * Omega[k * 9 + j];
      NtO[a * 9 + j] = acc;
    }
  double W[9], Winv[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) {
      double acc = 0.0;
      for (int k = 0; k < 9; ++k) acc += NtO[a * 9 + k] * Nn[k * 3 + b];
      W[a * 3 + b] = acc;
    }
  invert_symmetric3(W, Winv);
  double M[27];   // (-Winv) * NtOmega
  for (int a = 0; a < 3; ++a)
    for (int j = 0; j < 9; ++j) {
      double acc = 0.0;
      for (int k = 0; k < 3; ++k) acc += (-Winv[a * 3 + k]) * NtO[k * 9 + j];
      M[a * 9 + j] = acc;
    }
  double y[3];
  for (int a = 0; a < 3; ++a) {
    double acc = 0.0;
    for (int j = 0; j < 9; ++j) acc += M[a * 9 + j] * (delta[j] + r[j]);
    y[a] = acc;
  }
  for (int i = 0; i < 9; ++i) {
    double acc = 0.0;
    for (int a = 0; a < 3; ++a) acc += Nn[i * 3 + a] * y[a];
    delta[i] += acc;
  }
}

// sqpnp.cc:23-55.  `step++ < DEFAULT_SQP_SQUARED_TOLERANCE` (an int compared with
// 1e-10) lets exactly ONE iteration run: reference behaviour, kept.
RDEV void run_sqp(const double* r0, const double* Omega, Sol* sol) {
  double r[9], delta[9];
  for (int i = 0; i < 9; ++i) r[i] = r0[i];
  solve_sqp_system(r, Omega, delta);
  for (int i = 0; i < 9; ++i) r[i] += delta[i];
  for (int i = 0; i < 9; ++i) sol->r[i] = r[i];
  double det_r = det9(sol->r);
  if (det_r < 0) { for (int i = 0; i < 9; ++i) sol->r[i] = -r[i]; det_r = -det_r; }
  if (det_r > SQP_DET_THRESHOLD) nearest_rotation_svd(sol->r, sol->r_hat);
  else for (int i = 0; i < 9; ++i) sol->r_hat[i] = sol->r[i];
}

// sqpnp_helper.cc:58-101
RDEV void handle_solution(const double* Omega, const double* mean, Sol& s, Sol* sols, double& min_sq_error, int& nsol) {
  const double* r = s.r_hat;
  if (!(r[6] * mean[0] + r[7] * mean[1] + r[8] * mean[2] + s.t[2] > 0)) return;
  double acc = 0.0;
  for (int i = 0; i < 9; ++i) {
    double row = 0.0;
    for (int j = 0; j < 9; ++j) row += Omega[i * 9 + j] * r[j];
    acc += row * r[i];
  }
  s.sq_error = acc;
  if (fabs(min_sq_error - s.sq_error) > SQP_EQUAL_SQ_ERR) {
    if (min_sq_error > s.sq_error) { min_sq_error = s.sq_error; sols[0] = s; nsol = 1; }
  } else {
    bool found = false;
    for (int i = 0; i < nsol; ++i) {
      double d2 = 0.0;
      for (int k = 0; k < 9; ++k) d2 += (sols[i].r_hat[k] - s.r_hat[k]) * (sols[i].r_hat[k] - s.r_hat[k]);
      if (d2 < SQP_EQUAL_VEC_SQ) {
        if (sols[i].sq_error > s.sq_error) sols[i] = s;
        found = true;
        break;
      }
    }
    if (!found && nsol < 18) sols[nsol++] = s;   // the reference's array holds 18 and is not bounds-checked
    if (min_sq_error > s.sq_error) min_sq_error = s.sq_error;
  }
}

RDEV void apply_P(const double* P, const double* rh, double* t) {
  for (int a = 0; a < 3; ++a) {
    double acc = 0.0;
    for (int j = 0; j < 9; ++j) acc += P[a * 9 + j] * rh[j];
    t[a] = acc;
  }
}
}  // namespace sqp

// SQPnP (sqpnp.cc:58-353) in three pieces, so that the RANSAC fit can run the 9 x 9 SVD between them as a team of lanes
// (svd_team.h): sqpnp_pre = Omega, P and the centroid (sqpnp.cc:100-231), sqpnp_post = everything after the SVD
// (:236-353); sqpnp() = pre, svd_sq<9>, post in one thread (single-problem entry point, any number of points).
__device__ __attribute__((noinline)) bool sqpnp_pre(int n, const double* feat, const double* world, double* Om, double* P, double* mean) {
  using namespace sqp;
  if (n < 3) return false;
  double QA[27];
  for (int i = 0; i < 81; ++i) Om[i] = 0.0;
  for (int i = 0; i < 27; ++i) QA[i] = 0.0;
#define OM(i, j) Om[(i) * 9 + (j)]
#define QAA(i, j) QA[(i) * 9 + (j)]
  double sum_wx = 0, sum_wy = 0, sum_wx2y2 = 0, sum_w = 0, sum_X = 0, sum_Y = 0, sum_Z = 0;
  for (int i = 0; i < n; ++i) {
    const double w = 1.0;
    const double wx = feat[2 * i] * w, wy = feat[2 * i + 1] * w;
    const double wsq = w * (feat[2 * i] * feat[2 * i] + feat[2 * i + 1] * feat[2 * i + 1]);
    sum_wx += wx; sum_wy += wy; sum_wx2y2 += wsq; sum_w += w;
    const double X = world[3 * i], Y = world[3 * i + 1], Z = world[3 * i + 2];
    sum_X += X; sum_Y += Y; sum_Z += Z;
    const double X2 = X * X, XY = X * Y, XZ = X * Z, Y2 = Y * Y, YZ = Y * Z, Z2 = Z * Z;
    OM(0, 0) += w * X2; OM(0, 1) += w * XY; OM(0, 2) += w * XZ; OM(1, 1) += w * Y2; OM(1, 2) += w * YZ; OM(2, 2) += w * Z2;
    OM(0, 6) += -wx * X2; OM(0, 7) += -wx * XY; OM(0, 8) += -wx * XZ; OM(1, 7) += -wx * Y2; OM(1, 8) += -wx * YZ; OM(2, 8) += -wx * Z2;
    OM(3, 6) += -wy * X2; OM(3, 7) += -wy * XY; OM(3, 8) += -wy * XZ; OM(4, 7) += -wy * Y2; OM(4, 8) += -wy * YZ; OM(5, 8) += -wy * Z2;
    OM(6, 6) += wsq * X2; OM(6, 7) += wsq * XY; OM(6, 8) += wsq * XZ; OM(7, 7) += wsq * Y2; OM(7, 8) += wsq * YZ; OM(8, 8) += wsq * Z2;
    const double wX = w * X, wY = w * Y, wZ = w * Z;
    QAA(0, 0) += wX; QAA(0, 1) += wY; QAA(0, 2) += wZ; QAA(0, 6) += -wx * X; QAA(0, 7) += -wx * Y; QAA(0, 8) += -wx * Z;
    QAA(1, 3) += wX; QAA(1, 4) += wY; QAA(1, 5) += wZ; QAA(1, 6) += -wy * X; QAA(1, 7) += -wy * Y; QAA(1, 8) += -wy * Z;
    QAA(2, 0) += -wx * X; QAA(2, 1) += -wx * Y; QAA(2, 2) += -wx * Z; QAA(2, 3) += -wy * X; QAA(2, 4) += -wy * Y; QAA(2, 5) += -wy * Z;
    QAA(2, 6) += wsq * X; QAA(2, 7) += wsq * Y; QAA(2, 8) += wsq * Z;
  }
  OM(1, 6) = OM(0, 7); OM(2, 6) = OM(0, 8); OM(2, 7) = OM(1, 8);
  OM(4, 6) = OM(3, 7); OM(5, 6) = OM(3, 8); OM(5, 7) = OM(4, 8);
  OM(7, 6) = OM(6, 7); OM(8, 6) = OM(6, 8); OM(8, 7) = OM(7, 8);
  OM(3, 3) = OM(0, 0); OM(3, 4) = OM(0, 1); OM(3, 5) = OM(0, 2); OM(4, 4) = OM(1, 1); OM(4, 5) = OM(1, 2); OM(5, 5) = OM(2, 2);
  for (int i = 1; i < 9; ++i) for (int j = 0; j < i; ++j) OM(i, j) = OM(j, i);   // lower triangle (sqpnp.cc:178-214)
  const double Q[9] = {sum_w, 0, -sum_wx, 0, sum_w, -sum_wy, -sum_wx, -sum_wy, sum_wx2y2};
  double Qinv[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  invert_symmetric3(Q, Qinv);
  for (int a = 0; a < 3; ++a)
    for (int j = 0; j < 9; ++j) {
      double acc = 0.0;
      for (int k = 0; k < 3; ++k) acc += (-Qinv[a * 3 + k]) * QAA(k, j);
      P[a * 9 + j] = acc;
    }
  for (int i = 0; i < 9; ++i)
    for (int j = 0; j < 9; ++j) {
      double acc = 0.0;
      for (int k = 0; k < 3; ++k) acc += QAA(k, i) * P[k * 9 + j];
      OM(i, j) += acc;
    }
#undef OM
#undef QAA
  const double inv_n = 1.0 / n;
  mean[0] = sum_X * inv_n; mean[1] = sum_Y * inv_n; mean[2] = sum_Z * inv_n;
  return true;
}

// Writes up to 18 solutions: quaternions [w x y z] (of the row-major rotation r_hat) and translations.
// (the body is inlined into k_sqp_c: as a separate function it is compiled for the default 1024-thread workgroup, i.e. with a
// budget of 128 VGPRs, and spent its time on ~2000 spill reloads per call; a 64-thread kernel may use the whole register file)
__device__ __forceinline__ int sqpnp_post_impl(const double* Om, const double* P, const double* mean, const double* U, const double* S,
                                               double* quats, double* ts) {
  using namespace sqp;
  int num_null = 0;
  while (num_null <= 7 && S[7 - num_null] < SQP_RANK_TOL) num_null++;
  if (++num_null > 6) return 0;
  double min_sq_error = DBL_MAX;
  const int nep = num_null > 0 ? num_null : 1;
  Sol sols[18];
  int nsol = 0;
  const double sqrt3 = sqrt(3.0);
  for (int i = 9 - nep; i < 9; ++i) {
    double e[9], me[9];
    for (int k = 0; k < 9; ++k) { e[k] = sqrt3 * U[k * 9 + i]; me[k] = -e[k]; }
    Sol s0, s1;
    if (orthogonality_error(e) < SQP_ORTH_SQ_ERR) {
      const double de = det9(e);
      for (int k = 0; k < 9; ++k) { s0.r_hat[k] = de * e[k]; s0.r[k] = s0.r_hat[k]; }
      apply_P(P, s0.r_hat, s0.t);
      handle_solution(Om, mean, s0, sols, min_sq_error, nsol);
    } else {
      double r0[9];
      nearest_rotation_svd(e, r0);
      run_sqp(r0, Om, &s0);
      apply_P(P, s0.r_hat, s0.t);
      handle_solution(Om, mean, s0, sols, min_sq_error, nsol);
      nearest_rotation_svd(me, r0);
      run_sqp(r0, Om, &s1);
      apply_P(P, s1.r_hat, s1.t);
      handle_solution(Om, mean, s1, sols, min_sq_error, nsol);
    }
  }
  int c = 1;
  while (min_sq_error > 3 * S[9 - nep - c] && 9 - nep - c > 0) {
    const int index = 9 - nep - c;
    double e[9], me[9], r0[9];
    for (int k = 0; k < 9; ++k) { e[k] = U[k * 9 + index]; me[k] = -e[k]; }
    Sol s0, s1;
    nearest_rotation_svd(e, r0);
    run_sqp(r0, Om, &s0);
    apply_P(P, s0.r_hat, s0.t);
    handle_solution(Om, mean, s0, sols, min_sq_error, nsol);
    nearest_rotation_svd(me, r0);
    run_sqp(r0, Om, &s1);
    apply_P(P, s1.r_hat, s1.t);
    handle_solution(Om, mean, s1, sols, min_sq_error, nsol);
    c++;
  }
  for (int i = 0; i < nsol; ++i) {
    rot_to_quat(sols[i].r_hat, quats + 4 * i);
    for (int k = 0; k < 3; ++k) ts[3 * i + k] = sols[i].t[k];
  }
  return nsol;
}
__device__ __attribute__((noinline)) int sqpnp_post(const double* Om, const double* P, const double* mean, const double* U, const double* S,
                                                   double* quats, double* ts) {
  return sqpnp_post_impl(Om, P, mean, U, S, quats, ts);
}

// feat: n x [x y], world: n x [X Y Z].
RDEV int sqpnp(int n, const double* feat, const double* world, double* quats, double* ts) {
  double Om[81], P[27], mean[3];
  if (!sqpnp_pre(n, feat, world, Om, P, mean)) return 0;
  double U[81], S[9], V[81];
#ifdef THIP_SQPNP_SKIP_SVD   // development (timing only, wrong results): how much of the solver is the 9 x 9 SVD
  for (int i = 0; i < 81; ++i) { U[i] = (i % 10 == 0) ? 1.0 : 0.0; V[i] = U[i]; }
  for (int i = 0; i < 9; ++i) S[i] = fabs(Om[i * 10]) * (i < 6 ? 1.0 : 1e-12);
#else
  svd_sq<9>(Om, U, S, V);
#endif
  return sqpnp_post(Om, P, mean, U, S, quats, ts);
}

// ceres/rotation.h RotationMatrixToAngleAxis (via quaternion) and AngleAxisToRotationMatrix, as
// Camera::SetOrientationFromRotationMatrix / GetOrientationAsRotationMatrix use them
// (camera.cc:245-261).  R row-major.
RDEV void rot_to_angle_axis(const double* R, double* aa) {
  double q[4];
  const double trace = R[0] + R[4] + R[8];
  if (trace >= 0.0) {
    double t = sqrt(trace + 1.0);
    q[0] = 0.5 * t;
    t = 0.5 / t;
    q[1] = (R[7] - R[5]) * t; q[2] = (R[2] - R[6]) * t; q[3] = (R[3] - R[1]) * t;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[i * 3 + i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    double t = sqrt(R[i * 3 + i] - R[j * 3 + j] - R[k * 3 + k] + 1.0);
    q[i + 1] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (R[k * 3 + j] - R[j * 3 + k]) * t;
    q[j + 1] = (R[j * 3 + i] + R[i * 3 + j]) * t;
    q[k + 1] = (R[k * 3 + i] + R[i * 3 + k]) * t;
  }
  const double s2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  if (s2 > 0.0) {
    const double st = sqrt(s2), ct = q[0];
    const double two_theta = 2.0 * ((ct < 0.0) ? atan2(-st, -ct) : atan2(st, ct));
    const double k = two_theta / st;
    aa[0] = q[1] * k; aa[1] = q[2] * k; aa[2] = q[3] * k;
  } else {
    aa[0] = q[1] * 2.0; aa[1] = q[2] * 2.0; aa[2] = q[3] * 2.0;
  }
}
RDEV void angle_axis_to_rot(const double* aa, double* R) {
  const double theta2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
  if (theta2 > DBL_EPSILON) {
    const double theta = sqrt(theta2);
    const double wx = aa[0] / theta, wy = aa[1] / theta, wz = aa[2] / theta;
    const double c = cos(theta), s = sin(theta);
    R[0] = c + wx * wx * (1.0 - c);      R[3] = wz * s + wx * wy * (1.0 - c);  R[6] = -wy * s + wx * wz * (1.0 - c);
    R[1] = wx * wy * (1.0 - c) - wz * s; R[4] = c + wy * wy * (1.0 - c);       R[7] = wx * s + wy * wz * (1.0 - c);
    R[2] = wy * s + wx * wz * (1.0 - c); R[5] = -wx * s + wy * wz * (1.0 - c); R[8] = c + wz * wz * (1.0 - c);
  } else {
    R[0] = 1.0; R[3] = aa[2]; R[6] = -aa[1];
    R[1] = -aa[2]; R[4] = 1.0; R[7] = aa[0];
    R[2] = aa[1]; R[5] = -aa[0]; R[8] = 1.0;
  }
}

// Eigen::AngleAxisd(Matrix3d) (Quaternion.h quaternionbase_assign_impl<.., 3, 3> then AngleAxis.h
// operator=(QuaternionBase)) scaled to a rotation vector, and AngleAxisd::toRotationMatrix with
// angle = |v|, axis = v / angle -- RelativePoseEstimator::RefineModel converts this way
// (estimate_relative_pose.cc:116-118,131-134; a zero vector gives NaNs there too).  R row-major.
RDEV void eigen_rot_to_rotvec(const double* R, double* v) {
  double q[4];   // x y z w
  double t = (R[0] + R[4]) + R[8];
  if (t > 0.0) {
    t = sqrt(t + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (R[7] - R[5]) * t; q[1] = (R[2] - R[6]) * t; q[2] = (R[3] - R[1]) * t;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[i * 3 + i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = sqrt(R[i * 3 + i] - R[j * 3 + j] - R[k * 3 + k] + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (R[k * 3 + j] - R[j * 3 + k]) * t;
    q[j] = (R[j * 3 + i] + R[i * 3 + j]) * t;
    q[k] = (R[k * 3 + i] + R[i * 3 + k]) * t;
  }
  double n = sqrt((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]);
  double angle, axis[3];
  if (n != 0.0) {
    angle = 2.0 * atan2(n, fabs(q[3]));
    if (q[3] < 0.0) n = -n;
    axis[0] = q[0] / n; axis[1] = q[1] / n; axis[2] = q[2] / n;
  } else {
    angle = 0.0; axis[0] = 1.0; axis[1] = 0.0; axis[2] = 0.0;
  }
  v[0] = angle * axis[0]; v[1] = angle * axis[1]; v[2] = angle * axis[2];
}
RDEV void eigen_rotvec_to_rot(const double* v, double* R) {
  const double angle = sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
  const double ax[3] = {v[0] / angle, v[1] / angle, v[2] / angle};
  const double s = sin(angle), c = cos(angle);
  const double sa[3] = {s * ax[0], s * ax[1], s * ax[2]};
  const double ca[3] = {(1.0 - c) * ax[0], (1.0 - c) * ax[1], (1.0 - c) * ax[2]};
  double tmp;
  tmp = ca[0] * ax[1]; R[1] = tmp - sa[2]; R[3] = tmp + sa[2];
  tmp = ca[0] * ax[2]; R[2] = tmp + sa[1]; R[6] = tmp - sa[1];
  tmp = ca[1] * ax[2]; R[5] = tmp - sa[0]; R[7] = tmp + sa[0];
  R[0] = ca[0] * ax[0] + c; R[4] = ca[1] * ax[1] + c; R[8] = ca[2] * ax[2] + c;
}

// ------------------------------------------------------------ five point
RDEV void mul_deg1(const double* a, const double* b, double* o) {  // five_point_relative_pose.cc:68-92
  o[0] = a[0] * b[0];
  o[1] = a[0] * b[1] + a[1] * b[0];
  o[2] = a[1] * b[1];
  o[3] = a[0] * b[2] + a[2] * b[0];
  o[4] = a[1] * b[2] + a[2] * b[1];
  o[5] = a[2] * b[2];
  o[6] = a[0] * b[3] + a[3] * b[0];
  o[7] = a[1] * b[3] + a[3] * b[1];
  o[8] = a[2] * b[3] + a[3] * b[2];
  o[9] = a[3] * b[3];
}
RDEV void mul_deg2_deg1(const double* a, const double* b, double* o) {  // :96-140
  o[0] = a[0] * b[0];
  o[1] = a[0] * b[1] + a[1] * b[0];
  o[2] = a[1] * b[1] + a[2] * b[0];
  o[3] = a[2] * b[1];
  o[4] = a[0] * b[2] + a[3] * b[0];
  o[5] = a[1] * b[2] + a[3] * b[1] + a[4] * b[0];
  o[6] = a[2] * b[2] + a[4] * b[1];
  o[7] = a[3] * b[2] + a[5] * b[0];
  o[8] = a[4] * b[2] + a[5] * b[1];
  o[9] = a[5] * b[2];
  o[10] = a[0] * b[3] + a[6] * b[0];
  o[11] = a[1] * b[3] + a[6] * b[1] + a[7] * b[0];
  o[12] = a[2] * b[3] + a[7] * b[1];
  o[13] = a[3] * b[3] + a[6] * b[2] + a[8] * b[0];
  o[14] = a[4] * b[3] + a[7] * b[2] + a[8] * b[1];
  o[15] = a[5] * b[3] + a[8] * b[2];
  o[16] = a[6] * b[3] + a[9] * b[0];
  o[17] = a[7] * b[3] + a[9] * b[1];
  o[18] = a[8] * b[3] + a[9] * b[2];
  o[19] = a[9] * b[3];
}

// corr: 5 x [x1 y1 x2 y2]; E: up to 10 row-major 3x3.  Returns #solutions.
// Steps 1-3 of the five-point solver: null space N (9 x 4) and the 10 x 10 action matrix M (false: rank deficient)
__device__ __attribute__((noinline)) bool five_point_pre(const double* corr, double* N, double* M) {
  // Step 1: 5x9 epipolar constraint rows (:228-236)
  double A[45];
  for (int i = 0; i < 5; ++i) {
    const double x1 = corr[4 * i], y1 = corr[4 * i + 1], x2 = corr[4 * i + 2], y2 = corr[4 * i + 3];
    double* r = A + 9 * i;
    r[0] = x2 * x1; r[1] = y2 * x1; r[2] = x1; r[3] = x2 * y1; r[4] = y2 * y1; r[5] = y1; r[6] = x2; r[7] = y2; r[8] = 1.0;
  }
  // null space via full-pivot LU (Eigen FullPivLU::kernel, :242-247)
  FullPivLU f;
  fullpiv_lu(A, 5, 9, f);
  const double premult = fabs(f.maxpivot) * (DBL_EPSILON * 5.0);
  int rank = 0;
  for (int i = 0; i < f.nonzero_pivots; ++i) rank += fabs(A[i * 9 + i]) > premult;
  if (9 - rank != 4) return false;
  int cidx[9];
  for (int j = 0; j < 9; ++j) cidx[j] = j;
  for (int k = 0; k < f.size; ++k) dswap(cidx[k], cidx[f.colt[k]]);
  // solve U1 X = U2 (5x5 upper, 4 right-hand sides)
  double X[20];
  for (int c = 0; c < 4; ++c)
    for (int i = 4; i >= 0; --i) {
      double s = A[i * 9 + 5 + c];
      for (int j = i + 1; j < 5; ++j) s -= A[i * 9 + j] * X[j * 4 + c];
      X[i * 4 + c] = s / A[i * 9 + i];
    }
  for (int i = 0; i < 36; ++i) N[i] = 0.0;   // null_space 9 x 4
  for (int i = 0; i < 5; ++i) for (int c = 0; c < 4; ++c) N[cidx[i] * 4 + c] = -X[i * 4 + c];
  for (int c = 0; c < 4; ++c) N[cidx[5 + c] * 4 + c] = 1.0;
  // null_space_matrix[i][j] = row (3*j + i) of N  (:254-257)
  const double* ns[3][3];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) ns[i][j] = N + 4 * (3 * j + i);
  // Step 2: constraint matrix 10 x 20 (:142-206)
  double C[200];
  {
    double eet[3][3][10];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
      double t0[10], t1[10], t2[10];
      mul_deg1(ns[i][0], ns[j][0], t0); mul_deg1(ns[i][1], ns[j][1], t1); mul_deg1(ns[i][2], ns[j][2], t2);
      for (int k = 0; k < 10; ++k) eet[i][j][k] = 2 * ((t0[k] + t1[k]) + t2[k]);
    }
    double trace[10];
    for (int k = 0; k < 10; ++k) trace[k] = (eet[0][0][k] + eet[1][1][k]) + eet[2][2][k];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
      double a[20], b[20], c[20], d[20];
      mul_deg2_deg1(eet[i][0], ns[0][j], a); mul_deg2_deg1(eet[i][1], ns[1][j], b);
      mul_deg2_deg1(eet[i][2], ns[2][j], c); mul_deg2_deg1(trace, ns[i][j], d);
      double* row = C + 20 * (3 * i + j);
      for (int k = 0; k < 20; ++k) row[k] = ((a[k] + b[k]) + c[k]) - 0.5 * d[k];
    }
    double p0[10], p1[10], q[10], d0[20], d1[20], d2[20];
    mul_deg1(ns[0][1], ns[1][2], p0); mul_deg1(ns[0][2], ns[1][1], p1);
    for (int k = 0; k < 10; ++k) q[k] = p0[k] - p1[k];
    mul_deg2_deg1(q, ns[2][0], d0);
    mul_deg1(ns[0][2], ns[1][0], p0); mul_deg1(ns[0][0], ns[1][2], p1);
    for (int k = 0; k < 10; ++k) q[k] = p0[k] - p1[k];
    mul_deg2_deg1(q, ns[2][1], d1);
    mul_deg1(ns[0][0], ns[1][1], p0); mul_deg1(ns[0][1], ns[1][0], p1);
    for (int k = 0; k < 10; ++k) q[k] = p0[k] - p1[k];
    mul_deg2_deg1(q, ns[2][2], d2);
    for (int k = 0; k < 20; ++k) C[180 + k] = (d0[k] + d1[k]) + d2[k];
  }
  // Step 3: eliminated = lu(C[:, :10]).solve(C[:, 10:])  (:260-263)
  double L[100], B[100];
  for (int i = 0; i < 10; ++i) for (int j = 0; j < 10; ++j) { L[i * 10 + j] = C[i * 20 + j]; B[i * 10 + j] = C[i * 20 + 10 + j]; }
  FullPivLU g;
  fullpiv_lu(L, 10, 10, g);
  for (int k = 0; k < 10; ++k) if (g.rowt[k] != k) for (int j = 0; j < 10; ++j) dswap(B[k * 10 + j], B[g.rowt[k] * 10 + j]);
  for (int c = 0; c < 10; ++c) {
    for (int i = 0; i < 10; ++i) { double s = B[i * 10 + c]; for (int j = 0; j < i; ++j) s -= L[i * 10 + j] * B[j * 10 + c]; B[i * 10 + c] = s; }
    for (int i = 9; i >= 0; --i) { double s = B[i * 10 + c]; for (int j = i + 1; j < 10; ++j) s -= L[i * 10 + j] * B[j * 10 + c]; B[i * 10 + c] = s / L[i * 10 + i]; }
  }
  int gidx[10];
  for (int j = 0; j < 10; ++j) gidx[j] = j;
  for (int k = 0; k < 10; ++k) dswap(gidx[k], gidx[g.colt[k]]);
  double El[100];  // eliminated matrix: row gidx[i] = B row i
  for (int i = 0; i < 10; ++i) for (int j = 0; j < 10; ++j) El[gidx[i] * 10 + j] = B[i * 10 + j];
  // action matrix (:265-273)
  for (int i = 0; i < 100; ++i) M[i] = 0.0;
  const int src[6] = {0, 1, 2, 4, 5, 7};
  for (int r = 0; r < 6; ++r) for (int j = 0; j < 10; ++j) M[r * 10 + j] = El[src[r] * 10 + j];
  M[6 * 10 + 0] = -1.0; M[7 * 10 + 1] = -1.0; M[8 * 10 + 3] = -1.0; M[9 * 10 + 6] = -1.0;
  return true;
}
// the last four entries of the NORMALISED eigenvector in column i of Vv (10 x 10 row-major): Eigen normalises
RDEV void five_point_v4(const double* Vv, int i, double* v4) {
  double nrm = 0.0;
  for (int k = 0; k < 10; ++k) nrm += Vv[k * 10 + i] * Vv[k * 10 + i];
  nrm = sqrt(nrm);
  for (int k = 0; k < 4; ++k) v4[k] = Vv[(6 + k) * 10 + i] / nrm;
}
// Map<Matrix<9,1>>(ematrix.data()) = null_space * v  (column-major 3x3) -> E row-major
RDEV void five_point_E(const double* N, const double* v4, double* Eo) {
  double e9[9];
  for (int rI = 0; rI < 9; ++rI) e9[rI] = ((N[rI * 4] * v4[0] + N[rI * 4 + 1] * v4[1]) + N[rI * 4 + 2] * v4[2]) + N[rI * 4 + 3] * v4[3];
  for (int c = 0; c < 3; ++c) for (int rr = 0; rr < 3; ++rr) Eo[rr * 3 + c] = e9[c * 3 + rr];
}
RDEV int five_point(const double* corr, double* E) {
  double N[36], M[100];
  if (!five_point_pre(corr, N, M)) return 0;
  double wr[10], wi[10], Vv[100];
  if (!eig_real_general(10, M, wr, wi, Vv)) return 0;
  int ns_out = 0;
  for (int i = 0; i < 10; ++i) {
    if (wi[i] != 0) continue;  // only real solutions (:281-284)
    double v4[4];
    five_point_v4(Vv, i, v4);
    five_point_E(N, v4, E + 9 * ns_out);
    ns_out++;
  }
  return ns_out;
}

// triangulation.cc:216-232
RDEV bool in_front(const double* c, const double* R, const double* pos) {
  const double d1[3] = {c[0], c[1], 1.0};
  const double h2[3] = {c[2], c[3], 1.0};
  const double d2[3] = {(R[0] * h2[0] + R[3] * h2[1]) + R[6] * h2[2], (R[1] * h2[0] + R[4] * h2[1]) + R[7] * h2[2],
                        (R[2] * h2[0] + R[5] * h2[1]) + R[8] * h2[2]};
  const double d1sq = (d1[0] * d1[0] + d1[1] * d1[1]) + d1[2] * d1[2];
  const double d2sq = (d2[0] * d2[0] + d2[1] * d2[1]) + d2[2] * d2[2];
  const double d1d2 = (d1[0] * d2[0] + d1[1] * d2[1]) + d1[2] * d2[2];
  const double d1p = (d1[0] * pos[0] + d1[1] * pos[1]) + d1[2] * pos[2];
  const double d2p = (d2[0] * pos[0] + d2[1] * pos[1]) + d2[2] * pos[2];
  return (d2sq * d1p - d1d2 * d2p > 0) && (d1d2 * d1p - d1sq * d2p > 0);
}

// pose/util.cc:56-68, y^T F x with x = feature1, y = feature2
RDEV double sampson(const double* F, const double* c) {
  const double x0 = c[0], x1 = c[1], y0 = c[2], y1 = c[3];
  const double ex0 = (F[0] * x0 + F[1] * x1) + F[2];
  const double ex1 = (F[3] * x0 + F[4] * x1) + F[5];
  const double ex2 = (F[6] * x0 + F[7] * x1) + F[8];
  const double num = (y0 * ex0 + y1 * ex1) + ex2;
  const double dn0 = (y0 * F[0] + y1 * F[3]) + F[6];
  const double dn1 = (y0 * F[1] + y1 * F[4]) + F[7];
  const double den = ((dn0 * dn0 + dn1 * dn1) + ex0 * ex0) + ex1 * ex1;
  return num * num / den;
}

// The same error as numerator and denominator (r = a / den): the scoring kernel decides r < thresh without the FP64 division
// whenever a is clear of thresh * den by more than the roundings involved, and divides only at the boundary or when the value
// itself is needed (MLE score of an inlier) -- the decisions and sums are those of the divided form, bit for bit.
RDEV void sampson_parts(const double* F, const double* c, double* a, double* den_out) {
  const double x0 = c[0], x1 = c[1], y0 = c[2], y1 = c[3];
  const double ex0 = (F[0] * x0 + F[1] * x1) + F[2];
  const double ex1 = (F[3] * x0 + F[4] * x1) + F[5];
  const double ex2 = (F[6] * x0 + F[7] * x1) + F[8];
  const double num = (y0 * ex0 + y1 * ex1) + ex2;
  const double dn0 = (y0 * F[0] + y1 * F[3]) + F[6];
  const double dn1 = (y0 * F[1] + y1 * F[4]) + F[7];
  *den_out = ((dn0 * dn0 + dn1 * dn1) + ex0 * ex0) + ex1 * ex1;
  *a = num * num;
}
// fl(a / den) < thresh ?  *r receives the quotient when it had to be (or was asked to be) computed, else stays untouched.
// |fl(t den) - t den| <= eps t den and |fl(a / den) - a / den| <= eps a / den (eps = 2^-53): a margin of 2^-50 covers both.
RDEV bool quotient_below(double a, double den, double thresh, bool want_value, double* r) {
  const double td = thresh * den;
  const double lo = td * (1.0 - 0x1p-50), hi = td * (1.0 + 0x1p-50);
  if (!want_value && a > hi) return false;
  if (!want_value && a < lo) return true;
  if (a > hi) return false;          // an outlier's value is never used
  *r = a / den;
  return *r < thresh;
}

// ---------------------------------------------------------------------------
// Uncalibrated two-view models, plane and known-orientation position
// ---------------------------------------------------------------------------
// row-major 3x3 product, each entry accumulated left to right
RDEV void matmul3(const double* A, const double* B, double* C) {
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) C[3 * r + c] = (A[3 * r] * B[c] + A[3 * r + 1] * B[3 + c]) + A[3 * r + 2] * B[6 + c];
}

// NormalizeImagePoints (sfm/pose/util.cc:81-112): centroid to the origin, RMS
// distance sqrt(2).  pts: n points, `stride` doubles apart; T row-major 3x3.
RDEV void normalize_image_points(const double* pts, int stride, int n, double* out, double* T) {
  double cx = 0.0, cy = 0.0;
  for (int i = 0; i < n; ++i) { cx += pts[i * stride]; cy += pts[i * stride + 1]; }
  cx /= n; cy /= n;
  double ss = 0.0;
  for (int i = 0; i < n; ++i) {
    const double dx = pts[i * stride] - cx, dy = pts[i * stride + 1] - cy;
    ss += dx * dx;
    ss += dy * dy;
  }
  const double rms_mean_dist = sqrt(ss / n);
  const double norm_factor = sqrt(2.0) / rms_mean_dist;
  T[0] = norm_factor; T[1] = 0.0; T[2] = -1.0 * norm_factor * cx;
  T[3] = 0.0; T[4] = norm_factor; T[5] = -1.0 * norm_factor * cy;
  T[6] = 0.0; T[7] = 0.0; T[8] = 1.0;
  for (int i = 0; i < n; ++i) {
    const double x = pts[i * stride], y = pts[i * stride + 1];
    const double hx = (T[0] * x + T[1] * y) + T[2];
    const double hy = (T[3] * x + T[4] * y) + T[5];
    const double hz = (T[6] * x + T[7] * y) + T[8];
    out[2 * i] = hx / hz; out[2 * i + 1] = hy / hz;
  }
}

// NormalizedEightPointFundamentalMatrix for exactly 8 correspondences
// (sfm/pose/eight_point_fundamental_matrix.cc:50-112: the minimal case takes
// the kernel of the 8x9 constraint matrix from a full-pivot LU).
// corr: 8 x [x1 y1 x2 y2]; F row-major with x2^T F x1 = 0.
RDEV bool eight_point_fundamental(const double* corr, double* F) {
  double n1[16], n2[16], T1[9], T2[9];
  normalize_image_points(corr, 4, 8, n1, T1);
  normalize_image_points(corr + 2, 4, 8, n2, T2);
  double A[72];
  for (int i = 0; i < 8; ++i) {
    const double x1 = n1[2 * i], y1 = n1[2 * i + 1], x2 = n2[2 * i], y2 = n2[2 * i + 1];
    double* r = A + 9 * i;
    r[0] = x1 * x2; r[1] = y1 * x2; r[2] = x2;
    r[3] = x1 * y2; r[4] = y1 * y2; r[5] = y2;
    r[6] = x1; r[7] = y1; r[8] = 1.0;
  }
  FullPivLU f;
  fullpiv_lu(A, 8, 9, f);
  const double premult = fabs(f.maxpivot) * (DBL_EPSILON * 8.0);
  int rank = 0;
  for (int i = 0; i < f.nonzero_pivots; ++i) rank += fabs(A[i * 9 + i]) > premult;
  if (9 - rank != 1) return false;   // dimensionOfKernel() != 1
  int cidx[9];
  for (int j = 0; j < 9; ++j) cidx[j] = j;
  for (int k = 0; k < f.size; ++k) dswap(cidx[k], cidx[f.colt[k]]);
  double X[8];
  for (int i = 7; i >= 0; --i) {
    double s = A[i * 9 + 8];
    for (int j = i + 1; j < 8; ++j) s -= A[i * 9 + j] * X[j];
    X[i] = s / A[i * 9 + i];
  }
  double v[9];
  for (int i = 0; i < 9; ++i) v[i] = 0.0;
  for (int i = 0; i < 8; ++i) v[cidx[i]] = -X[i];
  v[cidx[8]] = 1.0;
  // rank-2 projection of the (lazily transposed, :96-105) matrix: row-major reading of v
  double U[9], S[3], V[9];
  svd3(v, U, S, V);
  S[2] = 0.0;
  double US[9], Fr[9];
  for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) US[3 * r + k] = U[3 * r + k] * S[k];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) Fr[3 * r + c] = (US[3 * r] * V[3 * c] + US[3 * r + 1] * V[3 * c + 1]) + US[3 * r + 2] * V[3 * c + 2];
  // undo the normalisation: T2^T F T1 (:107-109)
  double T2t[9], tmp[9];
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) T2t[3 * r + c] = T2[3 * c + r];
  matmul3(T2t, Fr, tmp);
  matmul3(tmp, T1, F);
  return true;
}

// Eigen's 3x3 inverse (Eigen/src/LU/InverseImpl.h compute_inverse<Matrix3d>:
// cofactors of column 0 give the determinant, inverse = cofactor^T / det)
RDEV double cofactor3(const double* m, int i, int j) {
  const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
  return m[3 * i1 + j1] * m[3 * i2 + j2] - m[3 * i1 + j2] * m[3 * i2 + j1];
}
RDEV void inverse3(const double* m, double* inv) {
  const double c00 = cofactor3(m, 0, 0), c10 = cofactor3(m, 1, 0), c20 = cofactor3(m, 2, 0);
  const double det = (c00 * m[0] + c10 * m[3]) + c20 * m[6];
  const double invdet = 1.0 / det;
  inv[0] = c00 * invdet; inv[1] = c10 * invdet; inv[2] = c20 * invdet;
  inv[3] = cofactor3(m, 0, 1) * invdet; inv[4] = cofactor3(m, 1, 1) * invdet; inv[5] = cofactor3(m, 2, 1) * invdet;
  inv[6] = cofactor3(m, 0, 2) * invdet; inv[7] = cofactor3(m, 1, 2) * invdet; inv[8] = cofactor3(m, 2, 2) * invdet;
}

// FourPointHomography for exactly 4 correspondences (sfm/pose/four_point_homography.cc:54-105):
// normalised DLT, null vector = last right singular vector of A^T A.
// corr: 4 x [x1 y1 x2 y2]; H row-major with x2 ~ H x1.
// four_point_homography in two pieces around the 9 x 9 SVD (the RANSAC fit runs it by teams of lanes, svd_team.h):
// M = A^T A of the normalised DLT system and the two normalisations, then H = T2^-1 Hn T1 from the last right singular vector
RDEV void four_point_homography_pre(const double* corr, double* M, double* T1, double* T2) {
  double n1[8], n2[8];
  normalize_image_points(corr, 4, 4, n1, T1);
  normalize_image_points(corr + 2, 4, 4, n2, T2);
  double A[72];
  for (int i = 0; i < 4; ++i) {
    const double x1 = n1[2 * i], y1 = n1[2 * i + 1], x2 = n2[2 * i], y2 = n2[2 * i + 1];
    double* r = A + 18 * i;
    r[0] = 0.0; r[1] = 0.0; r[2] = 0.0; r[3] = -x1; r[4] = -y1; r[5] = -1.0; r[6] = x1 * y2; r[7] = y1 * y2; r[8] = y2;
    r[9] = x1; r[10] = y1; r[11] = 1.0; r[12] = 0.0; r[13] = 0.0; r[14] = 0.0; r[15] = -x1 * x2; r[16] = -y1 * x2; r[17] = -x2;
  }
  for (int a = 0; a < 9; ++a)
    for (int b = 0; b < 9; ++b) {
      double s = 0.0;
      for (int k = 0; k < 8; ++k) s += A[9 * k + a] * A[9 * k + b];
      M[9 * a + b] = s;
    }
}
RDEV void four_point_homography_post(const double* Hn, const double* T1, const double* T2, double* H) {
  double T2i[9], tmp[9];
  inverse3(T2, T2i);
  matmul3(T2i, Hn, tmp);
  matmul3(tmp, T1, H);
}

RDEV bool four_point_homography(const double* corr, double* H) {
  double n1[8], n2[8], T1[9], T2[9];
  normalize_image_points(corr, 4, 4, n1, T1);
  normalize_image_points(corr + 2, 4, 4, n2, T2);
  double A[72];
  for (int i = 0; i < 4; ++i) {
    const double x1 = n1[2 * i], y1 = n1[2 * i + 1], x2 = n2[2 * i], y2 = n2[2 * i + 1];
    double* r = A + 18 * i;
    r[0] = 0.0; r[1] = 0.0; r[2] = 0.0; r[3] = -x1; r[4] = -y1; r[5] = -1.0; r[6] = x1 * y2; r[7] = y1 * y2; r[8] = y2;
    r[9] = x1; r[10] = y1; r[11] = 1.0; r[12] = 0.0; r[13] = 0.0; r[14] = 0.0; r[15] = -x1 * x2; r[16] = -y1 * x2; r[17] = -x2;
  }
  double M[81];
  for (int a = 0; a < 9; ++a)
    for (int b = 0; b < 9; ++b) {
      double s = 0.0;
      for (int k = 0; k < 8; ++k) s += A[9 * k + a] * A[9 * k + b];
      M[9 * a + b] = s;
    }
  double U[81], S[9], V[81];
  svd_sq<9>(M, U, S, V);
  double Hn[9];
  for (int k = 0; k < 9; ++k) Hn[k] = V[9 * k + 8];
  double T2i[9], tmp[9];
  inverse3(T2, T2i);
  matmul3(T2i, Hn, tmp);
  matmul3(tmp, T1, H);
  return true;
}

// asymmetric transfer error of HomographyEstimator::Error (estimate_homography.cc:106-112)
RDEV double homography_error(const double* H, const double* c) {
  const double x = c[0], y = c[1];
  const double px = (H[0] * x + H[1] * y) + H[2];
  const double py = (H[3] * x + H[4] * y) + H[5];
  const double pz = (H[6] * x + H[7] * y) + H[8];
  const double ex = c[2] - px / pz, ey = c[3] - py / pz;
  return ex * ex + ey * ey;
}

// DominantPlaneEstimator::EstimateModel (estimate_dominant_plane_from_points.cc:62-81).
// pts: 3 x [X Y Z]; model = point(3) unit_normal(3).
RDEV bool plane_from_three_points(const double* pts, double* model) {
  const double a[3] = {pts[3] - pts[0], pts[4] - pts[1], pts[5] - pts[2]};
  const double b[3] = {pts[6] - pts[0], pts[7] - pts[1], pts[8] - pts[2]};
  const double cr[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
  const double n2 = (cr[0] * cr[0] + cr[1] * cr[1]) + cr[2] * cr[2];
  if (n2 < 1e-6) return false;
  const double nrm = sqrt(n2);
  for (int k = 0; k < 3; ++k) { model[k] = pts[k]; model[3 + k] = cr[k] / nrm; }
  return true;
}
// point-to-plane distance (:84-86; NOT squared)
RDEV double plane_error(const double* model, const double* p) {
  return fabs((model[3] * (p[0] - model[0]) + model[4] * (p[1] - model[1])) + model[5] * (p[2] - model[2]));
}

// RelativePoseFromTwoPointsWithKnownRotation
// (sfm/pose/relative_pose_from_two_points_with_known_rotation.cc:51-88):
// kernel of the 2x3 epipolar constraint, normalised.  corr: 2 x [x1 y1 x2 y2].
RDEV bool two_point_relative_position(const double* corr, double* pos) {
  double A[6];
  for (int i = 0; i < 2; ++i) {
    const double x1 = corr[4 * i], y1 = corr[4 * i + 1], x2 = corr[4 * i + 2], y2 = corr[4 * i + 3];
    A[3 * i] = -y1 + y2;
    A[3 * i + 1] = -x2 + x1;
    A[3 * i + 2] = y1 * x2 - x1 * y2;
  }
  FullPivLU f;
  fullpiv_lu(A, 2, 3, f);
  const double premult = fabs(f.maxpivot) * (DBL_EPSILON * 2.0);
  int rank = 0;
  for (int i = 0; i < f.nonzero_pivots; ++i) rank += fabs(A[i * 3 + i]) > premult;
  if (3 - rank != 1) return false;
  int cidx[3] = {0, 1, 2};
  for (int k = 0; k < f.size; ++k) dswap(cidx[k], cidx[f.colt[k]]);
  double X[2];
  X[1] = A[5] / A[4];
  X[0] = (A[2] - A[1] * X[1]) / A[0];
  double v[3] = {0.0, 0.0, 0.0};
  v[cidx[0]] = -X[0]; v[cidx[1]] = -X[1]; v[cidx[2]] = 1.0;
  const double nrm = sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
  for (int k = 0; k < 3; ++k) pos[k] = v[k] / nrm;
  return true;
}
// Sampson distance under E = [-position]_x (estimate_relative_pose_with_known_orientation.cc:52-58)
RDEV double known_orientation_error(const double* pos, const double* c) {
  const double E[9] = {0.0, pos[2], -pos[1], -pos[2], 0.0, pos[0], pos[1], -pos[0], 0.0};
  return sampson(E, c);
}

// essential_matrix_utils.cc:57-80,109-149.  Returns #points in front.
RDEV int best_pose_from_E(const double* E, const double* corr, int ncorr, double* Rout, double* posout) {
  double U[9], S[3], V[9];
  svd3(E, U, S, V);
  if (det3(U) < 0) for (int k = 0; k < 3; ++k) U[k * 3 + 2] = -U[k * 3 + 2];
  if (det3(V) < 0) for (int k = 0; k < 3; ++k) V[k * 3 + 2] = -V[k * 3 + 2];
  // R1 = U d V^T, R2 = U d^T V^T, d = [0 1 0; -1 0 0; 0 0 1]
  double Ud[9], Udt[9];
  for (int i = 0; i < 3; ++i) {
    Ud[i * 3 + 0] = -U[i * 3 + 1]; Ud[i * 3 + 1] = U[i * 3 + 0]; Ud[i * 3 + 2] = U[i * 3 + 2];
    Udt[i * 3 + 0] = U[i * 3 + 1]; Udt[i * 3 + 1] = -U[i * 3 + 0]; Udt[i * 3 + 2] = U[i * 3 + 2];
  }
  double R[2][9];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
    R[0][i * 3 + j] = (Ud[i * 3] * V[j * 3] + Ud[i * 3 + 1] * V[j * 3 + 1]) + Ud[i * 3 + 2] * V[j * 3 + 2];
    R[1][i * 3 + j] = (Udt[i * 3] * V[j * 3] + Udt[i * 3 + 1] * V[j * 3 + 1]) + Udt[i * 3 + 2] * V[j * 3 + 2];
  }
  double t[3] = {U[2], U[5], U[8]};
  const double tn = sqrt((t[0] * t[0] + t[1] * t[1]) + t[2] * t[2]);
  t[0] /= tn; t[1] /= tn; t[2] /= tn;
  int best = -1, bestcount = -1;
  double bestpos[3] = {0, 0, 0};
  for (int k = 0; k < 4; ++k) {
    const double* Rk = R[k >> 1];
    const double sg = (k & 1) ? -1.0 : 1.0;
    const double tk[3] = {sg * t[0], sg * t[1], sg * t[2]};
    // position = -R^T * translation
    const double pos[3] = {-((Rk[0] * tk[0] + Rk[3] * tk[1]) + Rk[6] * tk[2]), -((Rk[1] * tk[0] + Rk[4] * tk[1]) + Rk[7] * tk[2]),
                           -((Rk[2] * tk[0] + Rk[5] * tk[1]) + Rk[8] * tk[2])};
    int cnt = 0;
    for (int i = 0; i < ncorr; ++i) cnt += in_front(corr + 4 * i, Rk, pos) ? 1 : 0;
    if (cnt > bestcount) { bestcount = cnt; best = k; bestpos[0] = pos[0]; bestpos[1] = pos[1]; bestpos[2] = pos[2]; }
  }
  for (int i = 0; i < 9; ++i) Rout[i] = R[best >> 1][i];
  posout[0] = bestpos[0]; posout[1] = bestpos[1]; posout[2] = bestpos[2];
  return bestcount;
}

// Eigen::ColPivHouseholderQR of a 4 x 3 system, rank() and solve() (Eigen/src/QR/ColPivHouseholderQR.h:
// computeInPlace, rank with the default threshold eps * diagonalSize, _solve_impl), restated for
// PositionFromTwoRays.  A row-major 4 x 3 (destroyed), b (destroyed).  Returns the rank.
RDEV int colpiv_qr_solve_4x3(double* A, double* b, double* x) {
  const int rows = 4, cols = 3, size = 3;
  double hcoef[3], norm_upd[3], norm_dir[3];
  int transp[3];
  for (int k = 0; k < cols; ++k) {
    double s2 = 0.0;
    for (int r = 0; r < rows; ++r) s2 += A[r * cols + k] * A[r * cols + k];
    norm_upd[k] = norm_dir[k] = sqrt(s2);
  }
  double maxn = fmax(norm_upd[0], fmax(norm_upd[1], norm_upd[2]));
  const double threshold_helper = (maxn * DBL_EPSILON) * (maxn * DBL_EPSILON) / (double)rows;
  const double norm_downdate_threshold = sqrt(DBL_EPSILON);
  int nonzero_pivots = size;
  double maxpivot = 0.0;
  for (int k = 0; k < size; ++k) {
    int big = k;
    for (int j = k + 1; j < cols; ++j) if (norm_upd[j] > norm_upd[big]) big = j;
    const double big_sq = norm_upd[big] * norm_upd[big];
    if (nonzero_pivots == size && big_sq < threshold_helper * (double)(rows - k)) nonzero_pivots = k;
    transp[k] = big;
    if (k != big) {
      for (int r = 0; r < rows; ++r) dswap(A[r * cols + k], A[r * cols + big]);
      dswap(norm_upd[k], norm_upd[big]);
      dswap(norm_dir[k], norm_dir[big]);
    }
    // makeHouseholderInPlace on A(k:, k)
    const double c0 = A[k * cols + k];
    double tail = 0.0;
    for (int r = k + 1; r < rows; ++r) tail += A[r * cols + k] * A[r * cols + k];
    double tau, beta;
    if (tail <= DBL_MIN) {
      tau = 0.0; beta = c0;
      for (int r = k + 1; r < rows; ++r) A[r * cols + k] = 0.0;
    } else {
      beta = sqrt(c0 * c0 + tail);
      if (c0 >= 0.0) beta = -beta;
      for (int r = k + 1; r < rows; ++r) A[r * cols + k] /= (c0 - beta);
      tau = (beta - c0) / beta;
    }
    hcoef[k] = tau;
    A[k * cols + k] = beta;
    if (fabs(beta) > maxpivot) maxpivot = fabs(beta);
    // apply H_k = I - tau v v^T (v = [1; essential]) to the remaining columns
    for (int j = k + 1; j < cols; ++j) {
      double tmp = A[k * cols + j];
      for (int r = k + 1; r < rows; ++r) tmp += A[r * cols + k] * A[r * cols + j];
      A[k * cols + j] -= tau * tmp;
      for (int r = k + 1; r < rows; ++r) A[r * cols + j] -= tau * A[r * cols + k] * tmp;
    }
    for (int j = k + 1; j < cols; ++j) {
      if (norm_upd[j] != 0.0) {
        double temp = fabs(A[k * cols + j]) / norm_upd[j];
        temp = (1.0 + temp) * (1.0 - temp);
        temp = temp < 0.0 ? 0.0 : temp;
        const double q = norm_upd[j] / norm_dir[j];
        const double temp2 = temp * (q * q);
        if (temp2 <= norm_downdate_threshold) {
          double s2 = 0.0;
          for (int r = k + 1; r < rows; ++r) s2 += A[r * cols + j] * A[r * cols + j];
          norm_dir[j] = sqrt(s2);
          norm_upd[j] = norm_dir[j];
        } else {
          norm_upd[j] *= sqrt(temp);
        }
      }
    }
  }
  const double premult = fabs(maxpivot) * (DBL_EPSILON * (double)size);
  int rank = 0;
  for (int i = 0; i < nonzero_pivots; ++i) rank += fabs(A[i * cols + i]) > premult;
  // c = Q^T b, then back-substitution on the leading block, then the column permutation
  for (int k = 0; k < nonzero_pivots; ++k) {
    double tmp = b[k];
    for (int r = k + 1; r < rows; ++r) tmp += A[r * cols + k] * b[r];
    b[k] -= hcoef[k] * tmp;
    for (int r = k + 1; r < rows; ++r) b[r] -= hcoef[k] * A[r * cols + k] * tmp;
  }
  double y[3] = {0.0, 0.0, 0.0};
  for (int i = nonzero_pivots - 1; i >= 0; --i) {
    double s = b[i];
    for (int j = i + 1; j < nonzero_pivots; ++j) s -= A[i * cols + j] * y[j];
    y[i] = s / A[i * cols + i];
  }
  int perm[3] = {0, 1, 2};
  for (int k = 0; k < size; ++k) dswap(perm[k], perm[transp[k]]);
  for (int i = 0; i < 3; ++i) x[i] = 0.0;
  for (int i = 0; i < nonzero_pivots; ++i) x[perm[i]] = y[i];
  return rank;
}

// PositionFromTwoRays (sfm/pose/position_from_two_rays.cc:55-83).  corr: 2 x [u v X Y Z], features
// already rotated into the world frame.
RDEV bool position_from_two_rays(const double* corr, double* pos) {
  double A[12], b[4];
  for (int i = 0; i < 2; ++i) {
    const double u = corr[5 * i], v = corr[5 * i + 1], X = corr[5 * i + 2], Y = corr[5 * i + 3], Z = corr[5 * i + 4];
    A[6 * i] = 1.0; A[6 * i + 1] = 0.0; A[6 * i + 2] = -u;
    A[6 * i + 3] = 0.0; A[6 * i + 4] = 1.0; A[6 * i + 5] = -v;
    b[2 * i] = X - u * Z; b[2 * i + 1] = Y - v * Z;
  }
  return colpiv_qr_solve_4x3(A, b, pos) == 3;
}
// AbsolutePoseWithKnownOrientationEstimator::Error (estimate_absolute_pose_with_known_orientation.cc:113-120)
RDEV double known_orientation_abs_error(const double* pos, const double* d) {
  const double px = d[2] - pos[0], py = d[3] - pos[1], pz = d[4] - pos[2];
  const double ex = px / pz - d[0], ey = py / pz - d[1];
  return ex * ex + ey * ey;
}

// ---- EstimateTriangulation (estimate_triangulation.cc:69-101, triangulation.cc:66-125,160-175,
// essential_matrix_utils.cc:86-107).  obs: two PointObservation rows of 33 doubles (layout: theia_hip.h).
// Triangulate(): E from the two projection matrices, Lindstrom's optimal image points, DLT null vector.
RDEV bool triangulate_two_views(const double* obs, double* X) {
  const double* P1 = obs;
  const double* P2 = obs + 33;
  const double x1[2] = {obs[12], obs[13]}, x2[2] = {obs[33 + 12], obs[33 + 13]};
  // relative_rotation = R1 R2^T, translation = (t1 - relative_rotation t2).normalized(), E = [t]x relative_rotation
  double Rr[9], t[3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Rr[3 * i + j] = (P1[4 * i] * P2[4 * j] + P1[4 * i + 1] * P2[4 * j + 1]) + P1[4 * i + 2] * P2[4 * j + 2];
  for (int i = 0; i < 3; ++i) t[i] = P1[4 * i + 3] - ((Rr[3 * i] * P2[3] + Rr[3 * i + 1] * P2[7]) + Rr[3 * i + 2] * P2[11]);
  const double tn = sqrt((t[0] * t[0] + t[1] * t[1]) + t[2] * t[2]);
  for (int i = 0; i < 3; ++i) t[i] /= tn;
  double E[9];
  for (int j = 0; j < 3; ++j) {
    E[j] = -t[2] * Rr[3 + j] + t[1] * Rr[6 + j];
    E[3 + j] = t[2] * Rr[j] - t[0] * Rr[6 + j];
    E[6 + j] = -t[1] * Rr[j] + t[0] * Rr[3 + j];
  }
  // FindOptimalImagePoints (triangulation.cc:66-103)
  const double p1[3] = {x1[0], x1[1], 1.0}, p2[3] = {x2[0], x2[1], 1.0};
  double l1[2], l2[2];   // epipolar_line1 = S E p2, epipolar_line2 = S E^T p1
  for (int i = 0; i < 2; ++i) {
    l1[i] = (E[3 * i] * p2[0] + E[3 * i + 1] * p2[1]) + E[3 * i + 2] * p2[2];
    l2[i] = (E[i] * p1[0] + E[3 + i] * p1[1]) + E[6 + i] * p1[2];
  }
  const double Et[4] = {E[0], E[1], E[3], E[4]};   // top-left 2 x 2
  const double a = l1[0] * (Et[0] * l2[0] + Et[1] * l2[1]) + l1[1] * (Et[2] * l2[0] + Et[3] * l2[1]);
  const double b = ((l1[0] * l1[0] + l1[1] * l1[1]) + (l2[0] * l2[0] + l2[1] * l2[1])) / 2.0;
  double c = 0.0;
  for (int i = 0; i < 3; ++i) c += p1[i] * ((E[3 * i] * p2[0] + E[3 * i + 1] * p2[1]) + E[3 * i + 2] * p2[2]);
  const double d = sqrt(b * b - a * c);
  double lambda = c / (b + d);
  {
    const double n1[2] = {l1[0] - lambda * (Et[0] * l1[0] + Et[1] * l1[1]), l1[1] - lambda * (Et[2] * l1[0] + Et[3] * l1[1])};
    const double n2[2] = {l2[0] - lambda * (Et[0] * l2[0] + Et[2] * l2[1]), l2[1] - lambda * (Et[1] * l2[0] + Et[3] * l2[1])};
    l1[0] = n1[0]; l1[1] = n1[1]; l2[0] = n2[0]; l2[1] = n2[1];
  }
  lambda *= (2.0 * d) / ((l1[0] * l1[0] + l1[1] * l1[1]) + (l2[0] * l2[0] + l2[1] * l2[1]));
  const double c1[2] = {p1[0] - lambda * l1[0], p1[1] - lambda * l1[1]};   // hnormalized(): the third coordinate stays 1
  const double c2[2] = {p2[0] - lambda * l2[0], p2[1] - lambda * l2[1]};
  // TriangulateDLT (triangulation.cc:160-175): last right singular vector of the 4 x 4 design matrix
  double A[16], U[16], S[4], V[16];
  for (int j = 0; j < 4; ++j) {
    A[j] = c1[0] * P1[8 + j] - P1[j];
    A[4 + j] = c1[1] * P1[8 + j] - P1[4 + j];
    A[8 + j] = c2[0] * P2[8 + j] - P2[j];
    A[12 + j] = c2[1] * P2[8 + j] - P2[4 + j];
  }
  svd_sq<4>(A, U, S, V);
  for (int i = 0; i < 4; ++i) X[i] = V[4 * i + 3];
  // IsPointInFrontOfCamera for both (estimate_triangulation.cc:62-66,82-87)
  const double z1 = ((X[0] * P1[8] + X[1] * P1[9]) + X[2] * P1[10]) + X[3] * P1[11];
  const double z2 = ((X[0] * P2[8] + X[1] * P2[9]) + X[2] * P2[10]) + X[3] * P2[11];
  return z1 > 0.0 && z2 > 0.0;
}

// FocalLengthsFromFundamentalMatrix (sfm/pose/fundamental_matrix_util.cc:57-130).
// F row-major.  Epipoles = last right singular vectors of F and F^T.
RDEV bool focal_lengths_from_fundamental(const double* F, double* f1, double* f2) {
  double U[9], S[3], V[9], Ft[9];
  svd3(F, U, S, V);
  const double e1[3] = {V[2], V[5], V[8]};
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Ft[3 * r + c] = F[3 * c + r];
  svd3(Ft, U, S, V);
  const double e2[3] = {V[2], V[5], V[8]};
  if (e1[0] == 0 || e2[0] == 0) return false;
  const double theta1 = atan2(-e1[1], e1[0]);
  const double theta2 = atan2(-e2[1], e2[0]);
  const double R1[9] = {cos(theta1), -sin(theta1), 0.0, sin(theta1), cos(theta1), 0.0, 0.0, 0.0, 1.0};
  const double R2[9] = {cos(theta2), -sin(theta2), 0.0, sin(theta2), cos(theta2), 0.0, 0.0, 0.0, 1.0};
  double R1t[9], tmp[9], rotF[9];
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) R1t[3 * r + c] = R1[3 * c + r];
  matmul3(R2, F, tmp);
  matmul3(tmp, R1t, rotF);
  double re1[3], re2[3];
  for (int r = 0; r < 3; ++r) {
    re1[r] = (R1[3 * r] * e1[0] + R1[3 * r + 1] * e1[1]) + R1[3 * r + 2] * e1[2];
    re2[r] = (R2[3 * r] * e2[0] + R2[3 * r + 1] * e2[1]) + R2[3 * r + 2] * e2[2];
  }
  const double d2i[3] = {1.0 / re2[2], 1.0, 1.0 / (-re2[0])};
  const double d1i[3] = {1.0 / re1[2], 1.0, 1.0 / (-re1[0])};
  const double a = (d2i[0] * rotF[0]) * d1i[0];
  const double b = (d2i[0] * rotF[1]) * d1i[1];
  const double c = (d2i[1] * rotF[3]) * d1i[0];
  const double d = (d2i[1] * rotF[4]) * d1i[1];
  const double f1sq = (-a * c * re1[0] * re1[0]) / (a * c * re1[2] * re1[2] + b * d);
  const double f2sq = (-a * b * re2[0] * re2[0]) / (a * b * re2[2] * re2[2] + c * d);
  if (f1sq < 0 || f2sq < 0) return false;
  *f1 = sqrt(f1sq);
  *f2 = sqrt(f2sq);
  return true;
}

// UncalibratedRelativePoseEstimator::EstimateModel (estimate_uncalibrated_relative_pose.cc:83-138).
// corr: 8 x [x1 y1 x2 y2], principal point removed; minmax = {min, max} focal length or null.
// model: F(9) R(9) position(3) focal_length1 focal_length2
RDEV bool uncalibrated_relative_pose(const double* corr, const double* minmax, double* model) {
  double* F = model;
  if (!eight_point_fundamental(corr, F)) return false;
  double f1, f2;
  if (!focal_lengths_from_fundamental(F, &f1, &f2)) return false;
  if (minmax && minmax[0] >= 1.0 && minmax[1] >= 1.0) {
    if (f1 < minmax[0] || f2 < minmax[0] || f1 > minmax[1] || f2 > minmax[1]) return false;
  }
  // EssentialMatrixFromFundamentalMatrix (fundamental_matrix_util.cc:239-249): diag(f2,f2,1) F diag(f1,f1,1)
  const double dl[3] = {f2, f2, 1.0}, dr[3] = {f1, f1, 1.0};
  double E[9];
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) E[3 * r + c] = (dl[r] * F[3 * r + c]) * dr[c];
  double nc[32];
  for (int i = 0; i < 8; ++i) {
    nc[4 * i] = corr[4 * i] / f1; nc[4 * i + 1] = corr[4 * i + 1] / f1;
    nc[4 * i + 2] = corr[4 * i + 2] / f2; nc[4 * i + 3] = corr[4 * i + 3] / f2;
  }
  best_pose_from_E(E, nc, 8, model + 9, model + 18);
  model[21] = f1; model[22] = f2;
  return true;
}

// UncalibratedRelativePoseEstimator::Error (:181-196): cheirality on the focal-normalised
// correspondence, then the Sampson distance under F in (centred) pixels
RDEV double uncalibrated_relative_pose_error(const double* m, const double* d) {
  const double nd[4] = {d[0] / m[21], d[1] / m[21], d[2] / m[22], d[3] / m[22]};
  if (!in_front(nd, m + 9, m + 18)) return DBL_MAX;
  return sampson(m, d);
}


// ----------------------------------------------------------------- P3P
// math/find_polynomial_roots_companion_matrix.cc for degree >= 3 (the quartic
// of P3P): normalise, companion matrix, power-of-two balancing (gamma = 0.9),
// eigenvalues; REAL PARTS of all roots are returned (reference quirk).
RDEV int poly_roots_real_parts(const double* poly_in, int size, double* real_out) {
  int lead = 0;
  while (lead < size - 1 && poly_in[lead] == 0) ++lead;  // RemoveLeadingZeros
  const double* poly = poly_in + lead;
  const int degree = size - lead - 1;
  if (degree == 0) return 0;
  if (degree == 1) { real_out[0] = -poly[1] / poly[0]; return 1; }
  if (degree == 2) {  // FindQuadraticPolynomialRoots (math/polynomial.cc)
    const double a = poly[0], b = poly[1], c = poly[2];
    const double D = b * b - 4 * a * c;
    const double sqrt_D = sqrt(fabs(D));
    if (D >= 0) {
      if (b >= 0) { real_out[0] = (-b - sqrt_D) / (2.0 * a); real_out[1] = (2.0 * c) / (-b - sqrt_D); }
      else { real_out[0] = (2.0 * c) / (-b + sqrt_D); real_out[1] = (-b + sqrt_D) / (2.0 * a); }
    } else { real_out[0] = -b / (2.0 * a); real_out[1] = -b / (2.0 * a); }
    return 2;
  }
  double p[EIG_MAXN + 1];
  for (int i = 0; i <= degree; ++i) p[i] = poly[i] / poly[0];
  double Cm[EIG_MAXN * EIG_MAXN];
  for (int i = 0; i < degree * degree; ++i) Cm[i] = 0.0;
  for (int i = 1; i < degree; ++i) Cm[i * degree + i - 1] = 1.0;
  for (int i = 0; i < degree; ++i) Cm[i * degree + degree - 1] = -p[degree - i];
  // BalanceCompanionMatrix
  {
    double Off[EIG_MAXN * EIG_MAXN];
    for (int i = 0; i < degree * degree; ++i) Off[i] = Cm[i];
    for (int i = 0; i < degree; ++i) Off[i * degree + i] = 0.0;
    const double gamma = 0.9;
    bool changed;
    do {
      changed = false;
      for (int i = 0; i < degree; ++i) {
        double row_norm = 0.0, col_norm = 0.0;
        for (int j = 0; j < degree; ++j) { row_norm += fabs(Off[i * degree + j]); col_norm += fabs(Off[j * degree + i]); }
        int exponent = 0;
        frexp(row_norm / col_norm, &exponent);
        exponent /= 2;
        if (exponent != 0) {
          const double scaled_col = ldexp(col_norm, exponent);
          const double scaled_row = ldexp(row_norm, -exponent);
          if (scaled_col + scaled_row < gamma * (col_norm + row_norm)) {
            changed = true;
            const double rs = ldexp(1.0, -exponent), cs = ldexp(1.0, exponent);
            for (int j = 0; j < degree; ++j) Off[i * degree + j] *= rs;
            for (int j = 0; j < degree; ++j) Off[j * degree + i] *= cs;
          }
        }
      }
    } while (changed);
    for (int i = 0; i < degree; ++i) Off[i * degree + i] = Cm[i * degree + i];
    for (int i = 0; i < degree * degree; ++i) Cm[i] = Off[i];
  }
  double wr[EIG_MAXN], wi[EIG_MAXN];
  if (!eig_real_general(degree, Cm, wr, wi, nullptr)) return 0;
  for (int i = 0; i < degree; ++i) real_out[i] = wr[i];
  return degree;
}

RDEV void cross3(const double* a, const double* b, double* o) {
  o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
RDEV double dot3(const double* a, const double* b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
RDEV void normalize3(double* a) { const double n = sqrt(dot3(a, a)); a[0] /= n; a[1] /= n; a[2] /= n; }

// perspective_three_point.cc:187-291.  corr: 3 x [u v X Y Z].
// R: up to 4 row-major 3x3 (world -> camera), t: up to 4 translations.
RDEV int p3p(const double* corr, double* Rs, double* ts) {
  double f[3][3], wp[3][3];
  for (int i = 0; i < 3; ++i) {
    f[i][0] = corr[5 * i]; f[i][1] = corr[5 * i + 1]; f[i][2] = 1.0;
    normalize3(f[i]);
    wp[i][0] = corr[5 * i + 2]; wp[i][1] = corr[5 * i + 3]; wp[i][2] = corr[5 * i + 4];
  }
  double w10[3], w20[3], cr[3];
  for (int k = 0; k < 3; ++k) { w10[k] = wp[1][k] - wp[0][k]; w20[k] = wp[2][k] - wp[0][k]; }
  cross3(w10, w20, cr);
  if (dot3(cr, cr) < 1e-6) return 0;
  double T[9];  // intermediate camera frame, rows
  auto build_T = [&]() {
    for (int k = 0; k < 3; ++k) T[k] = f[0][k];
    cross3(f[0], f[1], T + 6); normalize3(T + 6);
    cross3(T + 6, T, T + 3);
  };
  build_T();
  double ip[3] = {dot3(T, f[2]), dot3(T + 3, f[2]), dot3(T + 6, f[2])};
  if (ip[2] > 0) {
    for (int k = 0; k < 3; ++k) { dswap(f[0][k], f[1][k]); }
    build_T();
    ip[0] = dot3(T, f[2]); ip[1] = dot3(T + 3, f[2]); ip[2] = dot3(T + 6, f[2]);
    for (int k = 0; k < 3; ++k) dswap(wp[0][k], wp[1][k]);
    for (int k = 0; k < 3; ++k) { w10[k] = wp[1][k] - wp[0][k]; w20[k] = wp[2][k] - wp[0][k]; }
  }
  double Nw[9];  // intermediate world frame, rows
  for (int k = 0; k < 3; ++k) Nw[k] = w10[k];
  normalize3(Nw);
  cross3(Nw, w20, Nw + 6); normalize3(Nw + 6);
  cross3(Nw + 6, Nw, Nw + 3);
  const double iw[3] = {dot3(Nw, w20), dot3(Nw + 3, w20), dot3(Nw + 6, w20)};
  const double d_12 = sqrt(dot3(w10, w10));
  // SolvePlaneRotation (:56-137)
  const double f_1 = ip[0] / ip[2], f_2 = ip[1] / ip[2];
  const double p_1 = iw[0], p_2 = iw[1];
  const double cos_beta = dot3(f[0], f[1]);
  double b = 1.0 / (1.0 - cos_beta * cos_beta) - 1.0;
  b = (cos_beta < 0) ? -sqrt(b) : sqrt(b);
  const double f_1_pw2 = f_1 * f_1, f_2_pw2 = f_2 * f_2;
  const double p_1_pw2 = p_1 * p_1, p_1_pw3 = p_1_pw2 * p_1, p_1_pw4 = p_1_pw3 * p_1;
  const double p_2_pw2 = p_2 * p_2, p_2_pw3 = p_2_pw2 * p_2, p_2_pw4 = p_2_pw3 * p_2;
  const double d_12_pw2 = d_12 * d_12, b_pw2 = b * b;
  double co[5];
  co[0] = -f_2_pw2 * p_2_pw4 - p_2_pw4 * f_1_pw2 - p_2_pw4;
  co[1] = 2.0 * p_2_pw3 * d_12 * b + 2.0 * f_2_pw2 * p_2_pw3 * d_12 * b - 2.0 * f_2 * p_2_pw3 * f_1 * d_12;
  co[2] = -f_2_pw2 * p_2_pw2 * p_1_pw2 - f_2_pw2 * p_2_pw2 * d_12_pw2 * b_pw2 - f_2_pw2 * p_2_pw2 * d_12_pw2 +
          f_2_pw2 * p_2_pw4 + p_2_pw4 * f_1_pw2 + 2.0 * p_1 * p_2_pw2 * d_12 +
          2.0 * f_1 * f_2 * p_1 * p_2_pw2 * d_12 * b - p_2_pw2 * p_1_pw2 * f_1_pw2 +
          2.0 * p_1 * p_2_pw2 * f_2_pw2 * d_12 - p_2_pw2 * d_12_pw2 * b_pw2 - 2.0 * p_1_pw2 * p_2_pw2;
  co[3] = 2.0 * p_1_pw2 * p_2 * d_12 * b + 2.0 * f_2 * p_2_pw3 * f_1 * d_12 - 2.0 * f_2_pw2 * p_2_pw3 * d_12 * b -
          2.0 * p_1 * p_2 * d_12_pw2 * b;
  co[4] = -2 * f_2 * p_2_pw2 * f_1 * p_1 * d_12 * b + f_2_pw2 * p_2_pw2 * d_12_pw2 + 2.0 * p_1_pw3 * d_12 -
          p_1_pw2 * d_12_pw2 + f_2_pw2 * p_2_pw2 * p_1_pw2 - p_1_pw4 - 2.0 * f_2_pw2 * p_2_pw2 * p_1 * d_12 +
          p_2_pw2 * f_1_pw2 * p_1_pw2 + f_2_pw2 * p_2_pw2 * d_12_pw2 * b_pw2;
  double roots[4];
  const int nroots = poly_roots_real_parts(co, 5, roots);
  for (int i = 0; i < nroots; ++i) {
    const double cos_theta = roots[i];
    const double cot_alpha = (-f_1 * p_1 / f_2 - cos_theta * p_2 + d_12 * b) / (-f_1 * cos_theta * p_2 / f_2 + p_1 - d_12);
    // Backsubstitute (:142-183)
    const double sin_theta = sqrt(1.0 - cos_theta * cos_theta);
    const double sin_alpha = sqrt(1.0 / (cot_alpha * cot_alpha + 1.0));
    double cos_alpha = sqrt(1.0 - sin_alpha * sin_alpha);
    if (cot_alpha < 0) cos_alpha = -cos_alpha;
    const double c_nu[3] = {d_12 * cos_alpha * (sin_alpha * b + cos_alpha),
                            cos_theta * d_12 * sin_alpha * (sin_alpha * b + cos_alpha),
                            sin_theta * d_12 * sin_alpha * (sin_alpha * b + cos_alpha)};
    double tr[3];
    for (int k = 0; k < 3; ++k) tr[k] = wp[0][k] + ((Nw[k] * c_nu[0] + Nw[3 + k] * c_nu[1]) + Nw[6 + k] * c_nu[2]);
    const double Q[9] = {-cos_alpha, -sin_alpha * cos_theta, -sin_alpha * sin_theta,
                         sin_alpha, -cos_alpha * cos_theta, -cos_alpha * sin_theta,
                         0, -sin_theta, cos_theta};
    // rotation = (N^T Q^T T)^T = T^T Q N
    double QN[9];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) QN[r * 3 + c] = (Q[r * 3] * Nw[c] + Q[r * 3 + 1] * Nw[3 + c]) + Q[r * 3 + 2] * Nw[6 + c];
    double* R = Rs + 9 * i;
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) R[r * 3 + c] = (T[r] * QN[c] + T[3 + r] * QN[3 + c]) + T[6 + r] * QN[6 + c];
    double* t = ts + 3 * i;
    for (int r = 0; r < 3; ++r) t[r] = -((R[r * 3] * tr[0] + R[r * 3 + 1] * tr[1]) + R[r * 3 + 2] * tr[2]);
  }
  return nroots;
}



// ------------------------------------------------------------ radial-distortion homography
// sfm/pose/six_point_radial_distortion_homography.cc:62-241, estimators/estimate_radial_distortion_homography.cc:52-111.
//
// Right singular vectors by one-sided Jacobi (Hestenes) on the columns of an M x N row-major matrix: V (N x N, row-major),
// columns ordered by decreasing singular value S.  The reference asks Eigen's JacobiSVD for ComputeFullV of a 6 x 8 and a
// 6 x 5 matrix and reads the columns of the smallest singular values; the null space / least-squares direction they span is
// what enters the solver (any basis of the 6 x 8 matrix' null space gives the same homography up to scale).
template <int M, int N>
RDEV void right_singular_vectors(const double* Ain, double* V, double* S) {
  double W[M * N];
  double fro = 0.0;
  for (int i = 0; i < M * N; ++i) { W[i] = Ain[i]; fro += Ain[i] * Ain[i]; }
  for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) V[i * N + j] = (i == j) ? 1.0 : 0.0;
  const double tiny2 = (1e-28 * fro > DBL_MIN) ? 1e-28 * fro : DBL_MIN;   // columns below 1e-14 of the matrix norm are null columns
  for (int sweep = 0; sweep < 60; ++sweep) {
    bool rotated = false;
    for (int p = 0; p < N - 1; ++p)
      for (int q = p + 1; q < N; ++q) {
        double alpha = 0.0, beta = 0.0, gamma = 0.0;
        for (int i = 0; i < M; ++i) { const double a = W[i * N + p], b = W[i * N + q]; alpha += a * a; beta += b * b; gamma += a * b; }
        if (alpha < tiny2 || beta < tiny2) continue;
        if (!(fabs(gamma) > 2.0 * DBL_EPSILON * sqrt(alpha * beta))) continue;
        rotated = true;
        const double zeta = (beta - alpha) / (2.0 * gamma);
        const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
        for (int i = 0; i < M; ++i) {
          const double a = W[i * N + p], b = W[i * N + q];
          W[i * N + p] = c * a - s * b; W[i * N + q] = s * a + c * b;
        }
        for (int i = 0; i < N; ++i) {
          const double a = V[i * N + p], b = V[i * N + q];
          V[i * N + p] = c * a - s * b; V[i * N + q] = s * a + c * b;
        }
      }
    if (!rotated) break;
  }
  for (int j = 0; j < N; ++j) { double s2 = 0.0; for (int i = 0; i < M; ++i) s2 += W[i * N + j] * W[i * N + j]; S[j] = sqrt(s2); }
  for (int i = 0; i < N; ++i) {  // selection sort, descending
    int best = i;
    for (int j = i + 1; j < N; ++j) if (S[j] > S[best]) best = j;
    if (best != i) {
      dswap(S[i], S[best]);
      for (int k = 0; k < N; ++k) dswap(V[k * N + i], V[k * N + best]);
    }
  }
}

// Eigen's 3 x 3 inverse (Inverse_impl.h compute_inverse_size3_helper): cofactors over the determinant.  Row-major.
RDEV void inverse3_cofactor(const double* m, double* r) {
#define COF(i, j) (m[3 * (((i) + 1) % 3) + (((j) + 1) % 3)] * m[3 * (((i) + 2) % 3) + (((j) + 2) % 3)] - \
                   m[3 * (((i) + 1) % 3) + (((j) + 2) % 3)] * m[3 * (((i) + 2) % 3) + (((j) + 1) % 3)])
  const double c00 = COF(0, 0), c10 = COF(1, 0), c20 = COF(2, 0);
  const double det = (c00 * m[0] + c10 * m[3]) + c20 * m[6];
  const double invdet = 1.0 / det;
  r[0] = c00 * invdet; r[1] = c10 * invdet; r[2] = c20 * invdet;
  r[3] = COF(0, 1) * invdet; r[4] = COF(1, 1) * invdet; r[5] = COF(2, 1) * invdet;
  r[6] = COF(0, 2) * invdet; r[7] = COF(1, 2) * invdet; r[8] = COF(2, 2) * invdet;
#undef COF
}

constexpr int kRadHomDatum = 12;   // [feature_left (2) | feature_right (2) | normalized left (2) | normalized right (2) | f_left f_right | lmin lmax]
// SixPointRadialDistortionHomography on six data rows; models: up to two rows of `stride` doubles
// {H (9, row-major) | l1 | l2 | H^-1 (9)}.  Returns the number of results (the reference's bool is "the quadratic had a real
// root"; solutions outside [lmin, lmax] are dropped, so zero results with a true return is possible there as well).
RDEV int radial_homography_six_point(const double* d, double* models, int stride) {
  double Mx[48], u2[6], x2[6];
  const double lmin = d[10], lmax = d[11];
  for (int i = 0; i < 6; ++i) {
    const double* r = d + kRadHomDatum * i;
    const double x0 = r[4], x1 = r[5], u0 = r[6], u1 = r[7];
    u2[i] = u0 * u0 + u1 * u1;
    x2[i] = x0 * x0 + x1 * x1;
    double* m = Mx + 8 * i;
    m[0] = -x1 * u0; m[1] = -x1 * u1; m[2] = -x1; m[3] = x0 * u0; m[4] = x0 * u1; m[5] = x0; m[6] = -x1 * u2[i]; m[7] = x0 * u2[i];
  }
  double V1[64], S1[8];
  right_singular_vectors<6, 8>(Mx, V1, S1);
#define V1E(r, c) V1[(r) * 8 + (c)]
  const double a = -V1E(2, 6) * V1E(7, 6) + V1E(5, 6) * V1E(6, 6);
  const double b = ((-V1E(2, 6) * V1E(7, 7) - V1E(2, 7) * V1E(7, 6)) + V1E(5, 6) * V1E(6, 7)) + V1E(5, 7) * V1E(6, 6);
  const double c = -V1E(2, 7) * V1E(7, 7) + V1E(5, 7) * V1E(6, 7);
  const double dd = b * b - 4.0 * a * c;
  int nsols = 0;
  double rs[2] = {0.0, 0.0};
  if ((dd < 100.0 * DBL_EPSILON) && (-dd < 100.0 * DBL_EPSILON)) { nsols = 1; rs[0] = (-b) / (2.0 * a); }
  else if (dd > 0.0) { nsols = 2; const double d2 = sqrt(dd); rs[0] = (-b + d2) / (2.0 * a); rs[1] = (-b - d2) / (2.0 * a); }
  else return 0;
  int nres = 0;
  for (int s = 0; s < nsols; ++s) {
    double n[8];
    for (int k = 0; k < 8; ++k) n[k] = rs[s] * V1E(k, 6) + V1E(k, 7);
    const double l2 = n[6] / n[2];
    if (l2 < lmin || l2 > lmax || !(l2 == l2)) continue;
    double T[30];
    for (int i = 0; i < 6; ++i) {
      const double* r = d + kRadHomDatum * i;
      const double x0 = r[4], u0 = r[6], u1 = r[7];
      const double u3 = 1.0 + l2 * u2[i];
      const double rr = (n[0] * u0 + n[1] * u1) + n[2] * u3;
      T[5 * i + 0] = -Mx[8 * i + 3]; T[5 * i + 1] = -Mx[8 * i + 4]; T[5 * i + 2] = -x0 * u3; T[5 * i + 3] = x2[i] * rr; T[5 * i + 4] = rr;
    }
    double V2[25], S2[5];
    right_singular_vectors<6, 5>(T, V2, S2);
    const double v4 = V2[4 * 5 + 4];
    const double v[4] = {V2[0 * 5 + 4] / v4, V2[1 * 5 + 4] / v4, V2[2 * 5 + 4] / v4, V2[3 * 5 + 4] / v4};
    const double l1 = v[3];
    if (l1 < lmin || l1 > lmax || !(l1 == l1)) continue;
    double* m = models + (size_t)stride * nres;
    for (int k = 0; k < stride; ++k) m[k] = 0.0;
    for (int k = 0; k < 6; ++k) m[k] = n[k];
    m[6] = v[0]; m[7] = v[1]; m[8] = v[2];
    m[9] = l1; m[10] = l2;
    inverse3_cofactor(m, m + 11);
    nres++;
  }
#undef V1E
  return nres;
}

// division-model helpers of six_point_radial_distortion_homography.cc:151-193
RDEV void radhom_distort(const double* p3, double f, double l, double* out) {
  const double px = f * p3[0] / p3[2], py = f * p3[1] / p3[2];
  const double r_u_sq = px * px + py * py;
  const double denom = 2.0 * l * r_u_sq;
  const double inner = 1.0 - 4.0 * l * r_u_sq;
  if (fabs(denom) < DBL_EPSILON || inner < 0.0) { out[0] = px; out[1] = py; }
  else { const double scale = (1.0 - sqrt(inner)) / denom; out[0] = px * scale; out[1] = py * scale; }
}
RDEV void radhom_undistort(const double* p2, double f, double l, double* out3) {
  const double r_d_sq = p2[0] * p2[0] + p2[1] * p2[1];
  const double und = 1.0 / (1.0 + l * r_d_sq);
  out3[0] = p2[0] * und / f; out3[1] = p2[1] * und / f; out3[2] = 1.0;
}
// CheckRadialSymmetricError (:201-239) with the model row {H | l1 | l2 | H^-1}
RDEV double radial_homography_error(const double* m, const double* d) {
  const double f1 = d[8], f2 = d[9];
  const double l1s = m[9] / (f1 * f1), l2s = m[10] / (f2 * f2);
  double bl[3], br[3], y[3], z[3], pl[2], pr[2];
  radhom_undistort(d, f1, l1s, bl);
  radhom_undistort(d + 2, f2, l2s, br);
  for (int i = 0; i < 3; ++i) {
    y[i] = (m[3 * i] * br[0] + m[3 * i + 1] * br[1]) + m[3 * i + 2] * br[2];                   // ray 2 in camera 1
    z[i] = (m[11 + 3 * i] * bl[0] + m[11 + 3 * i + 1] * bl[1]) + m[11 + 3 * i + 2] * bl[2];    // ray 1 in camera 2
  }
  const double yz = y[2], zz = z[2];
  for (int i = 0; i < 3; ++i) { y[i] /= yz; z[i] /= zz; }
  radhom_distort(y, f1, l1s, pl);
  radhom_distort(z, f2, l2s, pr);
  const double dlx = d[0] - pl[0], dly = d[1] - pl[1], drx = d[2] - pr[0], dry = d[3] - pr[1];
  return 0.5 * ((dlx * dlx + dly * dly) + (drx * drx + dry * dry));
}

}  // namespace rsc
}  // namespace thip
