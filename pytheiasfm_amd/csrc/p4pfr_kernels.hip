// P4Pfr on the device: absolute pose + focal length + one radial-distortion coefficient from four 2D-3D correspondences
// (FourPointsPoseFocalLengthRadialDistortion, sfm/pose/four_point_focal_length_radial_distortion.cc:68-288 + _helper.cc;
// RadialDistUncalibratedAbsolutePoseEstimator, sfm/estimators/estimate_radial_dist_uncalibrated_absolute_pose.cc:76-160) behind
// THEIA_EST_RADIAL_DIST_UNCALIBRATED_ABSOLUTE_POSE.
//
// The polynomial system and the layout of the reference's 40 x 50 elimination template are in p4pfr_layout.h (derived from the
// geometry and checked symbolically against the reference's 327 generated formulas by scripts/gen_p4pfr_layout.py); the
// coefficients come from polynomial arithmetic, the elimination follows the reference's route (Eigen::FullPivLU of the 37 x 40
// transposed block, solve() with the free unknowns at zero).  Operation order = oracle/p4pfr_oracle.h, so that hypotheses, inlier
// sets and models agree with the oracle bit for bit.
//
//   k_p4pfr_pre  one THREAD per hypothesis: the normalisation (centroid, JacobiSVD rotation, scales), the 5 x 8 linear system, its
//                null space (Householder QR) rotated by the hypothesis' "random rotation" (a matrix the HOST makes from the three
//                RandDouble draws: sin / cos come from the same libm as the reference's), the particular solution, D
//   k_p4pfr_a    one WAVE per hypothesis, everything in LDS: the 21 products q_i q_j and the coefficients of the ten equations
//                (one lane per output coefficient, terms summed in the oracle's loop order from host-built gather lists), the
//                template, full-pivot LU of [C0^T | -I columns] with lane = column (the forward substitution rides along as seven
//                more columns), rank, back-substitution, RR = alpha^T C1, the 13 x 13 action matrix
//   k_p4pfr_b    teams of 8 lanes: the 13 x 13 eigen-decomposition (eig_team.h), eigenvector rows over the row of 1,
//                |Im a1| <= 1e-6, projection matrix -> (R, t, focal length, distortion), the metadata's range tests, models
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>
#include <cstdint>
#include <mutex>
#include <vector>

#include "eig_team.h"
#include "wave_reduce.h"
#include "p4pfr_layout.h"
#include "ransac_device.h"
#include "theia_hip_internal.h"

namespace thip {
namespace p4pfrdev {
using namespace thip::p4pfr_layout;

constexpr int kMaxModels = 13, kModel = 14;
// workspace per hypothesis: R0 (9) t0 (3) scale f0 k0 | Nn (8 x 4) | D (3 x 9) | d0 | U0 (3) | action matrix (13 x 13)
constexpr int kWsR0 = 0, kWsT0 = 9, kWsScale = 12, kWsF0 = 13, kWsK0 = 14, kWsN = 15, kWsD = 47, kWsD0 = 74, kWsU0 = 75, kWsAct = 78, kWs = 248;
// operand pool of k_p4pfr_a: Nn (32) | p3 rows x y z over tmp (27) | q_i q_j, i <= j (21 x 10) | 1
constexpr int kPoolN = 0, kPoolP3 = 32, kPoolQQ = 59, kPoolOne = 269, kPool = 270;
constexpr int kEqOut = 9 * kCols;        // equations 0 .. 8 through the gather lists (equation 9 is linear: written directly)
constexpr int kMaxTerms = 4096;

struct Tables {
  uint16_t qq_off[211];                  // products q_i q_j: terms (s, t) of output (pair, u) in loop order
  uint8_t qq_s[336], qq_t[336], qq_i[21], qq_j[21];
  uint16_t eq_off[kEqOut + 1];           // terms of coefficient (e, column): c * (pool[a] * pool[b])
  uint16_t ta[kMaxTerms], tb[kMaxTerms];
  int8_t tc[kMaxTerms];
};
__device__ Tables g_tab;

// ---------------------------------------------------------------- small dense algebra (oracle/p4pfr_oracle.h, same order)
struct Qr { int rows, cols, size, nonzero_pivots; double hcoef[8]; int transp[8]; };

RDEV void make_householder(double* A, int rows, int cols, int k, double* tau, double* beta) {
  const double c0 = A[k * cols + k];
  double tail = 0.0;
  for (int r = k + 1; r < rows; ++r) tail += A[r * cols + k] * A[r * cols + k];
  if (tail <= DBL_MIN) {
    *tau = 0.0; *beta = c0;
    for (int r = k + 1; r < rows; ++r) A[r * cols + k] = 0.0;
  } else {
    double b = sqrt(c0 * c0 + tail);
    if (c0 >= 0.0) b = -b;
    for (int r = k + 1; r < rows; ++r) A[r * cols + k] /= (c0 - b);
    *tau = (b - c0) / b; *beta = b;
  }
}
RDEV void apply_householder(const double* A, int rows, int cols, int k, double tau, double* B, int ldb, int j) {
  if (tau == 0.0) return;
  double tmp = 0.0;
  for (int r = k + 1; r < rows; ++r) tmp += A[r * cols + k] * B[r * ldb + j];
  tmp += B[k * ldb + j];
  B[k * ldb + j] -= tau * tmp;
  for (int r = k + 1; r < rows; ++r) B[r * ldb + j] -= (tau * A[r * cols + k]) * tmp;
}
RDEV void qr_factor(double* A, int rows, int cols, bool pivot, Qr& f) {
  f.rows = rows; f.cols = cols; f.size = rows < cols ? rows : cols; f.nonzero_pivots = f.size;
  double norm_upd[8], norm_dir[8];
  double threshold_helper = 0.0;
  const double norm_downdate_threshold = sqrt(DBL_EPSILON);
  if (pivot) {
    double maxn = 0.0;
    for (int k = 0; k < cols; ++k) {
      double s2 = 0.0;
      for (int r = 0; r < rows; ++r) s2 += A[r * cols + k] * A[r * cols + k];
      norm_upd[k] = norm_dir[k] = sqrt(s2);
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
        for (int r = 0; r < rows; ++r) rsc::dswap(A[r * cols + k], A[r * cols + big]);
        rsc::dswap(norm_upd[k], norm_upd[big]);
        rsc::dswap(norm_dir[k], norm_dir[big]);
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
        double temp = fabs(A[k * cols + j]) / norm_upd[j];
        temp = (1.0 + temp) * (1.0 - temp);
        temp = temp < 0.0 ? 0.0 : temp;
        const double q = norm_upd[j] / norm_dir[j];
        if (temp * (q * q) <= norm_downdate_threshold) {
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
RDEV void qr_q(const double* A, const Qr& f, double* Q) {
  const int n = f.rows;
  for (int i = 0; i < n * n; ++i) Q[i] = (i % (n + 1) == 0) ? 1.0 : 0.0;
  for (int k = f.size - 1; k >= 0; --k)
    for (int j = k; j < n; ++j) apply_householder(A, f.rows, f.cols, k, f.hcoef[k], Q, n, j);
}
RDEV void qr_solve(const double* A, const Qr& f, double* B, int nrhs, double* X) {
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
    for (int k = 0; k < f.size; ++k) rsc::dswap(perm[k], perm[f.transp[k]]);
    for (int i = 0; i < cols; ++i) X[i * nrhs + j] = 0.0;
    for (int i = 0; i < np; ++i) X[perm[i] * nrhs + j] = y[i];
  }
}
// FullPivLU::solve of a 5 x 5 system with one right-hand side (the particular solution)
RDEV void fullpiv_solve5(double* A, double* B, double* X) {
  const int n = 5;
  int rowt[5], colt[5];
  int nonzero = n;
  double maxpivot = 0.0;
  for (int k = 0; k < n; ++k) {
    double best = -1.0; int br = k, bc = k;
    for (int j = k; j < n; ++j)
      for (int i = k; i < n; ++i) {
        const double a = fabs(A[i * n + j]);
        if (a > best) { best = a; br = i; bc = j; }
      }
    if (best == 0.0) {
      nonzero = k;
      for (int i = k; i < n; ++i) { rowt[i] = i; colt[i] = i; }
      break;
    }
    if (best > maxpivot) maxpivot = best;
    rowt[k] = br; colt[k] = bc;
    if (br != k) for (int j = 0; j < n; ++j) rsc::dswap(A[k * n + j], A[br * n + j]);
    if (bc != k) for (int i = 0; i < n; ++i) rsc::dswap(A[i * n + k], A[i * n + bc]);
    if (k < n - 1) for (int i = k + 1; i < n; ++i) A[i * n + k] /= A[k * n + k];
    if (k < n - 1)
      for (int i = k + 1; i < n; ++i)
        for (int j = k + 1; j < n; ++j) A[i * n + j] -= A[i * n + k] * A[k * n + j];
  }
  const double premult = maxpivot * (DBL_EPSILON * (double)n);
  int rank = 0;
  for (int i = 0; i < nonzero; ++i) rank += fabs(A[i * n + i]) > premult;
  for (int i = 0; i < n; ++i) X[i] = 0.0;
  if (rank == 0) return;
  for (int k = 0; k < n; ++k) if (rowt[k] != k) rsc::dswap(B[k], B[rowt[k]]);
  for (int k = 0; k < n; ++k) for (int i = k + 1; i < n; ++i) B[i] -= A[i * n + k] * B[k];
  for (int k = rank - 1; k >= 0; --k) {
    B[k] /= A[k * n + k];
    for (int i = 0; i < k; ++i) B[i] -= A[i * n + k] * B[k];
  }
  int perm[5] = {0, 1, 2, 3, 4};
  for (int k = 0; k < n; ++k) rsc::dswap(perm[k], perm[colt[k]]);
  for (int i = 0; i < rank; ++i) X[perm[i]] = B[i];
}
RDEV void svd_u_3x4(const double* A, double* U) {
  double scale = 0.0;
  for (int i = 0; i < 12; ++i) scale = fmax(scale, fabs(A[i]));
  if (scale == 0.0) scale = 1.0;
  double At[12];
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) At[c * 3 + r] = A[r * 4 + c] / scale;
  Qr f;
  qr_factor(At, 4, 3, true, f);
  double W[9], V[9], S[3];
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) W[r * 3 + c] = (c <= r) ? At[c * 3 + r] : 0.0;
  int perm[3] = {0, 1, 2};
  for (int k = 0; k < 3; ++k) rsc::dswap(perm[k], perm[f.transp[k]]);
  for (int i = 0; i < 9; ++i) { U[i] = 0.0; V[i] = (i % 4 == 0) ? 1.0 : 0.0; }
  for (int j = 0; j < 3; ++j) U[perm[j] * 3 + j] = 1.0;
  rsc::svd3_sweeps(W, U, S, V);
}

// four_point_focal_length_radial_distortion.cc:87-215.  feat 4 x 2, world 4 x 3, Rr: the "random rotation" matrix; w: the workspace row
RDEV void normalise(const double* feat, const double* world, const double* Rr, double* w) {
  double* R0 = w + kWsR0; double* t0 = w + kWsT0; double* Nn = w + kWsN; double* Dm = w + kWsD;
  double d[4], u[2][4], Um[4][4];
  for (int i = 0; i < 4; ++i) d[i] = feat[2 * i] * feat[2 * i] + feat[2 * i + 1] * feat[2 * i + 1];
  for (int r = 0; r < 3; ++r) t0[r] = ((world[r] + world[3 + r]) + (world[6 + r] + world[9 + r])) / 4.0;
  double A[12];
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) A[r * 4 + c] = world[3 * c + r] - t0[r];
  double Us[9];
  svd_u_3x4(A, Us);
  if (rsc::det3(Us) < 0.0) for (int r = 0; r < 3; ++r) Us[3 * r] *= -1.0;
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) R0[r * 3 + c] = Us[c * 3 + r];
  for (int c = 0; c < 4; ++c) {
    for (int r = 0; r < 3; ++r) Um[r][c] = (R0[3 * r] * A[c] + R0[3 * r + 1] * A[4 + c]) + R0[3 * r + 2] * A[8 + c];
    Um[3][c] = 1.0;
  }
  double cn[4], fn[4];
  for (int c = 0; c < 4; ++c) {
    cn[c] = sqrt((Um[0][c] * Um[0][c] + Um[1][c] * Um[1][c]) + Um[2][c] * Um[2][c]);
    fn[c] = sqrt(feat[2 * c] * feat[2 * c] + feat[2 * c + 1] * feat[2 * c + 1]);
  }
  const double scale = ((cn[0] + cn[1]) + (cn[2] + cn[3])) / 4.0;
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) Um[r][c] /= scale;
  const double f0 = ((fn[0] + fn[1]) + (fn[2] + fn[3])) / 4.0;
  for (int c = 0; c < 4; ++c) { u[0][c] = feat[2 * c] / f0; u[1][c] = feat[2 * c + 1] / f0; }
  const double k0 = ((d[0] + d[1]) + (d[2] + d[3])) / 4.0;
  for (int c = 0; c < 4; ++c) d[c] /= k0;
  w[kWsScale] = scale; w[kWsF0] = f0; w[kWsK0] = k0;
  double Mt[40];
  for (int i = 0; i < 40; ++i) Mt[i] = 0.0;
  for (int c = 0; c < 4; ++c) { Mt[c * 5 + 0] = Um[c][0]; Mt[(4 + c) * 5 + 1] = Um[c][0]; }
  for (int k = 1; k < 4; ++k)
    for (int c = 0; c < 4; ++c) { Mt[c * 5 + k + 1] = u[1][k] * Um[c][k]; Mt[(4 + c) * 5 + k + 1] = -u[0][k] * Um[c][k]; }
  Qr f;
  qr_factor(Mt, 8, 5, false, f);
  double Q[64];
  qr_q(Mt, f, Q);
  for (int r = 0; r < 8; ++r)
    for (int c = 0; c < 3; ++c)
      Nn[r * 4 + c] = (Q[r * 8 + 5] * Rr[c] + Q[r * 8 + 6] * Rr[3 + c]) + Q[r * 8 + 7] * Rr[6 + c];
  double Rt[25], b[5] = {u[0][0], u[1][0], 0.0, 0.0, 0.0}, y[5];
  for (int r = 0; r < 5; ++r) for (int c = 0; c < 5; ++c) Rt[r * 5 + c] = (c <= r) ? Mt[c * 5 + r] : 0.0;
  fullpiv_solve5(Rt, b, y);
  for (int r = 0; r < 8; ++r) {
    double s = 0.0;
    for (int c = 0; c < 5; ++c) s += Q[r * 8 + c] * y[c];
    Nn[r * 4 + 3] = s;
  }
  double B[54], Cm[18];
  for (int h = 0; h < 2; ++h)
    for (int k = 0; k < 3; ++k) {
      double un[4];
      for (int c = 0; c < 4; ++c) {
        double s = 0.0;
        for (int r = 0; r < 4; ++r) s += Um[r][k + 1] * Nn[(4 * h + r) * 4 + c];
        un[c] = s;
      }
      double* row = B + (3 * h + k) * 9;
      for (int c = 0; c < 3; ++c) row[c] = un[c];
      for (int c = 0; c < 4; ++c) row[3 + c] = d[k + 1] * un[c];
      row[7] = -u[h][k + 1] * Um[2][k + 1];
      row[8] = un[3];
      Cm[(3 * h + k) * 3 + 0] = Um[0][k + 1] * u[h][k + 1];
      Cm[(3 * h + k) * 3 + 1] = Um[1][k + 1] * u[h][k + 1];
      Cm[(3 * h + k) * 3 + 2] = Um[3][k + 1] * u[h][k + 1];
    }
  Qr g;
  qr_factor(Cm, 6, 3, true, g);
  qr_solve(Cm, g, B, 9, Dm);
  w[kWsD0] = d[0];
  for (int r = 0; r < 3; ++r) w[kWsU0 + r] = Um[r][0];
}

__global__ __launch_bounds__(64) void k_p4pfr_pre(int nprob, int B, const int64_t* __restrict__ offsets, const double* __restrict__ data,
                                                  const int* __restrict__ samples, const int* __restrict__ active_iters,
                                                  const double* __restrict__ rot, double* __restrict__ ws) {
  const int p = blockIdx.y, b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B || b >= active_iters[p]) return;
  const size_t hyp = (size_t)p * B + b;
  const double* pd = data + (size_t)offsets[p] * 5;
  double feat[8], world[12], Rr[9];
  for (int i = 0; i < 4; ++i) {
    const double* d = pd + (size_t)samples[hyp * 4 + i] * 5;
    feat[2 * i] = d[0]; feat[2 * i + 1] = d[1];
    for (int k = 0; k < 3; ++k) world[3 * i + k] = d[2 + k];
  }
  for (int k = 0; k < 9; ++k) Rr[k] = rot[hyp * 9 + k];
  normalise(feat, world, Rr, ws + hyp * kWs);
}

// ---------------------------------------------------------------- template, elimination, action matrix: one wave per hypothesis
constexpr int kLd = kRows + kReduced;   // 47: the 40 columns of C0^T and the 7 right-hand sides; odd, so a column's rows spread over the banks
__global__ __launch_bounds__(64) void k_p4pfr_a(int B, const int* __restrict__ active_iters, double* __restrict__ ws) {
  __shared__ double pool[kPool], eq[10 * kCols], M[kElim * kLd], C1[kRows * kBasis], X[kRows * kReduced];
  __shared__ int perm[kRows];
  const int p = blockIdx.y, b = blockIdx.x, lane = threadIdx.x;
  if (b >= active_iters[p]) return;
  double* w = ws + ((size_t)p * B + b) * kWs;
  // operands: Nn | p3x p3y over tmp (D rows 0, 1), p3z = w = tmp[7] | 1
  if (lane < 32) pool[kPoolN + lane] = w[kWsN + lane];
  if (lane < 27) pool[kPoolP3 + lane] = lane < 18 ? w[kWsD + lane] : (lane - 18 == 7 ? 1.0 : 0.0);
  if (lane == 63) pool[kPoolOne] = 1.0;
  __syncthreads();
  // q_i q_j over the ten quadratic monomials: rows of [p1; p2] are Nn rows 0 1 2 4 5 6
  for (int o = lane; o < 210; o += 64) {
    const int pr = o / 10;
    const int ri = g_tab.qq_i[pr], rj = g_tab.qq_j[pr];
    double acc = 0.0;
    for (int t = g_tab.qq_off[o]; t < g_tab.qq_off[o + 1]; ++t) acc += pool[kPoolN + ri * 4 + g_tab.qq_s[t]] * pool[kPoolN + rj * 4 + g_tab.qq_t[t]];
    pool[kPoolQQ + o] = acc;
  }
  __syncthreads();
  for (int o = lane; o < kEqOut; o += 64) {
    double acc = 0.0;
    for (int t = g_tab.eq_off[o]; t < g_tab.eq_off[o + 1]; ++t) acc += (double)g_tab.tc[t] * (pool[g_tab.ta[t]] * pool[g_tab.tb[t]]);
    eq[o] = acc;
  }
  if (lane < kCols) eq[9 * kCols + lane] = 0.0;
  __syncthreads();
  if (lane < 9) {   // (1 + k d0) - (U0x p3x + U0y p3y + U0z w + p3w)
    const double* U0 = w + kWsU0; const double* D = w + kWsD;
    double v = -(((U0[0] * D[lane] + U0[1] * D[9 + lane]) + U0[2] * pool[kPoolP3 + 18 + lane]) + D[18 + lane]);
    if (lane == 8) v += 1.0;
    if (lane == 6) v += w[kWsD0];
    eq[9 * kCols + kTmpCol[lane]] = v;
  }
  __syncthreads();
  // M = [C0^T | b^T]: entry (i, r) = template (r, i), i < 37; right-hand side i of the reduced column 30 + i; C1 = template[:, 37:]
  for (int e = lane; e < kElim * kLd; e += 64) {
    const int i = e / kLd, r = e % kLd;
    double v;
    if (r < kRows) { const int s = kRowSrc[r][i]; v = s >= 0 ? eq[kRowEq[r] * kCols + s] : 0.0; }
    else v = (i == kFirstReduced + (r - kRows)) ? -1.0 : 0.0;
    M[e] = v;
  }
  for (int e = lane; e < kRows * kBasis; e += 64) {
    const int r = e / kBasis, c = e % kBasis;
    const int s = kRowSrc[r][kElim + c];
    C1[e] = s >= 0 ? eq[kRowEq[r] * kCols + s] : 0.0;
  }
  if (lane < kRows) perm[lane] = lane;
  __syncthreads();
  // Eigen::FullPivLU::compute on the 37 x 40 block, lane = column (the right-hand sides take the row operations along)
  int nonzero = kElim;
  double maxpivot = 0.0;
  // the search of step k rides in the update of step k - 1 (each lane sees the new entries of its column as it writes them);
  // only step 0 scans on its own.  best / brow: the first strict maximum of |column| over the rows that remain.
  double best = -1.0; int brow = 0;
  if (lane < kRows)
    for (int i = 0; i < kElim; ++i) {
      const double a = fabs(M[i * kLd + lane]);
      if (a > best) { best = a; brow = i; }
    }
  for (int k = 0; k < kElim; ++k) {
    // the first strict maximum in column-major order: the largest value, in the lowest column that holds it
    const double gbest = wave_max_abs(best);   // (wave_reduce.h: fmax over the lanes' bits, on the DPP network)
    const unsigned long long holders = __ballot(best == gbest && lane >= k && lane < kRows);
    if (gbest == 0.0 || holders == 0ull) { nonzero = k; break; }
    const int bc = __ffsll((long long)holders) - 1;
    const int br = __builtin_amdgcn_readlane(brow, bc);
    if (gbest > maxpivot) maxpivot = gbest;
    if (br != k && lane < kLd) { const double t = M[k * kLd + lane]; M[k * kLd + lane] = M[br * kLd + lane]; M[br * kLd + lane] = t; }
    __syncthreads();
    if (bc != k) {
      if (lane < kElim) { const double t = M[lane * kLd + k]; M[lane * kLd + k] = M[lane * kLd + bc]; M[lane * kLd + bc] = t; }
      if (lane == 63) { const int t = perm[k]; perm[k] = perm[bc]; perm[bc] = t; }
    }
    __syncthreads();
    const double piv = M[k * kLd + k];
    if (lane > k && lane < kElim) M[lane * kLd + k] /= piv;
    __syncthreads();
    best = -1.0; brow = k + 1;
    if (lane > k && lane < kLd) {
      const double ukj = M[k * kLd + lane];
      const bool searched = lane < kRows;
      for (int i = k + 1; i < kElim; ++i) {
        const double v = M[i * kLd + lane] - M[i * kLd + k] * ukj;
        M[i * kLd + lane] = v;
        const double a = fabs(v);
        if (searched && a > best) { best = a; brow = i; }
      }
    }
    __syncthreads();
  }
  // rank() with the default threshold eps * diagonalSize
  const double premult = maxpivot * (DBL_EPSILON * (double)kElim);
  const int rank = __popcll(__ballot(lane < nonzero && fabs(M[(lane < kElim ? lane : 0) * kLd + (lane < kElim ? lane : 0)]) > premult));
  for (int e = lane; e < kRows * kReduced; e += 64) X[e] = 0.0;
  __syncthreads();
  // the upper solve on the leading rank x rank block, column-oriented; lane = (right-hand side, row slice)
  const int rj = lane / 9, rs = lane % 9;
  for (int k = rank - 1; k >= 0; --k) {
    if (lane < kReduced) M[k * kLd + kRows + lane] /= M[k * kLd + k];
    __syncthreads();
    if (rj < kReduced) {
      const double xk = M[k * kLd + kRows + rj];
      for (int i = rs; i < k; i += 9) M[i * kLd + kRows + rj] -= M[i * kLd + k] * xk;
    }
    __syncthreads();
  }
  for (int e = lane; e < rank * kReduced; e += 64) {
    const int i = e / kReduced, j = e % kReduced;
    X[perm[i] * kReduced + j] = M[i * kLd + kRows + j];
  }
  __syncthreads();
  // RR = alpha^T C1 (rows 0 .. 6), identity below; the action matrix picks rows kAmRow
  for (int e = lane; e < kBasis * kBasis; e += 64) {
    const int i = e / kBasis, c = e % kBasis, src = kAmRow[i];
    double v;
    if (src < kReduced) {
      double s = 0.0;
      for (int r = 0; r < kRows; ++r) s += X[r * kReduced + src] * C1[r * kBasis + c];
      v = s;
    } else {
      v = (src - kReduced == c) ? 1.0 : 0.0;
    }
    w[kWsAct + e] = v;
  }
}

// ---------------------------------------------------------------- eigenvectors -> solutions -> models
constexpr int kTeam = 8, kTeamsPerWave = 64 / kTeam;
constexpr int kEigX = 16 + kBasis * (kModel + 1);          // the work array X of the eigen stage, reused for the candidates' slots (>= 13 x 13)
constexpr int kEigLds = 2 * kBasis * kBasis + kEigX + 3 * kBasis;
__global__ __launch_bounds__(64) void k_p4pfr_b(size_t nhyp, int B, const int* __restrict__ active_iters, const double* __restrict__ ws,
                                                double max_f, double min_f, double max_d, double min_d, double* __restrict__ models,
                                                int* __restrict__ counts, int* __restrict__ dense_count, int* __restrict__ tags, int* __restrict__ hyp_base,
                                                int* __restrict__ solver_counts) {
  __shared__ double lds[kTeamsPerWave][kEigLds];
  const int team = threadIdx.x / kTeam, tl = threadIdx.x % kTeam;
  const size_t hyp = (size_t)blockIdx.x * kTeamsPerWave + team;
  if (hyp >= nhyp) return;
  const int p = (int)(hyp / B), b = (int)(hyp % B);
  if (b >= active_iters[p]) { if (tl == 0) { counts[hyp] = 0; if (solver_counts) solver_counts[hyp] = 0; } return; }
  const double* w = ws + hyp * kWs;
  constexpr int n = kBasis;
  double* H = lds[team]; double* V = H + n * n; double* Xw = V + n * n; double* wr = Xw + kEigX; double* wi = wr + n; double* ort = wi + n;
  for (int e = tl; e < n * n; e += kTeam) H[e] = w[kWsAct + e];
  rsc::team_sync();
  const bool good = rsc::eig_team<kTeam, true>(n, H, V, Xw, wr, wi, ort, tl);
  rsc::team_sync();
  if (!good) { if (tl == 0) { counts[hyp] = 0; if (solver_counts) solver_counts[hyp] = 0; } return; }
  // helper.cc:1384-1408: columns over their first row, |Im a1| <= 1e-6, real parts (EigenSolver::eigenvectors(): a column is real
  // when |Im lambda| <= 1e-12 |Re lambda| or it is the last one, else columns j, j + 1 are re +- i im; each normalised).  The
  // thirteen candidates -- slot k = real column k, or the first / second member of the pair that covers column k -- are dealt to
  // the team's lanes (the eigen stage leaves seven of eight lanes idle otherwise); lane 0 then collects them in slot order, so the
  // models come out in the order of the one-lane loop, each from the same arithmetic.
  double* kind = Xw;                 // [n]: 0 real, 1 first of a pair, 2 second of a pair   (Xw is free after the eigen stage)
  double* slot = Xw + 16;            // [n][kModel + 1]: valid flag | model
  if (tl == 0)
    for (int j = 0; j < n; ++j) {
      const bool real = fabs(wi[j]) <= fabs(wr[j]) * 1e-12 || j + 1 == n;
      kind[j] = real ? 0.0 : 1.0;
      if (!real) { kind[j + 1] = 2.0; ++j; }
    }
  rsc::team_sync();
  const double* R0 = w + kWsR0; const double* t0 = w + kWsT0; const double* Nn = w + kWsN; const double* Dm = w + kWsD;
  const double scale = w[kWsScale], f0 = w[kWsF0], k0 = w[kWsK0];
  const int rows[4] = {kRowA1, kRowA2, kRowK, kRowW};
  for (int k = tl; k < n; k += kTeam) {
    double* out = slot + (kModel + 1) * k;
    out[0] = 0.0;
    const int kd = (int)kind[k];
    const bool real = kd == 0;
    const int j = kd == 2 ? k - 1 : k;             // first column of the pair (or the real column)
    const double sg = kd == 2 ? -1.0 : 1.0;        // the conjugate member
    double nrm2 = 0.0;
    for (int i = 0; i < n; ++i) nrm2 += real ? V[n * i + j] * V[n * i + j] : V[n * i + j] * V[n * i + j] + V[n * i + j + 1] * V[n * i + j + 1];
    const double nrm = sqrt(nrm2);
    const double v0r = V[j] / nrm, v0i = real ? 0.0 : sg * V[j + 1] / nrm;
    double re[4], im[4];
    for (int q = 0; q < 4; ++q) {
      const double xr = V[n * rows[q] + j] / nrm, xi = real ? 0.0 : sg * V[n * rows[q] + j + 1] / nrm;
      if (real) { re[q] = xr / v0r; im[q] = 0.0; }
      else rsc::eig_cdiv(xr, xi, v0r, v0i, &re[q], &im[q]);
    }
    if (im[0] < -1e-6 || im[0] > 1e-6) continue;
    out[0] = 2.0;   // one of the solver's valid_solutions (the function's return value counts these, :287); 1.0 below = kept
    // four_point_focal_length_radial_distortion.cc:219-285
    const double kk = re[2], P33 = re[3];
    const double alpha[4] = {re[0], re[1], wr[j], 1.0};
    double P[12];
    for (int r = 0; r < 8; ++r) {
      double a = 0.0;
      for (int c = 0; c < 4; ++c) a += Nn[r * 4 + c] * alpha[c];
      P[r] = a;
    }
    const double tmp[9] = {alpha[0], alpha[1], alpha[2], kk * alpha[0], kk * alpha[1], kk * alpha[2], kk, P33, 1.0};
    double p3[3];
    for (int r = 0; r < 3; ++r) {
      double a = 0.0;
      for (int c = 0; c < 9; ++c) a += Dm[r * 9 + c] * tmp[c];
      p3[r] = a;
    }
    P[8] = p3[0]; P[9] = p3[1]; P[10] = P33; P[11] = p3[2];
    const double n3 = sqrt((P[8] * P[8] + P[9] * P[9]) + P[10] * P[10]);
    for (int i = 0; i < 12; ++i) P[i] /= n3;
    const double f = sqrt((P[0] * P[0] + P[1] * P[1]) + P[2] * P[2]);
    const double focal = f * f0;
    if (focal < min_f || focal > max_f) continue;
    const double rd = kk / k0;
    if (rd < max_d || rd > min_d || rd > 0.0) continue;
    double Rt[12];
    const double kf = 1.0 / f;
    for (int c = 0; c < 4; ++c) { Rt[c] = kf * P[c]; Rt[4 + c] = kf * P[4 + c]; Rt[8 + c] = 1.0 * P[8 + c]; }
    const double R3[9] = {Rt[0], Rt[1], Rt[2], Rt[4], Rt[5], Rt[6], Rt[8], Rt[9], Rt[10]};
    if (rsc::det3(R3) < 0.0) for (int i = 0; i < 12; ++i) Rt[i] *= -1.0;
    double RR0[9];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) RR0[r * 3 + c] = (Rt[4 * r] * R0[c] + Rt[4 * r + 1] * R0[3 + c]) + Rt[4 * r + 2] * R0[6 + c];
    double* m = out + 1;
    for (int r = 0; r < 3; ++r) m[9 + r] = Rt[4 * r + 3] * scale - ((RR0[3 * r] * t0[0] + RR0[3 * r + 1] * t0[1]) + RR0[3 * r + 2] * t0[2]);
    for (int i = 0; i < 9; ++i) m[i] = RR0[i];
    m[12] = focal; m[13] = rd;
    out[0] = 1.0;
  }
  rsc::team_sync();
  if (tl != 0) return;
  int nm = 0, nsolver = 0;
  for (int k = 0; k < n; ++k) { nm += slot[(kModel + 1) * k] == 1.0; nsolver += slot[(kModel + 1) * k] != 0.0; }
  counts[hyp] = nm;
  if (solver_counts) solver_counts[hyp] = nsolver;
  if (nm == 0) return;
  const int base = atomicAdd(&dense_count[p], nm);
  hyp_base[hyp] = base;
  double* mo = models + ((size_t)p * B * kMaxModels + base) * (size_t)THEIA_RANSAC_MODEL_STRIDE;
  int* tg = tags + (size_t)p * B * kMaxModels + base;
  int jm = 0;
  for (int k = 0; k < n; ++k) {
    const double* sl = slot + (kModel + 1) * k;
    if (sl[0] != 1.0) continue;
    double* m = mo + (size_t)jm * THEIA_RANSAC_MODEL_STRIDE;
    for (int q = 0; q < kModel; ++q) m[q] = sl[1 + q];
    for (int q = kModel; q < THEIA_RANSAC_MODEL_STRIDE; ++q) m[q] = 0.0;
    tg[jm] = b * kMaxModels + jm;
    ++jm;
  }
}

// the gather lists of the run-time polynomial arithmetic, in the loop order of oracle/p4pfr_oracle.h build_template
bool build_tables(Tables* t) {
  int pair_index[6][6];
  int np = 0;
  for (int i = 0; i < 6; ++i) for (int j = i; j < 6; ++j) { t->qq_i[np] = (uint8_t)(i < 3 ? i : i + 1); t->qq_j[np] = (uint8_t)(j < 3 ? j : j + 1); pair_index[i][j] = np++; }
  int nt = 0;
  for (int pr = 0; pr < 21; ++pr)
    for (int u = 0; u < 10; ++u) {
      t->qq_off[pr * 10 + u] = (uint16_t)nt;
      for (int s = 0; s < 4; ++s) for (int tt = 0; tt < 4; ++tt) if (kMulAA[s][tt] == u) { t->qq_s[nt] = (uint8_t)s; t->qq_t[nt] = (uint8_t)tt; ++nt; }
    }
  t->qq_off[210] = (uint16_t)nt;
  if (nt != 336) return false;
  struct Term { int out, a, b, c; };
  std::vector<Term> terms;
  for (int i = 0; i < 3; ++i)
    for (int s = 0; s < 4; ++s)
      for (int tt = 0; tt < 9; ++tt) {
        if (i == 2 && tt != 7) continue;
        terms.push_back({0 * kCols + kMulATmp[s][tt], kPoolN + (4 + i) * 4 + s, kPoolP3 + i * 9 + tt, 1});
        terms.push_back({1 * kCols + kMulATmp[s][tt], kPoolN + i * 4 + s, kPoolP3 + i * 9 + tt, 1});
      }
  for (int i = 0; i < 3; ++i)
    for (int u = 0; u < 10; ++u) {
      terms.push_back({2 * kCols + kA2Col[u], kPoolQQ + pair_index[i][3 + i] * 10 + u, kPoolOne, 1});
      terms.push_back({3 * kCols + kA2Col[u], kPoolQQ + pair_index[i][i] * 10 + u, kPoolOne, 1});
    }
  for (int i = 0; i < 3; ++i) for (int u = 0; u < 10; ++u) terms.push_back({3 * kCols + kA2Col[u], kPoolQQ + pair_index[3 + i][3 + i] * 10 + u, kPoolOne, -1});
  for (int f = 0; f < 5; ++f)
    for (int n = 0; n < kCubicTerms[f]; ++n) {
      const int i = kCubic[f][n][0], j = kCubic[f][n][1], l = kCubic[f][n][2], c = kCubic[f][n][3];
      for (int u = 0; u < 10; ++u)
        for (int tt = 0; tt < 9; ++tt) {
          if (l == 2 && tt != 7) continue;
          terms.push_back({(4 + f) * kCols + kMulA2Tmp[u][tt], kPoolQQ + pair_index[i][j] * 10 + u, kPoolP3 + l * 9 + tt, c});
        }
    }
  if ((int)terms.size() > kMaxTerms) return false;
  nt = 0;
  for (int o = 0; o < kEqOut; ++o) {   // stable: the terms of one output keep the loop order
    t->eq_off[o] = (uint16_t)nt;
    for (const Term& m : terms) if (m.out == o) { t->ta[nt] = (uint16_t)m.a; t->tb[nt] = (uint16_t)m.b; t->tc[nt] = (int8_t)m.c; ++nt; }
  }
  t->eq_off[kEqOut] = (uint16_t)nt;
  return true;
}

}  // namespace p4pfrdev

int p4pfr_ensure_tables() {
  static std::once_flag once;
  static int rc = 0;
  std::call_once(once, [] {
    static p4pfrdev::Tables t;
    if (!p4pfrdev::build_tables(&t)) { rc = set_error(THEIA_HIP_ERR_INTERNAL, "P4Pfr term tables do not fit"); return; }
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(p4pfrdev::g_tab), &t, sizeof(t));
    if (e != hipSuccess) rc = set_error(THEIA_HIP_ERR_NO_DEVICE, "hipMemcpyToSymbol(P4Pfr tables) failed: %s", hipGetErrorString(e));
  });
  return rc;
}
int p4pfr_workspace_doubles() { return p4pfrdev::kWs; }

// Eigen::AngleAxisd(|v|, v).toRotationMatrix() for the raw (not normalised) vector v, as the reference forms it (:137-141); host
// side, so that sin / cos are the host libm's -- the reference's and the oracle's.  Row-major.
void p4pfr_rotation_from_draws(const double* v, double* R) {
  const double angle = std::sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
  // sincos(), not sin() and cos(): a GCC build of the reference merges Eigen's two calls into one sincos, and glibc's sincos differs
  // from its sin / cos by an ulp for ~0.1 % of the arguments (the oracle takes the same call)
  double s, c;
  sincos(angle, &s, &c);
  const double sa[3] = {s * v[0], s * v[1], s * v[2]};
  const double ca[3] = {(1.0 - c) * v[0], (1.0 - c) * v[1], (1.0 - c) * v[2]};
  double t;
  t = ca[0] * v[1]; R[1] = t - sa[2]; R[3] = t + sa[2];
  t = ca[0] * v[2]; R[2] = t + sa[1]; R[6] = t - sa[1];
  t = ca[1] * v[2]; R[5] = t - sa[0]; R[7] = t + sa[0];
  R[0] = ca[0] * v[0] + c; R[4] = ca[1] * v[1] + c; R[8] = ca[2] * v[2] + c;
}

void launch_p4pfr_fit(int nprob, int B, const int64_t* offsets, const double* data, const int* samples, const int* active_iters,
                      const double* rot, const double* limits, double* ws, double* models, int* counts, int* dense_count, int* tags,
                      int* hyp_base, hipStream_t st, int* solver_counts) {
  using namespace p4pfrdev;
  const size_t nh = (size_t)nprob * B;
  k_p4pfr_pre<<<dim3((B + 63) / 64, nprob), 64, 0, st>>>(nprob, B, offsets, data, samples, active_iters, rot, ws);
  k_p4pfr_a<<<dim3(B, nprob), 64, 0, st>>>(B, active_iters, ws);
  k_p4pfr_b<<<(unsigned)((nh + kTeamsPerWave - 1) / kTeamsPerWave), 64, 0, st>>>(nh, B, active_iters, ws, limits[0], limits[1], limits[2], limits[3],
                                                                               models, counts, dense_count, tags, hyp_base, solver_counts);
}

}  // namespace thip
