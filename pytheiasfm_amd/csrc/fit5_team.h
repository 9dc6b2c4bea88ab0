// Steps 1-3 of the five-point solver (five_point_pre, ransac_device.h; five_point_relative_pose.cc:228-273) by a TEAM of
// lanes with the work arrays in LDS: the same arithmetic per entry and the same order inside every sum as the one-thread
// routine, so the null space and the action matrix come out bit-identical -- but the 5 x 9 system, the nine E E^T
// polynomials, the 10 x 20 constraint matrix and its elimination (5 KB per hypothesis as per-lane scratch, which made the
// one-thread kernel HBM-bound on scratch traffic) stay on chip.
//
// The team is a contiguous, TEAM-aligned group of lanes of ONE wave (eig_team.h conventions: team_sync() only orders the
// LDS accesses; every lane keeps the same scalar state).  Work is dealt out by matrix column (pivot search, rank-1 updates,
// substitutions) or by polynomial (one E E^T entry / one constraint row per lane).
#ifndef THEIA_HIP_FIT5_TEAM_H_
#define THEIA_HIP_FIT5_TEAM_H_

#include "eig_team.h"
#include "wave_reduce.h"

namespace thip {
namespace rsc {

constexpr int kFit5TeamLds = 200 + 36 + 90 + 10 + 16;   // C (10 x 20; the 5 x 9 system first) | N | eet | trace | rowt, colt (ints)

// Full-pivot LU (fullpiv_lu above, Eigen::FullPivLU::compute) of the leading rows x cols block of a row-major LDS matrix
// with row stride ld.  Columns cols .. wide - 1 are right-hand sides: they take the row swaps and the elimination (= the
// forward substitution of the one-thread routine, the same subtractions in the same order), not the pivot search.
// rowt / colt: the transpositions (LDS ints).  Returns nonzero_pivots; *maxpivot as FullPivLU::maxPivot().
template <int TEAM>
__device__ int fullpiv_lu_team(double* __restrict__ A, int ld, int rows, int cols, int wide, int* __restrict__ rowt,
                               int* __restrict__ colt, double* maxpivot, int tl) {
  const int size = rows < cols ? rows : cols;
  int nonzero = size;
  double maxp = 0.0;
  for (int k = 0; k < size; ++k) {
    // pivot: the first strict maximum of the remaining corner scanned column by column
    double best = -1.0; int br = k, bc = k;
    for (int j = k + tl; j < cols; j += TEAM)
      for (int i = k; i < rows; ++i) {
        const double a = fabs(A[i * ld + j]);
        if (a > best) { best = a; br = i; bc = j; }
      }
    if constexpr (TEAM == 16) {
      // a team of 16 is one row of the DPP network: the maximum of |a| as two unsigned 32-bit all-reductions (wave_reduce.h),
      // then -- the lanes hold disjoint columns, so (value, column) is a total order and any reduction tree returns the
      // butterfly's answer -- the lowest column among the holders and that lane's row, two more
      const bool valid = best >= 0.0;
      const unsigned hi = valid ? (unsigned)__double2hiint(best) : 0u, lo = (unsigned)__double2loint(best);
      const unsigned hm = row16_max_u32(hi);
      const bool top = valid & (hi == hm);
      const unsigned lm = row16_max_u32(top ? lo : 0u);
      const bool holder = top & (lo == lm);
      const unsigned gbc = row16_min_u32(holder ? (unsigned)bc : 0xffffffffu);
      const unsigned gbr = row16_min_u32((holder & ((unsigned)bc == gbc)) ? (unsigned)br : 0xffffffffu);
      if (gbc != 0xffffffffu) { best = __hiloint2double((int)hm, (int)lm); br = (int)gbr; bc = (int)gbc; }
    } else {
    for (int o = TEAM / 2; o >= 1; o >>= 1) {
      const double ob = __shfl_xor(best, o, TEAM);
      const int obr = __shfl_xor(br, o, TEAM), obc = __shfl_xor(bc, o, TEAM);
      if (ob > best || (ob == best && obc < bc)) { best = ob; br = obr; bc = obc; }
    }
    }
    if (best == 0.0) {
      nonzero = k;
      if (tl == 0) for (int i = k; i < size; ++i) { rowt[i] = i; colt[i] = i; }
      break;
    }
    if (best > maxp) maxp = best;
    if (tl == 0) { rowt[k] = br; colt[k] = bc; }
    if (br != k)
      for (int j = tl; j < wide; j += TEAM) { const double t = A[k * ld + j]; A[k * ld + j] = A[br * ld + j]; A[br * ld + j] = t; }
    team_sync();
    if (bc != k)
      for (int i = tl; i < rows; i += TEAM) { const double t = A[i * ld + k]; A[i * ld + k] = A[i * ld + bc]; A[i * ld + bc] = t; }
    team_sync();
    if (k < rows - 1) {
      const double d = A[k * ld + k];
      team_sync();
      for (int i = k + 1 + tl; i < rows; i += TEAM) A[i * ld + k] /= d;
      team_sync();
      for (int j = k + 1 + tl; j < wide; j += TEAM) {
        const double u = A[k * ld + j];
        for (int i = k + 1; i < rows; ++i) A[i * ld + j] -= A[i * ld + k] * u;
      }
      team_sync();
    }
  }
  team_sync();
  *maxpivot = maxp;
  return nonzero;
}

// corr: the five correspondences [x1 y1 x2 y2] (global).  W: kFit5TeamLds doubles of LDS.  N_out (36) / M_out (100): global.
template <int TEAM>
__device__ bool five_point_pre_team(const double* __restrict__ pd, const int* __restrict__ sample, double* __restrict__ W,
                                    double* __restrict__ N_out, double* __restrict__ M_out, int tl) {
  static_assert(TEAM >= 10, "one lane per constraint row");
  double* C = W; double* N = C + 200; double* eet = N + 36; double* trace = eet + 90;
  int* rowt = reinterpret_cast<int*>(trace + 10); int* colt = rowt + 10;
  // Step 1: the 5 x 9 epipolar constraint rows, one per lane
  double* A = C;
  if (tl < 5) {
    const double* c = pd + (size_t)sample[tl] * 4;
    const double x1 = c[0], y1 = c[1], x2 = c[2], y2 = c[3];
    double* r = A + 9 * tl;
    r[0] = x2 * x1; r[1] = y2 * x1; r[2] = x1; r[3] = x2 * y1; r[4] = y2 * y1; r[5] = y1; r[6] = x2; r[7] = y2; r[8] = 1.0;
  }
  team_sync();
  double maxpivot;
  const int nzp = fullpiv_lu_team<TEAM>(A, 9, 5, 9, 9, rowt, colt, &maxpivot, tl);
  const double premult = fabs(maxpivot) * (DBL_EPSILON * 5.0);
  int rank = 0;
  for (int i = 0; i < nzp; ++i) rank += fabs(A[i * 9 + i]) > premult;
  if (9 - rank != 4) return false;
  int cidx[9];
#pragma unroll
  for (int j = 0; j < 9; ++j) cidx[j] = j;
  // the transpositions k <-> colt[k], k = 0 .. 4, on a register array: compare-and-select instead of dynamic indexing
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const int t = colt[k];
    int vt = 0;
#pragma unroll
    for (int j = 0; j < 9; ++j) if (j == t) vt = cidx[j];
    const int vk = cidx[k];
#pragma unroll
    for (int j = 0; j < 9; ++j) if (j == t) cidx[j] = vk;
    cidx[k] = vt;
  }
  for (int e = tl; e < 36; e += TEAM) N[e] = 0.0;
  team_sync();
  if (tl < 4) {   // U1 X = U2, one right-hand side per lane
    double X[5];
#pragma unroll
    for (int i = 4; i >= 0; --i) {
      double s = A[i * 9 + 5 + tl];
#pragma unroll
      for (int j = i + 1; j < 5; ++j) s -= A[i * 9 + j] * X[j];
      X[i] = s / A[i * 9 + i];
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) N[cidx[i] * 4 + tl] = -X[i];
    int cfree = 0;
#pragma unroll
    for (int j = 5; j < 9; ++j) if (j == 5 + tl) cfree = cidx[j];
    N[cfree * 4 + tl] = 1.0;
  }
  team_sync();
  for (int e = tl; e < 36; e += TEAM) N_out[e] = N[e];
  // Step 2: the constraint matrix.  ns(i, j) = row 3 j + i of N
#define NS(i, j) (N + 4 * (3 * (j) + (i)))
  if (tl < 9) {
    const int i = tl / 3, j = tl % 3;
    double t0[10], t1[10], t2[10];
    mul_deg1(NS(i, 0), NS(j, 0), t0); mul_deg1(NS(i, 1), NS(j, 1), t1); mul_deg1(NS(i, 2), NS(j, 2), t2);
#pragma unroll
    for (int k = 0; k < 10; ++k) eet[tl * 10 + k] = 2 * ((t0[k] + t1[k]) + t2[k]);
  }
  team_sync();
  if (tl < 10) trace[tl] = (eet[tl] + eet[40 + tl]) + eet[80 + tl];
  team_sync();   // (C aliases the 5 x 9 system: every lane is done with it)
  if (tl < 9) {
    const int i = tl / 3, j = tl % 3;
    // ((a + b) + c) - 0.5 d, one product live at a time (four at once are 160 VGPRs)
    double acc[20], t[20];
    mul_deg2_deg1(eet + 10 * (3 * i), NS(0, j), acc);
    mul_deg2_deg1(eet + 10 * (3 * i + 1), NS(1, j), t);
#pragma unroll
    for (int k = 0; k < 20; ++k) acc[k] = acc[k] + t[k];
    mul_deg2_deg1(eet + 10 * (3 * i + 2), NS(2, j), t);
#pragma unroll
    for (int k = 0; k < 20; ++k) acc[k] = acc[k] + t[k];
    mul_deg2_deg1(trace, NS(i, j), t);
    double* row = C + 20 * tl;
#pragma unroll
    for (int k = 0; k < 20; ++k) row[k] = acc[k] - 0.5 * t[k];
  } else if (tl == 9) {
    double p0[10], p1[10], q[10], acc[20], t[20];   // (d0 + d1) + d2
    mul_deg1(NS(0, 1), NS(1, 2), p0); mul_deg1(NS(0, 2), NS(1, 1), p1);
#pragma unroll
    for (int k = 0; k < 10; ++k) q[k] = p0[k] - p1[k];
    mul_deg2_deg1(q, NS(2, 0), acc);
    mul_deg1(NS(0, 2), NS(1, 0), p0); mul_deg1(NS(0, 0), NS(1, 2), p1);
#pragma unroll
    for (int k = 0; k < 10; ++k) q[k] = p0[k] - p1[k];
    mul_deg2_deg1(q, NS(2, 1), t);
#pragma unroll
    for (int k = 0; k < 20; ++k) acc[k] = acc[k] + t[k];
    mul_deg1(NS(0, 0), NS(1, 1), p0); mul_deg1(NS(0, 1), NS(1, 0), p1);
#pragma unroll
    for (int k = 0; k < 10; ++k) q[k] = p0[k] - p1[k];
    mul_deg2_deg1(q, NS(2, 2), t);
#pragma unroll
    for (int k = 0; k < 20; ++k) C[180 + k] = acc[k] + t[k];
  }
#undef NS
  team_sync();
  // Step 3: eliminated = lu(C[:, :10]).solve(C[:, 10:]): the right half rides the factorisation (row swaps + forward
  // substitution), then one lane per right-hand side substitutes back
  fullpiv_lu_team<TEAM>(C, 20, 10, 10, 20, rowt, colt, &maxpivot, tl);
  for (int e = tl; e < 100; e += TEAM) M_out[e] = 0.0;
  team_sync();
  if (tl < 10) {
    double Bc[10];
#pragma unroll
    for (int i = 9; i >= 0; --i) {
      double s = C[i * 20 + 10 + tl];
#pragma unroll
      for (int j = i + 1; j < 10; ++j) s -= C[i * 20 + j] * Bc[j];
      Bc[i] = s / C[i * 20 + i];
    }
    int gidx[10];
#pragma unroll
    for (int j = 0; j < 10; ++j) gidx[j] = j;
#pragma unroll
    for (int k = 0; k < 10; ++k) {
      const int t = colt[k];
      int vt = 0;
#pragma unroll
      for (int j = 0; j < 10; ++j) if (j == t) vt = gidx[j];
      const int vk = gidx[k];
#pragma unroll
      for (int j = 0; j < 10; ++j) if (j == t) gidx[j] = vk;
      gidx[k] = vt;
    }
    // action matrix: rows 0 .. 5 = rows {0, 1, 2, 4, 5, 7} of the eliminated matrix (row gidx[i] = substituted row i)
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      const int g = gidx[i];
      const int r = (g <= 2) ? g : (g == 4 ? 3 : (g == 5 ? 4 : (g == 7 ? 5 : -1)));
      if (r >= 0) M_out[r * 10 + tl] = Bc[i];
    }
  }
  team_sync();
  if (tl == 0) { M_out[6 * 10 + 0] = -1.0; M_out[7 * 10 + 1] = -1.0; M_out[8 * 10 + 3] = -1.0; M_out[9 * 10 + 6] = -1.0; }
  return true;
}

}  // namespace rsc
}  // namespace thip
#endif
