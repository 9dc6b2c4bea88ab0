// Two-sided Jacobi SVD of a 9 x 9 matrix by a TEAM of 9 lanes with the work matrices in LDS: the algorithm, the rotation
// order and the arithmetic per matrix entry of the one-thread svd_sq<9> (ransac_device.h: Eigen::JacobiSVD for square
// real input, as sqpnp.cc:234 runs it), so U and S come out with the same bits -- but a rotation updates its 2 x 9 row
// and column entries on nine lanes at once and the 81-double arrays live on chip instead of in per-lane scratch (the
// SVD was 77 % of the SQPnP RANSAC fit kernel).  V is not accumulated: SQPnP never reads it and the iteration does not
// depend on it.
//
// Every lane of the team keeps the same scalar state (rotation parameters are computed redundantly from LDS
// broadcasts); lane tl owns column tl of the row phase and row tl of the column phase.  Teams are contiguous,
// 9-aligned groups of lanes of ONE wave; team_sync() (eig_team.h) orders the LDS accesses between the phases.  Teams of
// one wave may diverge (different sweep counts, skipped rotations).
#ifndef THEIA_HIP_SVD_TEAM_H_
#define THEIA_HIP_SVD_TEAM_H_

#include "eig_team.h"

namespace thip {
namespace rsc {

// A: 81 doubles (row-major, global or LDS), W / U: 81 doubles of LDS each, S: 9 doubles of LDS.  tl in [0, 9).
// V (optional, 81 doubles of LDS): the right singular vectors, accumulated and sorted as svd_sq<9> does.
// TEAM (round 6): lanes per matrix.  Lane tl owns the entries e = tl, tl + TEAM, .. < 9 of every row / column phase (TEAM = 9:
// one each, the original; TEAM = 5: two, so that a wave holds twelve matrices and the rotation parameters -- seven divisions
// and three square roots that EVERY lane of a team computes -- are shared by twelve instead of seven).
template <int TEAM = 9>
__device__ inline void svd9_team(const double* __restrict__ A, double* __restrict__ W, double* __restrict__ U,
                                 double* __restrict__ S, int tl, double* __restrict__ V = nullptr) {
  constexpr int N = 9;
  for (int e = tl; e < N; e += TEAM)
    for (int i = 0; i < N; ++i) {   // column e
      W[i * N + e] = A[i * N + e];
      U[i * N + e] = (i == e) ? 1.0 : 0.0;
      if (V) V[i * N + e] = (i == e) ? 1.0 : 0.0;
    }
  team_sync();
  double scale = 0.0;
  for (int i = 0; i < N * N; ++i) scale = fmax(scale, fabs(W[i]));
  if (scale == 0.0) scale = 1.0;
  team_sync();
  for (int e = tl; e < N; e += TEAM)
    for (int i = 0; i < N; ++i) W[i * N + e] = W[i * N + e] / scale;
  team_sync();
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
        const double m01 = W[p * N + q], m10 = W[q * N + p];
        if (fabs(m01) > threshold || fabs(m10) > threshold) {
          finished = false;
          const double m00 = W[p * N + p], m11 = W[q * N + q];
          double r1c, r1s;
          const double tt = m00 + m11, dd = m10 - m01;
          if (fabs(dd) < DBL_MIN) { r1c = 1.0; r1s = 0.0; }
          else { const double u = tt / dd; const double tmp = sqrt(1.0 + u * u); r1s = 1.0 / tmp; r1c = u / tmp; }
          const double n00 = r1c * m00 + r1s * m10, n01 = r1c * m01 + r1s * m11;
          const double n11 = -r1s * m01 + r1c * m11;
          double jc, js;
          jacobi_rot_sym(n00, n01, n11, &jc, &js);
          const double lc = r1c * jc + r1s * js, ls = r1s * jc - r1c * js;
          team_sync();   // every lane has read the 2 x 2 block before the rows change
          for (int e = tl; e < N; e += TEAM) {   // rows p, q of W (entry e) and columns p, q of U (entry e)
            const double a = W[p * N + e], b = W[q * N + e];
            W[p * N + e] = lc * a + ls * b; W[q * N + e] = -ls * a + lc * b;
            const double ua = U[e * N + p], ub = U[e * N + q];
            U[e * N + p] = lc * ua + ls * ub; U[e * N + q] = -ls * ua + lc * ub;
          }
          team_sync();
          for (int e = tl; e < N; e += TEAM) {   // columns p, q of W (entry e)
            const double a = W[e * N + p], b = W[e * N + q];
            W[e * N + p] = jc * a - js * b; W[e * N + q] = js * a + jc * b;
            if (V) {
              const double va = V[e * N + p], vb = V[e * N + q];
              V[e * N + p] = jc * va - js * vb; V[e * N + q] = js * va + jc * vb;
            }
          }
          team_sync();
          maxdiag = fmax(maxdiag, fmax(fabs(W[p * N + p]), fabs(W[q * N + q])));
        }
      }
  }
  for (int e = tl; e < N; e += TEAM) {   // singular values, sign into U: column e
    const double a = W[e * N + e];
    S[e] = fabs(a);
    if (a < 0) for (int k = 0; k < N; ++k) U[k * N + e] = -U[k * N + e];
  }
  team_sync();
  for (int i = 0; i < N; ++i) {   // selection sort, descending (first maximum wins): lane tl swaps row tl of U's columns
    int best = i;
    for (int j = i + 1; j < N; ++j) if (S[j] > S[best]) best = j;
    team_sync();   // every lane has chosen before S changes
    if (best != i) {
      if (tl == 0) { const double t = S[i]; S[i] = S[best]; S[best] = t; }
      for (int e = tl; e < N; e += TEAM) {
        const double t = U[e * N + i]; U[e * N + i] = U[e * N + best]; U[e * N + best] = t;
        if (V) { const double v = V[e * N + i]; V[e * N + i] = V[e * N + best]; V[e * N + best] = v; }
      }
    }
    team_sync();
  }
  for (int e = tl; e < N; e += TEAM) S[e] *= scale;
  team_sync();
}

}  // namespace rsc
}  // namespace thip
#endif
