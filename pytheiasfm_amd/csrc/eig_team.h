// Real general eigen-decomposition by a TEAM of lanes with the matrices in LDS: the same algorithm, the same
// arithmetic per matrix entry and the same summation order as the one-thread eig_general_t (ransac_device.h: EISPACK
// orthes + hqr2 as Eigen's EigenSolver runs them), so the results are bit-identical to it -- but the n x n work arrays
// live on chip instead of in per-lane scratch, where the QR sweeps of 64 independent matrices per wave moved
// ~200 KB of HBM traffic per 10 x 10 matrix (measured on k_fit: FETCH + WRITE = 3.2 TB/s).
//
// Every lane of the team calls eig_team() with the same arguments and keeps the same scalar state; loops over a row or
// a column of the matrix are dealt out by team lane (tl), dot products stay sequential inside one lane.  The team is a
// contiguous, TEAM-aligned group of lanes of ONE wave (lockstep): team_sync() only has to stop the compiler from
// moving LDS accesses across it.  Teams of the same wave may diverge from each other (different deflation
// histories); nothing here uses wave-wide collectives.
//   H  n x n row-major, in: the matrix, out: quasi-triangular Schur form (destroyed)
//   V  n x n, out: eigenvectors (hqr2 column convention; complex pairs only with CPLX)
//   X  n x n work array (the back-substituted vectors before the back-transformation)
//   wr, wi, ort: n doubles each
#ifndef THEIA_HIP_EIG_TEAM_H_
#define THEIA_HIP_EIG_TEAM_H_

#include <hip/hip_runtime.h>

#include "ransac_device.h"

namespace thip {
namespace rsc {

__device__ __forceinline__ void team_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// Development (-DTHIP_EIG_STAMPS): s_memtime ticks per phase of eig_team, summed over the teams' lane 0 into g_eig_stamps
// {orthes, accumulate, hqr2 deflation scans + root branches, shift + start search, chase steps, back-substitution,
// back-transformation, calls}; read back with theia_hip_debug_eig_stamps (ransac.hip).
#ifdef THIP_EIG_STAMPS
__device__ unsigned long long g_eig_stamps[8];
#define EIG_STAMP_DECL unsigned long long st_t = __builtin_amdgcn_s_memtime(), st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define EIG_STAMP(k) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); st_acc[k] += n_ - st_t; st_t = n_; } while (0)
#define EIG_STAMP_FLUSH do { if (tl == 0) { for (int k_ = 0; k_ < 7; ++k_) atomicAdd(&g_eig_stamps[k_], st_acc[k_]); atomicAdd(&g_eig_stamps[7], 1ull); } } while (0)
#else
#define EIG_STAMP_DECL do {} while (0)
#define EIG_STAMP(k) do {} while (0)
#define EIG_STAMP_FLUSH do {} while (0)
#endif

// lane LANE of every row of 16 lanes, to the whole row (DPP row_newbcast: two v_mov_b32_dpp)
template <int LANE>
__device__ __forceinline__ double row16_bcast(double v) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x150 + LANE, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x150 + LANE, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

// lane LANE (0 .. 4) of every aligned group of 8 lanes, to the whole group: inside the quads by quad_perm, then the other
// quad of the group takes it over a row shift by four lanes under a bank mask (4 v_mov_b32_dpp per double)
template <int LANE>
__device__ __forceinline__ int oct_bcast_i(int v) {
  static_assert(LANE >= 0 && LANE <= 4, "lanes 0 .. 4");
  constexpr int L4 = LANE & 3;
  const int t = __builtin_amdgcn_mov_dpp(v, L4 * 0x55, 0xf, 0xf, false);          // quad_perm [L4, L4, L4, L4]
  if (LANE < 4) return __builtin_amdgcn_update_dpp(t, t, 0x114, 0xf, 0xA, false);  // row_shr:4 into the upper quads
  return __builtin_amdgcn_update_dpp(t, t, 0x104, 0xf, 0x5, false);                // row_shl:4 into the lower quads
}
template <int LANE>
__device__ __forceinline__ double oct_bcast(double v) {
  return __hiloint2double(oct_bcast_i<LANE>(__double2hiint(v)), oct_bcast_i<LANE>(__double2loint(v)));
}
// lane LANE (0 .. 4) of the team, to the team
template <int TEAM, int LANE>
__device__ __forceinline__ double team_bcast(double v) {
  if constexpr (TEAM % 16 == 0) return row16_bcast<LANE>(v); else return oct_bcast<LANE>(v);
}

// The team's slice of a wave ballot (bit i = team lane i; the team is contiguous and TEAM-aligned, TEAM <= 32)
template <int TEAM>
__device__ __forceinline__ unsigned team_ballot(bool pred, int tl) {
  const unsigned long long m = __ballot(pred);
  const int base = (int)__lane_id() - tl;
  return (unsigned)(m >> base) & (TEAM >= 32 ? 0xffffffffu : ((1u << (TEAM & 31)) - 1u));
}

//
// NR > 0 (a caller that reads only NR rows of the eigenvector matrix, e.g. DLS: rows 0, 9, 3, 1): once the Householder
// reflectors are accumulated, every later operation on V combines COLUMNS of one row -- the rows never mix again -- so
// only the NR wanted rows (keep[0 .. NR - 1]) are carried through hqr2 and the back-transformation, in Vk (NR x n, LDS).
// The full V of the accumulation lives in V, which may then be slow memory (the work array X is not in use yet: the two
// can share storage).  Same arithmetic per entry, so the kept rows equal those of the full computation bit for bit.
template <int TEAM, bool CPLX, int NR = 0>
__device__ __forceinline__ bool eig_team(int nn, double* __restrict__ H, double* V, double* X,
                         double* __restrict__ wr, double* __restrict__ wi, double* __restrict__ ort, int tl,
                         double* __restrict__ Vk = nullptr, const int* __restrict__ keep = nullptr) {
#define HH(i, j) H[(i) * nn + (j)]
#define VV(i, j) V[(i) * nn + (j)]
#define XX(i, j) X[(i) * nn + (j)]
  // rows of the eigenvector matrix after the accumulation: all of V, or the kept rows in Vk
  const int vrows = NR > 0 ? NR : nn;
  double* const VR = NR > 0 ? Vk : V;
#define VRR(i, j) VR[(i) * nn + (j)]
  const int low = 0, high = nn - 1;
  EIG_STAMP_DECL;
  // ---- orthes
  for (int m = low + 1; m <= high - 1; ++m) {
    double scale = 0.0;
    for (int i = m; i <= high; ++i) scale += fabs(HH(i, m - 1));
    if (scale != 0.0) {
      for (int i = m + tl; i <= high; i += TEAM) ort[i] = HH(i, m - 1) / scale;
      team_sync();
      double h = 0.0;
      for (int i = high; i >= m; --i) h += ort[i] * ort[i];
      double g = sqrt(h);
      const double om = ort[m];
      if (om > 0) g = -g;
      h = h - om * g;
      team_sync();
      if (tl == 0) ort[m] = om - g;
      team_sync();
      for (int j = m + tl; j < nn; j += TEAM) {
        double f = 0.0;
        for (int i = high; i >= m; --i) f += ort[i] * HH(i, j);
        f = f / h;
        for (int i = m; i <= high; ++i) HH(i, j) -= f * ort[i];
      }
      team_sync();
      for (int i = tl; i <= high; i += TEAM) {
        double f = 0.0;
        for (int j = high; j >= m; --j) f += ort[j] * HH(i, j);
        f = f / h;
        for (int j = m; j <= high; ++j) HH(i, j) -= f * ort[j];
      }
      team_sync();
      if (tl == 0) { ort[m] = scale * ort[m]; HH(m, m - 1) = scale * g; }
      team_sync();
    } else {
      if (tl == 0) ort[m] = 0.0;
      team_sync();
    }
  }
  EIG_STAMP(0);
  for (int e = tl; e < nn * nn; e += TEAM) V[e] = (e / nn == e % nn) ? 1.0 : 0.0;
  team_sync();
  for (int m = high - 1; m >= low + 1; --m) {
    if (HH(m, m - 1) != 0.0) {
      for (int i = m + 1 + tl; i <= high; i += TEAM) ort[i] = HH(i, m - 1);
      team_sync();
      for (int j = m + tl; j <= high; j += TEAM) {
        double g = 0.0;
        for (int i = m; i <= high; ++i) g += ort[i] * VV(i, j);
        g = (g / ort[m]) / HH(m, m - 1);
        for (int i = m; i <= high; ++i) VV(i, j) += g * ort[i];
      }
      team_sync();
    }
  }
  if constexpr (NR > 0) {
    for (int e = tl; e < NR * nn; e += TEAM) Vk[e] = VV(keep[e / nn], e % nn);
    team_sync();
  }
  EIG_STAMP(1);
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
    // Converged roots are peeled off in an inner loop until THIS team is due for a sweep (or done): the teams of a wave then
    // meet in the sweep code every pass of the outer loop, instead of idling through the others' sweeps each time they
    // split off a root (one matrix sees the same operations in the same order either way).
    int l = low;
    bool sweep_due = false;
    while (n >= low) {
    // the deflation scan "l = n; while (l > low && |H(l, l-1)| >= eps (|H(l-1, l-1)| + |H(l, l)|)) l--": every candidate l is
    // tested by one team lane (the same arithmetic per candidate), the answer is the LARGEST small one = the first set bit of
    // the team ballot.  The sequential loop ran ~(n - l) dependent iterations in every lane of the team.
    l = low;
    for (int lb = n; lb > low; lb -= TEAM) {
      const int lc = lb - tl;
      bool small = false;
      if (lc > low) {
        double ss = fabs(HH(lc - 1, lc - 1)) + fabs(HH(lc, lc));
        if (ss == 0.0) ss = norm;
        small = fabs(HH(lc, lc - 1)) < eps * ss;
      }
      const unsigned bm = team_ballot<TEAM>(small, tl);
      if (bm) { l = lb - (__ffs((int)bm) - 1); break; }
    }
    if (l == n) {  // one root
      const double v = HH(n, n) + exshift;
      team_sync();
      if (tl == 0) { HH(n, n) = v; wr[n] = v; wi[n] = 0.0; }
      team_sync();
      n--; iter = 0;
      continue;
    }
    if (l == n - 1) {  // two roots
      w = HH(n, n - 1) * HH(n - 1, n);
      p = (HH(n - 1, n - 1) - HH(n, n)) / 2.0;
      q = p * p + w;
      z = sqrt(fabs(q));
      const double hnn = HH(n, n) + exshift, hmm = HH(n - 1, n - 1) + exshift;
      x = hnn;
      const double xl = HH(n, n - 1);
      team_sync();
      if (tl == 0) { HH(n, n) = hnn; HH(n - 1, n - 1) = hmm; }
      team_sync();
      if (q >= 0) {  // real pair
        z = (p >= 0) ? p + z : p - z;
        double w0 = x + z, w1 = w0;
        if (z != 0.0) w1 = x - w / z;
        if (tl == 0) { wr[n - 1] = w0; wr[n] = w1; wi[n - 1] = 0.0; wi[n] = 0.0; }
        x = xl;
        s = fabs(x) + fabs(z);
        p = x / s; q = z / s;
        r = sqrt(p * p + q * q);
        p = p / r; q = q / r;
        for (int j = n - 1 + tl; j < nn; j += TEAM) { const double zz = HH(n - 1, j); HH(n - 1, j) = q * zz + p * HH(n, j); HH(n, j) = q * HH(n, j) - p * zz; }
        team_sync();
        for (int i = tl; i <= n; i += TEAM) { const double zz = HH(i, n - 1); HH(i, n - 1) = q * zz + p * HH(i, n); HH(i, n) = q * HH(i, n) - p * zz; }
        for (int i = tl; i < vrows; i += TEAM) { const double zz = VRR(i, n - 1); VRR(i, n - 1) = q * zz + p * VRR(i, n); VRR(i, n) = q * VRR(i, n) - p * zz; }
        team_sync();
      } else {  // complex pair
        if (tl == 0) { wr[n - 1] = x + p; wr[n] = x + p; wi[n - 1] = z; wi[n] = -z; }
        team_sync();
      }
      n = n - 2; iter = 0;
      continue;
    }
    sweep_due = true;
    break;
    }   // peel loop
    if (!sweep_due) break;
    {
      x = HH(n, n); y = 0.0; w = 0.0;
      if (l < n) { y = HH(n - 1, n - 1); w = HH(n, n - 1) * HH(n - 1, n); }
      if (iter == 10) {  // Wilkinson's original ad hoc shift
        exshift += x;
        team_sync();
        for (int i = low + tl; i <= n; i += TEAM) HH(i, i) -= x;
        team_sync();
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
          team_sync();
          for (int i = low + tl; i <= n; i += TEAM) HH(i, i) -= s;
          team_sync();
          exshift += s;
          x = y = w = 0.964;
        }
      }
      iter = iter + 1;
      if (++total_iter > 40 * nn) return false;
      EIG_STAMP(2);
      // The search for two consecutive small subdiagonal elements walks m = n - 2 down to l and stops at the first m that is l
      // or passes the test; each step is four divisions in every lane of the team.  Here every candidate m is evaluated by ONE
      // team lane with the sequential loop's arithmetic, the answer is the largest m that stops (first set bit of the ballot),
      // and its (p, q, r) reach the team through ort[0..2] (free since the accumulation).
      int m = l;
      for (int mb = n - 2; mb >= l; mb -= TEAM) {
        const int mc = mb - tl;
        bool hit = false;
        double pc = 0.0, qc = 0.0, rc = 0.0;
        if (mc >= l) {
          const double zc = HH(mc, mc);
          const double r0 = x - zc, s0 = y - zc;
          pc = (r0 * s0 - w) / HH(mc + 1, mc) + HH(mc, mc + 1);
          qc = HH(mc + 1, mc + 1) - zc - r0 - s0;
          rc = HH(mc + 2, mc + 1);
          const double sc = fabs(pc) + fabs(qc) + fabs(rc);
          pc = pc / sc; qc = qc / sc; rc = rc / sc;
          hit = (mc == l) || (fabs(HH(mc, mc - 1)) * (fabs(qc) + fabs(rc)) <
                              eps * (fabs(pc) * (fabs(HH(mc - 1, mc - 1)) + fabs(zc) + fabs(HH(mc + 1, mc + 1)))));
        }
        const unsigned bm = team_ballot<TEAM>(hit, tl);
        if (bm) {
          const int t0 = __ffs((int)bm) - 1;
          m = mb - t0;
          if (tl == t0) { ort[0] = pc; ort[1] = qc; ort[2] = rc; }
          break;
        }
      }
      team_sync();
      p = ort[0]; q = ort[1]; r = ort[2];
      team_sync();
      for (int i = m + 2 + tl; i <= n; i += TEAM) { HH(i, i - 2) = 0.0; if (i > m + 2) HH(i, i - 3) = 0.0; }
      team_sync();
      EIG_STAMP(3);
      for (int k = m; k <= n - 1; ++k) {
        const bool notlast = (k != n - 1);
        if (k != m) {
          p = HH(k, k - 1); q = HH(k + 1, k - 1);
          r = notlast ? HH(k + 2, k - 1) : 0.0;
          x = fabs(p) + fabs(q) + fabs(r);
          if (x == 0.0) continue;
          if constexpr (TEAM % 16 == 0 || TEAM == 8) {
            // The step's eight quotients are the same in every lane of the team, and their division sequences (11 instructions
            // each) were most of a chase step's instruction stream.  They are dealt to lanes -- lane j of every DPP row (of every
            // group of 8 lanes for the teams of 8) divides ONE pair and broadcasts the quotient -- : one division + 2 (4) moves
            // per value instead of 3 (here) and 5 (below) divisions; the same operands, so the same bits.
            const int l16 = tl & (TEAM == 8 ? 7 : 15);
            const double quot = (l16 == 0 ? p : (l16 == 1 ? q : r)) / x;
            p = team_bcast<TEAM, 0>(quot); q = team_bcast<TEAM, 1>(quot); r = team_bcast<TEAM, 2>(quot);
          } else {
          p = p / x; q = q / x; r = r / x;
          }
        }
        s = sqrt(p * p + q * q + r * r);
        if (p < 0) s = -s;
        if (s != 0) {
          team_sync();   // every lane has read H(k.., k - 1) before lane 0 overwrites it
          if (tl == 0) {
            if (k != m) HH(k, k - 1) = -s * x;
            else if (l != m) HH(k, k - 1) = -HH(k, k - 1);
          }
          p = p + s;
          if constexpr (TEAM % 16 == 0 || TEAM == 8) {   // x = p / s; y = q / s; z = r / s; q = q / p; r = r / p  -- lanes 0 .. 4
            const int l16 = tl & (TEAM == 8 ? 7 : 15);
            const double num = l16 == 0 ? p : ((l16 == 1 || l16 == 3) ? q : r), den = l16 < 3 ? s : p;
            const double quot = num / den;
            x = team_bcast<TEAM, 0>(quot); y = team_bcast<TEAM, 1>(quot); z = team_bcast<TEAM, 2>(quot);
            q = team_bcast<TEAM, 3>(quot); r = team_bcast<TEAM, 4>(quot);
          } else {
          x = p / s; y = q / s; z = r / s; q = q / p; r = r / p;
          }
          team_sync();
          for (int j = k + tl; j < nn; j += TEAM) {
            double pp = HH(k, j) + q * HH(k + 1, j);
            if (notlast) { pp = pp + r * HH(k + 2, j); HH(k + 2, j) = HH(k + 2, j) - pp * z; }
            HH(k, j) = HH(k, j) - pp * x;
            HH(k + 1, j) = HH(k + 1, j) - pp * y;
          }
          team_sync();
          // column modification of H (rows 0 .. imax) and of the eigenvector rows in ONE pass over "virtual rows": the two sets
          // are disjoint and see the same operation, and a team has lanes to spare (27 + 4 rows on 32 lanes: one round)
          const int imax = (n < k + 3) ? n : k + 3;
          const int nvrow = imax + 1 + vrows;
          for (int v = tl; v < nvrow; v += TEAM) {
            double* row = (v <= imax) ? (H + v * nn) : (VR + (v - imax - 1) * nn);
            double pp = x * row[k] + y * row[k + 1];
            if (notlast) { pp = pp + z * row[k + 2]; row[k + 2] = row[k + 2] - pp * r; }
            row[k] = row[k] - pp;
            row[k + 1] = row[k + 1] - pp * q;
          }
          team_sync();
        }
      }
      EIG_STAMP(4);
    }
  }
  team_sync();
  EIG_STAMP(2);
  if (norm == 0.0) return true;
  // ---- back-substitution: one lane per eigenvalue column.  The sequential routine overwrites column n of H with the
  // vector while columns < n still hold the triangular form; here the vectors go to X so that the columns are independent
  // (column n reads T(i, j) for j <= n only through H, and its own entries through X).
  for (int e = tl; e < nn * nn; e += TEAM) X[e] = 0.0;
  team_sync();
  for (n = nn - 1 - tl; n >= 0; n -= TEAM) {
    p = wr[n]; q = wi[n];
    if (CPLX && q < 0 && n > 0) {   // second member of a pair: real part in column n - 1, imaginary part in column n
      int l = n - 1;
      double a11, a12;
      if (fabs(HH(n, n - 1)) > fabs(HH(n - 1, n))) {
        a11 = q / HH(n, n - 1);
        a12 = -(HH(n, n) - p) / HH(n, n - 1);
      } else {
        eig_cdiv(0.0, -HH(n - 1, n), HH(n - 1, n - 1) - p, q, &a11, &a12);
      }
      XX(n - 1, n - 1) = a11; XX(n - 1, n) = a12;
      XX(n, n - 1) = 0.0; XX(n, n) = 1.0;
      double lastra = 0.0, lastsa = 0.0, lastw = 0.0;
      for (int i = n - 2; i >= 0; --i) {
        double ra = 0.0, sa = 0.0;
        for (int j = l; j <= n; ++j) { ra = ra + HH(i, j) * XX(j, n - 1); sa = sa + HH(i, j) * XX(j, n); }
        w = HH(i, i) - p;
        if (wi[i] < 0.0) { lastw = w; lastra = ra; lastsa = sa; continue; }
        l = i;
        double cr, ci;
        if (wi[i] == 0.0) {
          eig_cdiv(-ra, -sa, w, q, &cr, &ci);
          XX(i, n - 1) = cr; XX(i, n) = ci;
        } else {
          x = HH(i, i + 1); y = HH(i + 1, i);
          double vr = (wr[i] - p) * (wr[i] - p) + wi[i] * wi[i] - q * q;
          const double vi = (wr[i] - p) * 2.0 * q;
          if (vr == 0.0 && vi == 0.0) vr = eps * norm * (fabs(w) + fabs(q) + fabs(x) + fabs(y) + fabs(lastw));
          eig_cdiv(x * lastra - lastw * ra + q * sa, x * lastsa - lastw * sa - q * ra, vr, vi, &cr, &ci);
          XX(i, n - 1) = cr; XX(i, n) = ci;
          if (fabs(x) > (fabs(lastw) + fabs(q))) {
            XX(i + 1, n - 1) = (-ra - w * XX(i, n - 1) + q * XX(i, n)) / x;
            XX(i + 1, n) = (-sa - w * XX(i, n) - q * XX(i, n - 1)) / x;
          } else {
            eig_cdiv(-lastra - y * XX(i, n - 1), -lastsa - y * XX(i, n), lastw, q, &cr, &ci);
            XX(i + 1, n - 1) = cr; XX(i + 1, n) = ci;
          }
        }
        t = fmax(fabs(XX(i, n - 1)), fabs(XX(i, n)));
        if ((eps * t) * t > 1) for (int j = i; j <= n; ++j) { XX(j, n - 1) = XX(j, n - 1) / t; XX(j, n) = XX(j, n) / t; }
      }
      continue;
    }
    if (q != 0) continue;
    int l = n;
    XX(n, n) = 1.0;
    for (int i = n - 1; i >= 0; --i) {
      w = HH(i, i) - p;
      r = 0.0;
      for (int j = l; j <= n; ++j) r = r + HH(i, j) * XX(j, n);
      if (wi[i] < 0.0) { z = w; s = r; }
      else {
        l = i;
        if (wi[i] == 0.0) {
          if (w != 0.0) XX(i, n) = -r / w;
          else XX(i, n) = -r / (eps * norm);
        } else {
          x = HH(i, i + 1); y = HH(i + 1, i);
          q = (wr[i] - p) * (wr[i] - p) + wi[i] * wi[i];
          t = (x * s - z * r) / q;
          XX(i, n) = t;
          if (fabs(x) > fabs(z)) XX(i + 1, n) = (-r - w * t) / x;
          else XX(i + 1, n) = (-s - y * t) / z;
        }
        t = fabs(XX(i, n));
        if ((eps * t) * t > 1) for (int j = i; j <= n; ++j) XX(j, n) = XX(j, n) / t;
      }
    }
  }
  team_sync();
  EIG_STAMP(5);
  // ---- back-transformation V <- V X, row i by lane (column j in descending order, in place as in the sequential code)
  for (int i = tl; i < vrows; i += TEAM)
    for (int j = nn - 1; j >= low; --j) {
      if (!CPLX && wi[j] != 0) continue;
      z = 0.0;
      for (int k = low; k <= j; ++k) z = z + VRR(i, k) * XX(k, j);
      VRR(i, j) = z;
    }
  team_sync();
  EIG_STAMP(6);
  EIG_STAMP_FLUSH;
  return true;
#undef HH
#undef VV
#undef XX
#undef VRR
}

}  // namespace rsc
}  // namespace thip
#endif
