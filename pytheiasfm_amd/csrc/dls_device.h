// DLS-PnP on the device (the reference: sfm/pose/dls_pnp.cc:67-200).  Two stages per minimal problem:
//
//   stage A  one WORKGROUP of 192 threads per problem.  Cost matrix, the cost quartic, the three Jacobian cubics, then the
//            Schur complement of the Macaulay matrix by the REFERENCE'S route: a dense partial-pivot LU of the 93 x 93
//            block in the generated table's row / column order (dls_layout.h) with the 27 reduced columns as right-hand
//            sides (dls_pnp.cc:143-146), a column-oriented back-substitution and M00 - M01 X.  Every entry sees exactly
//            the operations of oracle/dls_oracle.h in the same order (first-maximum pivots, l = a / pivot, one fused
//            multiply-add per update, x = rhs / u_kk), so the 27 x 27 result -- and with it every hypothesis, inlier set
//            and cost -- is BIT-IDENTICAL to the sequential restatement (round 3 eliminated degree by degree: a faster
//            but different route, 978 of 1000 pairs equal).
//            The augmented 93 x 120 matrix lives in REGISTERS: lane = (column group g = tid / 32, row group rg = tid % 32)
//            holds rows 3 rg + {0, 1, 2} at the columns 6 i + g -- 60 doubles.  Columns are dealt cyclically so that all
//            groups shrink together; after every six steps the lane's columns move down one register (folded into the
//            destination of that step's multiply-adds), so the pivot column is always register 0 of the half-wave g = k % 6
//            and every register index is a compile-time constant inside a rolled loop.  Per step: pivot search inside one
//            half-wave (butterfly), the factors l = a / pivot to LDS, barrier, the six lanes that hold the pivot row
//            broadcast it through LDS (and keep its part right of the diagonal for the back-substitution: 34 KB
//            triangular store), barrier, 3 x (20 - k / 6) multiply-adds per lane.  Rows are never moved: the oracle's
//            swap is bookkeeping of positions (ties go to the smallest POSITION, as the sequential scan takes them).
//            40 KB of LDS per workgroup: four problems per CU.
//   stage B  real Schur form + eigenvectors of the 27 x 27 matrix (orthes + hqr2, complex pairs included), root
//            extraction and the reference's solution filter.  In the RANSAC path a TEAM of 32 lanes per matrix with the
//            work arrays in LDS (eig_team.h); one thread per problem with the arrays in scratch for the
//            directly bound solver.
#ifndef THEIA_HIP_DLS_DEVICE_H_
#define THEIA_HIP_DLS_DEVICE_H_

#include <hip/hip_runtime.h>

#include "dls_tables.h"
#include "ransac_device.h"

namespace thip {
namespace dlsdev {

using dls::kReduced;
using dls::kBlock;
constexpr int kThreads = dls::kThreads;
constexpr int kMaxSolutions = 27;
constexpr int kUSize = kBlock * (kBlock - 1) / 2;   // rows of U right of the diagonal, row k at u_off(k)
__host__ __device__ constexpr int u_off(int k) { return (kBlock - 1) * k - k * (k - 1) / 2; }

__constant__ dls::Tables c_tab;

struct WgLds {
  double U[kUSize + 2];
  union {
    struct { double Dm[81]; double J[36]; double hinv[16]; double traw[36]; } fe;   // front end
    double pbuf[2][144];                                                           // pivot row of a step, by step parity
  };
  double lbuf[2][96];    // factors of a step by parity; the back-substitution's solved row lives in lbuf[parity][0..31]
  double diag[96];
  double f[60];
  double T[27];
  double sf[9];          // gDLS: the scale factor row
  double u[4];
  int pinfo[2][2];       // (row that holds the pivot, its position) by parity
  int flag;
  unsigned char prow_of[96];
};

// Eigen's Matrix4d::inverse() restated as adjugate over determinant (oracle/dls_oracle.h: gdls_inverse4)
__device__ inline bool inverse4(const double* a, double* inv) {
  auto m3 = [&](int r0, int r1, int r2, int c0, int c1, int c2) {
    return a[4 * r0 + c0] * (a[4 * r1 + c1] * a[4 * r2 + c2] - a[4 * r1 + c2] * a[4 * r2 + c1]) -
           a[4 * r0 + c1] * (a[4 * r1 + c0] * a[4 * r2 + c2] - a[4 * r1 + c2] * a[4 * r2 + c0]) +
           a[4 * r0 + c2] * (a[4 * r1 + c0] * a[4 * r2 + c1] - a[4 * r1 + c1] * a[4 * r2 + c0]);
  };
  double cof[16];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int r0 = r == 0 ? 1 : 0, r1 = r <= 1 ? 2 : 1, r2 = r <= 2 ? 3 : 2;
      const int c0 = c == 0 ? 1 : 0, c1 = c <= 1 ? 2 : 1, c2 = c <= 2 ? 3 : 2;
      const double minor = m3(r0, r1, r2, c0, c1, c2);
      cof[4 * r + c] = ((r + c) & 1) ? -minor : minor;
    }
  const double det = ((a[0] * cof[0] + a[1] * cof[1]) + a[2] * cof[2]) + a[3] * cof[3];
  if (det == 0.0) return false;
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) inv[4 * r + c] = cof[4 * c + r] / det;
  return true;
}

// One elimination step k = 6 o + S with NL live registers per row (NL >= 20 - o).  SHIFT (S == 5): the step's results
// land one register lower, so that the next six steps find their pivot columns in register 0 again.
// pos[q]: position of the lane's row q in the oracle's (swapped) row order; a row pivoted at step j keeps pos = j, so
// "still a candidate" is k <= pos < 93 (the padding rows 93..95 start at their own index and never are).
#define THIP_DLS_FENCE() asm volatile("" ::: "memory")
template <int NL, int Q>
__device__ __forceinline__ void lu_pivot_row_out(WgLds& L, const double (&a)[3][20], int par, int k, int o, int g) {
  double* ub = L.U + (u_off(k) - k - 1);
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const int col = 6 * (i + o) + g;
    L.pbuf[par][col] = a[Q][i];
    if (col > k && col < kBlock) ub[col] = a[Q][i];
  }
}
template <int NL, int S>
__device__ __forceinline__ void lu_step(WgLds& L, double (&a)[3][20], int (&pos)[3], int k, int o, int g, int rg) {
  const int par = k & 1;
  if (g == S) {   // this half-wave holds column k in register 0: pivot = first maximum in POSITION order
    double bv = 0.0; int bp = 1 << 20;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const bool cand = pos[q] >= k && pos[q] < kBlock;
      if (cand && (fabs(a[q][0]) > fabs(bv) || (fabs(a[q][0]) == fabs(bv) && pos[q] < bp))) { bv = a[q][0]; bp = pos[q]; }
    }
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) {
      const double ov = __shfl_xor(bv, m, 32);
      const int op = __shfl_xor(bp, m, 32);
      if (fabs(ov) > fabs(bv) || (fabs(ov) == fabs(bv) && op < bp)) { bv = ov; bp = op; }
    }
    if (bv == 0.0) { L.flag = 1; bv = 1.0; }   // singular block: the oracle gives up (no models); finish harmlessly
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const bool cand = pos[q] >= k && pos[q] < kBlock, mine = cand && pos[q] == bp;
      L.lbuf[par][3 * rg + q] = (!cand || mine) ? 0.0 : a[q][0] / bv;
      if (mine) { L.pinfo[par][0] = 3 * rg + q; L.pinfo[par][1] = bp; L.diag[k] = bv; L.prow_of[k] = (unsigned char)(3 * rg + q); }
    }
  }
  __syncthreads();
  const int pr = __builtin_amdgcn_readfirstlane(L.pinfo[par][0]), pp = __builtin_amdgcn_readfirstlane(L.pinfo[par][1]);
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    if (pos[q] == pp) pos[q] = k;
    else if (pos[q] == k) pos[q] = pp;
  }
  // the six lanes of the pivot row broadcast it and keep its part right of the diagonal (pr is uniform: no selects)
  const int prg = pr / 3, pq = pr - 3 * prg;
  if (pq == 0) { if (rg == prg) lu_pivot_row_out<NL, 0>(L, a, par, k, o, g); }
  else if (pq == 1) { if (rg == prg) lu_pivot_row_out<NL, 1>(L, a, par, k, o, g); }
  else { if (rg == prg) lu_pivot_row_out<NL, 2>(L, a, par, k, o, g); }
  __syncthreads();
  double l[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) l[q] = -L.lbuf[par][3 * rg + q];
  const double* pb = &L.pbuf[par][6 * o + g];
  constexpr int I0 = (S == 5) ? 1 : 0, D = (S == 5) ? 1 : 0;
#pragma unroll
  for (int c0 = I0; c0 < NL; c0 += 4) {   // four columns at a time: their LDS reads in flight together, no more
    double pv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) if (c0 + j < NL) pv[j] = pb[6 * (c0 + j)];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (c0 + j < NL) {
#pragma unroll
        for (int q = 0; q < 3; ++q) a[q][c0 + j - D] = __builtin_fma(l[q], pv[j], a[q][c0 + j]);
      }
    THIP_DLS_FENCE();
  }
  if (S == 5) {
#pragma unroll
    for (int q = 0; q < 3; ++q) a[q][NL - 1] = 0.0;
  }
}

template <int NL>
__device__ __forceinline__ void lu_six(WgLds& L, double (&a)[3][20], int (&pos)[3], int o, int g, int rg) {
  const int k = 6 * o;
  lu_step<NL, 0>(L, a, pos, k, o, g, rg);
  lu_step<NL, 1>(L, a, pos, k + 1, o, g, rg);
  lu_step<NL, 2>(L, a, pos, k + 2, o, g, rg);
  if (k + 3 >= kBlock) return;   // 93 = 15 * 6 + 3
  lu_step<NL, 3>(L, a, pos, k + 3, o, g, rg);
  lu_step<NL, 4>(L, a, pos, k + 4, o, g, rg);
  lu_step<NL, 5>(L, a, pos, k + 5, o, g, rg);
}

// back-substitution step: the lanes of the row pivoted at step k (row Q of row group prg) solve its five right-hand sides
template <int Q>
__device__ __forceinline__ void bs_solve_row(WgLds& L, const double (&a)[3][20], int par, int k, int g, double* __restrict__ Xn, int slot) {
  const double dg = L.diag[k];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const double x = a[Q][i] / dg;
    L.lbuf[par][6 * i + g] = x;
    const int c = 6 * i + g - 3;
    if (slot != 255 && c >= 0 && c < kReduced) Xn[slot * kReduced + c] = x;
  }
}

// points: feat[i * fstride + {0,1}], world[i * wstride + {0,1,2}] for i = index ? index[k] : k, k < npts.
// Writes action[729] (row-major) and tfac[27]; returns false when a pivot vanished (degenerate sample).
// GDLS (GdlsSimilarityTransform, gdls_similarity_transform.cc:67-175): feat holds the UNIT ray direction (3), world the
// homogeneous point (4: hnormalized here), origin the ray origin (3); tfac = translation factor (27) | scale factor (9).
// Called by all 192 threads of a workgroup; every sum runs in the order of oracle/dls_oracle.h (points in sequence).
template <bool GDLS = false>
__device__ __forceinline__ bool stage_a(WgLds& L, int npts, const double* __restrict__ feat, int fstride,
                               const double* __restrict__ world, int wstride, const int* __restrict__ index,
                               const double* __restrict__ u4, double* __restrict__ action, double* __restrict__ tfac,
                               const double* __restrict__ origin = nullptr, int ostride = 0) {
  const int tid = threadIdx.x;
  const int g = tid >> 5, rg = tid & 31;
  const dls::Tables& tb = c_tab;
  if (tid < 4) L.u[tid] = u4[tid];
  if (tid == 0) L.flag = 0;
  if constexpr (GDLS) {
    // ---- sums over the rays: the 4 x 4 matrix H^-1 (:80-96) and the 4 x 9 helper (:101-117), one entry per lane
    if (tid < 16 + 36) {
      double acc = 0.0;
      for (int i = 0; i < npts; ++i) {
        const int id = index ? index[i] : i;
        const double* xx = feat + (size_t)id * fstride; const double* cc = origin + (size_t)id * ostride; const double* ww = world + (size_t)id * wstride;
        const double x[3] = {xx[0], xx[1], xx[2]}, c[3] = {cc[0], cc[1], cc[2]};
        const double cd = (c[0] * x[0] + c[1] * x[1]) + c[2] * x[2];
        if (tid < 16) {
          const int r = tid >> 2, k = tid & 3;
          if (tid == 0) acc += ((c[0] * c[0] + c[1] * c[1]) + c[2] * c[2]) - cd * cd;
          else if (r == 0 || k == 0) { const int a = r + k - 1; acc += -c[a] + cd * x[a]; }
          else acc += (r == k ? 1.0 : 0.0) - x[r - 1] * x[k - 1];
        } else {
          const int e = tid - 16, r = e / 9, col = e % 9, kk = col / 3;
          const double lx = ww[col % 3] / ww[3];
          if (r == 0) acc += (c[kk] - cd * x[kk]) * lx;
          else acc += (x[r - 1] * x[kk] - (r - 1 == kk ? 1.0 : 0.0)) * lx;
        }
      }
      if (tid < 16) L.fe.hinv[tid] = acc; else L.fe.traw[tid - 16] = acc;
    }
    __syncthreads();
    if (tid < 36) {
      double Hm[16], hs[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) hs[k] = L.fe.hinv[k];
      if (!inverse4(hs, Hm)) { L.flag = 1; for (int k = 0; k < 16; ++k) Hm[k] = 0.0; }
      const int r = tid / 9, col = tid % 9;
      double s2 = 0.0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        double hk = 0.0;   // Hm[4 r + k]: compile-time register indices only
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) hk = (rr == r) ? Hm[4 * rr + k] : hk;
        s2 += hk * L.fe.traw[9 * k + col];
      }
      if (r == 0) L.sf[col] = s2; else L.T[9 * (r - 1) + col] = s2;
    }
  } else {
    // ---- H^-1 = n I - sum n n^T (dls_pnp.cc:90-94) and sum (n n^T - I) L(X) (:98-103), one entry per lane
    if (tid < 9 + 27) {
      const int r = tid < 9 ? tid / 3 : (tid - 9) / 9, c = tid < 9 ? tid % 3 : ((tid - 9) % 9) / 3;
      double acc = (tid < 9 && r == c) ? (double)npts : 0.0;
      for (int i = 0; i < npts; ++i) {
        const int id = index ? index[i] : i;
        const double fx = feat[(size_t)id * fstride], fy = feat[(size_t)id * fstride + 1];
        const double nrm = sqrt((fx * fx + fy * fy) + 1.0);
        const double b[3] = {fx / nrm, fy / nrm, 1.0 / nrm};
        const double br = r == 0 ? b[0] : (r == 1 ? b[1] : b[2]), bc = c == 0 ? b[0] : (c == 1 ? b[1] : b[2]);
        if (tid < 9) acc -= br * bc;
        else acc += (br * bc - (r == c ? 1.0 : 0.0)) * world[(size_t)id * wstride + (tid - 9) % 3];
      }
      if (tid < 9) L.fe.hinv[tid] = acc; else L.fe.traw[tid - 9] = acc;
    }
    __syncthreads();
    if (tid < 27) {   // translation_factor = H * (...)  (:105), H by cofactors as Eigen's 3 x 3 inverse
      const double* a = L.fe.hinv;
      const double c00 = a[4] * a[8] - a[5] * a[7], c01 = a[5] * a[6] - a[3] * a[8], c02 = a[3] * a[7] - a[4] * a[6];
      const double det = (a[0] * c00 + a[1] * c01) + a[2] * c02;
      const double id = 1.0 / det;
      const int r = tid / 9, c = tid % 9;
      double h0, h1, h2;
      if (r == 0) { h0 = c00 * id; h1 = (a[2] * a[7] - a[1] * a[8]) * id; h2 = (a[1] * a[5] - a[2] * a[4]) * id; }
      else if (r == 1) { h0 = c01 * id; h1 = (a[0] * a[8] - a[2] * a[6]) * id; h2 = (a[2] * a[3] - a[0] * a[5]) * id; }
      else { h0 = c02 * id; h1 = (a[1] * a[6] - a[0] * a[7]) * id; h2 = (a[0] * a[4] - a[1] * a[3]) * id; }
      L.T[tid] = (h0 * L.fe.traw[c] + h1 * L.fe.traw[9 + c]) + h2 * L.fe.traw[18 + c];
    }
  }
  __syncthreads();
  // ---- D = sum W^T (I - n n^T) W, W = L(X) + T (dls_pnp.cc:111-118; gDLS: W = L(X) - c scale_factor + T, :123-133)
  if (tid < 81) {
    const int al = tid / 9, be = tid % 9;
    double d = 0.0;
    for (int i = 0; i < npts; ++i) {
      const int id = index ? index[i] : i;
      double n[3], wa[3], wb[3];
      if constexpr (GDLS) {
        const double* xx = feat + (size_t)id * fstride; const double* cc = origin + (size_t)id * ostride; const double* ww = world + (size_t)id * wstride;
        n[0] = xx[0]; n[1] = xx[1]; n[2] = xx[2];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          wa[a] = (al / 3 == a ? ww[al % 3] / ww[3] : 0.0) + (L.T[9 * a + al] - cc[a] * L.sf[al]);
          wb[a] = (be / 3 == a ? ww[be % 3] / ww[3] : 0.0) + (L.T[9 * a + be] - cc[a] * L.sf[be]);
        }
      } else {
        const double fx = feat[(size_t)id * fstride], fy = feat[(size_t)id * fstride + 1];
        const double nrm = sqrt((fx * fx + fy * fy) + 1.0);
        n[0] = fx / nrm; n[1] = fy / nrm; n[2] = 1.0 / nrm;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          wa[a] = (al / 3 == a ? world[(size_t)id * wstride + al % 3] : 0.0) + L.T[9 * a + al];
          wb[a] = (be / 3 == a ? world[(size_t)id * wstride + be % 3] : 0.0) + L.T[9 * a + be];
        }
      }
      // PW = (I - n n^T) W column be, then column al of W against it
      double s2 = 0.0;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        double pw = 0.0;
#pragma unroll
        for (int k = 0; k < 3; ++k) pw += ((r == k ? 1.0 : 0.0) - n[r] * n[k]) * wb[k];
        s2 += wa[r] * pw;
      }
      d += s2;
    }
    L.fe.Dm[tid] = d;
  }
  __syncthreads();
  // ---- the cost quartic J' = sum_ab D_ab (rbar_a rbar_b), one coefficient per lane, (a, b) in row-major order
  if (tid < dls::kJMono) {
    double s = 0.0;
    for (int ab = 0; ab < 81; ++ab) { const int c = tb.P[ab][tid]; if (c) s += L.fe.Dm[ab] * (double)c; }
    L.fe.J[tid] = s;
  }
  __syncthreads();
  if (tid < 60) L.f[tid] = (double)tb.fmul[tid] * L.fe.J[tb.fsrc[tid]];   // f_i = dJ'/ds_i
  __syncthreads();
  // ---- the augmented block [M11 | M10] into registers
  double a[3][20];
  {
    const uint32_t* code = reinterpret_cast<const uint32_t*>(tb.init[tid]);
#pragma unroll
    for (int w = 0; w < 15; ++w) {
      const uint32_t cw = code[w];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int cd = (cw >> (8 * b)) & 255;
        a[(4 * w + b) / 20][(4 * w + b) % 20] = cd ? L.f[cd - 1] : 0.0;
      }
    }
  }
  int pos[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) pos[q] = 3 * rg + q;
  __syncthreads();   // the front end's arrays share the pivot-row buffer
  // ---- elimination (oracle: dls_action_from_cost)
  for (int o = 0; o < 4; ++o) lu_six<20>(L, a, pos, o, g, rg);
  for (int o = 4; o < 8; ++o) lu_six<16>(L, a, pos, o, g, rg);
  for (int o = 8; o < 12; ++o) lu_six<12>(L, a, pos, o, g, rg);
  for (int o = 12; o < 16; ++o) lu_six<8>(L, a, pos, o, g, rg);
  // ---- back-substitution, column oriented: register i < 5 of a row now holds right-hand side 6 (i + 15) + g - 93
  double* Xn = action;   // the solved rows the result reads wait in the problem's own output slot
  __syncthreads();       // the solved rows go through the factor buffer the last step may still be read from
  for (int k = kBlock - 1; k >= 0; --k) {
    const int par = k & 1;
    const int pr = __builtin_amdgcn_readfirstlane((int)L.prow_of[k]);
    const int prg = pr / 3, pq = pr - 3 * prg, slot = tb.xslot[k];
    if (pq == 0) { if (rg == prg) bs_solve_row<0>(L, a, par, k, g, Xn, slot); }
    else if (pq == 1) { if (rg == prg) bs_solve_row<1>(L, a, par, k, g, Xn, slot); }
    else { if (rg == prg) bs_solve_row<2>(L, a, par, k, g, Xn, slot); }
    __syncthreads();
    double x[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) x[i] = L.lbuf[par][6 * i + g];
#pragma unroll
    for (int q = 0; q < 3; ++q)
      if (pos[q] < k) {
        const double u = -L.U[u_off(pos[q]) + k - pos[q] - 1];
#pragma unroll
        for (int i = 0; i < 5; ++i) a[q][i] = __builtin_fma(u, x[i], a[q][i]);
      }
  }
  __threadfence_block();
  __syncthreads();
  // ---- M00 - M01 X, the columns of M01 in ascending order
  double res[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int e = tid + kThreads * t;
    res[t] = 0.0;
    if (e < kReduced * kReduced) {
      const int r = e / kReduced, c = e % kReduced;
      const int cd = tb.m00[r][c];
      double s = cd ? L.u[cd - 1] : 0.0;
      for (int m = 0; m < tb.m01n[r]; ++m) s = __builtin_fma(-L.u[tb.m01q[r][m]], Xn[tb.xslot[tb.m01j[r][m]] * kReduced + c], s);
      res[t] = s;
    }
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 4; ++t) { const int e = tid + kThreads * t; if (e < kReduced * kReduced) action[e] = res[t]; }
  if (tid < 27) tfac[tid] = L.T[tid];
  if (GDLS && tid < 9) tfac[27 + tid] = L.sf[tid];
  return L.flag == 0;
}

// One eigenvector column -> (quaternion [w x y z], translation) if it is an admissible root (dls_pnp.cc:147-198):
// V (27 x 27, hqr2 column convention), wi: imaginary parts of the eigenvalues.
// COMPACT: V holds only the four rows read here, in the order {0, 9, 3, 1} (eig_team's kept rows).
constexpr int kKeptRows = 4;
__device__ __constant__ const int kKeptRow[kKeptRows] = {0, 9, 3, 1};
template <bool COMPACT = false>
__device__ inline bool column_solution(const double* V, const double* wi, int i, const double* __restrict__ tfac, int npts,
                                       const double* __restrict__ world, int wstride, const int* __restrict__ index,
                                       double* quat, double* tr) {
  const int re_col = wi[i] < 0 ? i - 1 : i;
  if (re_col < 0) return false;
  const bool cplx = wi[i] != 0.0;
  const double sg = wi[i] < 0 ? -1.0 : 1.0;
  const double d_re = V[re_col], d_im = cplx ? sg * V[re_col + 1] : 0.0;   // row 0
  if (d_re == 0.0 && d_im == 0.0) return false;
  double sr[3], si[3];
  const int rows[3] = {COMPACT ? 1 : 9, COMPACT ? 2 : 3, COMPACT ? 3 : 1};
  for (int k = 0; k < 3; ++k) {
    const double a = V[27 * rows[k] + re_col], b = cplx ? sg * V[27 * rows[k] + re_col + 1] : 0.0;
    rsc::eig_cdiv(a, b, d_re, d_im, &sr[k], &si[k]);
  }
  const double kEps = 1e-6;
  if (!(fabs(si[0]) < kEps && fabs(si[1]) < kEps && fabs(si[2]) < kEps)) return false;
  // Quaterniond(1, s1, s2, s3).inverse().normalized()
  const double n2 = ((1.0 + sr[0] * sr[0]) + sr[1] * sr[1]) + sr[2] * sr[2];
  const double qi[4] = {1.0 / n2, -sr[0] / n2, -sr[1] / n2, -sr[2] / n2};
  const double nq = sqrt(((qi[0] * qi[0] + qi[1] * qi[1]) + qi[2] * qi[2]) + qi[3] * qi[3]);
  const double qs[4] = {qi[0] / nq, qi[1] / nq, qi[2] / nq, qi[3] / nq};
  const double m2 = ((qs[0] * qs[0] + qs[1] * qs[1]) + qs[2] * qs[2]) + qs[3] * qs[3];
  const double qv[4] = {qs[0] / m2, -qs[1] / m2, -qs[2] / m2, -qs[3] / m2};
  double Rm[9], Rs[9], t[3];
  rsc::quat_to_rot(qv, Rm);
  for (int r = 0; r < 3; ++r) {
    double s = 0.0;
    for (int c = 0; c < 3; ++c) for (int k = 0; k < 3; ++k) s += tfac[9 * r + 3 * c + k] * Rm[3 * k + c];
    t[r] = s;
  }
  rsc::quat_to_rot(qs, Rs);
  for (int j = 0; j < npts; ++j) {
    const int id = index ? index[j] : j;
    const double* X = world + (size_t)id * wstride;
    const double z = ((Rs[6] * X[0] + Rs[7] * X[1]) + Rs[8] * X[2]) + t[2];
    if (z < 0) return false;
  }
  for (int k = 0; k < 4; ++k) quat[k] = qs[k];
  for (int k = 0; k < 3; ++k) tr[k] = t[k];
  return true;
}

// gDLS: one eigenvector column -> (quaternion [w x y z] = soln_rotation, translation, scale) if it is an admissible root
// (gdls_similarity_transform.cc:176-226).  V: the four kept rows {0, 9, 3, 1} (COMPACT layout of column_solution).
// tfac: translation factor (27) | scale factor (9).  Rays: dir / origin / homogeneous world point of the npts sampled data.
__device__ inline bool column_solution_gdls(const double* V, const double* wi, int i, const double* __restrict__ tfac, int npts,
                                            const double* __restrict__ data, int stride, int dir_off, int org_off, int wld_off,
                                            const int* __restrict__ index, double* quat, double* tr, double* scale) {
  const int re_col = wi[i] < 0 ? i - 1 : i;
  if (re_col < 0) return false;
  const bool cplx = wi[i] != 0.0;
  const double sg = wi[i] < 0 ? -1.0 : 1.0;
  const double d_re = V[re_col], d_im = cplx ? sg * V[re_col + 1] : 0.0;   // row 0
  if (d_re == 0.0 && d_im == 0.0) return false;
  double sr[3], si[3];
  for (int k = 0; k < 3; ++k) {
    const double a = V[27 * (k + 1) + re_col], b = cplx ? sg * V[27 * (k + 1) + re_col + 1] : 0.0;
    rsc::eig_cdiv(a, b, d_re, d_im, &sr[k], &si[k]);
  }
  const double kEps = 1e-6;
  if (!(fabs(si[0]) < kEps && fabs(si[1]) < kEps && fabs(si[2]) < kEps)) return false;
  const double n2 = ((1.0 + sr[0] * sr[0]) + sr[1] * sr[1]) + sr[2] * sr[2];
  const double qi[4] = {1.0 / n2, -sr[0] / n2, -sr[1] / n2, -sr[2] / n2};
  const double nq = sqrt(((qi[0] * qi[0] + qi[1] * qi[1]) + qi[2] * qi[2]) + qi[3] * qi[3]);
  const double qs[4] = {qi[0] / nq, qi[1] / nq, qi[2] / nq, qi[3] / nq};
  const double m2 = ((qs[0] * qs[0] + qs[1] * qs[1]) + qs[2] * qs[2]) + qs[3] * qs[3];
  const double qv[4] = {qs[0] / m2, -qs[1] / m2, -qs[2] / m2, -qs[3] / m2};
  double Rm[9], Rs[9], t[3], sc = 0.0;
  rsc::quat_to_rot(qv, Rm);
  for (int r = 0; r < 3; ++r) {
    double s = 0.0;
    for (int c = 0; c < 3; ++c) for (int k = 0; k < 3; ++k) s += tfac[9 * r + 3 * c + k] * Rm[3 * k + c];
    t[r] = s;
  }
  for (int c = 0; c < 3; ++c) for (int k = 0; k < 3; ++k) sc += tfac[27 + 3 * c + k] * Rm[3 * k + c];
  rsc::quat_to_rot(qs, Rs);
  for (int j = 0; j < npts; ++j) {   // every point in front of its ray: x . (R X + t - s c) >= 0
    const double* d = data + (size_t)(index ? index[j] : j) * stride;
    const double* x = d + dir_off; const double* c = d + org_off; const double* w = d + wld_off;
    const double X[3] = {w[0] / w[3], w[1] / w[3], w[2] / w[3]};
    double p[3];
    for (int r = 0; r < 3; ++r) p[r] = (((Rs[3 * r] * X[0] + Rs[3 * r + 1] * X[1]) + Rs[3 * r + 2] * X[2]) + t[r]) - sc * c[r];
    if ((x[0] * p[0] + x[1] * p[1]) + x[2] * p[2] < 0) return false;
  }
  for (int k = 0; k < 4; ++k) quat[k] = qs[k];
  for (int k = 0; k < 3; ++k) tr[k] = t[k];
  *scale = sc;
  return true;
}


// Stage B, one thread per problem (the directly bound solver): H, V = 729-double work arrays of the calling thread.
__device__ inline int stage_b(double* H, double* V, const double* __restrict__ tfac, int npts,
                              const double* __restrict__ world, int wstride, const int* __restrict__ index,
                              double* quats, double* ts) {
  double wr[27], wi[27];
  if (!rsc::eig_general_t<27, true>(27, H, wr, wi, V)) return 0;
  int ns = 0;
  for (int i = 0; i < 27; ++i)
    if (column_solution(V, wi, i, tfac, npts, world, wstride, index, quats + 4 * ns, ts + 3 * ns)) ns++;
  return ns;
}

}  // namespace dlsdev
}  // namespace thip
#endif
