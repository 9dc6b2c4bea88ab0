// UPnP minimal solver for THEIA_EST_RIGID_TRANSFORMATION_2D3D (estimate_rigid_transformation_2d_3d.cc:62-135 -> Upnp::EstimatePose,
// sfm/pose/upnp.cc:462-493 -> BuildActionMatrixUsingSymmetry, build_upnp_action_matrix_using_symmetry.cc:2344-2502 -> the reference's
// own GaussJordan, math/matrix/gauss_jordan.h:94-193).  Four kernels per round of hypotheses:
//   k_upnp_pre   one thread per hypothesis: H, V_i, G, J and the rows A_i, b_i of the four correspondences (upnp.cc:80-217)
//   k_upnp_scan  one workgroup per problem: the estimator's cost parameters only ever GROW (upnp.cc:191-200 adds to the member of the
//                one Upnp object the RANSAC estimator keeps), so hypothesis k is solved from the sum over the samples 0 .. k of its
//                Estimate() call: a running sum over the round in sample order, one correspondence at a time, the state carried from
//                round to round (the reference's quirk, reproduced)
//   k_upnp_a     one workgroup of 384 threads per hypothesis: the 8 x 24 input matrix and its reduction, then the 141 x 149 template
//                IN REGISTERS (thread (g, rg) holds rows 3 rg + {0, 1, 2} x columns 8 j + g: 57 doubles) and the top-down
//                Gauss-Jordan with the reference's rules -- first maximum over the positions k .. 139 (the last row is never searched),
//                rows swapped (a position table; nothing moves), the pivot row DIVIDED by the pivot, multipliers below 1e-9
//                skipped, whole rows updated --, then the bottom-up pass over the rows 140 .. 121 on a 20 x 28 block in LDS
//   k_upnp_b     teams of 8 lanes: the 8 x 8 eigen-decomposition (eig_team.h), quaternions from rows 4 .. 7 of the normalised
//                eigenvectors' real parts, duplicates removed, translations, the all-points-in-front test, models
// Every entry sees the operations of oracle/upnp_oracle.h in its order (plain FP64, no fused multiply-adds: build.sh), so the
// hypotheses are bit-identical to the oracle's.  Own translation unit for the compiler switch of dls_kernels.hip
// (-mllvm -simplifycfg-sink-common=false: the three copies of "publish my row i" must stay three register-indexed copies).
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstring>
#include <mutex>

#include "eig_team.h"
#include "wave_reduce.h"
#include "theia_hip_internal.h"
#include "upnp_layout.h"

namespace thip {
namespace upnpdev {

constexpr int kCG = 8, kRG = 48, kNRL = 3, kNCL = 19, kThreads = kCG * kRG;   // 384 threads = 6 waves
constexpr int kRowsPad = kRG * kNRL, kColsPad = kCG * kNCL;                   // 144 x 152
constexpr int kMaxShift = 14;   // the local column 0 stops at 112 + g: the columns 121 .. 148 stay for the bottom-up block
constexpr int kDatum = 26;      // the row of THEIA_EST_SIMILARITY_2D3D (theia_hip.h): dir | point (4) | pixel | extrinsics | model | intrinsics
// workspace per hypothesis: V_i (4 x 9) | T_i (3 x 10), tb_i (3) of the four correspondences | A (100), b (10) after the sample | action rows (4 x 8)
constexpr int kWsV = 0, kWsT = 36, kWsA = 168, kWsAct = 278, kWs = 310;
constexpr int kStateDoubles = 110;
constexpr int kMaxModels = 8;

struct Tables {
  uint8_t src[kRowsPad][kColsPad];   // template entry = input[src], 255 = zero
  double fixed[192];                 // rows 4 .. 7 of the input matrix (q_i (|q|^2 - 1)); rows 0 .. 3 are computed
  uint8_t nterm[96], coef[96][3], tq[96][3], tp[96][3];   // input (i, j), i < 4: sum of coef * A(q, p) (j < 20) or coef * b[p]
};
__device__ Tables g_tab;

__device__ __forceinline__ void quat_rotate(const double* q, const double* v, double* o) {   // Eigen's _transformVector
  double uv[3] = {q[2] * v[2] - q[3] * v[1], q[3] * v[0] - q[1] * v[2], q[1] * v[1] - q[2] * v[0]};
  for (int k = 0; k < 3; ++k) uv[k] += uv[k];
  const double c[3] = {q[2] * uv[2] - q[3] * uv[1], q[3] * uv[0] - q[1] * uv[2], q[1] * uv[1] - q[2] * uv[0]};
  for (int k = 0; k < 3; ++k) o[k] = (v[k] + q[0] * uv[k]) + c[k];
}

// ---------------------------------------------------------------- cost-parameter rows of one hypothesis
__global__ __launch_bounds__(64) void k_upnp_pre(int nprob, int B, const int64_t* __restrict__ offsets, const double* __restrict__ data,
                                                 const int* __restrict__ samples, const int* __restrict__ active_iters,
                                                 double* __restrict__ ws) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x, p = blockIdx.y;
  if (b >= B || p >= nprob || b >= active_iters[p]) return;
  const size_t hyp = (size_t)p * B + b;
  const double* pd = data + (size_t)offsets[p] * kDatum;
  double* w = ws + hyp * kWs;
  double o[4][3], f[4][3], X[4][3];
  for (int i = 0; i < 4; ++i) {
    const double* d = pd + (size_t)samples[hyp * 4 + i] * kDatum;
    for (int k = 0; k < 3; ++k) { f[i][k] = d[k]; X[i][k] = d[3 + k] / d[6]; o[i][k] = d[9 + k]; }
  }
  double Hinv[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Hinv[3 * r + c] -= f[i][r] * f[i][c];
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Hinv[3 * r + c] += 4.0 * (r == c ? 1.0 : 0.0);
  double H[9];
  {
    const double* m = Hinv;
#define COF(i1, j1, i2, j2) (m[3 * (i1) + (j1)] * m[3 * (i2) + (j2)] - m[3 * (i1) + (j2)] * m[3 * (i2) + (j1)])
    const double c0 = COF(1, 1, 2, 2), c1 = COF(2, 1, 0, 2), c2 = COF(0, 1, 1, 2);
    const double det = (c0 * m[0] + c1 * m[3]) + c2 * m[6];
    const double invdet = 1.0 / det;
    H[0] = c0 * invdet; H[1] = c1 * invdet; H[2] = c2 * invdet;
    H[3] = COF(1, 2, 2, 0) * invdet; H[4] = COF(2, 2, 0, 0) * invdet; H[5] = COF(0, 2, 1, 0) * invdet;
    H[6] = COF(1, 0, 2, 1) * invdet; H[7] = COF(2, 0, 0, 1) * invdet; H[8] = COF(0, 0, 1, 1) * invdet;
#undef COF
  }
  double G[30], J[3] = {0, 0, 0};
  for (int k = 0; k < 30; ++k) G[k] = 0.0;
  auto phi = [](const double* Xp, double* P) {   // LeftMultiply (upnp.cc:98-136)
    const double x = Xp[0], y = Xp[1], z = Xp[2];
    P[0] = x; P[1] = x; P[2] = -x; P[3] = -x; P[4] = 0.0; P[5] = 2 * z; P[6] = -2 * y; P[7] = 2 * y; P[8] = 2 * z; P[9] = 0.0;
    P[10] = y; P[11] = -y; P[12] = y; P[13] = -y; P[14] = -2.0 * z; P[15] = 0.0; P[16] = 2 * x; P[17] = 2 * x; P[18] = 0.0; P[19] = 2 * z;
    P[20] = z; P[21] = -z; P[22] = -z; P[23] = z; P[24] = 2.0 * y; P[25] = -2.0 * x; P[26] = 0.0; P[27] = 0.0; P[28] = 2.0 * x; P[29] = 2.0 * y;
  };
  for (int i = 0; i < 4; ++i) {
    double D[9], P[30], Vi[9];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) D[3 * r + c] = f[i][r] * f[i][c] - (r == c ? 1.0 : 0.0);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Vi[3 * r + c] = (H[3 * r] * D[c] + H[3 * r + 1] * D[3 + c]) + H[3 * r + 2] * D[6 + c];
    for (int k = 0; k < 9; ++k) w[kWsV + 9 * i + k] = Vi[k];
    phi(X[i], P);
    for (int r = 0; r < 3; ++r) J[r] += (Vi[3 * r] * o[i][0] + Vi[3 * r + 1] * o[i][1]) + Vi[3 * r + 2] * o[i][2];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 10; ++c) G[10 * r + c] += (Vi[3 * r] * P[c] + Vi[3 * r + 1] * P[10 + c]) + Vi[3 * r + 2] * P[20 + c];
  }
  for (int i = 0; i < 4; ++i) {
    double D[9], P[30], s[3];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) D[3 * r + c] = f[i][r] * f[i][c] - (r == c ? 1.0 : 0.0);
    phi(X[i], P);
    for (int k = 0; k < 30; ++k) P[k] = P[k] + G[k];
    double* T = w + kWsT + 33 * i;
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 10; ++c) T[10 * r + c] = (D[3 * r] * P[c] + D[3 * r + 1] * P[10 + c]) + D[3 * r + 2] * P[20 + c];
    for (int r = 0; r < 3; ++r) s[r] = o[i][r] + J[r];
    for (int r = 0; r < 3; ++r) T[30 + r] = ((-D[3 * r]) * s[0] + (-D[3 * r + 1]) * s[1]) + (-D[3 * r + 2]) * s[2];
  }
}

// ---------------------------------------------------------------- the estimator's accumulating cost parameters
// lanes 0 .. 99: A(p, q); 100 .. 109: b(p).  state: [nprob][110], the sums after the previous round.
__global__ __launch_bounds__(128) void k_upnp_scan(int B, const int* __restrict__ active_iters, double* __restrict__ state, double* __restrict__ ws) {
  const int p = blockIdx.x, e = threadIdx.x;
  if (e >= kStateDoubles) return;
  const int n = active_iters[p];
  double acc = state[(size_t)p * kStateDoubles + e];
  const int r = e < 100 ? e / 10 : e - 100, c = e < 100 ? e % 10 : 30;   // b: column "30" = tb
  for (int b = 0; b < n; ++b) {
    double* w = ws + ((size_t)p * B + b) * kWs;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const double* T = w + kWsT + 33 * i;
      // A: T^T T entry (r, c); b: T^T tb entry r (tb = T[30 .. 32])
      const double y0 = e < 100 ? T[c] : T[30], y1 = e < 100 ? T[10 + c] : T[31], y2 = e < 100 ? T[20 + c] : T[32];
      acc += (T[r] * y0 + T[10 + r] * y1) + T[20 + r] * y2;
    }
    w[kWsA + e] = acc;
  }
  state[(size_t)p * kStateDoubles + e] = acc;
}

// ---------------------------------------------------------------- input matrix, template, Gauss-Jordan
struct WgLds {
  double Ab[kStateDoubles];
  double M1[192];
  double LU[49], inv[49], out[168];
  int perm[8];
  double colbuf[2][kRowsPad];   // column k by POSITION
  double rowraw[2][kColsPad], rowbuf[2][kColsPad];
  double blk[20][29];
#ifdef THIP_UPNP_STAMPS
  unsigned long long st[8];
#endif
};

// wave-wide reductions on the DPP network (rows of 16 lanes, then the four row results through v_readlane)
template <int CTRL>
__device__ __forceinline__ double dpp_move_d(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_d(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ int wave_min_i(int v) {
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, false));
  return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

// One elimination step with the pivot column at local index LC (0 while the registers still shift) and NL live local columns
// (the rest hold the zeros that were shifted in).  mypos[i]: the POSITION of the thread's row i in the reference's swapped row
// order -- the column of step k is published by position, so the pivot search, the multipliers and the swap need no row table.
// Development (-DTHIP_UPNP_STAMPS): s_memtime ticks of thread 0 per section of a Gauss-Jordan step, summed over the workgroups:
// {column out + barrier, search, pivot row out + barrier, division + barrier, update, steps, workgroups}
#ifdef THIP_UPNP_STAMPS
__device__ unsigned long long g_upnp_stamps[8];
#define UPNP_T0 unsigned long long ut_ = __builtin_amdgcn_s_memtime()
#define UPNP_STAMP(k_) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) L.st[k_] += n_ - ut_; ut_ = n_; } while (0)
#else
#define UPNP_T0 do {} while (0)
#define UPNP_STAMP(k_) do {} while (0)
#endif
template <int LC, int NL>
__device__ __forceinline__ void gj_step(WgLds& L, double (&a)[kNRL][kNCL], int (&mypos)[kNRL], int k, int g) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int gk = k & (kCG - 1), buf = k & 1;
  UPNP_T0;
  if (g == gk) {
#pragma unroll
    for (int i = 0; i < kNRL; ++i) L.colbuf[buf][mypos[i]] = a[i][LC];
  }
  __syncthreads();
  UPNP_STAMP(0);
  double l[kNRL];
#pragma unroll
  for (int i = 0; i < kNRL; ++i) l[i] = L.colbuf[buf][mypos[i]];
  // first maximum over the positions k .. 139 (gauss_jordan.h:46-63 with ending_row = the last row: `row < ending_row`)
  int bpos = k;
  if (k < 140) {
    double babs = -1.0;
    int mine = 1 << 20;
    for (int pos = k + lane; pos < 140; pos += 64) {
      const double v = fabs(L.colbuf[buf][pos]);
      if (v > babs) { babs = v; mine = pos; }
    }
    // the maximum as two unsigned 32-bit reductions (high word, then the low word among the lanes at the high maximum: a
    // non-negative double orders like its bit pattern; a NaN is no candidate, as under `v > babs`), the position tie-break only
    // when two lanes hold the same |a|: the search is on every wave's critical path between two barriers (dls_stage_a.h)
    const bool valid = babs >= 0.0;
    const unsigned hi = valid ? (unsigned)__double2hiint(babs) : 0u, lo = (unsigned)__double2loint(babs);
    const unsigned hm = wave_max_u32(hi);
    const bool top = valid & (hi == hm);
    const unsigned lm = wave_max_u32(top ? lo : 0u);
    const bool match = top & (lo == lm);
    const unsigned long long own = __builtin_amdgcn_ballot_w64(match);
    if (__builtin_popcountll(own) == 1) bpos = __builtin_amdgcn_readlane(mine, (int)__builtin_ctzll(own));
    else bpos = wave_min_i(match ? mine : (1 << 20));
  }
  const double piv = L.colbuf[buf][bpos];
  UPNP_STAMP(1);
  // the pivot row as it stands (three register-indexed copies, at most one taken), then the swap of the two positions
  if (mypos[0] == bpos) {
#pragma unroll
    for (int j = 0; j < NL; ++j) L.rowraw[buf][kCG * j + g] = a[0][j];
  } else if (mypos[1] == bpos) {
#pragma unroll
    for (int j = 0; j < NL; ++j) L.rowraw[buf][kCG * j + g] = a[1][j];
  } else if (mypos[2] == bpos) {
#pragma unroll
    for (int j = 0; j < NL; ++j) L.rowraw[buf][kCG * j + g] = a[2][j];
  }
#pragma unroll
  for (int i = 0; i < kNRL; ++i) mypos[i] = mypos[i] == bpos ? k : (mypos[i] == k ? bpos : mypos[i]);
  __syncthreads();
  UPNP_STAMP(2);
  if (tid < kCG * NL) L.rowbuf[buf][tid] = (tid == kCG * LC + gk) ? 1.0 : L.rowraw[buf][tid] / piv;   // row /= pivot; (k, k) = 1
  __syncthreads();
  UPNP_STAMP(3);
  double prow[NL];
#pragma unroll
  for (int j = 0; j < NL; ++j) prow[j] = L.rowbuf[buf][kCG * j + g];
  if (mypos[0] == k) {
#pragma unroll
    for (int j = 0; j < NL; ++j) a[0][j] = prow[j];
  } else if (mypos[1] == k) {
#pragma unroll
    for (int j = 0; j < NL; ++j) a[1][j] = prow[j];
  } else if (mypos[2] == k) {
#pragma unroll
    for (int j = 0; j < NL; ++j) a[2][j] = prow[j];
  }
#pragma unroll
  for (int i = 0; i < kNRL; ++i) {
    if (mypos[i] > k && !(fabs(l[i]) < 1e-9)) {
#pragma unroll
      for (int j = 0; j < NL; ++j) a[i][j] = a[i][j] - l[i] * prow[j];
    }
  }
  UPNP_STAMP(4);
#ifdef THIP_UPNP_STAMPS
  if (threadIdx.x == 0) L.st[5] += 1ull;
#endif
}

template <int LC, int NL>
__device__ __forceinline__ void gj_steps(WgLds& L, double (&a)[kNRL][kNCL], int (&mypos)[kNRL], int k0, int k1, int g) {
#pragma nounroll
  for (int k = k0; k < k1; ++k) gj_step<LC, NL>(L, a, mypos, k, g);
}

// eight steps on local column 0, then the registers move one column down
template <int NL>
__device__ __forceinline__ void gj_block(WgLds& L, double (&a)[kNRL][kNCL], int (&mypos)[kNRL], int s0, int s1, int g) {
#pragma nounroll
  for (int s = s0; s < s1; ++s) {
    gj_steps<0, NL>(L, a, mypos, kCG * s, kCG * s + kCG, g);
    if (s < kMaxShift) {
#pragma unroll
      for (int i = 0; i < kNRL; ++i) {
#pragma unroll
        for (int j = 0; j + 1 < NL; ++j) a[i][j] = a[i][j + 1];
        a[i][NL - 1] = 0.0;
      }
    }
  }
}

__global__ __launch_bounds__(kThreads) void k_upnp_a(int B, const int* __restrict__ active_iters, double* __restrict__ ws) {
  __shared__ WgLds L;
  const int b = blockIdx.x, p = blockIdx.y;
  if (b >= active_iters[p]) return;
  double* w = ws + ((size_t)p * B + b) * kWs;
  const int tid = threadIdx.x, g = tid / kRG, rg = tid % kRG;
  if (tid < kStateDoubles) L.Ab[tid] = w[kWsA + tid];
  __syncthreads();
  // ---- the input matrix (build_upnp_action_matrix_using_symmetry.cc:2349-2468)
  if (tid < 96) {
    const int j = tid % 24;
    double acc = 0.0;
    const int nt = g_tab.nterm[tid];
    for (int t = 0; t < nt; ++t) {
      const double v = (double)g_tab.coef[tid][t] * (j < 20 ? L.Ab[10 * g_tab.tq[tid][t] + g_tab.tp[tid][t]] : L.Ab[100 + g_tab.tp[tid][t]]);
      acc = t == 0 ? v : acc + v;
    }
    L.M1[tid] = acc;
  } else if (tid < 192) L.M1[tid] = g_tab.fixed[tid];
  __syncthreads();
  // ---- block<7, 24> = block<7, 7>^-1 block<7, 24> (:2470-2471) and the cancellation of column 10 (:2474-2476), by wave 0
  if (tid < 64) {
    const int r = tid / 7, c = tid % 7;
    if (tid < 49) L.LU[tid] = L.M1[24 * r + c];
    if (tid < 7) L.perm[tid] = tid;
    rsc::team_sync();
    for (int k = 0; k < 7; ++k) {
      int best = k;
      for (int rr = k + 1; rr < 7; ++rr) if (fabs(L.LU[7 * rr + k]) > fabs(L.LU[7 * best + k])) best = rr;
      rsc::team_sync();
      if (best != k) {
        if (tid < 7) { const double t0 = L.LU[7 * k + tid]; L.LU[7 * k + tid] = L.LU[7 * best + tid]; L.LU[7 * best + tid] = t0; }
        if (tid == 7) { const int t0 = L.perm[k]; L.perm[k] = L.perm[best]; L.perm[best] = t0; }
      }
      rsc::team_sync();
      double nv = 0.0;
      const bool mine = tid < 49 && r > k && c >= k;
      if (mine) {
        const double l = L.LU[7 * r + k] / L.LU[7 * k + k];
        nv = c == k ? l : L.LU[7 * r + c] - l * L.LU[7 * k + c];
      }
      rsc::team_sync();
      if (mine) L.LU[7 * r + c] = nv;
      rsc::team_sync();
    }
    if (tid < 7) {   // column tid of the inverse: L y = P e, U x = y (x lands in inv[.][tid] as it is found)
      double y[7];
#pragma unroll
      for (int rr = 0; rr < 7; ++rr) {
        double v = L.perm[rr] == tid ? 1.0 : 0.0;
#pragma unroll
        for (int kk = 0; kk < 7; ++kk) if (kk < rr) v -= L.LU[7 * rr + kk] * y[kk];
        y[rr] = v;
      }
#pragma unroll
      for (int rr = 6; rr >= 0; --rr) {
        double v = y[rr];
        for (int kk = rr + 1; kk < 7; ++kk) v -= L.LU[7 * rr + kk] * L.inv[7 * kk + tid];
        L.inv[7 * rr + tid] = v / L.LU[7 * rr + rr];
      }
    }
    rsc::team_sync();
    for (int e = tid; e < 168; e += 64) {
      const int rr = e / 24, cc = e % 24;
      double acc = L.inv[7 * rr] * L.M1[cc];
      for (int kk = 1; kk < 7; ++kk) acc += L.inv[7 * rr + kk] * L.M1[24 * kk + cc];
      L.out[e] = acc;
    }
    rsc::team_sync();
    for (int e = tid; e < 168; e += 64) L.M1[e] = L.out[e];
    rsc::team_sync();
    for (int e = tid; e < 168; e += 64) {
      const int rr = e / 24, cc = e % 24;
      const double fct = L.M1[24 * rr + 10] / L.M1[24 * 7 + 10];
      L.out[e] = L.M1[e] - fct * L.M1[24 * 7 + cc];
    }
    rsc::team_sync();
    for (int e = tid; e < 168; e += 64) L.M1[e] = L.out[e];
  }
  __syncthreads();
  // ---- the template (:83-2340 through upnp_layout.h), in registers
  double a[kNRL][kNCL];
  int mypos[kNRL];
#pragma unroll
  for (int i = 0; i < kNRL; ++i) {
    mypos[i] = kNRL * rg + i;   // (the three padding rows 141 .. 143 hold zeros: their multipliers are always skipped)
#pragma unroll
    for (int j = 0; j < kNCL; ++j) {
      const int s = g_tab.src[kNRL * rg + i][kCG * j + g];
      a[i][j] = s == 255 ? 0.0 : L.M1[s];
    }
  }
#ifdef THIP_UPNP_STAMPS
  if (tid == 0) for (int k_ = 0; k_ < 8; ++k_) L.st[k_] = 0;
#endif
  // ---- GaussJordan(140, 121): top-down over all 141 rows.  After s shifts the local column j stands for 8 (j + s) + g <= 148,
  // i.e. 19 - s live registers per row
  gj_block<19>(L, a, mypos, 0, 4, g);
  gj_block<15>(L, a, mypos, 4, 8, g);
  gj_block<11>(L, a, mypos, 8, 12, g);
  gj_block<7>(L, a, mypos, 12, kMaxShift + 1, g);
  gj_steps<1, 5>(L, a, mypos, 120, 128, g);
  gj_steps<2, 5>(L, a, mypos, 128, 136, g);
  gj_steps<3, 5>(L, a, mypos, 136, 141, g);
#ifdef THIP_UPNP_STAMPS
  if (tid == 0) { for (int k_ = 0; k_ < 6; ++k_) atomicAdd(&g_upnp_stamps[k_], L.st[k_]); atomicAdd(&g_upnp_stamps[6], 1ull); }
#endif
  // ---- bottom-up over the rows 140 .. 121 (gauss_jordan.h:152-193), columns 121 .. 148 (the others are never read again)
#pragma unroll
  for (int i = 0; i < kNRL; ++i) {
    if (mypos[i] >= 121 && mypos[i] <= 140) {
#pragma unroll
      for (int j = 1; j < 5; ++j) {
        const int col = kCG * (kMaxShift + j) + g;
        if (col >= 121 && col <= 148) L.blk[mypos[i] - 121][col - 121] = a[i][j];
      }
    }
  }
  __syncthreads();
  for (int cur = 19; cur >= 1; --cur) {
    double nv[2] = {0.0, 0.0};
    bool upd[2] = {false, false};
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int e = tid + t * kThreads;
      if (e < cur * 28) {
        const int r = e / 28, c = e % 28;
        const double l = L.blk[r][cur];
        if (!(fabs(l) < 1e-9)) { nv[t] = L.blk[r][c] - l * L.blk[cur][c]; upd[t] = true; }
      }
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int e = tid + t * kThreads;
      if (upd[t]) L.blk[e / 28][e % 28] = nv[t];
    }
    __syncthreads();
  }
  if (tid < 32) w[kWsAct + tid] = 0.0 - L.blk[tid / 8][20 + tid % 8];   // action(r, s) -= template(121 + r, 141 + s)
}

// ---------------------------------------------------------------- eigenvectors -> rotations -> poses -> models
constexpr int kTeam = 8, kTeamsPerWave = 64 / kTeam;
__global__ __launch_bounds__(64) void k_upnp_b(size_t nhyp, int B, const int64_t* __restrict__ offsets, const double* __restrict__ data,
                                               const int* __restrict__ samples, const int* __restrict__ active_iters,
                                               const double* __restrict__ ws, double* __restrict__ models, int* __restrict__ counts,
                                               int* __restrict__ dense_count, int* __restrict__ tags, int* __restrict__ hyp_base) {
  __shared__ double lds[kTeamsPerWave][64 * 3 + 24];   // H | V | X | wr | wi | ort
  const int team = threadIdx.x / kTeam, tl = threadIdx.x % kTeam;
  const size_t hyp = (size_t)blockIdx.x * kTeamsPerWave + team;
  if (hyp >= nhyp) return;
  const int p = (int)(hyp / B), b = (int)(hyp % B);
  if (b >= active_iters[p]) { if (tl == 0) counts[hyp] = 0; return; }
  const double* w = ws + hyp * kWs;
  double* H = lds[team]; double* V = H + 64; double* X = V + 64; double* wr = X + 64; double* wi = wr + 8; double* ort = wi + 8;
  for (int e = tl; e < 64; e += kTeam) {   // rows 0 .. 3 from the elimination, rows 4 .. 7 the shift (:2488-2499)
    const int r = e / 8, c = e % 8;
    H[e] = r < 4 ? w[kWsAct + e] : (c == r - 4 ? 1.0 : 0.0);
  }
  rsc::team_sync();
  const bool good = rsc::eig_team<kTeam, true>(8, H, V, X, wr, wi, ort, tl);
  rsc::team_sync();
  if (tl != 0) return;
  if (!good) { counts[hyp] = 0; return; }
  // upnp.cc:411-437 with EigenSolver::eigenvectors(): a column is "real" when |imag| <= 1e-12 |real| (or it is the last one),
  // otherwise columns j, j + 1 are re +- i im; each is normalised, then the real parts of rows 4 .. 7 make the quaternion
  double cand[32];
  for (int j = 0; j < 8; ++j) {
    const bool real = fabs(wi[j]) <= fabs(wr[j]) * 1e-12 || j + 1 == 8;
    double nrm2 = 0.0;
    for (int i = 0; i < 8; ++i) nrm2 += real ? V[8 * i + j] * V[8 * i + j] : V[8 * i + j] * V[8 * i + j] + V[8 * i + j + 1] * V[8 * i + j + 1];
    const double nrm = sqrt(nrm2);
    double q[4];
    for (int k = 0; k < 4; ++k) q[k] = V[8 * (4 + k) + j] / nrm;
    const double qn = sqrt(((q[1] * q[1] + q[2] * q[2]) + q[3] * q[3]) + q[0] * q[0]);
    for (int k = 0; k < 4; ++k) cand[4 * j + k] = q[k] / qn;
    if (!real) { for (int k = 0; k < 4; ++k) cand[4 * (j + 1) + k] = cand[4 * j + k]; ++j; }
  }
  const double kAngleThreshold = 0.1 * (3.14159265358979323846 / 180.0);   // RemoveDuplicateRotations (:318-351)
  double quats[32];
  int nq = 0;
  for (int i = 0; i < 8; ++i) {
    bool dup = false;
    for (int j = nq - 1; j >= 0 && !dup; --j) {
      const double* qa = cand + 4 * i; const double* qb = quats + 4 * j;   // angularDistance: a * conj(b)
      const double d0 = qa[0] * qb[0] - qa[1] * -qb[1] - qa[2] * -qb[2] - qa[3] * -qb[3];
      const double d1 = qa[0] * -qb[1] + qa[1] * qb[0] + qa[2] * -qb[3] - qa[3] * -qb[2];
      const double d2 = qa[0] * -qb[2] + qa[2] * qb[0] + qa[3] * -qb[1] - qa[1] * -qb[3];
      const double d3 = qa[0] * -qb[3] + qa[3] * qb[0] + qa[1] * -qb[2] - qa[2] * -qb[1];
      dup = 2.0 * atan2(sqrt((d1 * d1 + d2 * d2) + d3 * d3), fabs(d0)) < kAngleThreshold;
    }
    if (!dup) { for (int k = 0; k < 4; ++k) quats[4 * nq + k] = cand[4 * i + k]; ++nq; }
  }
  const double* pd = data + (size_t)offsets[p] * kDatum;
  double o[4][3], f[4][3], Xw[4][3];
  for (int i = 0; i < 4; ++i) {
    const double* d = pd + (size_t)samples[hyp * 4 + i] * kDatum;
    for (int k = 0; k < 3; ++k) { f[i][k] = d[k]; Xw[i][k] = d[3 + k] / d[6]; o[i][k] = d[9 + k]; }
  }
  double keepq[32], keept[24];
  int nm = 0;
  for (int s = 0; s < nq; ++s) {
    const double* q = quats + 4 * s;
    double t[3] = {0, 0, 0};
    for (int i = 0; i < 4; ++i) {   // ComputeTranslation (:218-227)
      double rx[3], d[3];
      quat_rotate(q, Xw[i], rx);
      for (int k = 0; k < 3; ++k) d[k] = rx[k] - o[i][k];
      const double* Vi = w + kWsV + 9 * i;
      for (int r = 0; r < 3; ++r) t[r] += (Vi[3 * r] * d[0] + Vi[3 * r + 1] * d[1]) + Vi[3 * r + 2] * d[2];
    }
    bool front = true;
    for (int i = 0; i < 4 && front; ++i) {   // DiscardBadSolutions (:286-343): the point, seen along its ray, has z >= 0
      double rx[3], pt[3];
      quat_rotate(q, Xw[i], rx);
      for (int k = 0; k < 3; ++k) pt[k] = (rx[k] + t[k]) - o[i][k];
      const double fn = sqrt((f[i][0] * f[i][0] + f[i][1] * f[i][1]) + f[i][2] * f[i][2]);
      const double v0[3] = {f[i][0] / fn, f[i][1] / fn, f[i][2] / fn};
      const double c = v0[2];
      double u[4];
      if (c < -1.0 + 1e-12) {
        const double w2 = (1.0 + c) * 0.5;
        u[0] = sqrt(w2 > 0 ? w2 : 0.0);
        u[1] = sqrt(1.0 - w2); u[2] = 0.0; u[3] = 0.0;
      } else {
        const double sq = sqrt((1.0 + c) * 2.0), invs = 1.0 / sq;
        u[1] = v0[1] * invs; u[2] = -v0[0] * invs; u[3] = 0.0 * invs; u[0] = sq * 0.5;
      }
      double rp[3];
      quat_rotate(u, pt, rp);
      if (rp[2] < 0) front = false;
    }
    if (front) { for (int k = 0; k < 4; ++k) keepq[4 * nm + k] = q[k]; for (int k = 0; k < 3; ++k) keept[3 * nm + k] = t[k]; ++nm; }
  }
  counts[hyp] = nm;
  if (nm == 0) return;
  const int base = atomicAdd(&dense_count[p], nm);
  hyp_base[hyp] = base;
  double* mo = models + ((size_t)p * B * kMaxModels + base) * (size_t)THEIA_RANSAC_MODEL_STRIDE;
  int* tg = tags + (size_t)p * B * kMaxModels + base;
  for (int j = 0; j < nm; ++j) {
    double* m = mo + (size_t)j * THEIA_RANSAC_MODEL_STRIDE;
    rsc::quat_to_rot(keepq + 4 * j, m);
    for (int k = 0; k < 3; ++k) m[9 + k] = keept[3 * j + k];
    for (int k = 12; k < THEIA_RANSAC_MODEL_STRIDE; ++k) m[k] = 0.0;
    tg[j] = b * kMaxModels + j;
  }
}

// the tables from the layout (the oracle builds the same ones from the same header: oracle/upnp_oracle.h)
void build_tables(Tables* t) {
  static const int kS[10][4] = {{2, 0, 0, 0}, {0, 2, 0, 0}, {0, 0, 2, 0}, {0, 0, 0, 2}, {1, 1, 0, 0}, {1, 0, 1, 0}, {1, 0, 0, 1}, {0, 1, 1, 0}, {0, 1, 0, 1}, {0, 0, 1, 1}};
  int mono[24][4];
  int n = 0;
  for (int e3 = 0; e3 <= 3; ++e3) for (int e2 = 0; e2 + e3 <= 3; ++e2) for (int e1 = 0; e1 + e2 + e3 <= 3; ++e1) {
    mono[n][0] = 3 - e1 - e2 - e3; mono[n][1] = e1; mono[n][2] = e2; mono[n][3] = e3; ++n;
  }
  for (int k = 0; k < 4; ++k) for (int x = 0; x < 4; ++x) mono[20 + k][x] = x == k;
  std::memset(t, 0, sizeof(*t));
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 24; ++j) {
      int cnt = 0;
      const int e = 24 * i + j;
      for (int q = 9; q >= 0; --q)
        for (int p = 0; p < 10; ++p) {
          if (kS[p][i] == 0) continue;
          bool hit = true;
          for (int x = 0; x < 4; ++x) hit = hit && (kS[p][x] - (x == i) + (j < 20 ? kS[q][x] : 0) == mono[j][x]);
          if (j >= 20 && q != 9) hit = false;   // the linear columns have one term per p
          if (hit) { t->coef[e][cnt] = (uint8_t)(2 * kS[p][i]); t->tq[e][cnt] = (uint8_t)q; t->tp[e][cnt] = (uint8_t)p; ++cnt; }
        }
      t->nterm[e] = (uint8_t)cnt;
    }
  for (int i = 0; i < 4; ++i) {
    for (int k = 0; k < 4; ++k) {
      int ex[4] = {0, 0, 0, 0}; ex[k] += 2; ex[i] += 1;
      for (int j = 0; j < 20; ++j) if (mono[j][0] == ex[0] && mono[j][1] == ex[1] && mono[j][2] == ex[2] && mono[j][3] == ex[3]) t->fixed[24 * (4 + i) + j] = 1.0;
    }
    t->fixed[24 * (4 + i) + 20 + i] = -1.0;
  }
  std::memset(t->src, 255, sizeof(t->src));
  for (int r = 0; r < upnp_layout::kRows; ++r) {
    const int eq = upnp_layout::kRowEq[r];
    for (int j = 0; j < 24; ++j) {
      const bool in = eq == 7 ? (j == 10 || j == 12 || j == 15 || j == 19 || j == 23) : (j == eq || (j >= 7 && j <= 9) || j >= 11);
      if (!in) continue;
      for (int c = 0; c < upnp_layout::kCols; ++c) {
        bool same = true;
        for (int x = 0; x < 4; ++x) same = same && upnp_layout::kColMono[c][x] == upnp_layout::kRowMul[r][x] + mono[j][x];
        if (same) { t->src[r][c] = (uint8_t)(eq * 24 + j); break; }
      }
    }
  }
}

}  // namespace upnpdev

int upnp_ensure_tables() {
  static std::once_flag once;
  static int rc = 0;
  std::call_once(once, [] {
    static upnpdev::Tables t;
    upnpdev::build_tables(&t);
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(upnpdev::g_tab), &t, sizeof(t));
    if (e != hipSuccess) rc = set_error(THEIA_HIP_ERR_NO_DEVICE, "hipMemcpyToSymbol(upnp tables) failed: %s", hipGetErrorString(e));
  });
  return rc;
}
int upnp_workspace_doubles() { return upnpdev::kWs; }
int upnp_state_doubles() { return upnpdev::kStateDoubles; }

void launch_upnp_fit(int nprob, int B, const int64_t* offsets, const double* data, const int* samples, const int* active_iters,
                     double* state, double* ws, double* models, int* counts, int* dense_count, int* tags, int* hyp_base, hipStream_t st) {
  using namespace upnpdev;
  const size_t nh = (size_t)nprob * B;
  k_upnp_pre<<<dim3((B + 63) / 64, nprob), 64, 0, st>>>(nprob, B, offsets, data, samples, active_iters, ws);
  k_upnp_scan<<<nprob, 128, 0, st>>>(B, active_iters, state, ws);
  k_upnp_a<<<dim3(B, nprob), kThreads, 0, st>>>(B, active_iters, ws);
  k_upnp_b<<<(unsigned)((nh + kTeamsPerWave - 1) / kTeamsPerWave), 64, 0, st>>>(nh, B, offsets, data, samples, active_iters, ws, models, counts,
                                                                              dense_count, tags, hyp_base);
}

}  // namespace thip

#ifdef THIP_UPNP_STAMPS
extern "C" int theia_hip_debug_upnp_stamps(unsigned long long* out) {
  unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(thip::upnpdev::g_upnp_stamps), sizeof(z)) != hipSuccess) return 1;
  return hipMemcpyToSymbol(HIP_SYMBOL(thip::upnpdev::g_upnp_stamps), z, sizeof(z)) != hipSuccess;
}
#endif
