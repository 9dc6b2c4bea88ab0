// cholesky_device.h -- device code shared by the dense and the tile-sparse
// Cholesky of the reduced camera system (K3): the one-wavefront 64x64 diagonal
// block factorisation + inverse, and the FP64 MFMA helpers.
#pragma once
#include <hip/hip_runtime.h>

namespace thip {
namespace chol {

constexpr int NB = 64;
constexpr int LDP = NB + 1;  // LDS row pitch (doubles)

typedef double double4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double readlane_d(double v, int l) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, l);
  hi = __builtin_amdgcn_readlane(hi, l);
  return __hiloint2double(hi, lo);
}

// 32 x 32 x 32 product on the FP64 matrix core by ONE wavefront, operands in LDS:
//   C[i][j] (+)= sum_k Aop(i,k) * Bop(k,j),  i,j,k in [0,32)
// a-lane = A[i = l&15][k = l>>4], b-lane = B[k = l>>4][j = l&15];
// C/D: col = l&15, row = (l>>4) + 4*reg (f64 layout).  Result tiles acc[ti][tj].
template <typename FA, typename FB>
__device__ __forceinline__ void wave_gemm32(FA Aop, FB Bop, double4_t (&acc)[2][2], int lane) {
  const int li = lane & 15, lk = lane >> 4;
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) acc[ti][tj] = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int kk = 0; kk < 32; kk += 4) {
    const double a0 = Aop(li, kk + lk), a1 = Aop(16 + li, kk + lk);
    const double b0 = Bop(kk + lk, li), b1 = Bop(kk + lk, 16 + li);
    acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
  }
}

// 16 x 16 x 16 product on the matrix core (4 MFMAs), accumulated onto c.
template <typename FA, typename FB>
__device__ __forceinline__ double4_t wave_gemm16(FA Aop, FB Bop, double4_t c, int lane) {
  const int li = lane & 15, lk = lane >> 4;
#pragma unroll
  for (int kk = 0; kk < 16; kk += 4) c = __builtin_amdgcn_mfma_f64_16x16x4f64(Aop(li, kk + lk), Bop(kk + lk, li), c, 0, 0, 0);
  return c;
}

// One wavefront: factor the diagonal block at k0 (nb <= 64 valid rows; the rest
// is padded with the identity), write L11 back, and write L11^-1 (64 x 64,
// row-major, zero upper part) to Linv.
//
// The block lives in LDS; lane i owns row i.  Right-looking over four 16-wide
// panels: the lane loads its 16 panel entries into registers, a rank-1 sweep
// (pivot row broadcast with v_readlane, one rsqrt per column) produces the
// diagonal 16x16 factor AND the rows below it at once, and the trailing
// 16x16 tiles are updated on the FP64 matrix core.  L^-1 is built bottom-up:
// the four 16x16 triangular inverses by quarter-waves, then
//   [A 0; B C]^-1 = [A^-1 0; -C^-1 B A^-1  C^-1]
// at 32 and at 64 with MFMA products.  The code is small (16-wide unrolling
// only): a single wave executes it once per launch, so instruction fetch and
// register spills, not arithmetic, were what the older 32-wide version paid for.
__device__ __forceinline__ void potrf64_wave(double* __restrict__ A, int lda, int k0, int nb,
                                             double* __restrict__ Linv, double* __restrict__ fail_flag) {
  __shared__ double Ls[NB][LDP];
  __shared__ double Zs[NB][LDP];
  __shared__ double rdiag[NB];
  const int i = threadIdx.x;
  const int li = i & 15, lk = i >> 4;
#ifdef THIP_POTRF_STAMPS
  long long tq_ = clock64();
#define PSTAMP(k_) do { const long long tn_ = clock64(); if (i == 0) THIP_POTRF_STAMPS[k_] = tn_ - tq_; tq_ = tn_; } while (0)
  long long tsub_[3] = {0, 0, 0}, tp_ = 0;
#define PSUB(k_) do { const long long tn_ = clock64(); tsub_[k_] += tn_ - tp_; tp_ = tn_; } while (0)
#define PSUB0() do { tp_ = clock64(); } while (0)
#else
#define PSTAMP(k_) do {} while (0)
#define PSUB(k_) do {} while (0)
#define PSUB0() do {} while (0)
#endif
  // coalesced load: one 512-B row per instruction.  The loads are UNCONDITIONAL
  // (clamped address, value selected afterwards): a predicated load compiles to
  // branch + load + s_waitcnt per row, i.e. 64 serialised memory round trips.
  {
    // all 64 row loads are in flight before the first use: one exposed memory latency
    const int ic = min(i, nb - 1);
    double v[NB];
#pragma unroll
    for (int r = 0; r < NB; ++r) v[r] = A[(size_t)(k0 + min(r, nb - 1)) * lda + k0 + ic];
#pragma unroll
    for (int r = 0; r < NB; ++r) {
      Ls[r][i] = (r < nb && i <= r) ? v[r] : ((r == i) ? 1.0 : 0.0);
      Zs[r][i] = 0.0;
    }
  }
  __syncthreads();
  PSTAMP(0);
  int bad = 0;
  for (int p = 0; p < 4; ++p) {
    const int c0 = 16 * p;
    PSUB0();
    double r16[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) r16[j] = Ls[i][c0 + j];
    PSUB(0);
    // right-looking inside the panel with the scaling deferred: the rank-1
    // update of column j uses s_ij s_kj / d_j, so the per-column dependent chain
    // is pivot -> 1/d -> one multiply -> one FMA; the rsqrt needed for
    // L = s / sqrt(d) is computed off that chain and applied at the end.
    double rs[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      double d = readlane_d(r16[j], c0 + j);
      if (!(d > 0.0)) { bad = 1; d = 1.0; }
      // the row's entry over the pivot first, then ONE FMA per remaining column (s_ij / d_j) s_kj -- the older form
      // (s_ij s_kj) / d_j took a multiply and an FMA per entry and step, and a wave issues one FP64 instruction per ~6 cycles
      // whatever its parallelism: the sweep is bound by its instruction count.  Chain per column:
      // pivot -> rcp -> one Newton step (v_rcp_f64 carries ~26 bits, one step leaves < 2 ulp) -> multiply -> FMA
      double w = __builtin_amdgcn_rcp(d);
      w = __builtin_fma(w, __builtin_fma(-d, w, 1.0), w);
      const double t = -(r16[j] * w);
#pragma unroll
      for (int k = j + 1; k < 16; ++k) r16[k] = __builtin_fma(t, readlane_d(r16[j], c0 + k), r16[k]);
      rs[j] = d;
    }
    // 16 independent rsqrt refinements (v_rsq_f64 seed + two Newton steps, full
    // FP64, no sqrt / divide): kept out of the in-order sweep above
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const double d = rs[j];
      double rinv = __builtin_amdgcn_rsq(d);
      rinv = rinv * (1.5 - (0.5 * d) * (rinv * rinv));
      rinv = rinv * (1.5 - (0.5 * d) * (rinv * rinv));
      rs[j] = rinv;
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (i == 0) rdiag[c0 + j] = rs[j];
      r16[j] = (i >= c0 + j) ? r16[j] * rs[j] : 0.0;
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) Ls[i][c0 + j] = r16[j];
    __syncthreads();
    PSUB(1);
    // trailing tiles (ti >= tj > p):  C -= L[ti][p] L[tj][p]^T
    for (int ti = p + 1; ti < 4; ++ti)
      for (int tj = p + 1; tj <= ti; ++tj) {
        double4_t c;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) c[reg] = Ls[16 * ti + lk + 4 * reg][16 * tj + li];
        c = wave_gemm16([&](int r, int k) { return -Ls[16 * ti + r][c0 + k]; },
                        [&](int k, int cc) { return Ls[16 * tj + cc][c0 + k]; }, c, i);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) Ls[16 * ti + lk + 4 * reg][16 * tj + li] = c[reg];
      }
    __syncthreads();
    PSUB(2);
  }
  PSTAMP(1);
#ifdef THIP_POTRF_STAMPS
  if (i == 0) { THIP_POTRF_STAMPS[8] = tsub_[0]; THIP_POTRF_STAMPS[9] = tsub_[1]; THIP_POTRF_STAMPS[10] = tsub_[2]; }
#endif
  // Whole rows are stored (the zeros of Ls above the diagonal land in the tile's upper triangle, which
  // nothing reads: S is symmetric-lower and rebuilt by every linearisation), so the only predicates are
  // the lane mask i < nb (hoisted) and the uniform r < nb; LDS reads are batched ahead of the stores.
  if (i < nb) {
#pragma unroll
    for (int rb = 0; rb < NB; rb += 16) {
      double v[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) v[q] = Ls[rb + q][i];
#pragma unroll
      for (int q = 0; q < 16; ++q)
        if (rb + q < nb) A[(size_t)(k0 + rb + q) * lda + k0 + i] = v[q];
    }
  }
  if (bad && i == 0) unsafeAtomicAdd(fail_flag, 1.0);
  PSTAMP(2);
  // ---- inverse: four 16 x 16 triangular inverses, lane (q, c) owns column c of block q
  {
    const int q = i >> 4, c = i & 15;
    double z[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      double s0 = (r == c) ? 1.0 : 0.0, s1 = 0.0;
#pragma unroll
      for (int k = 0; k < r; ++k) {
        const double t = Ls[16 * q + r][16 * q + k] * z[k];
        if (k & 1) s1 -= t; else s0 -= t;
      }
      z[r] = (s0 + s1) * rdiag[16 * q + r];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) Zs[16 * q + r][16 * q + c] = z[r];
  }
  __syncthreads();
  PSTAMP(3);
  // ---- 32-level: Z[q1][q0] = -Z[q1][q1] (L[q1][q0] Z[q0][q0]) for (q0, q1) = (0, 1), (2, 3);
  // the product T is parked in the (zero, unused) upper part of Ls
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int q0 = 2 * h, q1 = 2 * h + 1;
    double4_t t = (double4_t){0.0, 0.0, 0.0, 0.0};
    t = wave_gemm16([&](int r, int k) { return Ls[16 * q1 + r][16 * q0 + k]; },
                    [&](int k, int cc) { return Zs[16 * q0 + k][16 * q0 + cc]; }, t, i);
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) Ls[16 * q0 + lk + 4 * reg][16 * q1 + li] = t[reg];
  }
  __syncthreads();
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int q0 = 2 * h, q1 = 2 * h + 1;
    double4_t t = (double4_t){0.0, 0.0, 0.0, 0.0};
    t = wave_gemm16([&](int r, int k) { return Zs[16 * q1 + r][16 * q1 + k]; },
                    [&](int k, int cc) { return Ls[16 * q0 + k][16 * q1 + cc]; }, t, i);
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) Zs[16 * q1 + lk + 4 * reg][16 * q0 + li] = -t[reg];
  }
  __syncthreads();
  PSTAMP(4);
  // ---- 64-level: Z21 = -Z22 (L21 Z11), 32 x 32 blocks; T parked in Ls[0..31][32..63]
  {
    double4_t acc[2][2];
    wave_gemm32([&](int r, int k) { return Ls[32 + r][k]; }, [&](int k, int cc) { return Zs[k][cc]; }, acc, i);
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
      for (int tj = 0; tj < 2; ++tj)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) Ls[16 * ti + lk + 4 * reg][32 + 16 * tj + li] = acc[ti][tj][reg];
    __syncthreads();
    wave_gemm32([&](int r, int k) { return Zs[32 + r][32 + k]; }, [&](int k, int cc) { return Ls[k][32 + cc]; }, acc, i);
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
      for (int tj = 0; tj < 2; ++tj)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) Zs[32 + 16 * ti + lk + 4 * reg][16 * tj + li] = -acc[ti][tj][reg];
  }
  __syncthreads();
  PSTAMP(5);
#pragma unroll 16
  for (int r = 0; r < NB; ++r) Linv[r * NB + i] = Zs[r][i];
  PSTAMP(6);
#undef PSTAMP
#undef PSUB
#undef PSUB0
}


// The same factorisation + inverse by a 256-thread workgroup (4 waves).  The in-panel sweep is inherently
// one wave's work (wave 0); everything else is spread over the four waves: the tile load and the stores
// (16 rows per wave), the trailing 16 x 16 MFMA updates of a panel (6, 3, 1 tiles: one tile per wave and
// round), the two 32-level inverse products (one per wave) and the four 16 x 16 tiles of each 64-level
// product.  Same arithmetic per entry as potrf64_wave (the MFMA tiles are the same tiles), so the two
// functions return identical bits.
// potrf64_wg_core: the work arrays are the caller's (Ls ends as scratch, Zs holds the inverse factor when it returns,
// after a workgroup barrier).  STORE_L = false leaves A untouched (k_sp_potrf_trsm: other workgroups factor the same tile at
// the same time); Linv / fail_flag may be null.
template <bool STORE_L>
__device__ __forceinline__ void potrf64_wg_core(double* __restrict__ A, int lda, int k0, int nb, double (*Ls)[LDP], double (*Zs)[LDP],
                                                double* rdiag, double* __restrict__ Linv, double* __restrict__ fail_flag) {
  const int tid = threadIdx.x, wv = tid >> 6, i = tid & 63;
  const int li = i & 15, lk = i >> 4;
  {
    // wave w loads rows 16w .. 16w+15 (unconditional clamped loads, all in flight)
    const int ic = min(i, nb - 1);
    double v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = A[(size_t)(k0 + min(16 * wv + q, nb - 1)) * lda + k0 + ic];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int r = 16 * wv + q;
      Ls[r][i] = (r < nb && i <= r) ? v[q] : ((r == i) ? 1.0 : 0.0);
      Zs[r][i] = 0.0;
    }
  }
  __syncthreads();
  int bad = 0;
  for (int p = 0; p < 4; ++p) {
    const int c0 = 16 * p;
    if (wv == 0) {
      double r16[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) r16[j] = Ls[i][c0 + j];
      double rs[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        // column j goes through LDS: every lane stores its entry, the pivot and the 15 - j multipliers come back as
        // broadcast reads (one ds_read_b64 each, issued back to back) instead of two v_readlane + hazard slots per value
        Ls[i][c0 + j] = r16[j];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        double d = Ls[c0 + j][c0 + j];   // (the pivot by v_readlane instead, to start 1 / d early: measured slower, 33.9 k against 33.4 k cycles)
        if (!(d > 0.0)) { bad = 1; d = 1.0; }
        // (s_ij / d_j) s_kj: one FMA per entry and step (see potrf64_wave); the multipliers are read while 1 / d is computed
        double mk[16];
#pragma unroll
        for (int k = j + 1; k < 16; ++k) mk[k] = Ls[c0 + k][c0 + j];
        double w = __builtin_amdgcn_rcp(d);
        w = __builtin_fma(w, __builtin_fma(-d, w, 1.0), w);
        const double t = -(r16[j] * w);
#pragma unroll
        for (int k = j + 1; k < 16; ++k) r16[k] = __builtin_fma(t, mk[k], r16[k]);
        rs[j] = d;
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const double d = rs[j];
        double rinv = __builtin_amdgcn_rsq(d);
        rinv = rinv * (1.5 - (0.5 * d) * (rinv * rinv));
        rinv = rinv * (1.5 - (0.5 * d) * (rinv * rinv));
        rs[j] = rinv;
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (i == 0) rdiag[c0 + j] = rs[j];
        r16[j] = (i >= c0 + j) ? r16[j] * rs[j] : 0.0;
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) Ls[i][c0 + j] = r16[j];
    }
    __syncthreads();
    // trailing tiles (ti >= tj > p), one per wave and round:  C -= L[ti][p] L[tj][p]^T
    {
      int t = 0;
      for (int ti = p + 1; ti < 4; ++ti)
        for (int tj = p + 1; tj <= ti; ++tj, ++t) {
          if ((t & 3) != wv) continue;
          double4_t c;
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) c[reg] = Ls[16 * ti + lk + 4 * reg][16 * tj + li];
          c = wave_gemm16([&](int r, int k) { return -Ls[16 * ti + r][c0 + k]; },
                          [&](int k, int cc) { return Ls[16 * tj + cc][c0 + k]; }, c, i);
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) Ls[16 * ti + lk + 4 * reg][16 * tj + li] = c[reg];
        }
    }
    __syncthreads();
  }
  // L back to global: wave w stores rows 16w .. 16w+15 (whole rows, see potrf64_wave)
  if (STORE_L && i < nb) {
    double v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = Ls[16 * wv + q][i];
#pragma unroll
    for (int q = 0; q < 16; ++q)
      if (16 * wv + q < nb) A[(size_t)(k0 + 16 * wv + q) * lda + k0 + i] = v[q];
  }
  if (bad && tid == 0 && fail_flag) unsafeAtomicAdd(fail_flag, 1.0);
  // ---- inverse: the four 16 x 16 triangular inverses (wave 0, lane (q, c) owns column c of block q)
  if (wv == 0) {
    const int q = i >> 4, c = i & 15;
    double z[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      double s0 = (r == c) ? 1.0 : 0.0, s1 = 0.0;
#pragma unroll
      for (int k = 0; k < r; ++k) {
        const double t = Ls[16 * q + r][16 * q + k] * z[k];
        if (k & 1) s1 -= t; else s0 -= t;
      }
      z[r] = (s0 + s1) * rdiag[16 * q + r];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) Zs[16 * q + r][16 * q + c] = z[r];
  }
  __syncthreads();
  // ---- 32-level (waves 0 and 1, one diagonal pair each): Z[q1][q0] = -Z[q1][q1] (L[q1][q0] Z[q0][q0])
  if (wv < 2) {
    const int q0 = 2 * wv, q1 = 2 * wv + 1;
    double4_t t = (double4_t){0.0, 0.0, 0.0, 0.0};
    t = wave_gemm16([&](int r, int k) { return Ls[16 * q1 + r][16 * q0 + k]; },
                    [&](int k, int cc) { return Zs[16 * q0 + k][16 * q0 + cc]; }, t, i);
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) Ls[16 * q0 + lk + 4 * reg][16 * q1 + li] = t[reg];
  }
  __syncthreads();
  if (wv < 2) {
    const int q0 = 2 * wv, q1 = 2 * wv + 1;
    double4_t t = (double4_t){0.0, 0.0, 0.0, 0.0};
    t = wave_gemm16([&](int r, int k) { return Zs[16 * q1 + r][16 * q1 + k]; },
                    [&](int k, int cc) { return Ls[16 * q0 + k][16 * q1 + cc]; }, t, i);
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) Zs[16 * q1 + lk + 4 * reg][16 * q0 + li] = -t[reg];
  }
  __syncthreads();
  // ---- 64-level: Z21 = -Z22 (L21 Z11); wave w owns the 16 x 16 tile (ti, tj) = (w >> 1, w & 1) of both products
  {
    const int ti = wv >> 1, tj = wv & 1;
    double4_t acc = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 32; kk += 4)
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Ls[32 + 16 * ti + li][kk + lk], Zs[kk + lk][16 * tj + li], acc, 0, 0, 0);
    __syncthreads();   // every wave has read its L21 rows / Z11 columns before T overwrites Ls[0..31][32..63]
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) Ls[16 * ti + lk + 4 * reg][32 + 16 * tj + li] = acc[reg];
    __syncthreads();
    acc = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 32; kk += 4)
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Zs[32 + 16 * ti + li][32 + kk + lk], Ls[kk + lk][32 + 16 * tj + li], acc, 0, 0, 0);
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) Zs[32 + 16 * ti + lk + 4 * reg][16 * tj + li] = -acc[reg];
  }
  __syncthreads();
  if (Linv) {
#pragma unroll
    for (int q = 0; q < 16; ++q) Linv[(16 * wv + q) * NB + i] = Zs[16 * wv + q][i];
  }
}

__device__ __forceinline__ void potrf64_wg(double* __restrict__ A, int lda, int k0, int nb,
                                           double* __restrict__ Linv, double* __restrict__ fail_flag) {
  __shared__ double Ls[NB][LDP];
  __shared__ double Zs[NB][LDP];
  __shared__ double rdiag[NB];
  potrf64_wg_core<true>(A, lda, k0, nb, Ls, Zs, rdiag, Linv, fail_flag);
}

}  // namespace chol
}  // namespace thip
