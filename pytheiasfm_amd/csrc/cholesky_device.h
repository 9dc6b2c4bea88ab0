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

// column-Crout over columns [J0, J0 + 32) of the rows held one per lane; the dot
// products run over k in [K0, j).  rdiag[j] = 1 / L[j][j].
template <int J0, int K0>
__device__ __forceinline__ void crout32(double (&row)[NB], double (&rdiag)[NB], int i, int* bad) {
#pragma unroll
  for (int j = J0; j < J0 + 32; ++j) {
    double s[8];
    s[0] = row[j];
#pragma unroll
    for (int q = 1; q < 8; ++q) s[q] = 0.0;
#pragma unroll
    for (int k = K0; k < j; ++k) s[(k - K0) & 7] -= row[k] * readlane_d(row[k], j);
    const double sj = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    double d = readlane_d(sj, j);
    if (!(d > 0.0)) { *bad = 1; d = 1.0; }
    // hardware v_rsq_f64 seed + two Newton steps (full FP64), no sqrt / divide
    double rinv = __builtin_amdgcn_rsq(d);
    rinv = rinv * (1.5 - (0.5 * d) * (rinv * rinv));
    rinv = rinv * (1.5 - (0.5 * d) * (rinv * rinv));
    rdiag[j] = rinv;
    row[j] = (i == j) ? d * rinv : (i > j ? sj * rinv : 0.0);
  }
}

// One wavefront: factor the diagonal block at k0 (nb <= 64 valid rows; the rest
// is padded with the identity), write L11 back, and write L11^-1 (64 x 64,
// row-major, zero upper part) to Linv.  Two-level recursion on 32-blocks
//   L = [A 0; B C]:  Crout on columns 0..31 (all 64 lanes: A and B = A21 A^-T),
//   C -= B B^T on the matrix core, Crout on columns 32..63 (dot products over
//   32..j only);  L^-1 = [A^-1 0; -C^-1 B A^-1  C^-1] with the two 32x32
//   triangular inverses formed by the two half-waves at once.
__device__ __forceinline__ void potrf64_wave(double* __restrict__ A, int lda, int k0, int nb,
                                             double* __restrict__ Linv, double* __restrict__ fail_flag) {
  __shared__ double Ls[NB][LDP];
  __shared__ double Zs[NB][LDP];
  const int i = threadIdx.x;
  // coalesced load: one 512-B row per instruction, then lane i picks up row i
#pragma unroll 16
  for (int r = 0; r < NB; ++r)
    Ls[r][i] = (r < nb && i <= r) ? A[(size_t)(k0 + r) * lda + k0 + i] : ((r == i) ? 1.0 : 0.0);
  __syncthreads();
  double row[NB], rdiag[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) row[j] = Ls[i][j];
  int bad = 0;
  crout32<0, 0>(row, rdiag, i, &bad);
  // B = rows 32..63, columns 0..31 -> LDS;  C -= B B^T
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 32; ++j) Ls[i][j] = row[j];
  __syncthreads();
  {
    double4_t acc[2][2];
    wave_gemm32([&](int r, int k) { return Ls[32 + r][k]; }, [&](int k, int c) { return Ls[32 + c][k]; }, acc, i);
    const int li = i & 15, lk = i >> 4;
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
      for (int tj = 0; tj < 2; ++tj)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) Zs[16 * ti + lk + 4 * reg][16 * tj + li] = acc[ti][tj][reg];
  }
  __syncthreads();
  if (i >= 32) {
#pragma unroll
    for (int j = 32; j < NB; ++j) row[j] -= Zs[i - 32][j - 32];
  }
  crout32<32, 32>(row, rdiag, i, &bad);
  if (i < nb) {
#pragma unroll
    for (int j = 0; j < NB; ++j) if (j <= i) A[(size_t)(k0 + i) * lda + k0 + j] = row[j];
  }
  if (bad && i == 0) unsafeAtomicAdd(fail_flag, 1.0);
  __syncthreads();
#pragma unroll
  for (int j = 32; j < NB; ++j) Ls[i][j] = row[j];
  __syncthreads();
  // triangular inverses of A (lanes 0..31) and C (lanes 32..63): lane owns a column
  const int h = i >> 5, c = i & 31;
  double z[32];
#pragma unroll
  for (int r = 0; r < 32; ++r) {
    double s[4];
    s[0] = (r == c) ? 1.0 : 0.0;
    s[1] = s[2] = s[3] = 0.0;
#pragma unroll
    for (int k = 0; k < r; ++k) s[k & 3] -= Ls[32 * h + r][32 * h + k] * z[k];
    const double sr = (s[0] + s[1]) + (s[2] + s[3]);
    z[r] = sr * (h ? rdiag[32 + r] : rdiag[r]);
  }
  // Zs <- [A^-1 0; 0 C^-1]
#pragma unroll
  for (int r = 0; r < 32; ++r) { Zs[32 * h + r][32 * h + c] = z[r]; Zs[32 * h + r][32 * (1 - h) + c] = 0.0; }
  __syncthreads();
  {
    // T = B A^-1  (B in Ls[32+r][k]),  Z21 = -C^-1 T
    double4_t acc[2][2];
    wave_gemm32([&](int r, int k) { return Ls[32 + r][k]; }, [&](int k, int cc) { return Zs[k][cc]; }, acc, i);
    const int li = i & 15, lk = i >> 4;
    __syncthreads();
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
      for (int tj = 0; tj < 2; ++tj)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) Ls[16 * ti + lk + 4 * reg][16 * tj + li] = acc[ti][tj][reg];  // T over the dead A
    __syncthreads();
    wave_gemm32([&](int r, int k) { return Zs[32 + r][32 + k]; }, [&](int k, int cc) { return Ls[k][cc]; }, acc, i);
    __syncthreads();
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
      for (int tj = 0; tj < 2; ++tj)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) Zs[32 + 16 * ti + lk + 4 * reg][16 * tj + li] = -acc[ti][tj][reg];
  }
  __syncthreads();
#pragma unroll 16
  for (int r = 0; r < NB; ++r) Linv[r * NB + i] = Zs[r][i];
}

}  // namespace chol
}  // namespace thip
