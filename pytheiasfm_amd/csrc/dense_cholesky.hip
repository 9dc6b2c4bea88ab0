// dense_cholesky.hip -- K3: dense SPD solve of the reduced camera system on
// gfx950, FP64.  Replaces Ceres' SPARSE_SCHUR + EIGEN_SPARSE SimplicialLDLT of
// the same matrix (bundle_adjustment.h:98,104; selected in bundle_adjuster.cc:63-89).
//
// Blocked right-looking Cholesky on the lower triangle of a row-major matrix.
// The right-hand side is stored as row n of the same (n+1) x lda array, so the
// forward substitution  y = L^-1 b  falls out of the panel solves for free;
// only the backward substitution  x = L^-T y  is separate.
//
// Everything at this size (n = 6 x #cameras ~ 10^3) is LATENCY bound: measured
// on MI355X a dependent FP64 op costs ~22 cycles and sqrt/div several hundred,
// with one wave per SIMD nothing hides it.  Hence:
//   k_potrf : ONE wavefront factors the 64x64 diagonal block in registers
//             (lane = row, pivot row broadcast by v_readlane, 8 partial sums,
//             one rsqrt per column) and also forms its inverse L11^-1;
//   k_trsm  : A21 <- A21 L11^-T as a GEMM with that inverse on the FP64 matrix
//             core (v_mfma_f64_16x16x4_f64): no dependent chains;
//   k_syrk  : trailing update A22 -= L21 L21^T, 64x64 tiles, FP64 MFMA;
//   backward substitution in 64-wide steps with the same block inverses.
#include "ba_kernels.h"
#include "cholesky_device.h"

#include <algorithm>

namespace thip {
namespace {

using namespace chol;

__global__ __launch_bounds__(256) void k_potrf(double* __restrict__ A, int lda, int k0, int nb,
                                               double* __restrict__ Linv, double* __restrict__ fail_flag) {
  potrf64_wg(A, lda, k0, nb, Linv, fail_flag);
}

// A21 <- A21 * L11^-T for rows [k0 + nb, nrows): X[r][j] = sum_i P[r][i] Z[j][i].
// 128 rows per workgroup; wave w owns rows [32w, 32w + 32).
__global__ __launch_bounds__(256) void k_trsm(double* __restrict__ A, int lda, int nrows, int k0, int nb,
                                              const double* __restrict__ Linv) {
  __shared__ double P[128][LDP];
  __shared__ double Z[NB][LDP];
  const int tid = threadIdx.x;
  const int r0 = k0 + nb + blockIdx.x * 128;
  for (int t = tid; t < NB * NB; t += 256) Z[t / NB][t % NB] = Linv[t];
  for (int t = tid; t < 128 * NB; t += 256) {
    const int rr = t / NB, j = t % NB;
    P[rr][j] = (r0 + rr < nrows && j < nb) ? A[(size_t)(r0 + rr) * lda + k0 + j] : 0.0;
  }
  __syncthreads();
  const int wv = tid >> 6, lane = tid & 63;
  const int li = lane & 15, lk = lane >> 4;
  double4_t acc[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[a][q] = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int kk = 0; kk < NB; kk += 4) {
    const double a0 = P[32 * wv + li][kk + lk];
    const double a1 = P[32 * wv + 16 + li][kk + lk];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const double b = Z[16 * q + li][kk + lk];
      acc[0][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b, acc[0][q], 0, 0, 0);
      acc[1][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b, acc[1][q], 0, 0, 0);
    }
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int row = r0 + 32 * wv + 16 * a + lk + 4 * reg;
        const int col = 16 * q + li;
        if (row < nrows && col < nb) A[(size_t)row * lda + k0 + col] = acc[a][q][reg];
      }
}

// Trailing update with FP64 MFMA.  Tile = 64 x 64 outputs per workgroup of 256
// threads (4 waves); wave w owns output rows [16w, 16w+16) x 64 columns = four
// 16x16 MFMA accumulators.  C[i][j] -= sum_k P[i][k] * P[j][k], k < nb.
//   v_mfma_f64_16x16x4_f64: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15],
//   C/D: col = l&15, row = (l>>4) + 4*reg   (f64 layout, see CDNA guide sec.3)
__global__ __launch_bounds__(256) void k_syrk(double* __restrict__ A, int lda, int nrows, int k0, int nb) {
  const int base = k0 + nb;  // first trailing row/col
  const int ti = blockIdx.y, tj = blockIdx.x;
  if (tj > ti) return;  // lower tiles only
  const int r0 = base + ti * 64, c0 = base + tj * 64;
  __shared__ double Pr[64][LDP];  // rows of the panel for this tile's rows
  __shared__ double Pc[64][LDP];  // rows of the panel for this tile's cols
  const int tid = threadIdx.x;
  const int wv = tid >> 6, lane = tid & 63;
  const int li = lane & 15, lk = lane >> 4;
  // prefetch the C tile (independent of the panel loads) to overlap one memory round trip
  double4_t cold[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int row = r0 + 16 * wv + lk + 4 * reg;
      const int col = c0 + 16 * q + li;
      const bool ok = row < nrows && col <= row && col < nrows - 1;
      cold[q][reg] = ok ? A[(size_t)row * lda + col] : 0.0;
    }
  for (int t = tid; t < 64 * NB; t += 256) {
    const int i = t / NB, k = t % NB;
    const int rr = r0 + i, cc = c0 + i;
    Pr[i][k] = (rr < nrows && k < nb) ? A[(size_t)rr * lda + k0 + k] : 0.0;
    Pc[i][k] = (cc < nrows && k < nb) ? A[(size_t)cc * lda + k0 + k] : 0.0;
  }
  __syncthreads();
  double4_t acc[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) acc[q] = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int kk = 0; kk < NB; kk += 4) {
    const double a = Pr[16 * wv + li][kk + lk];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const double b = Pc[16 * q + li][kk + lk];
      acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[q], 0, 0, 0);
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int row = r0 + 16 * wv + lk + 4 * reg;
      const int col = c0 + 16 * q + li;
      // the rhs row (row == nrows-1) has columns < nrows-1 only
      if (row < nrows && col <= row && col < nrows - 1) A[(size_t)row * lda + col] = cold[q][reg] - acc[q][reg];
    }
}

// Backward substitution x = L^-T y in 64-wide block steps with the block
// inverses saved by k_potrf: x_k = Linv_k^T y_k, then y_j -= L[k-rows][j]^T x_k
// spread over one workgroup per 256 columns.
__global__ __launch_bounds__(256) void k_back_step(const double* __restrict__ A, int lda, int n, int kb,
                                                   const double* __restrict__ Linv, double* __restrict__ y,
                                                   double* __restrict__ x) {
  __shared__ double yk[NB], xk[NB], part[4][NB];
  const int k0 = kb * NB;
  const int nb = min(NB, n - k0);
  const int tid = threadIdx.x;
  if (tid < NB) yk[tid] = (tid < nb) ? y[k0 + tid] : 0.0;
  __syncthreads();
  {
    // x_k[i] = sum_r Linv[r][i] y_k[r] : 4 row chunks of 16 per column i
    const int i = tid & 63, ch = tid >> 6;
    const double* Z = Linv + (size_t)kb * NB * NB;
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) { const int r = ch * 16 + q; s += Z[r * NB + i] * yk[r]; }
    part[ch][i] = s;
  }
  __syncthreads();
  if (tid < NB) {
    const double s = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
    xk[tid] = s;
    if (blockIdx.x == 0 && tid < nb) x[k0 + tid] = s;
  }
  __syncthreads();
  const int j = blockIdx.x * 256 + tid;
  if (j < k0) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    const double* col = A + (size_t)k0 * lda + j;
    if (nb == NB) {
#pragma unroll
      for (int r = 0; r < NB; r += 4) {
        s0 += col[(size_t)r * lda] * xk[r]; s1 += col[(size_t)(r + 1) * lda] * xk[r + 1];
        s2 += col[(size_t)(r + 2) * lda] * xk[r + 2]; s3 += col[(size_t)(r + 3) * lda] * xk[r + 3];
      }
    } else {
      for (int r = 0; r < nb; ++r) s0 += col[(size_t)r * lda] * xk[r];
    }
    y[j] -= (s0 + s1) + (s2 + s3);
  }
}

}  // namespace

size_t dense_cholesky_workspace(int n) {
  const int nblk = (n + NB - 1) / NB;
  return (size_t)std::max(1, nblk) * NB * NB + (size_t)n + 8;
}

void dense_cholesky_solve(int n, double* A, int lda, double* b, double* work, double* fail_flag, hipStream_t st) {
  if (n <= 0) return;
  const int nrows = n + 1;  // row n = right-hand side (b must alias A + n*lda)
  const int nblk = (n + NB - 1) / NB;
  double* Linv = work;
  double* x = work + (size_t)nblk * NB * NB;
  for (int kb = 0; kb < nblk; ++kb) {
    const int k0 = kb * NB;
    const int nb = (n - k0 < NB) ? (n - k0) : NB;
    const int below = nrows - (k0 + nb);
    double* Zk = Linv + (size_t)kb * NB * NB;
    k_potrf<<<1, 256, 0, st>>>(A, lda, k0, nb, Zk, fail_flag);
    if (below > 0) {
      k_trsm<<<(below + 127) / 128, 256, 0, st>>>(A, lda, nrows, k0, nb, Zk);
      const int tiles = (below + 63) / 64;
      dim3 grid(tiles, tiles);
      k_syrk<<<grid, 256, 0, st>>>(A, lda, nrows, k0, nb);
    }
  }
  double* y = A + (size_t)n * lda;
  for (int kb = nblk - 1; kb >= 0; --kb) {
    const int k0 = kb * NB;
    const int grid = k0 > 0 ? (k0 + 255) / 256 : 1;
    k_back_step<<<grid, 256, 0, st>>>(A, lda, n, kb, Linv, y, x);
  }
  (void)hipMemcpyAsync(b, x, sizeof(double) * n, hipMemcpyDeviceToDevice, st);
}

}  // namespace thip
