// dense_cholesky.hip -- K3: dense SPD solve of the reduced camera system on
// gfx950, FP64.  Replaces Ceres' SPARSE_SCHUR + EIGEN_SPARSE SimplicialLDLT of
// the same matrix (bundle_adjustment.h:98,104; selected in bundle_adjuster.cc:63-89).
//
// Blocked right-looking Cholesky on the lower triangle of a row-major matrix.
// The right-hand side is stored as row n of the same (n+1) x lda array, so the
// forward substitution  y = L^-1 b  falls out of the panel solves for free;
// only the backward substitution  x = L^-T y  is a separate kernel.
//
//   per panel (NB = 32 columns):
//     k_panel : every workgroup re-factors the 32x32 diagonal block in LDS
//               (11 kflop, cheaper than a launch boundary) and solves
//               X L11^T = A21 for its own 256 rows;
//     k_syrk  : trailing update A22 -= L21 L21^T on 64x64 tiles with the FP64
//               matrix core  v_mfma_f64_16x16x4_f64  (4x4 MFMA tiles per
//               wave-quadrant), lower tiles only.
#include "ba_kernels.h"

namespace thip {
namespace {

constexpr int NB = 32;

// Factor the NB x NB diagonal block held in LDS (lower, in place). 256 threads.
__device__ void factor_diag_lds(double (*D)[NB + 1], int nb, int tid, int nthreads, int* bad) {
  for (int j = 0; j < nb; ++j) {
    __syncthreads();
    if (tid == 0) {
      const double d = D[j][j];
      if (!(d > 0.0)) { *bad = 1; D[j][j] = 1.0; }
      else D[j][j] = sqrt(d);
    }
    __syncthreads();
    const double inv = 1.0 / D[j][j];
    for (int i = j + 1 + tid; i < nb; i += nthreads) D[i][j] *= inv;
    __syncthreads();
    // trailing update of the block: D[i][k] -= D[i][j] * D[k][j], j < k <= i
    const int m = nb - j - 1;
    for (int t = tid; t < m * m; t += nthreads) {
      const int i = j + 1 + t / m, k = j + 1 + t % m;
      if (k <= i) D[i][k] -= D[i][j] * D[k][j];
    }
  }
  __syncthreads();
}

// Panel step at column block k0: rows [k0, nrows). Workgroup b handles rows
// k0 + nb + b*256 + tid; workgroup 0 also writes back the factored diagonal.
__global__ __launch_bounds__(256) void k_panel(double* __restrict__ A, int lda, int nrows, int k0, int nb,
                                               double* __restrict__ fail_flag) {
  __shared__ double D[NB][NB + 1];
  __shared__ int bad;
  const int tid = threadIdx.x;
  if (tid == 0) bad = 0;
  for (int t = tid; t < nb * nb; t += 256) {
    const int i = t / nb, j = t % nb;
    D[i][j] = (j <= i) ? A[(size_t)(k0 + i) * lda + k0 + j] : 0.0;
  }
  factor_diag_lds(D, nb, tid, 256, &bad);
  if (blockIdx.x == 0) {
    for (int t = tid; t < nb * nb; t += 256) {
      const int i = t / nb, j = t % nb;
      if (j <= i) A[(size_t)(k0 + i) * lda + k0 + j] = D[i][j];
    }
    if (tid == 0 && bad) unsafeAtomicAdd(fail_flag, 1.0);
  }
  const int r = k0 + nb + blockIdx.x * 256 + tid;
  if (r >= nrows) return;
  double* row = A + (size_t)r * lda + k0;
  double x[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) x[j] = (j < nb) ? row[j] : 0.0;
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    if (j < nb) {
      double s = x[j];
#pragma unroll
      for (int i = 0; i < j; ++i) s -= x[i] * D[j][i];
      x[j] = s / D[j][j];
    }
  }
#pragma unroll
  for (int j = 0; j < NB; ++j) if (j < nb) row[j] = x[j];
}

typedef double double4_t __attribute__((ext_vector_type(4)));

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
  __shared__ double Pr[64][NB + 1];  // rows of the panel for this tile's rows
  __shared__ double Pc[64][NB + 1];  // rows of the panel for this tile's cols
  const int tid = threadIdx.x;
  for (int t = tid; t < 64 * NB; t += 256) {
    const int i = t / NB, k = t % NB;
    const int rr = r0 + i, cc = c0 + i;
    Pr[i][k] = (rr < nrows && k < nb) ? A[(size_t)rr * lda + k0 + k] : 0.0;
    Pc[i][k] = (cc < nrows && k < nb) ? A[(size_t)cc * lda + k0 + k] : 0.0;
  }
  __syncthreads();
  const int wv = tid >> 6, lane = tid & 63;
  const int li = lane & 15, lk = lane >> 4;
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
  for (int q = 0; q < 4; ++q) {
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int row = r0 + 16 * wv + lk + 4 * reg;
      const int col = c0 + 16 * q + li;
      if (row < nrows && col <= row && col < nrows - 0) {
        // the rhs row (row == nrows-1) only has columns < nrows-1
        if (!(row == nrows - 1 && col >= nrows - 1))
          A[(size_t)row * lda + col] -= acc[q][reg];
      }
    }
  }
}

// Backward substitution x = L^-T y, y in row n of A (A[n*lda + j]); result in b.
// Single workgroup; block-column sweep from the last panel to the first.
__global__ __launch_bounds__(1024) void k_backward(const double* __restrict__ A, int lda, int n,
                                                   double* __restrict__ b) {
  extern __shared__ double sh[];  // y[n] | xk[NB] | D[NB][NB+1]
  double* y = sh;
  double* xk = sh + n;
  double* D = xk + NB;
  const int tid = threadIdx.x;
  for (int j = tid; j < n; j += 1024) y[j] = A[(size_t)n * lda + j];
  __syncthreads();
  const int nblk = (n + NB - 1) / NB;
  for (int kb = nblk - 1; kb >= 0; --kb) {
    const int k0 = kb * NB;
    const int nb = min(NB, n - k0);
    for (int t = tid; t < nb * nb; t += 1024) {
      const int i = t / nb, j = t % nb;
      D[i * (NB + 1) + j] = (j <= i) ? A[(size_t)(k0 + i) * lda + k0 + j] : 0.0;
    }
    __syncthreads();
    if (tid < 64) {
      // solve L11^T x = y_k : x_i = (y_i - sum_{r>i} L[r][i] x_r) / L[i][i]
      for (int i = nb - 1; i >= 0; --i) {
        double part = 0.0;
        for (int r = i + 1 + tid; r < nb; r += 64) part += D[r * (NB + 1) + i] * xk[r];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
        if (tid == 0) xk[i] = (y[k0 + i] - part) / D[i * (NB + 1) + i];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
      }
    }
    __syncthreads();
    for (int i = tid; i < nb; i += 1024) b[k0 + i] = xk[i];
    // y_j -= sum_r L[k0+r][j] x_r  for j < k0
    for (int j = tid; j < k0; j += 1024) {
      double s = 0.0;
#pragma unroll 8
      for (int r = 0; r < nb; ++r) s += A[(size_t)(k0 + r) * lda + j] * xk[r];
      y[j] -= s;
    }
    __syncthreads();
  }
}

}  // namespace

void dense_cholesky_solve(int n, double* A, int lda, double* b, double* fail_flag, hipStream_t st) {
  if (n <= 0) return;
  const int nrows = n + 1;  // row n = right-hand side (b must alias A + n*lda)
  (void)b;
  for (int k0 = 0; k0 < n; k0 += NB) {
    const int nb = (n - k0 < NB) ? (n - k0) : NB;
    const int below = nrows - (k0 + nb);
    const int pblocks = below > 0 ? (below + 255) / 256 : 1;
    k_panel<<<pblocks, 256, 0, st>>>(A, lda, nrows, k0, nb, fail_flag);
    if (below > 0) {
      const int tiles = (below + 63) / 64;
      dim3 grid(tiles, tiles);
      k_syrk<<<grid, 256, 0, st>>>(A, lda, nrows, k0, nb);
    }
  }
  const size_t shmem = (size_t)(n + NB + NB * (NB + 1)) * sizeof(double);
  k_backward<<<1, 1024, shmem, st>>>(A, lda, n, A + (size_t)n * lda);
}

}  // namespace thip
