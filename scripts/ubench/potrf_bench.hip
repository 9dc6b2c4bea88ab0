// Development micro-benchmark: cycle stamps of the one-wavefront 64x64 potrf
// (pytheiasfm_amd/csrc/cholesky_device.h).  hipcc --offload-arch=gfx950 -O3
//   -ffp-contract=off -I pytheiasfm_amd/csrc scripts/ubench/potrf_bench.hip -o scripts/ubench/potrf_bench
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__device__ long long g_stamps[16];
#define THIP_POTRF_STAMPS g_stamps
#include "cholesky_device.h"
using namespace thip::chol;
__global__ __launch_bounds__(64) void k(double* A, int lda, double* Linv, double* flag) { potrf64_wave(A, lda, 0, 64, Linv, flag); }
__global__ __launch_bounds__(256) void kwg(double* A, int lda, double* Linv, double* flag) { potrf64_wg(A, lda, 0, 64, Linv, flag); }
// the factorisation alone, REPS times back to back in one launch (launch overhead amortised): cycles per factorisation
__global__ __launch_bounds__(256) void kwg_loop(double* A, int lda, double* Linv, long long* cyc, int reps) {
  __shared__ double Ls[NB][LDP];
  __shared__ double Zs[NB][LDP];
  __shared__ double rdiag[NB];
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) {
    potrf64_wg_core<false>(A, lda, 0, 64, Ls, Zs, rdiag, Linv, nullptr);
    __syncthreads();
  }
  if (threadIdx.x == 0) *cyc = (__builtin_amdgcn_s_memtime() - t0) / reps;
}
int main() {
  const int n = 64;
  std::vector<double> A(n * n), L(n * n), Z(n * n);
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) A[i * n + j] = (i == j) ? n + 1.0 : 1.0 / (1 + abs(i - j));
  double *dA, *dL, *df;
  hipMalloc(&dA, n * n * 8); hipMalloc(&dL, n * n * 8); hipMalloc(&df, 8); hipMemset(df, 0, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int rep = 0; rep < 20; ++rep) {
    hipMemcpy(dA, A.data(), n * n * 8, hipMemcpyHostToDevice);
    hipEventRecord(e0); k<<<1, 64>>>(dA, n, dL, df); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  long long st[16]; hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamps), sizeof(st));
  hipMemcpy(L.data(), dA, n * n * 8, hipMemcpyDeviceToHost); hipMemcpy(Z.data(), dL, n * n * 8, hipMemcpyDeviceToHost);
  double err = 0, errz = 0;
  for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) {
    double s = 0; for (int q = 0; q <= j; ++q) s += L[i * n + q] * L[j * n + q];
    err = fmax(err, fabs(s - A[i * n + j]));
    double t = 0; for (int q = j; q <= i; ++q) t += L[i * n + q] * Z[q * n + j];
    errz = fmax(errz, fabs(t - (i == j ? 1.0 : 0.0)));
  }
  printf("potrf64: %.2f us  |LL^T-A| %.2e  |L Linv - I| %.2e\n", best * 1e3, err, errz);
  {   // the 4-wave version: same bits, shorter
    std::vector<double> L2(n * n), Z2(n * n);
    float bw = 1e9;
    for (int rep = 0; rep < 20; ++rep) {
      hipMemcpy(dA, A.data(), n * n * 8, hipMemcpyHostToDevice);
      hipEventRecord(e0); kwg<<<1, 256>>>(dA, n, dL, df); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < bw) bw = ms;
    }
    hipMemcpy(L2.data(), dA, n * n * 8, hipMemcpyDeviceToHost); hipMemcpy(Z2.data(), dL, n * n * 8, hipMemcpyDeviceToHost);
    // residuals of the 4-wave result and its largest relative difference from the 1-wave factor (the rows below a diagonal
    // block are products with the block's inverse in the 4-wave code, a substitution in the 1-wave code)
    double e2 = 0, ez2 = 0, dl = 0, dz = 0, check = 0;
    for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) {
      double s = 0; for (int q = 0; q <= j; ++q) s += L2[i * n + q] * L2[j * n + q];
      e2 = fmax(e2, fabs(s - A[i * n + j]));
      double t = 0; for (int q = j; q <= i; ++q) t += L2[i * n + q] * Z2[q * n + j];
      ez2 = fmax(ez2, fabs(t - (i == j ? 1.0 : 0.0)));
      dl = fmax(dl, fabs(L2[i * n + j] - L[i * n + j]) / fmax(fabs(L[i * n + j]), 1e-300));
      dz = fmax(dz, fabs(Z2[i * n + j] - Z[i * n + j]) / fmax(fabs(Z[i * n + j]), 1e-300));
    }
    for (int i = 0; i < n; ++i) for (int j = i + 1; j < n; ++j) check = fmax(check, fmax(fabs(L2[i * n + j]), fabs(Z2[i * n + j])));
    printf("potrf64_wg (4 waves): %.2f us  |LL^T-A| %.2e  |L Linv - I| %.2e  max rel diff from the 1-wave result: L %.2e  Linv %.2e  upper part max %.1e\n",
           bw * 1e3, e2, ez2, dl, dz, check);
  }
  {
    long long* dc; hipMalloc(&dc, 8);
    hipMemcpy(dA, A.data(), n * n * 8, hipMemcpyHostToDevice);
    kwg_loop<<<1, 256>>>(dA, n, dL, dc, 64); hipDeviceSynchronize();
    long long c; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    printf("potrf64_wg_core in a loop: %lld s_memtime ticks per factorisation + inverse\n", c);
  }
  const char* nm[7] = {"load", "factor (4 panels)", "store L", "16x16 inverses", "32-level", "64-level", "store Linv"};
  for (int q = 0; q < 7; ++q) printf("  %-18s %8lld cycles\n", nm[q], st[q]);
  printf("  panels: load %lld  sweep %lld  mfma update %lld cycles\n", st[8], st[9], st[10]);
  return 0;
}
