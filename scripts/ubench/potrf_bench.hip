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
    int diffL = 0, diffZ = 0;
    for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) { diffL += L2[i * n + j] != L[i * n + j]; diffZ += Z2[i * n + j] != Z[i * n + j]; }
    printf("potrf64_wg (4 waves): %.2f us  entries differing from the 1-wave result: L %d  Linv %d\n", bw * 1e3, diffL, diffZ);
  }
  const char* nm[7] = {"load", "factor (4 panels)", "store L", "16x16 inverses", "32-level", "64-level", "store Linv"};
  for (int q = 0; q < 7; ++q) printf("  %-18s %8lld cycles\n", nm[q], st[q]);
  printf("  panels: load %lld  sweep %lld  mfma update %lld cycles\n", st[8], st[9], st[10]);
  return 0;
}
