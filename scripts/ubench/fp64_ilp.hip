// micro-benchmark: FP64 FMA issue rate of ONE wave as a function of instruction-level parallelism (K independent
// chains), and of W waves per SIMD.  hipcc --offload-arch=gfx950 -O3 scripts/ubench/fp64_ilp.hip -o scripts/ubench/fp64_ilp
#include <hip/hip_runtime.h>
#include <cstdio>
template <int K>
__global__ void chains(double* out, long long* t, int n) {
  double x[K];
#pragma unroll
  for (int k = 0; k < K; ++k) x[k] = out[k] + threadIdx.x;
  long long c0 = clock64();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int k = 0; k < K; ++k) x[k] = __builtin_fma(x[k], 1.0000001, 1e-9);
  }
  long long c1 = clock64();
  double s = 0;
#pragma unroll
  for (int k = 0; k < K; ++k) s += x[k];
  if (s == 12345.0) out[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = c1 - c0;
}
template <int K> void run(double* d, long long* t, int waves_per_simd) {
  const int n = 20000;
  long long h;
  // one workgroup of 64 * 4 * waves_per_simd threads on one CU
  chains<K><<<1, 256 * waves_per_simd>>>(d, t, n);
  hipDeviceSynchronize();
  chains<K><<<1, 256 * waves_per_simd>>>(d, t, n);
  hipDeviceSynchronize();
  hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
  printf("K=%2d waves/SIMD=%d: %.2f cycles per FMA per wave, %.2f cycles per FMA per SIMD\n", K, waves_per_simd, (double)h / n / K,
         (double)h / n / K / waves_per_simd);
}
int main() {
  double* d; long long* t; hipMalloc(&d, 1024); hipMalloc(&t, 64); hipMemset(d, 0, 1024);
  for (int w = 1; w <= 4; ++w) { run<1>(d, t, w); run<2>(d, t, w); run<4>(d, t, w); run<8>(d, t, w); run<16>(d, t, w); }
  return 0;
}
