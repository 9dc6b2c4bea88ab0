// micro-benchmark: issue rate / dependent latency of v_mfma_f64_16x16x4 and v_mfma_f64_4x4x4 on gfx950, one wave per SIMD
// and two.  hipcc --offload-arch=gfx950 -O3 scripts/ubench/mfma_f64.hip -o scripts/ubench/mfma_f64
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int CHAINS>
__global__ void k16(double* out, long long* t, int n) {
  d4 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) acc[c] = (d4){0.0, 0.0, 0.0, 0.0};
  double a = out[threadIdx.x & 63], b = out[64 + (threadIdx.x & 63)];
  long long c0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
  }
  long long c1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  if (s == 12345.0) out[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = c1 - c0;
}
template <int CHAINS>
__global__ void k4(double* out, long long* t, int n) {
  double acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) acc[c] = 0.0;
  double a = out[threadIdx.x & 63], b = out[64 + (threadIdx.x & 63)];
  long long c0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[c], 0, 0, 0);
  }
  long long c1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int c = 0; c < CHAINS; ++c) s += acc[c];
  if (s == 12345.0) out[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = c1 - c0;
}
template <class K> void run(const char* name, K kern, int chains, double* d, long long* t, int waves) {
  const int n = 4000; long long h;
  kern<<<1, 64 * waves>>>(d, t, n); hipDeviceSynchronize();
  kern<<<1, 64 * waves>>>(d, t, n); hipDeviceSynchronize();
  hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
  printf("%s chains %d, %d waves on one CU: %.1f ticks per MFMA per wave\n", name, chains, waves, (double)h / ((double)n * chains));
}
int main() {
  double* d; long long* t;
  hipMalloc(&d, 8 * 256); hipMalloc(&t, 8);
  hipMemset(d, 0, 8 * 256);
  for (int waves : {1, 4, 8}) {
    run("f64 16x16x4", k16<1>, 1, d, t, waves);
    run("f64 16x16x4", k16<3>, 3, d, t, waves);
    run("f64 4x4x4  ", k4<1>, 1, d, t, waves);
    run("f64 4x4x4  ", k4<4>, 4, d, t, waves);
  }
  return 0;
}
