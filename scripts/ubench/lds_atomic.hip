// micro-benchmark: cycles per wave instruction of ds_add_f64 (no return) with distinct addresses per lane, against a
// plain ds_read_b64 + v_add_f64 + ds_write_b64 of the same words.  hipcc --offload-arch=gfx950 -O3 ... -o scripts/ubench/lds_atomic
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_atomic(long long* t, double* out, int n, int waves) {
  __shared__ double s[64 * 16 * 16];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 64 * 16 * 16; i += blockDim.x) s[i] = 0.0;
  __syncthreads();
  double* base = s + wv * 64 * 16;
  long long c0 = clock64();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) __hip_atomic_fetch_add(&base[k * 64 + lane], 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __syncthreads();
  long long c1 = clock64();
  if (threadIdx.x == 0) t[0] = c1 - c0;
  out[threadIdx.x] = base[lane];
}
__global__ void k_rmw(long long* t, double* out, int n, int waves) {
  __shared__ double s[64 * 16 * 16];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 64 * 16 * 16; i += blockDim.x) s[i] = 0.0;
  __syncthreads();
  volatile double* base = s + wv * 64 * 16;
  long long c0 = clock64();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) base[k * 64 + lane] = base[k * 64 + lane] + 1.0;
  }
  __syncthreads();
  long long c1 = clock64();
  if (threadIdx.x == 0) t[0] = c1 - c0;
  out[threadIdx.x] = base[lane];
}
int main() {
  long long* t; double* o; long long h;
  hipMalloc(&t, 8); hipMalloc(&o, 8 * 1024);
  const int n = 2000;
  for (int waves : {1, 4, 8, 16}) {
    k_atomic<<<1, 64 * waves>>>(t, o, n, waves); hipDeviceSynchronize();
    k_atomic<<<1, 64 * waves>>>(t, o, n, waves); hipDeviceSynchronize();
    hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
    const double a = (double)h / (n * 16.0);
    k_rmw<<<1, 64 * waves>>>(t, o, n, waves); hipDeviceSynchronize();
    k_rmw<<<1, 64 * waves>>>(t, o, n, waves); hipDeviceSynchronize();
    hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
    printf("%2d waves on one CU: ds_add_f64 %.1f clock64 ticks per wave-instruction per wave, read+add+write %.1f\n", waves, a, (double)h / (n * 16.0));
  }
  return 0;
}
