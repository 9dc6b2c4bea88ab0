// wave_sum_butterfly / wave_max_butterfly (pytheiasfm_amd/csrc/wave_reduce.h: permlane swaps + DPP) against the __shfl_xor loops they replace, bit for bit.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I../../pytheiasfm_amd/csrc wave_butterfly_exact.hip -o wave_butterfly_exact   (0 lanes differ of 262 144)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <random>
#include "wave_reduce.h"
__device__ __forceinline__ double wsum_ref(double v) { for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64); return v; }
__device__ __forceinline__ double wmax_ref(double v) { for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64)); return v; }
__device__ __forceinline__ int wsumi_ref(int v) { for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64); return v; }
__global__ void k(const double* a, double* o, int* bad) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  const double v = a[i];
  const double r = wsum_ref(v), n = thip::wave_sum_butterfly(v), mr = wmax_ref(v), mn = thip::wave_max_butterfly(v);
  const int ir = wsumi_ref(threadIdx.x * 3 + 1), in = thip::wave_sum_butterfly((int)threadIdx.x * 3 + 1);
  if (__double_as_longlong(r) != __double_as_longlong(n) || __double_as_longlong(mr) != __double_as_longlong(mn) || ir != in) atomicAdd(bad, 1);
  o[i] = n;
}
int main() {
  const int n = 64 * 4096;
  std::vector<double> a(n); std::mt19937_64 g(3); std::uniform_real_distribution<double> u(-1.0, 1.0);
  for (int i = 0; i < n; ++i) a[i] = std::ldexp(u(g), (int)(g() % 40) - 20);
  double *da, *dо; int* db; hipMalloc(&da, 8 * n); hipMalloc(&dо, 8 * n); hipMalloc(&db, 4); hipMemset(db, 0, 4);
  hipMemcpy(da, a.data(), 8 * n, hipMemcpyHostToDevice);
  k<<<n / 64, 64>>>(da, dо, db);
  int bad = -1; hipMemcpy(&bad, db, 4, hipMemcpyDeviceToHost);
  printf("wave_sum / wave_max butterfly on permlane swaps + DPP vs the shfl_xor loops: %d lanes differ of %d\n", bad, n);
  return bad != 0;
}
