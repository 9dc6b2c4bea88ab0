// micro-benchmark: effective shader clock seen by a single-wave dependent chain
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void chain(double* out, long long* t, int n) {
  double x = out[0];
  long long c0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < n; ++i) x = x * 1.0000001 + 1e-9;
  long long c1 = clock64(), w1 = wall_clock64();
  out[0] = x; t[0] = c1 - c0; t[1] = w1 - w0;
}
__global__ void busy(double* out, int n) {
  double x = threadIdx.x;
  for (int i = 0; i < n; ++i) x = x * 1.0000001 + 1e-9;
  if (x == 123.0) out[1] = x;
}
int main() {
  double* d; long long* t; hipMalloc(&d, 64); hipMalloc(&t, 64); hipMemset(d, 0, 64);
  long long h[2];
  for (int rep = 0; rep < 3; ++rep) {
    for (int n : {1000, 10000, 100000}) {
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
      hipEventRecord(a); chain<<<1, 64>>>(d, t, n); hipEventRecord(b); hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, a, b); hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
      printf("rep %d n %d: shader cycles %lld (%.2f/iter) wallclock ticks %lld -> %.1f us by 100MHz, event %.1f us, eff clock %.0f MHz\n", rep, n, h[0],
             (double)h[0] / n, h[1], h[1] / 100.0, ms * 1e3, h[0] / (h[1] / 100.0));
    }
    if (rep == 1) { busy<<<4096, 256>>>(d, 2000000); printf("-- after heavy kernel enqueue (no sync)\n"); }
  }
  hipDeviceSynchronize();
  return 0;
}
