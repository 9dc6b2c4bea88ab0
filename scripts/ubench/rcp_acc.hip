// micro-benchmark: accuracy of v_rcp_f64 / v_rsq_f64 raw and after k Newton steps (max relative error over a sweep),
// and the cycles of a dependent chain of each.  hipcc --offload-arch=gfx950 -O3 scripts/ubench/rcp_acc.hip -o scripts/ubench/rcp_acc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k_acc(const double* x, double* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double v = x[i];
  double y = __builtin_amdgcn_rcp(v);
  out[8 * i + 0] = y;
  double e = __builtin_fma(-v, y, 1.0); y = __builtin_fma(y, e, y);
  out[8 * i + 1] = y;
  e = __builtin_fma(-v, y, 1.0); y = __builtin_fma(y, e, y);
  out[8 * i + 2] = y;
  e = __builtin_fma(-v, y, 1.0); y = __builtin_fma(y, e, y);
  out[8 * i + 3] = y;
  double r = __builtin_amdgcn_rsq(v);
  out[8 * i + 4] = r;
  for (int k = 0; k < 3; ++k) {
    const double t = v * r;
    const double ee = __builtin_fma(-t, r, 1.0);
    r = __builtin_fma(0.5 * r, ee, r);
    out[8 * i + 5 + k] = r;
  }
}
int main() {
  const int n = 1 << 20;
  double* hx = new double[n]; double* ho = new double[8 * n];
  unsigned long long s = 88172645463325252ull;
  for (int i = 0; i < n; ++i) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    const double u = (double)(s >> 11) / 9007199254740992.0;
    hx[i] = std::ldexp(1.0 + u, (int)(s % 40) - 20);
  }
  double *dx, *dout;
  hipMalloc(&dx, n * 8); hipMalloc(&dout, 8 * n * 8);
  hipMemcpy(dx, hx, n * 8, hipMemcpyHostToDevice);
  k_acc<<<n / 256, 256>>>(dx, dout, n);
  hipMemcpy(ho, dout, 8 * n * 8, hipMemcpyDeviceToHost);
  double mx[8] = {0};
  for (int i = 0; i < n; ++i) {
    const long double rc = 1.0L / hx[i], rs = 1.0L / sqrtl((long double)hx[i]);
    for (int k = 0; k < 4; ++k) mx[k] = std::fmax(mx[k], (double)fabsl((ho[8 * i + k] - rc) / rc));
    for (int k = 4; k < 8; ++k) mx[k] = std::fmax(mx[k], (double)fabsl((ho[8 * i + k] - rs) / rs));
  }
  printf("rcp: raw %.3e  1NR %.3e  2NR %.3e  3NR %.3e   (eps = 1.1e-16)\n", mx[0], mx[1], mx[2], mx[3]);
  printf("rsq: raw %.3e  1NR %.3e  2NR %.3e  3NR %.3e\n", mx[4], mx[5], mx[6], mx[7]);
  return 0;
}
