// Are the device's FP64 sqrt and division correctly rounded (= the host's bits)?  The micro-solvers compare with the CPU oracle
// bit for bit and rely on it.  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off sqrt_div_exact.hip -o sqrt_div_exact
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
__global__ void k(const double* a, const double* b, double* s, double* d, double* r, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  s[i] = sqrt(a[i]);
  d[i] = a[i] / b[i];
  r[i] = 1.0 / sqrt(a[i] * a[i] + 1.0);
}
int main() {
  const int n = 1 << 24;
  std::vector<double> a(n), b(n), s(n), d(n), r(n);
  std::mt19937_64 g(7);
  std::uniform_real_distribution<double> u(-30.0, 30.0), m(1.0, 2.0);
  for (int i = 0; i < n; ++i) { a[i] = std::ldexp(m(g), (int)u(g)); b[i] = std::ldexp(m(g), (int)u(g)) * ((i & 1) ? -1.0 : 1.0); }
  double *da, *db, *ds, *dd, *dr;
  hipMalloc(&da, 8 * n); hipMalloc(&db, 8 * n); hipMalloc(&ds, 8 * n); hipMalloc(&dd, 8 * n); hipMalloc(&dr, 8 * n);
  hipMemcpy(da, a.data(), 8 * n, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 8 * n, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(da, db, ds, dd, dr, n);
  hipMemcpy(s.data(), ds, 8 * n, hipMemcpyDeviceToHost); hipMemcpy(d.data(), dd, 8 * n, hipMemcpyDeviceToHost); hipMemcpy(r.data(), dr, 8 * n, hipMemcpyDeviceToHost);
  long bs = 0, bd = 0, br = 0;
  for (int i = 0; i < n; ++i) {
    bs += s[i] != std::sqrt(a[i]);
    bd += d[i] != a[i] / b[i];
    br += r[i] != 1.0 / std::sqrt(a[i] * a[i] + 1.0);
  }
  printf("n = %d: sqrt differs %ld, division differs %ld, 1/sqrt(a^2+1) differs %ld\n", n, bs, bd, br);
  return 0;
}
