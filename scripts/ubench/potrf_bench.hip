#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
constexpr int NB = 64; constexpr int LDP = NB + 1;
__device__ __forceinline__ double readlane_d(double v, int l) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, l);
  hi = __builtin_amdgcn_readlane(hi, l);
  return __hiloint2double(hi, lo);
}

// One wavefront: factor the diagonal block at k0 (nb <= 64 valid rows; the rest
// is padded with the identity), write L11 back, and write L11^-1 (64 x 64,
// row-major, zero upper part) to Linv.
__global__ __launch_bounds__(64) void k_potrf(double* __restrict__ A, int lda, int k0, int nb,
                                              double* __restrict__ Linv, double* __restrict__ fail_flag, long long* tstamp) {
  long long tl = clock64();
  __shared__ double Ls[NB][LDP];
  const int i = threadIdx.x;
  double row[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j)
    row[j] = (i < nb && j <= i) ? A[(size_t)(k0 + i) * lda + k0 + j] : ((i == j) ? 1.0 : 0.0);
  long long t0 = clock64(); if (i == 0) tstamp[4] = t0 - tl;
  int bad = 0;
  double rdiag[NB];  // 1 / L[j][j] (wave-uniform), reused by the inverse below
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    double s[8];
    s[0] = row[j];
#pragma unroll
    for (int q = 1; q < 8; ++q) s[q] = 0.0;
#pragma unroll
    for (int k = 0; k < j; ++k) s[k & 7] -= row[k] * readlane_d(row[k], j);
    const double sj = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    double d = readlane_d(sj, j);
    if (!(d > 0.0)) { bad = 1; d = 1.0; }
    // hardware v_rsq_f64 seed + two Newton steps (full FP64), no sqrt / divide
    double rinv = __builtin_amdgcn_rsq(d);
    rinv = rinv * (1.5 - (0.5 * d) * (rinv * rinv));
    rinv = rinv * (1.5 - (0.5 * d) * (rinv * rinv));
    rdiag[j] = rinv;
    row[j] = (i == j) ? d * rinv : (i > j ? sj * rinv : 0.0);
  }
  long long t1 = clock64();
  if (i < nb) {
#pragma unroll
    for (int j = 0; j < NB; ++j) if (j <= i) A[(size_t)(k0 + i) * lda + k0 + j] = row[j];
  }
#pragma unroll
  for (int j = 0; j < NB; ++j) Ls[i][j] = row[j];
  if (bad && i == 0) unsafeAtomicAdd(fail_flag, 1.0);
  __syncthreads();
  long long t2 = clock64();
  // lane c computes column c of Z = L^-1:  z[r] = (delta_rc - sum_{k<r} L[r][k] z[k]) / L[r][r]
  const int c = i;
  double z[NB];
#pragma unroll
  for (int r = 0; r < NB; ++r) {
    double s[8];
    s[0] = (r == c) ? 1.0 : 0.0;
#pragma unroll
    for (int q = 1; q < 8; ++q) s[q] = 0.0;
#pragma unroll
    for (int k = 0; k < r; ++k) s[k & 7] -= Ls[r][k] * z[k];
    const double sr = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    z[r] = sr * rdiag[r];
  }
  long long t3 = clock64();
#pragma unroll
  for (int r = 0; r < NB; ++r) Linv[r * NB + c] = z[r];
  if (i == 0) { tstamp[0] = t1 - t0; tstamp[1] = t2 - t1; tstamp[2] = t3 - t2; tstamp[3] = clock64() - t3; }
}


int main() {
  const int n = 64; std::vector<double> A(n * n);
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) A[i * n + j] = (i == j) ? n + 1.0 : 1.0 / (1 + abs(i - j));
  double *dA, *dL, *df; long long* dt; hipMalloc(&dA, n * n * 8); hipMalloc(&dL, n * n * 8); hipMalloc(&df, 8); hipMalloc(&dt, 64);
  for (int rep = 0; rep < 3; ++rep) {
    hipMemcpy(dA, A.data(), n * n * 8, hipMemcpyHostToDevice); hipMemset(df, 0, 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a); k_potrf<<<1, 64>>>(dA, n, 0, n, dL, df, dt); hipEventRecord(b); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, a, b); long long h[5]; hipMemcpy(h, dt, 40, hipMemcpyDeviceToHost);
    printf("event %.1f us | cycles: load %lld factor %lld store+lds %lld inverse %lld store %lld\n", ms * 1e3, h[4], h[0], h[1], h[2], h[3]);
  }
}
