// relpos_irls.hip -- batches of INDEPENDENT OptimizeRelativePositionWithKnownRotation solves, one per wavefront.
//
// Replaces N calls of OptimizeRelativePositionWithKnownRotation
// (src/theia/sfm/bundle_adjustment/optimize_relative_position_with_known_rotation.cc:116-191; pybind sfm.cc:1621-1622).
// The global pipeline calls it once per view pair of the view graph from a thread pool after the rotations are known
// (RefineRelativeTranslationsWithKnownRotations, sfm/reconstruction_estimator_utils.cc:255-297): thousands of tiny
// iteratively re-weighted least-squares problems, each a 3 x 3 null-vector computation per iteration over a few hundred
// epipolar constraints -- the "many micro-solves in one launch" shape of ba_batch.hip / twoview_lm.hip.
//
// Per pair (lane = correspondence, strided):  c_i = R1 ((R2^T f2_i) x (R1^T f1_i))   (CreateConstraintMatrix, :52-80), kept in
// an HBM scratch row that stays in L2 over the iterations;  A = sum_i c_i c_i^T / w_i  by per-lane partial sums and an XOR
// butterfly (every lane ends with the same nine doubles);  t = third column of U of the two-sided Jacobi SVD of A
// (rsc::svd3 = the restatement of Eigen's JacobiSVD the RANSAC solvers use), computed redundantly by every lane;
// w_i = max(|t . c_i|, 1e-7);  stop after ten consecutive iterations with max(|cost - new cost|, 1 - |t|^2) <= 1e-5 or at 100;
// sign by the majority cheirality vote (MajorityOfPointsInFrontOfCameras, :86-110).
// Built with -ffp-contract=off like the other micro-solvers; the sums run in wave order (strided partials + butterfly) where
// Eigen's product kernel runs blocked: bit-identical to the oracle run in wave order, ~1e-4 .. 1e-15 from the index-order run
// depending on the pair's noise (the IRLS amplifies summation-order roundings: tests/test_relpos_gpu.py).
#include "ransac_device.h"
#include "wave_reduce.h"
#include "eig_team.h"
#include "theia_hip_internal.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <vector>

#define HIP_TRY(expr)                                                                             \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess) return thip::set_error(THEIA_HIP_ERR_INTERNAL, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

namespace thip {
namespace {

// the XOR butterfly over the 64 lanes, on permlane swaps + DPP (wave_reduce.h: the bits of the __shfl_xor loop)
__device__ __forceinline__ double wsum(double v) { return wave_sum_butterfly(v); }
__device__ __forceinline__ int wsum_i(int v) { return wave_sum_butterfly(v); }

__global__ __launch_bounds__(256) void k_relpos_irls(int num, const int64_t* __restrict__ offsets, const double4* __restrict__ corr,
                                                     const double* __restrict__ rotm, double* __restrict__ C,
                                                     double* __restrict__ pos, int* __restrict__ iters, int maxit) {
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (p >= num) return;
  const int64_t base = offsets[p];
  const int n = (int)(offsets[p + 1] - base);
  double R1[9], R2[9];   // row-major, from the host (sin / cos of the host's libm: the oracle's bits)
#pragma unroll
  for (int k = 0; k < 9; ++k) { R1[k] = rotm[18 * (size_t)p + k]; R2[k] = rotm[18 * (size_t)p + 9 + k]; }
  double* Cp = C + 3 * base;
  for (int i = lane; i < n; i += 64) {
    const double4 m = corr[base + i];
    const double f1[3] = {m.x, m.y, 1.0}, f2[3] = {m.z, m.w, 1.0};
    double a[3], b[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      a[k] = (R1[k] * f1[0] + R1[3 + k] * f1[1]) + R1[6 + k] * f1[2];
      b[k] = (R2[k] * f2[0] + R2[3 + k] * f2[1]) + R2[6 + k] * f2[2];
    }
    const double x[3] = {b[1] * a[2] - b[2] * a[1], b[2] * a[0] - b[0] * a[2], b[0] * a[1] - b[1] * a[0]};
#pragma unroll
    for (int k = 0; k < 3; ++k) Cp[3 * i + k] = (R1[3 * k] * x[0] + R1[3 * k + 1] * x[1]) + R1[3 * k + 2] * x[2];
  }
  // (a lane reads back only the rows it wrote: no fence needed)
  double cost = 0.0, t[3] = {0.0, 0.0, 0.0};
  int inner = 0, it = 0;
  bool first = true;
  for (; it < maxit && inner < 10; ++it) {
    double A[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) A[k] = 0.0;
    for (int i = lane; i < n; i += 64) {
      const double c[3] = {Cp[3 * i], Cp[3 * i + 1], Cp[3 * i + 2]};
      double w = first ? 1.0 : fabs((t[0] * c[0] + t[1] * c[1]) + t[2] * c[2]);
      if (w < 1e-7) w = 1e-7;
      const double iw = 1.0 / w;
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int q = 0; q < 3; ++q) A[3 * r + q] += (c[r] * iw) * c[q];
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) A[k] = wsum(A[k]);
    double U[9], S[3], V[9];
    rsc::svd3(A, U, S, V);
    const double nt[3] = {U[2], U[5], U[8]};
    double ncost = 0.0;
    for (int i = lane; i < n; i += 64) ncost += fabs((nt[0] * Cp[3 * i] + nt[1] * Cp[3 * i + 1]) + nt[2] * Cp[3 * i + 2]);
    ncost = wsum(ncost);
    const double delta = fmax(fabs(cost - ncost), 1.0 - ((nt[0] * nt[0] + nt[1] * nt[1]) + nt[2] * nt[2]));
    if (delta <= 1e-5) ++inner; else inner = 0;
    cost = ncost;
    t[0] = nt[0]; t[1] = nt[1]; t[2] = nt[2];
    first = false;
  }
  double Rr[9];   // R2 R1^T
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int q = 0; q < 3; ++q) Rr[3 * r + q] = (R2[3 * r] * R1[3 * q] + R2[3 * r + 1] * R1[3 * q + 1]) + R2[3 * r + 2] * R1[3 * q + 2];
  int front = 0;
  for (int i = lane; i < n; i += 64) {
    const double4 m = corr[base + i];
    const double c4[4] = {m.x, m.y, m.z, m.w};
    front += rsc::in_front(c4, Rr, t) ? 1 : 0;
  }
  front = wsum_i(front);
  if (!(front > n / 2)) { t[0] = -t[0]; t[1] = -t[1]; t[2] = -t[2]; }
  if (lane == 0) {
    pos[3 * (size_t)p] = t[0]; pos[3 * (size_t)p + 1] = t[1]; pos[3 * (size_t)p + 2] = t[2];
    if (iters) iters[p] = it;
  }
}

// ceres::AngleAxisToRotationMatrix (rotation.h), row-major out; on the host so that sin / cos are libm's
void host_angle_axis_to_rot(const double* aa, double* R) {
  const double theta2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
  if (theta2 > DBL_EPSILON) {
    const double theta = std::sqrt(theta2);
    const double wx = aa[0] / theta, wy = aa[1] / theta, wz = aa[2] / theta;
    // sincos(), not sin() and cos(): a GCC build (the reference's, and the oracle's) merges the two calls of
    // ceres::AngleAxisToRotationMatrix into one sincos, and glibc's sincos differs from cos() in the last bit on ~0.15 % of
    // the arguments -- enough to move the IRLS result of one pair in six hundred
    double s, c;
    ::sincos(theta, &s, &c);
    R[0] = c + wx * wx * (1.0 - c);      R[3] = wz * s + wx * wy * (1.0 - c);  R[6] = -wy * s + wx * wz * (1.0 - c);
    R[1] = wx * wy * (1.0 - c) - wz * s; R[4] = c + wy * wy * (1.0 - c);       R[7] = wx * s + wy * wz * (1.0 - c);
    R[2] = wy * s + wx * wz * (1.0 - c); R[5] = -wx * s + wy * wz * (1.0 - c); R[8] = c + wz * wz * (1.0 - c);
  } else {
    R[0] = 1.0; R[3] = aa[2]; R[6] = -aa[1];
    R[1] = -aa[2]; R[4] = 1.0; R[7] = aa[0];
    R[2] = aa[1]; R[5] = -aa[0]; R[8] = 1.0;
  }
}

template <class T>
struct DevBuf {
  T* p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  int alloc(size_t n) {
    if (hipMalloc((void**)&p, std::max<size_t>(1, n) * sizeof(T)) != hipSuccess)
      return set_error(THEIA_HIP_ERR_OUT_OF_MEMORY, "hipMalloc(%zu) failed", n * sizeof(T));
    return 0;
  }
  int up(const void* src, size_t n) {
    int rc = alloc(n);
    if (rc) return rc;
    if (n && hipMemcpy(p, src, n * sizeof(T), hipMemcpyHostToDevice) != hipSuccess)
      return set_error(THEIA_HIP_ERR_INTERNAL, "hipMemcpy H2D failed");
    return 0;
  }
};

}  // namespace
}  // namespace thip

using namespace thip;

extern "C" int theia_hip_optimize_relative_position_batch(int32_t num_problems, const int64_t* offsets, const double* correspondences,
                                                          const double* rotations, double* relative_positions,
                                                          int32_t* num_iterations) {
  const int num = num_problems;
  if (num < 0) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "negative num_problems");
  if (num == 0) return 0;
  if (!offsets || !rotations || !relative_positions) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "null array");
  if (offsets[0] != 0) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "offsets[0] must be 0");
  for (int i = 0; i < num; ++i)
    if (offsets[i + 1] < offsets[i]) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "offsets must be non-decreasing");
  const int64_t total = offsets[num];
  if (total > 0 && !correspondences) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "null correspondences");
  for (int i = 0; i < num; ++i)
    if (offsets[i + 1] - offsets[i] > (int64_t)1 << 30) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "more than 2^30 correspondences in one pair");
  int rc = thip::ensure_device();
  if (rc) return rc;
  static const int maxit = [] { const char* e = getenv("THEIA_HIP_RELPOS_MAXIT"); return e ? atoi(e) : 100; }();   // development: kMaxIterations of the reference is 100
  std::vector<double> rotm((size_t)18 * num);
  for (int i = 0; i < num; ++i) {
    host_angle_axis_to_rot(rotations + 6 * (size_t)i, rotm.data() + 18 * (size_t)i);
    host_angle_axis_to_rot(rotations + 6 * (size_t)i + 3, rotm.data() + 18 * (size_t)i + 9);
  }
  DevBuf<int64_t> d_off; DevBuf<double> d_corr, d_rot, d_C, d_pos; DevBuf<int> d_it;
  if ((rc = d_off.up(offsets, num + 1)) || (rc = d_corr.up(correspondences, 4 * (size_t)total)) || (rc = d_rot.up(rotm.data(), rotm.size())) ||
      (rc = d_C.alloc(3 * (size_t)total)) || (rc = d_pos.alloc(3 * (size_t)num)) || (rc = d_it.alloc(num)))
    return rc;
  k_relpos_irls<<<(num + 3) / 4, 256, 0, nullptr>>>(num, d_off.p, reinterpret_cast<const double4*>(d_corr.p), d_rot.p, d_C.p, d_pos.p, d_it.p, maxit);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(relative_positions, d_pos.p, sizeof(double) * 3 * num, hipMemcpyDeviceToHost));
  if (num_iterations) HIP_TRY(hipMemcpy(num_iterations, d_it.p, sizeof(int32_t) * num, hipMemcpyDeviceToHost));
  return 0;
}

// ---- theia_hip_selftest_wave_primitives: wave_reduce.h / eig_team.h's broadcasts against shuffle loops
namespace thip {
namespace {
__global__ __launch_bounds__(64) void k_selftest_wave(const double* __restrict__ in, int* __restrict__ bad) {
  const int lane = threadIdx.x;
  const double v = in[(size_t)blockIdx.x * 64 + lane];
  double rs = v, rm = fabs(v);
  int ri = lane * 3 + 1;
  for (int off = 32; off > 0; off >>= 1) { rs += __shfl_xor(rs, off, 64); rm = fmax(rm, __shfl_xor(rm, off, 64)); ri += __shfl_xor(ri, off, 64); }
  int wrong = 0;
  wrong += __double_as_longlong(wave_sum_butterfly(v)) != __double_as_longlong(rs);
  wrong += wave_sum_butterfly(lane * 3 + 1) != ri;
  wrong += __double_as_longlong(wave_max_butterfly(fabs(v))) != __double_as_longlong(rm);
  wrong += __double_as_longlong(wave_max_abs(fabs(v))) != __double_as_longlong(rm);
  wrong += __double_as_longlong(wave_max_abs(lane & 1 ? fabs(v) : -1.0)) != __double_as_longlong([&] { double m = lane & 1 ? fabs(v) : -1.0; for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off, 64)); return m; }());
  wrong += __double_as_longlong(rsc::row16_bcast<5>(v)) != __double_as_longlong(__shfl(v, (lane & ~15) + 5, 64));
  wrong += __double_as_longlong(rsc::oct_bcast<2>(v)) != __double_as_longlong(__shfl(v, (lane & ~7) + 2, 64));
  wrong += __double_as_longlong(rsc::oct_bcast<4>(v)) != __double_as_longlong(__shfl(v, (lane & ~7) + 4, 64));
  wrong += __double_as_longlong(lane_value(v, 37)) != __double_as_longlong(__shfl(v, 37, 64));
  if (wrong) atomicAdd(bad, wrong);
}
}  // namespace
}  // namespace thip

extern "C" int theia_hip_selftest_wave_primitives(int32_t count, int32_t* mismatches) {
  if (count < 1 || !mismatches) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "count < 1 or null output");
  int rc = thip::ensure_device();
  if (rc) return rc;
  std::vector<double> h((size_t)count * 64);
  uint64_t x = 0x9E3779B97F4A7C15ull;
  for (auto& d : h) {   // doubles of mixed sign and magnitude (xorshift), a few exact ties
    x ^= x << 13; x ^= x >> 7; x ^= x << 17;
    const double m = 1.0 + (double)(x >> 12) * 0x1p-52;
    d = std::ldexp((x & 1) ? -m : m, (int)((x >> 3) % 41) - 20);
    if (((x >> 9) & 31) == 0) d = 1.5;
  }
  DevBuf<double> d_in; DevBuf<int> d_bad;
  if ((rc = d_in.up(h.data(), h.size())) || (rc = d_bad.alloc(1))) return rc;
  HIP_TRY(hipMemset(d_bad.p, 0, sizeof(int)));
  k_selftest_wave<<<count, 64, 0, nullptr>>>(d_in.p, d_bad.p);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(mismatches, d_bad.p, sizeof(int), hipMemcpyDeviceToHost));
  return 0;
}
