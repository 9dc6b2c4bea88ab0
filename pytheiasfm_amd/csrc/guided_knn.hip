// guided_knn.hip -- the descriptor search of GuidedEpipolarMatcher::GetMatches as one launch.
//
// TwoViewMatchGeometricVerification::VerifyMatches (sfm/two_view_match_geometric_verification.cc:157-170) can extend the
// verified matches by a guided search: the unmatched features of image 1 are grouped by their epipolar lines, the features
// of image 2 near each group's line are collected from a grid, and FindKNearestNeighbors
// (matching/guided_epipolar_matcher.cc:356-412) takes, for every query feature of a group, the TWO nearest candidates by
// squared L2 distance of the float descriptors (partial_sort of (distance, candidate position)) for Lowe's ratio test.  The
// geometry (lines, groups, grid walk) is host work of a few thousand elements (pytheiasfm_amd/twoview.py); the search is
// sum_groups |queries| x |candidates| x dim multiply-adds and runs here: one wavefront per query, lane = candidate (strided),
// every lane's distance summed over the dimensions IN SEQUENCE in float (no FMA: the file is built with -ffp-contract=off) --
// the order the oracle uses; Eigen's squaredNorm() sums in packets, unpinned (stated in DESIGN.md) -- then the lanes' two best
// are merged by a butterfly under the (distance, position) order of the reference's pair comparison.
#include "theia_hip_internal.h"

#include <algorithm>
#include <cfloat>
#include <vector>

#define HIP_TRY(expr)                                                                             \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess) return thip::set_error(THEIA_HIP_ERR_INTERNAL, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

namespace thip {
namespace {

struct Top2 { float d0, d1; int p0, p1; };
__device__ __forceinline__ bool before(float da, int pa, float db, int pb) { return da < db || (da == db && pa < pb); }
// the two smallest of {a.0, a.1, b.0, b.1} (each list sorted) under (distance, position)
__device__ __forceinline__ Top2 merge2(const Top2& a, float bd0, int bp0, float bd1, int bp1) {
  Top2 r;
  if (before(a.d0, a.p0, bd0, bp0)) {
    r.d0 = a.d0; r.p0 = a.p0;
    if (before(a.d1, a.p1, bd0, bp0)) { r.d1 = a.d1; r.p1 = a.p1; } else { r.d1 = bd0; r.p1 = bp0; }
  } else {
    r.d0 = bd0; r.p0 = bp0;
    if (before(a.d0, a.p0, bd1, bp1)) { r.d1 = a.d0; r.p1 = a.p0; } else { r.d1 = bd1; r.p1 = bp1; }
  }
  return r;
}

__global__ __launch_bounds__(256) void k_guided_knn(int nq, const int* __restrict__ q_group, const int* __restrict__ q_idx,
                                                    const int64_t* __restrict__ c_off, const int* __restrict__ c_idx, int dim,
                                                    const float* __restrict__ desc1, const float* __restrict__ desc2,
                                                    float* __restrict__ nn_dist, int* __restrict__ nn_index) {
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (q >= nq) return;
  const int g = q_group[q];
  const int64_t c0 = c_off[g];
  const int nc = (int)(c_off[g + 1] - c0);
  const float* qd = desc1 + (size_t)q_idx[q] * dim;
  constexpr int kNone = 0x7fffffff;
  Top2 best{FLT_MAX, FLT_MAX, kNone, kNone};
  for (int ci = lane; ci < nc; ci += 64) {
    const float* cd = desc2 + (size_t)c_idx[c0 + ci] * dim;
    float acc = 0.0f;
    for (int d = 0; d < dim; ++d) { const float df = qd[d] - cd[d]; acc = acc + df * df; }
    if (before(acc, ci, best.d0, best.p0)) { best.d1 = best.d0; best.p1 = best.p0; best.d0 = acc; best.p0 = ci; }
    else if (before(acc, ci, best.d1, best.p1)) { best.d1 = acc; best.p1 = ci; }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float od0 = __shfl_xor(best.d0, off, 64), od1 = __shfl_xor(best.d1, off, 64);
    const int op0 = __shfl_xor(best.p0, off, 64), op1 = __shfl_xor(best.p1, off, 64);
    best = merge2(best, od0, op0, od1, op1);
  }
  if (lane == 0) {
    nn_dist[2 * (size_t)q] = best.d0; nn_dist[2 * (size_t)q + 1] = best.d1;
    nn_index[2 * (size_t)q] = best.p0 == kNone ? -1 : c_idx[c0 + best.p0];
    nn_index[2 * (size_t)q + 1] = best.p1 == kNone ? -1 : c_idx[c0 + best.p1];
  }
}

template <class T>
struct DevBuf {
  T* p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  int up(const void* src, size_t n) {
    if (hipMalloc((void**)&p, std::max<size_t>(1, n) * sizeof(T)) != hipSuccess)
      return set_error(THEIA_HIP_ERR_OUT_OF_MEMORY, "hipMalloc(%zu) failed", n * sizeof(T));
    if (n && src && hipMemcpy(p, src, n * sizeof(T), hipMemcpyHostToDevice) != hipSuccess)
      return set_error(THEIA_HIP_ERR_INTERNAL, "hipMemcpy H2D failed");
    return 0;
  }
};

}  // namespace
}  // namespace thip

using namespace thip;

extern "C" int theia_hip_guided_knn(int32_t num_groups, const int64_t* q_off, const int32_t* q_idx, const int64_t* c_off,
                                    const int32_t* c_idx, int32_t n1, int32_t n2, int32_t dim, const float* desc1, const float* desc2,
                                    float* nn_dist, int32_t* nn_index) {
  if (num_groups < 0 || dim < 1 || n1 < 0 || n2 < 0) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "bad sizes");
  if (num_groups == 0) return 0;
  if (!q_off || !c_off || q_off[0] != 0 || c_off[0] != 0) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "offsets must start at 0");
  for (int g = 0; g < num_groups; ++g)
    if (q_off[g + 1] < q_off[g] || c_off[g + 1] < c_off[g]) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "offsets must be non-decreasing");
  const int64_t nq = q_off[num_groups], ncand = c_off[num_groups];
  if (nq == 0) return 0;
  if (nq >= ((int64_t)1 << 30) || ncand >= ((int64_t)1 << 31)) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "too many queries / candidates");
  if (!q_idx || (ncand && !c_idx) || !desc1 || (ncand && !desc2) || !nn_dist || !nn_index) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "null array");
  for (int64_t i = 0; i < nq; ++i) if (q_idx[i] < 0 || q_idx[i] >= n1) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "query index out of range");
  for (int64_t i = 0; i < ncand; ++i) if (c_idx[i] < 0 || c_idx[i] >= n2) return set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "candidate index out of range");
  int rc = thip::ensure_device();
  if (rc) return rc;
  std::vector<int> qg((size_t)nq);
  for (int g = 0; g < num_groups; ++g) for (int64_t i = q_off[g]; i < q_off[g + 1]; ++i) qg[(size_t)i] = g;
  DevBuf<int> d_qg, d_qi, d_ci, d_ni; DevBuf<int64_t> d_co; DevBuf<float> d_d1, d_d2, d_nd;
  if ((rc = d_qg.up(qg.data(), qg.size())) || (rc = d_qi.up(q_idx, (size_t)nq)) || (rc = d_ci.up(c_idx, (size_t)ncand)) ||
      (rc = d_co.up(c_off, (size_t)num_groups + 1)) || (rc = d_d1.up(desc1, (size_t)n1 * dim)) || (rc = d_d2.up(desc2, (size_t)n2 * dim)) ||
      (rc = d_nd.up(nullptr, 2 * (size_t)nq)) || (rc = d_ni.up(nullptr, 2 * (size_t)nq)))
    return rc;
  k_guided_knn<<<(unsigned)((nq + 3) / 4), 256, 0, nullptr>>>((int)nq, d_qg.p, d_qi.p, d_co.p, d_ci.p, dim, d_d1.p, d_d2.p, d_nd.p, d_ni.p);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(nn_dist, d_nd.p, sizeof(float) * 2 * nq, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(nn_index, d_ni.p, sizeof(int32_t) * 2 * nq, hipMemcpyDeviceToHost));
  return 0;
}
