// DLS-PnP / gDLS stage A kernels (dls_stage_a.h): one workgroup of 192 threads per minimal problem.  Their own translation
// unit because of one compiler switch (csrc/build.sh: -mllvm -simplifycfg-sink-common=false, see dls_stage_a.h).
#include "dls_stage_a.h"

#include <mutex>

#include "theia_hip_internal.h"

namespace thip {
namespace {

// ---- DLS-PnP hypotheses (estimate_calibrated_absolute_pose.cc:89-97): this stage, then the eigen stage of ransac.hip.
// k_dls_a: one workgroup of 192 threads per (problem, iteration); uvals holds the four Macaulay terms of every DlsPnp call of a process
// (iteration it of a problem = call it: the reference never seeds rand(), and one Estimate() is one process here).
__global__ __launch_bounds__(dlsdev::kThreads, 3) void k_dls_a(int nprob, int B, const int64_t* __restrict__ offsets,
                                              const double* __restrict__ data, const int* __restrict__ samples,
                                              const int* __restrict__ active_iters, const int* __restrict__ iter_base,
                                              const double* __restrict__ uvals, double* __restrict__ action,
                                              double* __restrict__ tfac, int* __restrict__ okflag) {
  __shared__ dlsdev::WgLds L;
  const int b = blockIdx.x, p = blockIdx.y;
  if (b >= active_iters[p]) return;
  const size_t hyp = (size_t)p * B + b;
  const double* pd = data + (size_t)offsets[p] * 5;
  const bool ok = dlsdev::stage_a(L, 3, pd, 5, pd + 2, 5, samples + hyp * 3, uvals + 4 * (size_t)(iter_base[p] + b),
                                  action + hyp * 729, tfac + hyp * 27);
  if (threadIdx.x == 0) okflag[hyp] = ok ? 1 : 0;
}

// ---- gDLS similarity hypotheses (estimate_similarity_transformation_2d_3d.cc:85-133): the DLS pipeline on four
// camera-bearing correspondences -- stage A with the generalised cost matrix, the same eigen stage, solutions with scale
__global__ __launch_bounds__(dlsdev::kThreads, 3) void k_gdls_a(int nprob, int B, int stride, const int64_t* __restrict__ offsets,
                                               const double* __restrict__ data, const int* __restrict__ samples,
                                               const int* __restrict__ active_iters, const int* __restrict__ iter_base,
                                               const double* __restrict__ uvals, double* __restrict__ action,
                                               double* __restrict__ tfac, int* __restrict__ okflag) {
  __shared__ dlsdev::WgLds L;
  const int b = blockIdx.x, p = blockIdx.y;
  if (b >= active_iters[p]) return;
  const size_t hyp = (size_t)p * B + b;
  const double* pd = data + (size_t)offsets[p] * stride;
  const bool ok = dlsdev::stage_a<true>(L, 4, pd, stride, pd + 3, stride, samples + hyp * 4, uvals + 4 * (size_t)(iter_base[p] + b),
                                        action + hyp * 729, tfac + hyp * 36, pd + 9, stride);
  if (threadIdx.x == 0) okflag[hyp] = ok ? 1 : 0;
}

// DlsPnp on problems of any size (the directly bound solver, sfm.cc:577): a wave per problem, then a thread per problem
__global__ __launch_bounds__(dlsdev::kThreads, 3) void k_dls_solve_a(const int64_t* __restrict__ offsets, const double* __restrict__ feat,
                                                    const double* __restrict__ world, const double* __restrict__ uvals,
                                                    double* __restrict__ action, double* __restrict__ tfac, int* __restrict__ okflag) {
  __shared__ dlsdev::WgLds L;
  const int i = blockIdx.x;
  const int64_t o = offsets[i];
  const int n = (int)(offsets[i + 1] - o);
  bool ok = false;
  if (n >= 3) ok = dlsdev::stage_a(L, n, feat + 2 * o, 2, world + 3 * o, 3, nullptr, uvals + 4 * (size_t)i, action + (size_t)i * 729, tfac + (size_t)i * 27);
  if (threadIdx.x == 0) okflag[i] = ok ? 1 : 0;
}
}  // namespace

int dls_ensure_tables() {
  static std::once_flag once;
  static int rc = 0;
  std::call_once(once, [] {
    dls::Tables t;
    dls::build_tables(&t);
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(dlsdev::c_tab), &t, sizeof(t));
    if (e != hipSuccess) rc = set_error(THEIA_HIP_ERR_NO_DEVICE, "hipMemcpyToSymbol(dls tables) failed: %s", hipGetErrorString(e));
  });
  return rc;
}

void launch_dls_stage_a(bool gdls, int datum_stride, int nprob, int B, const int64_t* offsets, const double* data, const int* samples,
                        const int* active_iters, const int* iter_base, const double* uvals, double* action, double* tfac, int* okflag,
                        hipStream_t st) {
  if (gdls) k_gdls_a<<<dim3(B, nprob), dlsdev::kThreads, 0, st>>>(nprob, B, datum_stride, offsets, data, samples, active_iters, iter_base, uvals, action, tfac, okflag);
  else k_dls_a<<<dim3(B, nprob), dlsdev::kThreads, 0, st>>>(nprob, B, offsets, data, samples, active_iters, iter_base, uvals, action, tfac, okflag);
}
void launch_dls_solve_a(int num, const int64_t* offsets, const double* feat, const double* world, const double* uvals, double* action,
                        double* tfac, int* okflag, hipStream_t st) {
  k_dls_solve_a<<<num, dlsdev::kThreads, 0, st>>>(offsets, feat, world, uvals, action, tfac, okflag);
}

}  // namespace thip

#ifdef THIP_DLS_STAMPS
// development: the section stamps of stage_a (dls_stage_a.h) summed since the last call; out[16]
extern "C" int theia_hip_debug_dls_stamps(unsigned long long* out) {
  unsigned long long zero[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(thip::dlsdev::g_dls_stamps), sizeof(zero)) != hipSuccess) return -1;
  (void)hipMemcpyToSymbol(HIP_SYMBOL(thip::dlsdev::g_dls_stamps), zero, sizeof(zero));
  return 0;
}
#endif
