// wave_reduce.h -- the pivot searches' wave-wide maximum of |a| on the DPP network.
// A double with a clear sign bit orders like its bit pattern, so the maximum over the 64 lanes is two unsigned 32-bit
// reductions -- the high words, then the low words of the lanes that hold the high maximum -- of four `v_max_u32_dpp` and four
// `v_readlane_b32` each, instead of six rounds of two `ds_bpermute_b32` (a `__shfl_xor` of a double) and a canonicalising
// `v_max_f64`: the searches sit on the critical path of every elimination step (dls_stage_a.h has the measurement).
// The result is the bit pattern `fmax` over the lanes returns: the same comparisons follow.
#ifndef THEIA_HIP_WAVE_REDUCE_H_
#define THEIA_HIP_WAVE_REDUCE_H_
#include <hip/hip_runtime.h>

namespace thip {

__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
  v = max(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
  v = max(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
  v = max(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xf, 0xf, true));   // row_half_mirror
  v = max(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x140, 0xf, 0xf, true));   // row_mirror
  return max(max((unsigned)__builtin_amdgcn_readlane((int)v, 0), (unsigned)__builtin_amdgcn_readlane((int)v, 16)),
             max((unsigned)__builtin_amdgcn_readlane((int)v, 32), (unsigned)__builtin_amdgcn_readlane((int)v, 48)));
}

// all-reduce inside every row of 16 lanes (= a 16-lane team aligned to a DPP row)
__device__ __forceinline__ unsigned row16_max_u32(unsigned v) {
  v = max(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true));
  v = max(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xf, 0xf, true));
  v = max(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xf, 0xf, true));
  v = max(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x140, 0xf, 0xf, true));
  return v;
}
__device__ __forceinline__ unsigned row16_min_u32(unsigned v) {
  v = min(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true));
  v = min(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xf, 0xf, true));
  v = min(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xf, 0xf, true));
  v = min(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x140, 0xf, 0xf, true));
  return v;
}

// the value lane `l` holds (l uniform): two v_readlane_b32 instead of the two ds_bpermute_b32 of a __shfl
__device__ __forceinline__ double lane_value(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

// max over the wave of the lanes' values that are >= 0 with a clear sign bit (|a|; a negative sentinel or a NaN is no
// candidate, as under `a > best`); -1.0 when no lane has one
__device__ __forceinline__ double wave_max_abs(double v) {
  const bool valid = v >= 0.0;
  const unsigned hi = valid ? (unsigned)__double2hiint(v) : 0u, lo = (unsigned)__double2loint(v);
  const unsigned hm = wave_max_u32(hi);
  const unsigned lm = wave_max_u32((valid & (hi == hm)) ? lo : 0u);
  return __builtin_amdgcn_ballot_w64(valid) != 0ull ? __hiloint2double((int)hm, (int)lm) : -1.0;
}

}  // namespace thip
#endif
