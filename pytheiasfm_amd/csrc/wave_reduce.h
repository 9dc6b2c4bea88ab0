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

// The XOR butterfly  for (off = 32; off > 0; off >>= 1) v += shfl_xor(v, off)  without its twelve ds_bpermute_b32: the levels 32
// and 16 on gfx950's v_permlane32_swap / v_permlane16_swap (both halves of the swap added: lane l gets v[l] + v[l ^ off],
// the addition commutes), the level 8 by a row rotation, and the levels 4, 2, 1 by ANY lane of the partner's residue class
// (after the levels above it every lane of a class holds the same sum): row_ror:4 and two quad_perms.  The same association as
// the butterfly, hence its bits (the oracle's wave-order sums restate the butterfly: oracle/ransac_oracle.cpp sum_in_order).
template <int CTRL>
__device__ __forceinline__ double wr_dpp_d(double v) {
  return __hiloint2double(__builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, false),
                          __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ double wave_sum_butterfly(double v) {
  {
    const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(v), __double2loint(v), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(v), __double2hiint(v), false, false);
    v = __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
  }
  {
    const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(v), __double2loint(v), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(v), __double2hiint(v), false, false);
    v = __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
  }
  v += wr_dpp_d<0x128>(v);   // row_ror:8
  v += wr_dpp_d<0x124>(v);   // row_ror:4
  v += wr_dpp_d<0x4E>(v);    // quad_perm [2,3,0,1]
  v += wr_dpp_d<0xB1>(v);    // quad_perm [1,0,3,2]
  return v;
}
__device__ __forceinline__ int wave_sum_butterfly(int v) {
  { const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false); v = (int)r[0] + (int)r[1]; }
  { const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false); v = (int)r[0] + (int)r[1]; }
  v += __builtin_amdgcn_mov_dpp(v, 0x128, 0xf, 0xf, false);
  v += __builtin_amdgcn_mov_dpp(v, 0x124, 0xf, 0xf, false);
  v += __builtin_amdgcn_mov_dpp(v, 0x4E, 0xf, 0xf, false);
  v += __builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, false);
  return v;
}
__device__ __forceinline__ double wave_max_butterfly(double v) {   // fmax over the 64 lanes (any tree: fmax is exact)
  {
    const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(v), __double2loint(v), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(v), __double2hiint(v), false, false);
    v = fmax(__hiloint2double((int)hi[0], (int)lo[0]), __hiloint2double((int)hi[1], (int)lo[1]));
  }
  {
    const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(v), __double2loint(v), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(v), __double2hiint(v), false, false);
    v = fmax(__hiloint2double((int)hi[0], (int)lo[0]), __hiloint2double((int)hi[1], (int)lo[1]));
  }
  v = fmax(v, wr_dpp_d<0x128>(v));
  v = fmax(v, wr_dpp_d<0x124>(v));
  v = fmax(v, wr_dpp_d<0x4E>(v));
  v = fmax(v, wr_dpp_d<0xB1>(v));
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
