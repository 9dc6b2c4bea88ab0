// ba_lane.h -- per-lane building blocks shared by the BA kernels (ba_kernels.hip, ba_fused.hip):
// wave tiles (lane = observation), the loss-corrected / Jacobi-scaled linearisation of one observation,
// track segments inside a tile and their fixed-order reductions, the small SPD inverse.
// Reference arithmetic: src/theia/sfm/camera/reprojection_error.h:54-110 (see ba_device.h).
#pragma once
#include "ba_kernels.h"
#include "ba_device.h"

#include <type_traits>

namespace thip {
namespace {

constexpr int kWave = 64;
constexpr int kWavesPerBlock = 4;
constexpr int kBlock = kWave * kWavesPerBlock;

THIP_DEV double shfl_d(double v, int src) { return __shfl(v, src, kWave); }

THIP_DEV double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
  return v;
}
THIP_DEV double wave_max(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, kWave));
  return v;
}

// The same reductions without the LDS crossbar (ds_bpermute): DPP moves inside the rows of 16 lanes, then the four row
// totals through v_readlane.  ALL 64 lanes must be active at the call (a disabled lane's row total would be stale);
// every lane receives the same bits.  The summation tree differs from wave_sum's butterfly.
template <int CTRL>
THIP_DEV double dpp_move(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
THIP_DEV double readlane_d(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
THIP_DEV double wave_sum_all(double v) {
  v += dpp_move<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_move<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp_move<0x141>(v);   // row_half_mirror
  v += dpp_move<0x140>(v);   // row_mirror
  return (readlane_d(v, 0) + readlane_d(v, 16)) + (readlane_d(v, 32) + readlane_d(v, 48));
}
THIP_DEV double wave_max_all(double v) {
  v = fmax(v, dpp_move<0xB1>(v));
  v = fmax(v, dpp_move<0x4E>(v));
  v = fmax(v, dpp_move<0x141>(v));
  v = fmax(v, dpp_move<0x140>(v));
  return fmax(fmax(readlane_d(v, 0), readlane_d(v, 16)), fmax(readlane_d(v, 32), readlane_d(v, 48)));
}
THIP_DEV double wave_count(bool pred) { return (double)__popcll(__ballot(pred)); }

THIP_DEV void atomic_add(double* p, double v) { unsafeAtomicAdd(p, v); }

// item types / flags of the gather lists (see "gather-based Schur assembly with intrinsics")
enum { IT_CC = 0, IT_CG = 1, IT_GG0 = 2, IT_GG1 = 3, IT_CD = 4, IT_CGD = 5, IT_GD0 = 6, IT_GD1 = 7, IT_GV = 8 };
enum { ITF_ATOMIC = 1, ITF_LOWER = 2 };

template <int PD, bool INTR = false>
struct LaneLin {
  double r[2];
  double Jc[12];
  double Jt[2 * PD];
  double Jk[INTR ? 2 * THEIA_MAX_INTRINSICS : 1];  // 2 x 10 wrt the intrinsics block (INTR only)
  double X[4];
  double mc[2];   // DIRC only: F y_c (loss-corrected), the camera part of the model residual
  double cost;
  int c, p, rc;
  int g, gr;   // intrinsics group and its reduced index (-1 = constant)
  bool active, valid, pconst;
};

// Load one observation and linearise it: loss-corrected, column-masked,
// Jacobi-scaled, tangent-space blocks.  WANT_JAC=false: residual/cost only.
// ROT: `cam` is the per-camera block array of k_cam_prep (kCamRot doubles per camera) instead of [nc][6].
// DIRC (with ROT): the camera block is replaced by its product with the camera's step (P.camdir, k_cam_update): L.mc = F y_c.
template <int PD, bool WANT_JAC, bool INTR = false, bool ROT = false, unsigned MODELS = kModelsAll, bool DIRC = false>
THIP_DEV void lane_linearize(const DevProblem& P, const double* __restrict__ cam,
                             const double* __restrict__ pts, int o, bool active, int lane,
                             LaneLin<PD, INTR>& L) {
  L.active = active;
  L.g = 0; L.gr = -1;
  if (INTR) {
#pragma unroll
    for (int i = 0; i < (INTR ? 2 * THEIA_MAX_INTRINSICS : 1); ++i) L.Jk[i] = 0.0;
  }
  L.valid = true;
  L.cost = 0.0;
  L.r[0] = L.r[1] = 0.0;
  L.rc = -1;
  L.c = 0;
  L.p = -1 - lane;  // unique sentinel: inactive lanes are their own segment
  L.pconst = true;
  if (WANT_JAC) {
#pragma unroll
    for (int i = 0; i < 12; ++i) L.Jc[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 2 * PD; ++i) L.Jt[i] = 0.0;
  }
  L.X[0] = L.X[1] = L.X[2] = 0.0; L.X[3] = 1.0;
  L.mc[0] = L.mc[1] = 0.0;
  if (!active) return;
  int c = P.obs_cam[o];
  int p = P.obs_pt[o];
  if constexpr (ROT) { if (P.fused_dbg & 4) { c = 0; p = lane & 7; } }   // development: cache-resident gathers (wrong results)
  L.c = c; L.p = p;
  if constexpr (!ROT) L.rc = P.cam_red[c];
  L.pconst = P.pt_const[p] != 0;
  const double2 uv = P.obs_uv[o];
  double six = 1.0, siy = 1.0;
  if (P.obs_si) { const double2 s = P.obs_si[o]; six = s.x; siy = s.y; }
  const double4 Xv = reinterpret_cast<const double4*>(pts)[p];
  L.X[0] = Xv.x; L.X[1] = Xv.y; L.X[2] = Xv.z; L.X[3] = Xv.w;
  double ext[6];
  RotTerms rt;
  double blk[20];          // ROT: [scale (6) | intrinsics (10) | model, reduced index, pad] of the camera's block
  const bool depth_row = P.obs_kind && P.obs_kind[o];
  int g = 0, model;
  const double* intr;
  if constexpr (ROT) {
    const double* cr = cam + (size_t)kCamRot * c;
    camrot_load(cr, ext, rt);
    load_d2<20>(cr + kCamRotScale, blk);
    L.rc = (int)blk[kCamRotRed - kCamRotScale];
    if constexpr (INTR) g = (int)blk[kCamRotGroup - kCamRotScale];
    model = depth_row ? THIP_MODEL_DEPTH_ROW : (int)blk[kCamRotModel - kCamRotScale];
    intr = blk + (kCamRotIntr - kCamRotScale);
  } else {
#pragma unroll
    for (int i = 0; i < 6; ++i) ext[i] = cam[6 * c + i];
    g = P.cam_group[c];
    model = depth_row ? THIP_MODEL_DEPTH_ROW : P.group_model[g];
    intr = P.intr + (size_t)g * THEIA_MAX_INTRINSICS;
  }
  L.g = g;
  typename std::conditional<INTR, ObsLinK, ObsLin>::type ol;
  if constexpr (ROT && DIRC) {
    double dir[12];
    load_d2<12>(P.camdir + (size_t)12 * c, dir);
    observe_rot<WANT_JAC, INTR && WANT_JAC, typename std::conditional<INTR, ObsLinK, ObsLin>::type, MODELS, true>(model, ext, rt, intr, L.X, uv.x, uv.y, six, siy, ol, dir);
  } else if constexpr (ROT) observe_rot<WANT_JAC, INTR && WANT_JAC, typename std::conditional<INTR, ObsLinK, ObsLin>::type, MODELS>(model, ext, rt, intr, L.X, uv.x, uv.y, six, siy, ol);
  else observe<WANT_JAC, INTR && WANT_JAC>(model, ext, intr, L.X, uv.x, uv.y, six, siy, ol);
  L.valid = ol.valid;
  const double s = ol.r[0] * ol.r[0] + ol.r[1] * ol.r[1];
  double rho1 = 1.0, rho = s, sr = 1.0;
  if (P.loss_type != THEIA_LOSS_TRIVIAL) {   // uniform branch: the trivial loss needs no corrector (and no sqrt)
    rho = loss_eval(P.loss_type, depth_row ? P.loss_width_depth : P.loss_width, s, &rho1);
    sr = fsqrt(rho1);
  }
  L.cost = 0.5 * rho;
  L.r[0] = sr * ol.r[0];
  L.r[1] = sr * ol.r[1];
  if (WANT_JAC) {
    unsigned mask = 0u;
    if constexpr (!ROT) mask = P.cam_mask[c];
    if constexpr (DIRC) {   // y = -(delta / scale): F y_c = sr J (scale y) = -sr J delta
      L.mc[0] = -sr * ol.dc0; L.mc[1] = -sr * ol.dc1;
    } else {
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      double sc;
      if constexpr (ROT) sc = sr * blk[q];   // frozen columns carry a zero scale in the block
      else sc = ((mask >> q) & 1u) ? 0.0 : sr * P.scale_c[6 * c + q];
      L.Jc[q] = ol.Jc[q] * sc;
      L.Jc[6 + q] = ol.Jc[6 + q] * sc;
    }
    }
    if constexpr (INTR) {
      L.gr = P.grp_red[g];
      if (L.gr >= 0) {
        const unsigned fm = P.grp_free[g];
#pragma unroll
        for (int q = 0; q < THEIA_MAX_INTRINSICS; ++q) {
          const double sc = ((fm >> q) & 1u) ? sr * P.scale_i[(size_t)g * THEIA_MAX_INTRINSICS + q] : 0.0;
          L.Jk[q] = ol.Jk[q] * sc;
          L.Jk[THEIA_MAX_INTRINSICS + q] = ol.Jk[THEIA_MAX_INTRINSICS + q] * sc;
        }
      }
    }
    if (!L.pconst) {
      if (PD == 3) {
        double Jt[6];
        to_tangent(L.X, ol.Jx, Jt);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const double sp = sr * P.scale_p[(size_t)3 * p + q];
          L.Jt[q] = Jt[q] * sp;
          L.Jt[PD + q] = Jt[3 + q] * sp;
        }
      } else {
#pragma unroll
        for (int q = 0; q < PD; ++q) {
          const double sp = sr * P.scale_p[(size_t)PD * p + q];
          L.Jt[q] = ol.Jx[q] * sp;
          L.Jt[PD + q] = ol.Jx[4 + q] * sp;
        }
      }
    }
  }
}

// The 40-double block of one camera (layout: ba_device.h, kCamRot): extrinsics + rotation terms, masked Jacobi scale,
// the intrinsics of its group, model id, reduced index.  k_cam_prep (state) and k_cam_update (candidate) write them.
THIP_DEV void cam_prep_one(const DevProblem& P, int c, const double* ext, const double* __restrict__ intr, double* __restrict__ camrot) {
  double* o = camrot + (size_t)kCamRot * c;
  camrot_store(ext, o);
  const unsigned mask = P.cam_mask[c];
  for (int q = 0; q < 6; ++q) o[kCamRotScale + q] = ((mask >> q) & 1u) ? 0.0 : P.scale_c[6 * c + q];
  const int g = P.cam_group[c];
  for (int q = 0; q < THEIA_MAX_INTRINSICS; ++q) o[kCamRotIntr + q] = intr[(size_t)g * THEIA_MAX_INTRINSICS + q];
  o[kCamRotModel] = (double)P.group_model[g];
  o[kCamRotRed] = (double)P.cam_red[c];
  o[kCamRotGroup] = (double)g; o[39] = 0.0;
}

// the same with the intrinsics of the camera's group given directly (k_cam_update: the candidate's, not yet in memory)
THIP_DEV void cam_prep_one_k(const DevProblem& P, int c, const double* ext, const double (&kgroup)[THEIA_MAX_INTRINSICS], double* __restrict__ camrot) {
  double* o = camrot + (size_t)kCamRot * c;
  camrot_store(ext, o);
  const unsigned mask = P.cam_mask[c];
  for (int q = 0; q < 6; ++q) o[kCamRotScale + q] = ((mask >> q) & 1u) ? 0.0 : P.scale_c[6 * c + q];
  const int g = P.cam_group[c];
  for (int q = 0; q < THEIA_MAX_INTRINSICS; ++q) o[kCamRotIntr + q] = kgroup[q];
  o[kCamRotModel] = (double)P.group_model[g];
  o[kCamRotRed] = (double)P.cam_red[c];
  o[kCamRotGroup] = (double)g; o[39] = 0.0;
}

struct Segment {
  int start, len, maxlen, rank;  // rank = index of the track inside its tile
  bool head;
};

// Segment (= track) geometry of a lane inside its wave tile.
THIP_DEV Segment lane_segment(int p, int lane) {
  const int prev = __shfl_up(p, 1, kWave);
  const bool head = (lane == 0) || (p != prev);
  const unsigned long long H = __ballot(head);
  const unsigned long long low = (lane == 63) ? ~0ull : ((1ull << (lane + 1)) - 1ull);
  Segment s;
  s.head = head;
  s.start = 63 - __clzll((long long)(H & low));
  s.rank = __popcll(H & low) - 1;
  const unsigned long long Hn = H & ~low;
  const int end = Hn ? (__ffsll((long long)Hn) - 1) : 64;
  s.len = end - s.start;
  int m = s.len;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_xor(m, off, kWave));
  s.maxlen = m;
  return s;
}

// lane_segment for call sites where all 64 lanes are active: the neighbour through a DPP wave shift and the longest
// segment through DPP maxima instead of seven ds_bpermute round trips.
THIP_DEV Segment lane_segment_all(int p, int lane) {
  const int prev = __builtin_amdgcn_update_dpp(p, p, 0x138, 0xf, 0xf, false);   // wave_shr:1 (lane 0 keeps its own value)
  const bool head = (lane == 0) || (p != prev);
  const unsigned long long H = __ballot(head);
  const unsigned long long low = (lane == 63) ? ~0ull : ((1ull << (lane + 1)) - 1ull);
  Segment s;
  s.head = head;
  s.start = 63 - __clzll((long long)(H & low));
  s.rank = __popcll(H & low) - 1;
  const unsigned long long Hn = H & ~low;
  const int end = Hn ? (__ffsll((long long)Hn) - 1) : 64;
  s.len = end - s.start;
  int m = s.len;
  m = max(m, __builtin_amdgcn_update_dpp(0, m, 0xB1, 0xf, 0xf, false));
  m = max(m, __builtin_amdgcn_update_dpp(0, m, 0x4E, 0xf, 0xf, false));
  m = max(m, __builtin_amdgcn_update_dpp(0, m, 0x141, 0xf, 0xf, false));
  m = max(m, __builtin_amdgcn_update_dpp(0, m, 0x140, 0xf, 0xf, false));
  s.maxlen = max(max(__builtin_amdgcn_readlane(m, 0), __builtin_amdgcn_readlane(m, 16)),
                 max(__builtin_amdgcn_readlane(m, 32), __builtin_amdgcn_readlane(m, 48)));
  return s;
}

// Sum `N` per-lane values over the lanes of the segment, in lane order; every
// lane of the segment receives the (bitwise identical) total.
template <int N>
THIP_DEV void segment_allsum(const Segment& s, const double (&in)[N], double (&out)[N]) {
#pragma unroll
  for (int k = 0; k < N; ++k) out[k] = 0.0;
  for (int j = 0; j < s.maxlen; ++j) {
    const int src = (s.start + j) & 63;
    const bool take = j < s.len;
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const double v = shfl_d(in[k], src);
      if (take) out[k] += v;
    }
  }
}

// The same totals by a log-step suffix scan inside the segment (ceil(log2(maxlen)) shuffle rounds instead of
// maxlen), then a broadcast from the segment head: fixed (tree) order, every lane of a track gets identical bits.
template <int N>
THIP_DEV void segment_allsum_log(const Segment& s, int lane, double (&v)[N]) {
  const int pos = lane - s.start;
  for (int d = 1; d < s.maxlen; d <<= 1) {
    const bool take = pos + d < s.len;
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const double o = __shfl_down(v[k], d, kWave);
      if (take) v[k] += o;
    }
  }
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] = shfl_d(v[k], s.start);
}

template <int PD> constexpr int tri() { return PD * (PD + 1) / 2; }
THIP_DEV constexpr int lidx(int a, int b) { return a * (a + 1) / 2 + b; }  // a >= b

// SPD inverse (packed lower) through Cholesky; false if not positive definite.
template <int PD>
THIP_DEV bool invert_spd(const double (&V)[PD * (PD + 1) / 2], double (&Vi)[PD * (PD + 1) / 2], double (*Li_out)[PD] = nullptr) {
  double Lm[PD][PD];
  bool ok = true;
#pragma unroll
  for (int i = 0; i < PD; ++i) {
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      double s = V[lidx(i, j)];
#pragma unroll
      for (int k = 0; k < j; ++k) s -= Lm[i][k] * Lm[j][k];
      if (i == j) { if (!(s > 0.0)) ok = false; Lm[i][i] = fsqrt(s); }
      else Lm[i][j] = s / Lm[j][j];
    }
  }
  // inverse of L (lower triangular)
  double Li[PD][PD];
#pragma unroll
  for (int i = 0; i < PD; ++i) {
    Li[i][i] = 1.0 / Lm[i][i];
#pragma unroll
    for (int j = 0; j < i; ++j) {
      double s = 0.0;
#pragma unroll
      for (int k = j; k < i; ++k) s -= Lm[i][k] * Li[k][j];
      Li[i][j] = s / Lm[i][i];
    }
  }
  // Vi = Li^T Li
#pragma unroll
  for (int a = 0; a < PD; ++a)
#pragma unroll
    for (int b = 0; b <= a; ++b) {
      double s = 0.0;
#pragma unroll
      for (int k = a; k < PD; ++k) s += Li[k][a] * Li[k][b];
      Vi[lidx(a, b)] = s;
    }
  if (Li_out) {
#pragma unroll
    for (int a = 0; a < PD; ++a)
#pragma unroll
      for (int b = 0; b < PD; ++b) Li_out[a][b] = (b <= a) ? Li[a][b] : 0.0;
  }
  return ok;
}

template <int PD>
THIP_DEV double sym_get(const double (&V)[PD * (PD + 1) / 2], int a, int b) {
  return a >= b ? V[lidx(a, b)] : V[lidx(b, a)];
}

}  // namespace
}  // namespace thip
