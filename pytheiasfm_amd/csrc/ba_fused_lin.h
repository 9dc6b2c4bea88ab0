// ba_fused_lin.h -- what the fused linearise + Schur kernels share on the observation side (ba_fused.hip: k_lin_schur,
// k_backsub_runs; ba_fused_intr.hip: k_lin_schur_i): the prefetched observation stream of a sub-chunk, the per-camera blocks
// of a run staged in LDS, and the closed-form linearisation of reprojection_error.h:54-110 on them.
#pragma once
#include "ba_lane.h"

namespace thip {

template <bool ON> struct Stamp {
  unsigned long long t;
  THIP_DEV void start() { if constexpr (ON) t = __builtin_amdgcn_s_memtime(); }
  THIP_DEV void lap(unsigned long long (&acc)[12], int k) {
    if constexpr (ON) { const unsigned long long n = __builtin_amdgcn_s_memtime(); acc[k] += n - t; t = n; }
  }
};
constexpr int kCamLds = 42;   // doubles per staged camera block: 40 + 2, so that 16 consecutive blocks start on 16 distinct
                              // 16-B slots of the 256-B bank row (a ds_read_b128 of one field by 16 cameras is conflict free)

template <int PD>
struct LanePre {      // one lane's observation of a sub-chunk, loaded ahead of its phase L
  double2 uv, si;
  double4 X;
  double sp[PD];
  int p;
  unsigned lc, tl;
  bool active, pconst, depth;
};

template <int PD, int TPS>
THIP_DEV void pre_level1(const DevProblem& P, const FusedRun& run, int sc, int wv, int lane, int& tile_out, bool& tile_ok_out, LanePre<PD>& q) {
  const int tile = run.tile0 + TPS * sc + wv;
  const bool tile_ok = tile < run.tile0 + run.ntiles;
  const int tile_s = tile_ok ? tile : run.tile0;          // (loads are unconditional: a clamped, always valid address)
  const int cnt = tile_ok ? P.tile_count[tile_s] : 0;
  const int start = P.tile_start[tile_s];
  const int o = start + min(lane, max(cnt, 1) - 1);
  q.uv = P.obs_uv[o];
  q.p = P.obs_pt[o];
  q.lc = P.obs_lc[o];
  q.tl = P.obs_tl[o];
  q.si = make_double2(1.0, 1.0);
  if (P.obs_si) q.si = P.obs_si[o];
  q.depth = P.obs_kind && P.obs_kind[o];
  q.active = lane < cnt && !(P.fused_dbg & 2);
  tile_out = tile; tile_ok_out = tile_ok;
}
template <int PD>
THIP_DEV void pre_level2(const DevProblem& P, const double* __restrict__ pts, LanePre<PD>& q) {
  q.X = reinterpret_cast<const double4*>(pts)[q.p];
  q.pconst = P.pt_const[q.p] != 0;
#pragma unroll
  for (int k = 0; k < PD; ++k) q.sp[k] = P.scale_p[(size_t)PD * q.p + k];
}

// Phase L of one wave tile on prefetched registers: the linearisation of reprojection_error.h:54-110 (closed form, as
// lane_linearize / observe_rot: the camera block comes from LDS), the track sums, V^-1, the record and the slot table.
// LOSSK: 0 = trivial loss (no corrector code), 1 = Huber / SoftLOne / Tukey / Truncated, 2 = Cauchy / Arctan (log / atan2:
// their polynomial constants are hoisted into registers for the whole kernel -- an instance of its own keeps them out of
// the others' allocation).
template <int LOSSK>
THIP_DEV double loss_eval_k(int type, double a, double s, double* rho1) {
  if constexpr (LOSSK == 1) {
    switch (type) {
      case THEIA_LOSS_HUBER: case THEIA_LOSS_SOFTLONE: case THEIA_LOSS_TUKEY: case THEIA_LOSS_TRUNCATED: return loss_eval(type, a, s, rho1);
      default: *rho1 = 1.0; return s;   // (Cauchy / Arctan: not in this instance)
    }
  } else {
    return loss_eval(type, a, s, rho1);
  }
}
inline int loss_class(int type) {
  if (type == THEIA_LOSS_TRIVIAL) return 0;
  return (type == THEIA_LOSS_CAUCHY || type == THEIA_LOSS_ARCTAN) ? 2 : 1;
}

// The linearisation of one observation on prefetched registers (lane = observation): residual, cost, the 2 x 6 camera block
// (Jacobi-scaled through the block's scale row: 0 = frozen column), the 2 x PD tangent-space point block (zero for a constant
// point), and with KJAC the 2 x THEIA_MAX_INTRINSICS block wrt the intrinsics.  cb: the camera's block in LDS (kCamLds pitch).
// LOSSK: 0 = trivial loss (no corrector code), 1 = Huber / SoftLOne / Tukey / Truncated, 2 = Cauchy / Arctan (log / atan2:
// their polynomial constants are hoisted into registers for the whole kernel -- an instance of its own keeps them out of
// the others' allocation).
template <int PD, unsigned MODELS, int LOSSK, bool KJAC>
THIP_DEV void lin5(const DevProblem& P, const LanePre<PD>& c, const double* __restrict__ cb, double (&r)[2], double& cost, bool& valid,
                   double (&Jc)[12], double (&Jt)[2 * PD], double* __restrict__ Jk) {
  const bool active = c.active;
  const double X[4] = {c.X.x, c.X.y, c.X.z, c.X.w};
  // p = X - w C,  q = R p
  const double2 c01 = *reinterpret_cast<const double2*>(cb), c23 = *reinterpret_cast<const double2*>(cb + 2), c45 = *reinterpret_cast<const double2*>(cb + 4);
  const double C[3] = {c01.x, c01.y, c23.x}, w[3] = {c23.y, c45.x, c45.y};
  const double p[3] = {X[0] - X[3] * C[0], X[1] - X[3] * C[1], X[2] - X[3] * C[2]};
  const double sq = p[0] * p[0] + p[1] * p[1] + p[2] * p[2];
  const bool behind = sq < 1e-8;   // reprojection_error.h:78-80: the functor returns false, nothing is evaluated
  double R[9];
  {
    const double2 r0 = *reinterpret_cast<const double2*>(cb + 6), r1 = *reinterpret_cast<const double2*>(cb + 8), r2 = *reinterpret_cast<const double2*>(cb + 10),
                  r3 = *reinterpret_cast<const double2*>(cb + 12), r4 = *reinterpret_cast<const double2*>(cb + 14);
    R[0] = r0.x; R[1] = r0.y; R[2] = r1.x; R[3] = r1.y; R[4] = r2.x; R[5] = r2.y; R[6] = r3.x; R[7] = r3.y; R[8] = r4.x;
  }
  const double q[3] = {R[0] * p[0] + R[1] * p[1] + R[2] * p[2], R[3] * p[0] + R[4] * p[1] + R[5] * p[2], R[6] * p[0] + R[7] * p[1] + R[8] * p[2]};
  const int model = c.depth ? THIP_MODEL_DEPTH_ROW : (int)cb[kCamRotModel];
  double uvp[2], Jq[6];
  valid = project<true, KJAC, MODELS>(model, cb + kCamRotIntr, q, uvp, Jq, KJAC ? Jk : nullptr);
  r[0] = c.si.x * (uvp[0] - c.uv.x); r[1] = c.si.y * (uvp[1] - c.uv.y);
  const double s2 = r[0] * r[0] + r[1] * r[1];
  double rho1 = 1.0, rho = s2, sr = 1.0;
  if constexpr (LOSSK != 0) {
    rho = loss_eval_k<LOSSK>(P.loss_type, c.depth ? P.loss_width_depth : P.loss_width, s2, &rho1);
    sr = fsqrt(rho1);
    r[0] *= sr; r[1] *= sr;
  }
  cost = 0.5 * rho;
  // camera block 2 x 6 and ambient point block 2 x 4
  {
    // d(R p)/d(omega) applied from the left to a row jq:  A (p x jq) + B ((jq . w) p + (w . p) jq) + (jq . h) w,
    // h = -A p + cA (w x p) + cB (w . p) w   (0 for small angles: rotation_dq_dw, ba_device.h, written per row)
    const double rA = cb[15], rB = cb[16], rcA = cb[17], rcB = cb[18];
    const bool small = cb[19] != 0.0;
    const double wxp[3] = {w[1] * p[2] - w[2] * p[1], w[2] * p[0] - w[0] * p[2], w[0] * p[1] - w[1] * p[0]};
    const double d = w[0] * p[0] + w[1] * p[1] + w[2] * p[2];
    double h[3] = {0.0, 0.0, 0.0};
    if (!small) {
#pragma unroll
      for (int i = 0; i < 3; ++i) h[i] = -rA * p[i] + rcA * wxp[i] + rcB * d * w[i];
    }
    double scl[6];
    {
      const double2 s0 = *reinterpret_cast<const double2*>(cb + kCamRotScale), s1 = *reinterpret_cast<const double2*>(cb + kCamRotScale + 2),
                    s2v = *reinterpret_cast<const double2*>(cb + kCamRotScale + 4);
      scl[0] = s0.x; scl[1] = s0.y; scl[2] = s1.x; scl[3] = s1.y; scl[4] = s2v.x; scl[5] = s2v.y;   // 0 = frozen column
    }
    double v[4] = {X[0], X[1], X[2], 1.0}, beta = 0.0, nx = 1.0;   // SphereManifold<4>: householder4 / to_tangent (ba_device.h)
    if constexpr (PD == 3) {
      const double sigma = X[0] * X[0] + X[1] * X[1] + X[2] * X[2];
      nx = fsqrt(X[3] * X[3] + sigma);           // |x| (= mu of the Householder vector)
      if (sigma <= DBL_EPSILON) { if (X[3] < 0.0) beta = 2.0; }
      else {
        const double vp = (X[3] <= 0.0) ? X[3] - nx : -sigma / (X[3] + nx);
        beta = 2.0 * vp * vp / (sigma + vp * vp);
        const double ivp = 1.0 / vp;
        v[0] *= ivp; v[1] *= ivp; v[2] *= ivp;
      }
    }
    const double sia[2] = {c.si.x * sr, c.si.y * sr};
    if constexpr (KJAC) {   // 2 x THEIA_MAX_INTRINSICS wrt the intrinsics: sqrt-information and loss weights here, Jacobi scale / mask by the caller
#pragma unroll
      for (int i = 0; i < THEIA_MAX_INTRINSICS; ++i) { Jk[i] *= sia[0]; Jk[THEIA_MAX_INTRINSICS + i] *= sia[1]; }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const double* jq = Jq + 3 * a;
      const double A0 = jq[0] * R[0] + jq[1] * R[3] + jq[2] * R[6];   // A = Jq R  (1 x 3)
      const double A1 = jq[0] * R[1] + jq[1] * R[4] + jq[2] * R[7];
      const double A2 = jq[0] * R[2] + jq[1] * R[5] + jq[2] * R[8];
      const double sa = sia[a], sw = -sa * X[3];
      Jc[6 * a + 0] = sw * A0 * scl[0];   // dq/dC = -w R
      Jc[6 * a + 1] = sw * A1 * scl[1];
      Jc[6 * a + 2] = sw * A2 * scl[2];
      const double pxj[3] = {p[1] * jq[2] - p[2] * jq[1], p[2] * jq[0] - p[0] * jq[2], p[0] * jq[1] - p[1] * jq[0]};
      const double jw = jq[0] * w[0] + jq[1] * w[1] + jq[2] * w[2];
      const double jh = jq[0] * h[0] + jq[1] * h[1] + jq[2] * h[2];
#pragma unroll
      for (int k = 0; k < 3; ++k)
        Jc[6 * a + 3 + k] = sa * scl[3 + k] * (rA * pxj[k] + rB * (jw * p[k] + d * jq[k]) + jh * w[k]);
      // dq/dX = [R | -R C] from the right: [A | -A . C]
      const double j4[4] = {A0, A1, A2, -(A0 * C[0] + A1 * C[1] + A2 * C[2])};
      if constexpr (PD == 3) {
        const double jv = j4[0] * v[0] + j4[1] * v[1] + j4[2] * v[2] + j4[3] * v[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) Jt[3 * a + k] = (sa * c.sp[k]) * (nx * (j4[k] - beta * v[k] * jv));
      } else {
#pragma unroll
        for (int k = 0; k < PD; ++k) Jt[PD * a + k] = (sa * c.sp[k]) * j4[k];
      }
    }
  }
  if (behind) {
    valid = false; r[0] = 0.0; r[1] = 0.0; cost = 0.0;   // (rho(0) = 0 for every loss)
#pragma unroll
    for (int i = 0; i < 12; ++i) Jc[i] = 0.0;
    if constexpr (KJAC) {
#pragma unroll
      for (int i = 0; i < 2 * THEIA_MAX_INTRINSICS; ++i) Jk[i] = 0.0;
    }
  }
  const bool pzero = !active || c.pconst || behind;   // no point block: constant point, or nothing evaluated
  if (pzero) {
#pragma unroll
    for (int i = 0; i < 2 * PD; ++i) Jt[i] = 0.0;
  }
  if (!active) { cost = 0.0; r[0] = 0.0; r[1] = 0.0; }
}

}  // namespace thip
