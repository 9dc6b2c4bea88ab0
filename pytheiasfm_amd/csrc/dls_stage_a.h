// DLS-PnP stage A on the device: cost matrix -> cost quartic -> Jacobian cubics -> the reference's dense partial-pivot LU
// of the 93 x 93 Macaulay block with the augmented matrix in registers (the design is described at the top of
// dls_device.h).  Included by dls_kernels.hip only: that file is compiled with SimplifyCFG's common-code sinking off --
// the sinking merges the three "this lane's row q is the pivot row" blocks into one block that indexes the register
// array by a runtime row, and a runtime index sends all 60 doubles of every lane to scratch.
#ifndef THEIA_HIP_DLS_STAGE_A_H_
#define THEIA_HIP_DLS_STAGE_A_H_

#include "dls_device.h"

namespace thip {
namespace dlsdev {

constexpr int kUStore = 4644;   // pivot rows, see u_base()

__constant__ dls::Tables c_tab;

struct WgLds {
  union {
    struct { double Dm[81]; double J[36]; double hinv[16]; double traw[36]; double f[60]; } fe;   // front end
    double U[kUStore];                                                                          // elimination
  };
  double lbuf[2][96];    // factors of a step by parity
  double diag[96];
  double T[27];
  double sf[9];          // gDLS: the scale factor row
  double u[4];
  int pinfo[2][2];       // (row that holds the pivot, its position) by parity
  int flag;
  unsigned char prow_of[96];
#ifdef THIP_DLS_STAMPS
  unsigned long long st[5];
#endif
};

// Eigen's Matrix4d::inverse() restated as adjugate over determinant (oracle/dls_oracle.h: gdls_inverse4)
__device__ inline bool inverse4(const double* a, double* inv) {
  auto m3 = [&](int r0, int r1, int r2, int c0, int c1, int c2) {
    return a[4 * r0 + c0] * (a[4 * r1 + c1] * a[4 * r2 + c2] - a[4 * r1 + c2] * a[4 * r2 + c1]) -
           a[4 * r0 + c1] * (a[4 * r1 + c0] * a[4 * r2 + c2] - a[4 * r1 + c2] * a[4 * r2 + c0]) +
           a[4 * r0 + c2] * (a[4 * r1 + c0] * a[4 * r2 + c1] - a[4 * r1 + c1] * a[4 * r2 + c0]);
  };
  double cof[16];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int r0 = r == 0 ? 1 : 0, r1 = r <= 1 ? 2 : 1, r2 = r <= 2 ? 3 : 2;
      const int c0 = c == 0 ? 1 : 0, c1 = c <= 1 ? 2 : 1, c2 = c <= 2 ? 3 : 2;
      const double minor = m3(r0, r1, r2, c0, c1, c2);
      cof[4 * r + c] = ((r + c) & 1) ? -minor : minor;
    }
  const double det = ((a[0] * cof[0] + a[1] * cof[1]) + a[2] * cof[2]) + a[3] * cof[3];
  if (det == 0.0) return false;
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) inv[4 * r + c] = cof[4 * c + r] / det;
  return true;
}

// Storage of the pivot rows: the row of step k (o = k / 6) is written ONCE, whole -- every live register of its six
// lanes, i.e. the columns 6 o .. 119 + garbage above -- at u_base(k) + (column - 6 o); the other lanes read it from there
// during the step, and the back-substitution reads its entries right of the diagonal again.  What a row writes beyond
// column 92 lands in the regions of later rows, which are written later: nothing is predicated, every offset is static.
__host__ __device__ constexpr int u_base(int k) { return (k / 6) * 558 - 18 * (k / 6) * (k / 6 - 1) + (k % 6) * (kBlock - 6 * (k / 6)); }
static_assert(u_base(6) == 6 * 93 && u_base(12) == 6 * 93 + 6 * 87 && u_base(92) + 48 <= kUStore, "pivot row store");

template <int CTRL>
__device__ __forceinline__ double dpp_move_d(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_d(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ double dls_dpp_max16(double v) {   // all-reduce inside every row of 16 lanes
  v = fmax(v, dpp_move_d<0xB1>(v));    // quad_perm [1,0,3,2]
  v = fmax(v, dpp_move_d<0x4E>(v));    // quad_perm [2,3,0,1]
  v = fmax(v, dpp_move_d<0x141>(v));   // row_half_mirror
  v = fmax(v, dpp_move_d<0x140>(v));   // row_mirror
  return v;
}
__device__ __forceinline__ unsigned dls_dpp_umax16(unsigned v) {   // all-reduce inside every row of 16 lanes
  v = max(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true));
  v = max(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xf, 0xf, true));
  v = max(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xf, 0xf, true));
  v = max(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x140, 0xf, 0xf, true));
  return v;
}
__device__ __forceinline__ int dls_dpp_min16(int v) {
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, false));
  return v;
}

// One elimination step k = 6 o + s with NL live registers per row (NL >= 20 - o).  SHIFT (s == 5): the step's results
// land one register lower, so that the next six steps find their pivot columns in register 0 again.
// pos[q]: position of the lane's row q in the oracle's (swapped) row order; a row pivoted at step j keeps pos = j, so
// "still a candidate" is k <= pos < 93 (the padding rows 93..95 start at their own index and never are).
#define THIP_DLS_FENCE() asm volatile("" ::: "memory")
#ifdef THIP_DLS_STAMPS
__device__ unsigned long long g_dls_stamps[16];   // [8 ..]: one elimination step, thread 0: {pivot block, barrier, row out, update, steps}
#define DLS_STEP_T0 unsigned long long st_ = __builtin_amdgcn_s_memtime()
#define DLS_STEP(k) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) L.st[k] += n_ - st_; st_ = n_; } while (0)
#else
#define DLS_STEP_T0 do {} while (0)
#define DLS_STEP(k) do {} while (0)
#endif
// (the three instances carry different asm comments: identical blocks are merged by the compiler's code sinking into ONE
// block that indexes a[][] by a runtime row, which sends the whole register array to scratch)
template <int NL, int Q>
__device__ __forceinline__ void lu_pivot_row_out(double* __restrict__ ub, const double (&a)[3][20]) {
  if (Q == 0) asm volatile("; pivot row = register row 0"); else if (Q == 1) asm volatile("; pivot row = register row 1"); else asm volatile("; pivot row = register row 2");
#pragma unroll
  for (int i = 0; i < NL; ++i) ub[6 * i] = a[Q][i];
}
// The pivot block of step k (column k = register 0 of column group s): search and factors, published in the step's parity
// buffers.  Called by the wave that holds the column (the half that does not rides along).
__device__ __forceinline__ void lu_pivot_block(WgLds& L, const double (&a)[3][20], const int (&pos)[3], int k, int s, int g, int rg) {
  const int par = k & 1;
    // pivot = first maximum in POSITION order: largest |a|, then smallest position.  The block below is the critical path of a
    // step -- one wave runs it while the other two wait at the barrier -- so it is written for dependent depth: no branches in
    // the choice among a lane's three rows, the maximum of the 32 lanes as two unsigned 32-bit reductions (high word, then the
    // low word among the lanes that hold the high maximum: |a| >= 0 orders like its bit pattern; a NaN never is a candidate,
    // as under the comparison av > bav), the position tie-break only when two lanes hold the same |a|, and the step's four
    // divisions side by side.
    __builtin_amdgcn_s_setprio(3);   // the workgroup's other waves wait for this block: ahead of the other workgroups' updates
    double bv = 0.0, bav = -1.0; int bp = 1 << 20;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const bool cand = (pos[q] >= k) & (pos[q] < kBlock);
      const double av = fabs(a[q][0]);
      const bool better = cand & ((av > bav) | ((av == bav) & (pos[q] < bp)));
      bv = better ? a[q][0] : bv; bav = better ? av : bav; bp = better ? pos[q] : bp;
    }
    const int base = (s / 3) << 5;
    const bool valid = bav >= 0.0;
    const unsigned hi = valid ? (unsigned)__double2hiint(bav) : 0u, lo = (unsigned)__double2loint(bav);
    const unsigned h16 = dls_dpp_umax16(hi);
    const unsigned hm = max((unsigned)__builtin_amdgcn_readlane((int)h16, base), (unsigned)__builtin_amdgcn_readlane((int)h16, base + 16));
    const bool top = valid & (hi == hm);
    const unsigned l16 = dls_dpp_umax16(top ? lo : 0u);
    const unsigned lm = max((unsigned)__builtin_amdgcn_readlane((int)l16, base), (unsigned)__builtin_amdgcn_readlane((int)l16, base + 16));
    const bool match = top & (lo == lm);
    unsigned own = (unsigned)(__builtin_amdgcn_ballot_w64(match) >> base);
    if (__builtin_popcount(own) != 1) {   // the same |a| in two lanes (or no candidate that is a number): smallest position
      const int p16 = dls_dpp_min16(match ? bp : (1 << 20));
      const int gpt = min(__builtin_amdgcn_readlane(p16, base), __builtin_amdgcn_readlane(p16, base + 16));
      own = (unsigned)(__builtin_amdgcn_ballot_w64(match & (bp == gpt)) >> base);
    }
    const int src = base + (own ? __builtin_ctz(own) : 0);
    const int gp = __builtin_amdgcn_readlane(bp, src);
    double piv = readlane_d(bv, src);
    if (g == s) {
      if (!(own != 0u) || piv == 0.0) { L.flag = 1; piv = 1.0; }   // singular block: the oracle gives up (no models); finish harmlessly
      // (the four quotients over one pivot on a shared reciprocal -- the division sequence's first six instructions once, its
      // last five per numerator, bit-identical to `/` on 2^24 operand pairs -- measured 0.6 % SLOWER than the plain divisions)
      double qd[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) qd[q] = a[q][0] / piv;
      const double ip = 1.0 / piv;
      int mq = -1;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const bool cand = (pos[q] >= k) & (pos[q] < kBlock), mine = cand & (pos[q] == gp);
        L.lbuf[par][3 * rg + q] = (!cand | mine) ? 0.0 : qd[q];
        mq = mine ? q : mq;
      }
      if (mq >= 0) { L.pinfo[par][0] = 3 * rg + mq; L.pinfo[par][1] = gp; L.diag[k] = ip; L.prow_of[k] = (unsigned char)(3 * rg + mq); }
    }
    __builtin_amdgcn_s_setprio(0);
}

// a step whose pivot-row hand-over and update a wave has put off (see lu_step)
struct LuPending { bool on; double l[3]; int pr, urow, skipg; };

// The pivot row of a step out to the store and back, and the step's multiply-adds on this lane's rows.  l: the NEGATED factors;
// skipg: the column group whose register 0 has taken this step already (look-ahead), or -1.
template <int NL>
__device__ __forceinline__ void lu_row_out(WgLds& L, const double (&a)[3][20], int pr, int urow, int g, int rg) {
  // the six lanes of the pivot row put it where everybody reads it (pr is uniform: no selects)
  double* ub = L.U + urow + g;   // urow = u_base(k)
  const int prg = pr / 3, pq = pr - 3 * prg;
  if (pq == 0) { if (rg == prg) lu_pivot_row_out<NL, 0>(ub, a); }
  else if (pq == 1) { if (rg == prg) lu_pivot_row_out<NL, 1>(ub, a); }
  else { if (rg == prg) lu_pivot_row_out<NL, 2>(ub, a); }
}
// ROW_IS_OUT: the pivot row went out earlier (a deferred step: right behind the look-ahead, so that its way through LDS is
// behind this wave by the time it catches up)
template <int NL, bool SHIFT, bool ROW_IS_OUT = false>
__device__ __forceinline__ void lu_row_and_update(WgLds& L, double (&a)[3][20], const double (&l)[3], int pr, int urow, int g, int rg, int skipg) {
  double* ub = L.U + urow + g;
  if (!ROW_IS_OUT) lu_row_out<NL>(L, a, pr, urow, g, rg);
  // The piece of the pivot row at ub (= the columns of group g) is written by lane (g, prg) and read, in this step, by the lanes
  // of column group g only: writer and readers are lanes of ONE wave, whose LDS operations complete in order -- a wavefront
  // fence orders them; the workgroup barrier that stood here (round 4 - 5) made every step wait twice for its slowest wave.
  // (Other waves read these rows in the back-substitution, many barriers later.)
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  constexpr int I0 = SHIFT ? 1 : 0, D = SHIFT ? 1 : 0;
  const bool skip0 = (g == skipg);
#pragma unroll
  for (int c0 = I0; c0 < NL; c0 += 4) {   // four columns at a time: their LDS reads in flight together, no more
    double pv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) if (c0 + j < NL) pv[j] = ub[6 * (c0 + j)];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (c0 + j < NL) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const double v = __builtin_fma(l[q], pv[j], a[q][c0 + j]);
          if (!SHIFT && c0 + j == 0) a[q][0] = skip0 ? a[q][0] : v; else a[q][c0 + j - D] = v;
        }
      }
    THIP_DLS_FENCE();
  }
  if (SHIFT) {
#pragma unroll
    for (int q = 0; q < 3; ++q) a[q][NL - 1] = 0.0;
  }
}

// One elimination step.  The step is a dependent chain: pivot block (one wave) -> barrier -> pivot position -> pivot row out
// and back -> multiply-adds; with the column groups dealt to the waves so that consecutive pivot columns sit in DIFFERENT waves,
// the wave that holds column k + 1 takes step k on that column alone (the pivot row's entry comes from its own registers),
// runs the pivot block of step k + 1 at once -- while the other two waves do their hand-over and update of step k -- and puts
// its own hand-over and update of step k off until after the next barrier (`pend`): the chain of a step is the pivot block
// alone.  Every entry still takes the steps in order, with the same operands: the same bits.  lookahead: step k + 1 belongs to
// this group of six (no look-ahead across the register shift).
template <int NL, bool SHIFT>
__device__ __forceinline__ void lu_step(WgLds& L, double (&a)[3][20], int (&pos)[3], int k, int s, int g, int rg, int urow, LuPending& pend,
                                        bool lookahead) {
  const int par = k & 1, wave = (int)(threadIdx.x >> 6);
  DLS_STEP_T0;
  if (s == 0 && wave == 0) lu_pivot_block(L, a, pos, k, 0, g, rg);   // (the later steps' blocks ran one step ahead)
  DLS_STEP(0);
  __syncthreads();
  DLS_STEP(1);
  double l[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) l[q] = -L.lbuf[par][3 * rg + q];
  const int pr = __builtin_amdgcn_readfirstlane(L.pinfo[par][0]), pp = __builtin_amdgcn_readfirstlane(L.pinfo[par][1]);
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    if (pos[q] == pp) pos[q] = k;
    else if (pos[q] == k) pos[q] = pp;
  }
  if (pend.on) {   // the step this wave put off: before anything of step k touches its rows
    lu_row_and_update<NL, false, true>(L, a, pend.l, pend.pr, pend.urow, g, rg, pend.skipg);
    pend.on = false;
  }
  if (!SHIFT && lookahead && wave == (s + 1) % 3) {
    // column k + 1 = register 0 of group s + 1: step k on it now, with the pivot row's entry from lane prg of that half
    const int prg = pr / 3, pq = pr - 3 * prg, src = (((s + 1) / 3) << 5) + prg;
    const double u = pq == 0 ? readlane_d(a[0][0], src) : (pq == 1 ? readlane_d(a[1][0], src) : readlane_d(a[2][0], src));
    if (g == s + 1) {
#pragma unroll
      for (int q = 0; q < 3; ++q) a[q][0] = __builtin_fma(l[q], u, a[q][0]);
    }
    lu_pivot_block(L, a, pos, k + 1, s + 1, g, rg);
    lu_row_out<NL>(L, a, pr, urow, g, rg);   // (step k leaves its pivot row as it is: the row can go out now, the update waits)
    pend.on = true; pend.pr = pr; pend.urow = urow; pend.skipg = s + 1;
#pragma unroll
    for (int q = 0; q < 3; ++q) pend.l[q] = l[q];
  } else {
    lu_row_and_update<NL, SHIFT>(L, a, l, pr, urow, g, rg, -1);
  }
  DLS_STEP(3);
#ifdef THIP_DLS_STAMPS
  if (threadIdx.x == 0) L.st[4] += 1ull;
#endif
}

template <int NL>
__device__ __forceinline__ void lu_six(WgLds& L, double (&a)[3][20], int (&pos)[3], int o, int g, int rg, int& urow, LuPending& pend) {
  const int k = 6 * o, full = (k + 6 <= kBlock), ns = full ? 5 : kBlock - k, len = kBlock - k;   // 93 = 15 * 6 + 3
#pragma nounroll
  for (int s = 0; s < ns; ++s, urow += len) lu_step<NL, false>(L, a, pos, k + s, s, g, rg, urow, pend, full || s + 1 < ns);
  if (full) { lu_step<NL, true>(L, a, pos, k + 5, 5, g, rg, urow, pend, false); urow += len; }
}

// back-substitution step: row Q of every lane times 1 / u_kk -- only the lanes of row group prg hold the row pivoted at
// step k; the others compute along and their values are never read
template <int Q>
__device__ __forceinline__ void bs_scale_row(const double (&a)[3][20], double inv_ukk, double (&xq)[5]) {
#pragma unroll
  for (int i = 0; i < 5; ++i) xq[i] = a[Q][i] * inv_ukk;
}

// Development (-DTHIP_DLS_STAMPS): s_memtime ticks of thread 0 per section of stage_a, summed over the workgroups:
// {front end, register load, elimination, back-substitution, M00 - M01 X + stores, calls}
#ifdef THIP_DLS_STAMPS
#define DLS_STAMP_DECL unsigned long long ds_t = __builtin_amdgcn_s_memtime(), ds_acc[5] = {0, 0, 0, 0, 0}; if (threadIdx.x == 0) for (int k_ = 0; k_ < 5; ++k_) L.st[k_] = 0
#define DLS_STAMP(k) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); ds_acc[k] += n_ - ds_t; ds_t = n_; } while (0)
#define DLS_STAMP_FLUSH do { if (threadIdx.x == 0) { for (int k_ = 0; k_ < 5; ++k_) atomicAdd(&g_dls_stamps[k_], ds_acc[k_]); atomicAdd(&g_dls_stamps[5], 1ull); for (int k_ = 0; k_ < 5; ++k_) atomicAdd(&g_dls_stamps[8 + k_], L.st[k_]); } } while (0)
#else
#define DLS_STAMP_DECL do {} while (0)
#define DLS_STAMP(k) do {} while (0)
#define DLS_STAMP_FLUSH do {} while (0)
#endif

// points: feat[i * fstride + {0,1}], world[i * wstride + {0,1,2}] for i = index ? index[k] : k, k < npts.
// Writes action[729] (row-major) and tfac[27]; returns false when a pivot vanished (degenerate sample).
// GDLS (GdlsSimilarityTransform, gdls_similarity_transform.cc:67-175): feat holds the UNIT ray direction (3), world the
// homogeneous point (4: hnormalized here), origin the ray origin (3); tfac = translation factor (27) | scale factor (9).
// Called by all 192 threads of a workgroup; every sum runs in the order of oracle/dls_oracle.h (points in sequence).
template <bool GDLS = false>
__device__ __forceinline__ bool stage_a(WgLds& L, int npts, const double* __restrict__ feat, int fstride,
                               const double* __restrict__ world, int wstride, const int* __restrict__ index,
                               const double* __restrict__ u4, double* __restrict__ action, double* __restrict__ tfac,
                               const double* __restrict__ origin = nullptr, int ostride = 0) {
  const int tid = threadIdx.x;
  // column group g = lower halves of the waves 0 1 2, then their upper halves: the owners of consecutive steps are different waves
  const int g = (tid >> 6) + 3 * ((tid >> 5) & 1), rg = tid & 31;
  const dls::Tables& tb = c_tab;
  DLS_STAMP_DECL;
  if (tid < 4) L.u[tid] = u4[tid];
  if (tid == 0) L.flag = 0;
  if constexpr (GDLS) {
    // ---- sums over the rays: the 4 x 4 matrix H^-1 (:80-96) and the 4 x 9 helper (:101-117), one entry per lane
    if (tid < 16 + 36) {
      double acc = 0.0;
      for (int i = 0; i < npts; ++i) {
        const int id = index ? index[i] : i;
        const double* xx = feat + (size_t)id * fstride; const double* cc = origin + (size_t)id * ostride; const double* ww = world + (size_t)id * wstride;
        const double x[3] = {xx[0], xx[1], xx[2]}, c[3] = {cc[0], cc[1], cc[2]};
        const double cd = (c[0] * x[0] + c[1] * x[1]) + c[2] * x[2];
        if (tid < 16) {
          const int r = tid >> 2, k = tid & 3;
          if (tid == 0) acc += ((c[0] * c[0] + c[1] * c[1]) + c[2] * c[2]) - cd * cd;
          else if (r == 0 || k == 0) { const int a = r + k - 1; acc += -c[a] + cd * x[a]; }
          else acc += (r == k ? 1.0 : 0.0) - x[r - 1] * x[k - 1];
        } else {
          const int e = tid - 16, r = e / 9, col = e % 9, kk = col / 3;
          const double lx = ww[col % 3] / ww[3];
          if (r == 0) acc += (c[kk] - cd * x[kk]) * lx;
          else acc += (x[r - 1] * x[kk] - (r - 1 == kk ? 1.0 : 0.0)) * lx;
        }
      }
      if (tid < 16) L.fe.hinv[tid] = acc; else L.fe.traw[tid - 16] = acc;
    }
    __syncthreads();
    if (tid < 36) {
      double Hm[16], hs[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) hs[k] = L.fe.hinv[k];
      if (!inverse4(hs, Hm)) { L.flag = 1; for (int k = 0; k < 16; ++k) Hm[k] = 0.0; }
      const int r = tid / 9, col = tid % 9;
      double s2 = 0.0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        double hk = 0.0;   // Hm[4 r + k]: compile-time register indices only
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) hk = (rr == r) ? Hm[4 * rr + k] : hk;
        s2 += hk * L.fe.traw[9 * k + col];
      }
      if (r == 0) L.sf[col] = s2; else L.T[9 * (r - 1) + col] = s2;
    }
  } else {
    // ---- H^-1 = n I - sum n n^T (dls_pnp.cc:90-94) and sum (n n^T - I) L(X) (:98-103), one entry per lane
    if (tid < 9 + 27) {
      const int r = tid < 9 ? tid / 3 : (tid - 9) / 9, c = tid < 9 ? tid % 3 : ((tid - 9) % 9) / 3;
      double acc = (tid < 9 && r == c) ? (double)npts : 0.0;
      for (int i = 0; i < npts; ++i) {
        const int id = index ? index[i] : i;
        const double fx = feat[(size_t)id * fstride], fy = feat[(size_t)id * fstride + 1];
        const double nrm = sqrt((fx * fx + fy * fy) + 1.0);
        const double b[3] = {fx / nrm, fy / nrm, 1.0 / nrm};
        const double br = r == 0 ? b[0] : (r == 1 ? b[1] : b[2]), bc = c == 0 ? b[0] : (c == 1 ? b[1] : b[2]);
        if (tid < 9) acc -= br * bc;
        else acc += (br * bc - (r == c ? 1.0 : 0.0)) * world[(size_t)id * wstride + (tid - 9) % 3];
      }
      if (tid < 9) L.fe.hinv[tid] = acc; else L.fe.traw[tid - 9] = acc;
    }
    __syncthreads();
    if (tid < 27) {   // translation_factor = H * (...)  (:105), H by cofactors as Eigen's 3 x 3 inverse
      const double* a = L.fe.hinv;
      const double c00 = a[4] * a[8] - a[5] * a[7], c01 = a[5] * a[6] - a[3] * a[8], c02 = a[3] * a[7] - a[4] * a[6];
      const double det = (a[0] * c00 + a[1] * c01) + a[2] * c02;
      const double id = 1.0 / det;
      const int r = tid / 9, c = tid % 9;
      double h0, h1, h2;
      if (r == 0) { h0 = c00 * id; h1 = (a[2] * a[7] - a[1] * a[8]) * id; h2 = (a[1] * a[5] - a[2] * a[4]) * id; }
      else if (r == 1) { h0 = c01 * id; h1 = (a[0] * a[8] - a[2] * a[6]) * id; h2 = (a[2] * a[3] - a[0] * a[5]) * id; }
      else { h0 = c02 * id; h1 = (a[1] * a[6] - a[0] * a[7]) * id; h2 = (a[0] * a[4] - a[1] * a[3]) * id; }
      L.T[tid] = (h0 * L.fe.traw[c] + h1 * L.fe.traw[9 + c]) + h2 * L.fe.traw[18 + c];
    }
  }
  __syncthreads();
  // ---- D = sum W^T (I - n n^T) W, W = L(X) + T (dls_pnp.cc:111-118; gDLS: W = L(X) - c scale_factor + T, :123-133)
  if (tid < 81) {
    const int al = tid / 9, be = tid % 9;
    double d = 0.0;
    for (int i = 0; i < npts; ++i) {
      const int id = index ? index[i] : i;
      double n[3], wa[3], wb[3];
      if constexpr (GDLS) {
        const double* xx = feat + (size_t)id * fstride; const double* cc = origin + (size_t)id * ostride; const double* ww = world + (size_t)id * wstride;
        n[0] = xx[0]; n[1] = xx[1]; n[2] = xx[2];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          wa[a] = (al / 3 == a ? ww[al % 3] / ww[3] : 0.0) + (L.T[9 * a + al] - cc[a] * L.sf[al]);
          wb[a] = (be / 3 == a ? ww[be % 3] / ww[3] : 0.0) + (L.T[9 * a + be] - cc[a] * L.sf[be]);
        }
      } else {
        const double fx = feat[(size_t)id * fstride], fy = feat[(size_t)id * fstride + 1];
        const double nrm = sqrt((fx * fx + fy * fy) + 1.0);
        n[0] = fx / nrm; n[1] = fy / nrm; n[2] = 1.0 / nrm;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          wa[a] = (al / 3 == a ? world[(size_t)id * wstride + al % 3] : 0.0) + L.T[9 * a + al];
          wb[a] = (be / 3 == a ? world[(size_t)id * wstride + be % 3] : 0.0) + L.T[9 * a + be];
        }
      }
      // PW = (I - n n^T) W column be, then column al of W against it
      double s2 = 0.0;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        double pw = 0.0;
#pragma unroll
        for (int k = 0; k < 3; ++k) pw += ((r == k ? 1.0 : 0.0) - n[r] * n[k]) * wb[k];
        s2 += wa[r] * pw;
      }
      d += s2;
    }
    L.fe.Dm[tid] = d;
  }
  __syncthreads();
  // ---- the cost quartic J' = sum_ab D_ab (rbar_a rbar_b), one coefficient per lane, (a, b) in row-major order
  if (tid < dls::kJMono) {
    double s = 0.0;
    for (int ab = 0; ab < 81; ++ab) { const int c = tb.P[ab][tid]; if (c) s += L.fe.Dm[ab] * (double)c; }
    L.fe.J[tid] = s;
  }
  __syncthreads();
  if (tid < 60) L.fe.f[tid] = (double)tb.fmul[tid] * L.fe.J[tb.fsrc[tid]];   // f_i = dJ'/ds_i
  __syncthreads();
  DLS_STAMP(0);
  // ---- the augmented block [M11 | M10] into registers
  double a[3][20];
  {
    const uint32_t* code = reinterpret_cast<const uint32_t*>(tb.init[32 * g + rg]);   // (the table is laid out by column group)
#pragma unroll
    for (int w = 0; w < 15; ++w) {
      const uint32_t cw = code[w];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int cd = (cw >> (8 * b)) & 255;
        a[(4 * w + b) / 20][(4 * w + b) % 20] = cd ? L.fe.f[cd - 1] : 0.0;
      }
    }
  }
  int pos[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) pos[q] = 3 * rg + q;
  __syncthreads();   // the front end's arrays share the pivot-row store
  DLS_STAMP(1);
  // ---- elimination (oracle: dls_action_from_cost)
  int urow = 0;   // u_base(k), carried along
  LuPending pend; pend.on = false; pend.pr = 0; pend.urow = 0; pend.skipg = -1; pend.l[0] = pend.l[1] = pend.l[2] = 0.0;
  for (int o = 0; o < 4; ++o) lu_six<20>(L, a, pos, o, g, rg, urow, pend);
  for (int o = 4; o < 8; ++o) lu_six<16>(L, a, pos, o, g, rg, urow, pend);
  for (int o = 8; o < 12; ++o) lu_six<12>(L, a, pos, o, g, rg, urow, pend);
  for (int o = 12; o < 16; ++o) lu_six<8>(L, a, pos, o, g, rg, urow, pend);
  // ---- back-substitution, column oriented: register i < 5 of a row now holds right-hand side 6 (i + 15) + g - 93
  // A right-hand side never leaves its column group: the 32 lanes of a half-wave hold its entries of all 96 rows, so every
  // half-wave runs the whole substitution on its own five columns: nothing crosses a wave and there is no barrier in the loop
  // (the pivot-row store is read-only here).
  DLS_STAMP(2);
  double* Xn = action;   // the solved rows the result reads wait in the problem's own output slot
  __syncthreads();   // the factor buffer of the last elimination step becomes the hand-over buffer below
  int ucol[3];   // entry (pos[q], k) of the pivot-row store sits at ucol[q] + k
#pragma unroll
  for (int q = 0; q < 3; ++q) ucol[q] = pos[q] < kBlock ? u_base(pos[q]) - 6 * (pos[q] / 6) : 0;
  // what a step reads from LDS that does not depend on the step before it -- the pivot's row, 1 / u_kk, the three U entries of
  // this lane's rows -- is requested one step ahead: the chain of a step is the hand-over and its FMAs only
  int pr_n = (int)L.prow_of[kBlock - 1];
  double inv_n = L.diag[kBlock - 1];   // 1 / u_kk, inverted once when the pivot was found
  double u_n[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) u_n[q] = L.U[ucol[q] + kBlock - 1];
  for (int k = kBlock - 1; k >= 0; --k) {
    const int pr = __builtin_amdgcn_readfirstlane(pr_n);
    const double inv_ukk = inv_n;
    const double u0 = u_n[0], u1 = u_n[1], u2 = u_n[2];
    if (k > 0) {
      pr_n = (int)L.prow_of[k - 1]; inv_n = L.diag[k - 1];
#pragma unroll
      for (int q = 0; q < 3; ++q) u_n[q] = L.U[ucol[q] + k - 1];   // (in bounds for every row; used where pos[q] < k - 1)
    }
    const int prg = pr / 3, pq = pr - 3 * prg, slot = tb.xslot[k];
    double xq[5], x[5];
    if (pq == 0) bs_scale_row<0>(a, inv_ukk, xq); else if (pq == 1) bs_scale_row<1>(a, inv_ukk, xq); else bs_scale_row<2>(a, inv_ukk, xq);
    // the lane of row group prg hands its five values to the rest of its half-wave through LDS: writer and readers are
    // lanes of one wave (LDS operations of a wave complete in order), so a wavefront fence is all the ordering it takes
    double* xb = &L.lbuf[0][8 * g];
    if (rg == prg) {
#pragma unroll
      for (int i = 0; i < 5; ++i) xb[i] = xq[i];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
    for (int i = 0; i < 5; ++i) x[i] = xb[i];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (rg == prg && slot != 255) {
#pragma unroll
      for (int i = 0; i < 5; ++i) { const int c = 6 * i + g - 3; if (c >= 0 && c < kReduced) Xn[slot * kReduced + c] = x[i]; }
    }
#pragma unroll
    for (int q = 0; q < 3; ++q)
      if (pos[q] < k) {
        const double u = -(q == 0 ? u0 : (q == 1 ? u1 : u2));
#pragma unroll
        for (int i = 0; i < 5; ++i) a[q][i] = __builtin_fma(u, x[i], a[q][i]);
      }
  }
  __threadfence_block();
  __syncthreads();
  DLS_STAMP(3);
  // ---- M00 - M01 X, the columns of M01 in ascending order
  double res[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int e = tid + kThreads * t;
    res[t] = 0.0;
    if (e < kReduced * kReduced) {
      const int r = e / kReduced, c = e % kReduced;
      const int cd = tb.m00[r][c];
      double s = cd ? L.u[cd - 1] : 0.0;
      for (int m = 0; m < tb.m01n[r]; ++m) s = __builtin_fma(-L.u[tb.m01q[r][m]], Xn[tb.xslot[tb.m01j[r][m]] * kReduced + c], s);
      res[t] = s;
    }
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 4; ++t) { const int e = tid + kThreads * t; if (e < kReduced * kReduced) action[e] = res[t]; }
  if (tid < 27) tfac[tid] = L.T[tid];
  if (GDLS && tid < 9) tfac[27 + tid] = L.sf[tid];
  DLS_STAMP(4);
  DLS_STAMP_FLUSH;
  return L.flag == 0;
}

}  // namespace dlsdev
}  // namespace thip
#endif
