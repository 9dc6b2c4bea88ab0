// sparse_cholesky.hip -- K3 for structured reduced camera systems: a tile-sparse
// (64 x 64 tiles), nested-dissection ordered, LEVEL-SCHEDULED Cholesky.
//
// The reference hands the reduced camera matrix to Ceres' SPARSE_SCHUR with a
// fill-reducing ordering (bundle_adjustment.h:98-104, bundle_adjuster.cc:63-89),
// i.e. it exploits the camera co-visibility sparsity.  A dense right-looking
// factorisation is a chain of n/64 dependent panel steps and on MI355X every
// step is latency bound (dense_cholesky.hip).  Here the 64-wide tiles of S are
// reordered by nested dissection of the tile co-visibility graph (host, once
// per problem), the tile elimination tree is cut into levels, and ALL tiles of
// a level are factored by one launch: the dependent chain is the tree height
// (5-8 for ring / band like camera graphs), not the tile count.
//
// Storage is unchanged: S row-major n x lda with the rhs as row n; the
// permutation is applied by tile indirection only (tile (I,J) of P S P^T lives
// at physical tile (perm[I], perm[J]); tiles that land in the physical upper
// triangle are first mirrored from the lower one).
#include "ba_kernels.h"
#include "cholesky_device.h"

#include <algorithm>
#include <cstdlib>
#include <cstdio>
#include <functional>
#include <map>
#include <vector>

namespace thip {
namespace {

using namespace chol;

// ---- per-level work lists (int32 records in one device array) ----
// potrf : {k0, nb, slot}
// trsm  : {row0, h, k0, nb, slot}
// upd   : {row0, h, col0, w, sbeg, send, diag}   sources: {k0, nb}
// back  : {k0, nb, slot, sbeg, send}             sources: {row0, h}
// symm  : {r0, h, c0, w}   dst tile (r0, c0) <- transpose of tile (c0, r0)

__global__ __launch_bounds__(256) void k_sp_potrf(double* __restrict__ A, int lda, const int* __restrict__ items,
                                                  double* __restrict__ Linv, double* __restrict__ fail_flag) {
  const int* it = items + 3 * blockIdx.x;
  potrf64_wg(A, lda, it[0], it[1], Linv + (size_t)it[2] * NB * NB, fail_flag);
}

// stage a 64 x 64 tile (h valid rows, w valid columns, zero padded) in LDS; 16
// independent loads per thread are in flight before the first LDS store
__device__ __forceinline__ void load_tile(double (*T)[LDP], const double* __restrict__ A, int lda, int row0, int h,
                                          int col0, int w, int tid) {
  double v[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int t = tid + 256 * q, r = t >> 6, c = t & 63;
    v[q] = A[(size_t)(row0 + min(r, h - 1)) * lda + col0 + min(c, w - 1)];   // unconditional: no branch + wait per load
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int t = tid + 256 * q, r = t >> 6, c = t & 63;
    T[r][c] = (r < h && c < w) ? v[q] : 0.0;
  }
}

// X = P * Linv^T for one tile: X[r][j] = sum_i P[r][i] Z[j][i]
__global__ __launch_bounds__(256) void k_sp_trsm(double* __restrict__ A, int lda, const int* __restrict__ items,
                                                 const double* __restrict__ Linv) {
  __shared__ double P[NB][LDP];
  __shared__ double Z[NB][LDP];
  const int* it = items + 5 * blockIdx.x;
  const int row0 = it[0], h = it[1], k0 = it[2], nb = it[3];
  const int tid = threadIdx.x;
  load_tile(Z, Linv + (size_t)it[4] * NB * NB, NB, 0, NB, 0, NB, tid);
  load_tile(P, A, lda, row0, h, k0, nb, tid);
  __syncthreads();
  const int wv = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  if (16 * wv >= h) return;
  double4_t acc[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) acc[q] = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int kk = 0; kk < NB; kk += 4) {
    const double a = P[16 * wv + li][kk + lk];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Z[16 * q + li][kk + lk], acc[q], 0, 0, 0);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int r = 16 * wv + lk + 4 * reg, c = 16 * q + li;
      if (r < h && c < nb) A[(size_t)(row0 + r) * lda + k0 + c] = acc[q][reg];
    }
}

// k_sp_potrf and k_sp_trsm of a level in ONE launch: the workgroup of an off-diagonal tile (I, K) factors the diagonal tile
// K itself -- in LDS, the same arithmetic as its owner, so the same bits -- and goes straight on to X = P Linv^T with the
// inverse factor still on chip.  Redundant arithmetic on otherwise idle CUs for one dependent launch less per level
// (a level has <= ~130 such tiles on 256 CUs).  Nobody stores L(K, K): the diagonal tile is read by every workgroup of its
// column while it is being factored, and neither the updates nor the back-substitution read it (they use Linv).
//   blocks [0, npotrf): owners {k0, nb, slot} -> Linv, fail flag;  blocks npotrf ..: {row0, h, k0, nb, slot}
__global__ __launch_bounds__(256) void k_sp_potrf_trsm(double* __restrict__ A, int lda, const int* __restrict__ potrf_items, int npotrf,
                                                       const int* __restrict__ trsm_items, double* __restrict__ Linv,
                                                       double* __restrict__ fail_flag) {
  __shared__ double Ls[NB][LDP];
  __shared__ double Zs[NB][LDP];
  __shared__ double rdiag[NB];
  if ((int)blockIdx.x < npotrf) {
    const int* it = potrf_items + 3 * blockIdx.x;
    potrf64_wg_core<false>(A, lda, it[0], it[1], Ls, Zs, rdiag, Linv + (size_t)it[2] * NB * NB, fail_flag);
    return;
  }
  // FOUR workgroups per tile, 16 of its rows each (wave = 16 of the 64 columns): the 64 x 64 x 64 product would keep this
  // CU's matrix cores busy for 4096 cycles behind the factorisation (k_sp_update has the arithmetic)
  const int* it = trsm_items + 5 * ((blockIdx.x - npotrf) >> 2);
  const int rb = (blockIdx.x - npotrf) & 3;
  const int row0 = it[0], h = it[1], k0 = it[2], nb = it[3];
  if (16 * rb >= h) return;
  const int tid = threadIdx.x;
  // the workgroup's rows of P are fetched before the factorisation and parked in registers: their HBM latency hides behind
  // the panel chain
  double pv[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int t = tid + 256 * q, r = 16 * rb + (t >> 6), c = t & 63;
    pv[q] = A[(size_t)(row0 + min(r, h - 1)) * lda + k0 + min(c, nb - 1)];
  }
  potrf64_wg_core<false>(A, lda, k0, nb, Ls, Zs, rdiag, nullptr, nullptr);
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int t = tid + 256 * q, r = t >> 6, c = t & 63;
    Ls[r][c] = (16 * rb + r < h && c < nb) ? pv[q] : 0.0;   // rows 0 .. 15 of Ls: the workgroup's rows of P
  }
  __syncthreads();
  const int wv = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  double4_t acc = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int kk = 0; kk < NB; kk += 4)
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Ls[li][kk + lk], Zs[16 * wv + li][kk + lk], acc, 0, 0, 0);
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    const int r = 16 * rb + lk + 4 * reg, c = 16 * wv + li;
    if (r < h && c < nb) A[(size_t)(row0 + r) * lda + k0 + c] = acc[reg];
  }
}

// C(I,J) -= sum over the level's sources K of L(I,K) L(J,K)^T, fixed order.  FOUR workgroups per target, one 32 x 32
// quadrant each (wave = one 16 x 16 MFMA tile of it): a 64 x 64 x 64 FP64 product keeps the matrix cores of ONE CU busy for
// 4096 cycles (4 SIMDs x one v_mfma_f64_16x16x4 per 64 cycles) whatever the number of waves, so the product of a source is
// spread over four CUs instead (1024 cycles each, half the operand rows staged per workgroup).  Every entry sees the same
// MFMA sequence as before.
__global__ __launch_bounds__(256) void k_sp_update(double* __restrict__ A, int lda, const int* __restrict__ items,
                                                   const int* __restrict__ srcs) {
  __shared__ double Pr[32][LDP];
  __shared__ double Pc[32][LDP];
  const int* it = items + 7 * (blockIdx.x >> 2);
  const int qr = (blockIdx.x >> 1) & 1, qc = blockIdx.x & 1;
  const int row0 = it[0], h = it[1], col0 = it[2], w = it[3], sbeg = it[4], send = it[5], diag = it[6];
  if ((diag && qc > qr) || 32 * qr >= h || 32 * qc >= w) return;   // upper quadrant of a diagonal target / beyond a short tile
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  const int tr = wv >> 1, tc = wv & 1;
  const int hr = min(32, h - 32 * qr), wc = min(32, w - 32 * qc);   // valid rows of the two staged operand blocks
  double4_t cold, acc = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    const int r = 32 * qr + 16 * tr + lk + 4 * reg, c = 32 * qc + 16 * tc + li;
    cold[reg] = A[(size_t)(row0 + min(r, h - 1)) * lda + col0 + min(c, w - 1)];
  }
  // the operand rows of source s + 1 are in flight (registers) while the matrix cores work on source s
  double vr[8], vc[8];
  int nb = 0;
  auto fetch = [&](int s) {
    const int k0 = srcs[2 * s];
    nb = srcs[2 * s + 1];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int t = tid + 256 * q, r = t >> 6, c = min(t & 63, nb - 1);
      vr[q] = A[(size_t)(row0 + 32 * qr + min(r, hr - 1)) * lda + k0 + c];
      vc[q] = A[(size_t)(col0 + 32 * qc + min(r, wc - 1)) * lda + k0 + c];
    }
  };
  if (sbeg < send) fetch(sbeg);
  for (int s = sbeg; s < send; ++s) {
    if (s > sbeg) __syncthreads();
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int t = tid + 256 * q, r = t >> 6, c = t & 63;
      Pr[r][c] = (r < hr && c < nb) ? vr[q] : 0.0;
      Pc[r][c] = (r < wc && c < nb) ? vc[q] : 0.0;
    }
    if (s + 1 < send) fetch(s + 1);
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < NB; kk += 4)
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Pr[16 * tr + li][kk + lk], Pc[16 * tc + li][kk + lk], acc, 0, 0, 0);
  }
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    const int r = 32 * qr + 16 * tr + lk + 4 * reg, c = 32 * qc + 16 * tc + li;
    if (r < h && c < w && (!diag || c <= r)) A[(size_t)(row0 + r) * lda + col0 + c] = cold[reg] - acc[reg];
  }
}

// Deferred updates of the dense border (the shared-intrinsics tiles, ordered last): a target (I, J) with both tiles in
// the border receives a source from EVERY camera tile, and it is not read before the border's own levels.  Instead of one
// workgroup walking 30+ sources per level, the whole list is cut into chunks summed by separate workgroups
// (k_sp_update_partial -> scratch tiles) and added in chunk order (k_sp_update_reduce) right before the first border level.
// partial item: {row0, h, col0, w, sbeg, send, slot}; reduce item: {row0, h, col0, w, slot0, nslots, diag}
__global__ __launch_bounds__(256) void k_sp_update_partial(const double* __restrict__ A, int lda, const int* __restrict__ items,
                                                           const int* __restrict__ srcs, double* __restrict__ scratch) {
  // four workgroups per chunk, one 32 x 32 quadrant each, the next source's operand rows in flight (as k_sp_update)
  __shared__ double Pr[32][LDP];
  __shared__ double Pc[32][LDP];
  const int* it = items + 7 * (blockIdx.x >> 2);
  const int qr = (blockIdx.x >> 1) & 1, qc = blockIdx.x & 1;
  const int row0 = it[0], h = it[1], col0 = it[2], w = it[3], sbeg = it[4], send = it[5];
  double* out = scratch + (size_t)it[6] * NB * NB;
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  const int tr = wv >> 1, tc = wv & 1;
  double4_t acc = (double4_t){0.0, 0.0, 0.0, 0.0};
  if (32 * qr < h && 32 * qc < w) {     // (a quadrant beyond a short tile stores zeros: the reduction reads whole tiles)
    const int hr = min(32, h - 32 * qr), wc = min(32, w - 32 * qc);
    double vr[8], vc[8];
    int nb = 0;
    auto fetch = [&](int s) {
      const int k0 = srcs[2 * s];
      nb = srcs[2 * s + 1];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int t = tid + 256 * q, r = t >> 6, c = min(t & 63, nb - 1);
        vr[q] = A[(size_t)(row0 + 32 * qr + min(r, hr - 1)) * lda + k0 + c];
        vc[q] = A[(size_t)(col0 + 32 * qc + min(r, wc - 1)) * lda + k0 + c];
      }
    };
    if (sbeg < send) fetch(sbeg);
    for (int s = sbeg; s < send; ++s) {
      if (s > sbeg) __syncthreads();
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int t = tid + 256 * q, r = t >> 6, c = t & 63;
        Pr[r][c] = (r < hr && c < nb) ? vr[q] : 0.0;
        Pc[r][c] = (r < wc && c < nb) ? vc[q] : 0.0;
      }
      if (s + 1 < send) fetch(s + 1);
      __syncthreads();
#pragma unroll
      for (int kk = 0; kk < NB; kk += 4)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Pr[16 * tr + li][kk + lk], Pc[16 * tc + li][kk + lk], acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) out[(32 * qr + 16 * tr + lk + 4 * reg) * NB + 32 * qc + 16 * tc + li] = acc[reg];
}

__global__ __launch_bounds__(256) void k_sp_update_reduce(double* __restrict__ A, int lda, const int* __restrict__ items,
                                                          const double* __restrict__ scratch) {
  // 16 workgroups per target, one element per thread, the chunk tiles added in slot order with eight loads in flight
  const int* it = items + 7 * (blockIdx.x >> 4);
  const int row0 = it[0], h = it[1], col0 = it[2], w = it[3], slot0 = it[4], nslots = it[5], diag = it[6];
  const int e = (blockIdx.x & 15) * 256 + threadIdx.x, r = e >> 6, c = e & 63;
  if (r >= h || c >= w || (diag && c > r)) return;
  const double* src = scratch + (size_t)slot0 * NB * NB + e;
  double v = 0.0;
  int s = 0;
  for (; s + 8 <= nslots; s += 8) {
    double t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = src[(size_t)(s + u) * NB * NB];
#pragma unroll
    for (int u = 0; u < 8; ++u) v += t[u];
  }
  for (; s < nslots; ++s) v += src[(size_t)s * NB * NB];
  A[(size_t)(row0 + r) * lda + col0 + c] -= v;
}

// x_K = Linv_K^T (y_K - sum_I L(I,K)^T x_I), I over the (already solved) ancestors.
// A level of the back-substitution is a chain of dependent global-memory round trips (item -> source list -> operands);
// everything that does not depend on the previous link is requested up front: the tile of the inverse factor and y_K
// with the first source's operands, a source's L(I,K) columns together with its x_I.
__global__ __launch_bounds__(256) void k_sp_back(const double* __restrict__ A, int lda, const int* __restrict__ items,
                                                 const int* __restrict__ srcs, const double* __restrict__ Linv,
                                                 const double* y, double* x) {   // x may alias y (in place)
  __shared__ double xs[2][NB], yk[NB], part[4][NB];
  const int* it = items + 5 * blockIdx.x;
  const int k0 = it[0], nb = it[1], sbeg = it[3], send = it[4];
  const double* Z = Linv + (size_t)it[2] * NB * NB;
  const int tid = threadIdx.x, j = tid & 63, ch = tid >> 6;
  double zt[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) zt[q] = Z[(ch * 16 + q) * NB + j];
  const double yv = (tid < nb) ? y[k0 + tid] : 0.0;
  double s = 0.0;
  for (int q = sbeg; q < send; ++q) {
    const int row0 = srcs[2 * q], h = srcs[2 * q + 1];
    // unconditional (clamped) loads; xs is zero beyond h and columns >= nb are dropped below
    const double* col = A + (size_t)row0 * lda + k0 + min(j, nb - 1);
    double v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = col[(size_t)min(ch * 16 + r, h - 1) * lda];
    double* xb = xs[(q - sbeg) & 1];   // two buffers: one barrier per source
    if (tid < NB) xb[tid] = (tid < h) ? x[row0 + tid] : 0.0;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) s += v[r] * xb[ch * 16 + r];
  }
  part[ch][j] = s;
  __syncthreads();
  if (tid < NB) yk[tid] = (tid < nb) ? yv - ((part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid])) : 0.0;
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int q = 0; q < 16; ++q) { const int r = ch * 16 + q; t += zt[q] * yk[r]; }
  __syncthreads();
  part[ch][j] = t;
  __syncthreads();
  if (tid < nb) x[k0 + tid] = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
}

__global__ __launch_bounds__(256) void k_sp_symm(double* __restrict__ A, int lda, const int* __restrict__ items) {
  __shared__ double T[NB][LDP];
  const int* it = items + 4 * blockIdx.x;
  const int r0 = it[0], h = it[1], c0 = it[2], w = it[3];
  load_tile(T, A, lda, c0, w, r0, h, threadIdx.x);   // source tile (c0, r0): w rows, h columns
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int t = threadIdx.x + 256 * q, r = t >> 6, c = t & 63;
    if (r < h && c < w) A[(size_t)(r0 + r) * lda + c0 + c] = T[c][r];
  }
}

// ------------------------------------------------------------------ host
struct Level {
  int potrf_off = 0, npotrf = 0, trsm_off = 0, ntrsm = 0, upd_off = 0, nupd = 0, upd_src_off = 0;
  int back_off = 0, nback = 0, back_src_off = 0;
};

// nested dissection by BFS level-set separators on the tile graph
struct Orderer {
  const std::vector<std::vector<int>>& nbr;
  std::vector<int> mark, dist, out;
  int stamp = 0;
  explicit Orderer(const std::vector<std::vector<int>>& g) : nbr(g), mark(g.size(), 0), dist(g.size(), 0) {}

  // BFS inside the node set marked with `tag`; returns visit order, fills dist
  std::vector<int> bfs(int src, int tag) {
    std::vector<int> q{src};
    const int seen = ++stamp;
    std::vector<int>& d = dist;
    d[src] = 0;
    vis.resize(nbr.size());
    vis[src] = seen;
    for (size_t i = 0; i < q.size(); ++i)
      for (int v : nbr[q[i]])
        if (mark[v] == tag && vis[v] != seen) { vis[v] = seen; d[v] = d[q[i]] + 1; q.push_back(v); }
    return q;
  }
  std::vector<int> vis;

  // Small node sets are ordered EXACTLY: the height of the elimination tree is the tree-depth of the induced graph,
  //   td(G) = max over the components of G;  td(connected G) = 1 + min over v of td(G - v),
  // by memoised recursion over vertex subsets (<= 8 nodes: 256 subsets).  The separator heuristic below leaves pieces
  // like {a - b - c with a ~ c} (the short last tile of a ring makes one) that cost a level more than they have to.
  static constexpr int kExact = 8;
  void order_exact(const std::vector<int>& nodes) {
    const int m = (int)nodes.size();
    std::vector<unsigned> adjm(m, 0);
    for (int i = 0; i < m; ++i)
      for (int j = 0; j < m; ++j)
        if (i != j && std::find(nbr[nodes[i]].begin(), nbr[nodes[i]].end(), nodes[j]) != nbr[nodes[i]].end()) adjm[i] |= 1u << j;
    std::vector<signed char> td(1u << m, -1), pick(1u << m, -1);
    auto component = [&](unsigned mask, int seed) {
      unsigned comp = 1u << seed, frontier = comp;
      while (frontier) {
        unsigned next = 0;
        for (int i = 0; i < m; ++i) if (frontier >> i & 1u) next |= adjm[i] & mask & ~comp;
        comp |= next; frontier = next;
      }
      return comp;
    };
    // depth of a CONNECTED subset
    std::function<int(unsigned)> solve = [&](unsigned mask) -> int {
      if (td[mask] >= 0) return td[mask];
      if (!(mask & (mask - 1))) { td[mask] = 1; pick[mask] = (signed char)__builtin_ctz(mask); return 1; }
      int best = 127, bv = -1;
      for (int v = 0; v < m; ++v) {
        if (!(mask >> v & 1u)) continue;
        unsigned rest = mask & ~(1u << v);
        int worst = 0;
        while (rest && worst < best - 1) {
          const unsigned c = component(rest, __builtin_ctz(rest));
          worst = std::max(worst, solve(c));
          rest &= ~c;
        }
        if (!rest && worst + 1 < best) { best = worst + 1; bv = v; }
      }
      td[mask] = (signed char)best; pick[mask] = (signed char)bv;
      return best;
    };
    std::function<void(unsigned)> emit = [&](unsigned mask) {   // mask: connected
      const int v = pick[mask];
      unsigned rest = mask & ~(1u << v);
      while (rest) { const unsigned c = component(rest, __builtin_ctz(rest)); emit(c); rest &= ~c; }
      out.push_back(nodes[v]);
    };
    unsigned all = (m >= 32) ? ~0u : ((1u << m) - 1u);
    while (all) { const unsigned c = component(all, __builtin_ctz(all)); solve(c); emit(c); all &= ~c; }
  }

  void order(std::vector<int> nodes) {
    if ((int)nodes.size() <= kExact) { std::sort(nodes.begin(), nodes.end()); order_exact(nodes); return; }
    const int tag = ++stamp;
    for (int v : nodes) mark[v] = tag;
    std::vector<int> comp = bfs(nodes[0], tag);
    if (comp.size() < nodes.size()) {   // disconnected: independent subtrees
      std::vector<std::vector<int>> comps;
      std::vector<char> done(nbr.size(), 0);
      for (int v : nodes) {
        if (done[v]) continue;
        std::vector<int> c = bfs(v, tag);
        for (int u : c) done[u] = 1;
        std::sort(c.begin(), c.end());
        comps.push_back(c);
      }
      for (auto& c : comps) order(c);
      return;
    }
    const int far = comp.back();
    std::vector<int> lv = bfs(far, tag);
    const int d = dist[lv.back()];
    if (d < 2) { for (int v : nodes) out.push_back(v); return; }
    std::vector<int> cnt(d + 1, 0);
    for (int v : lv) cnt[dist[v]]++;
    int best = -1;
    const int lo = std::max(1, d / 3), hi = std::min(d - 1, (2 * d + 2) / 3);
    for (int m = lo; m <= hi; ++m) {
      if (best < 0 || cnt[m] < cnt[best] || (cnt[m] == cnt[best] && std::abs(2 * m - d) < std::abs(2 * best - d))) best = m;
    }
    std::vector<int> sep, rest;
    for (int v : nodes) (dist[v] == best ? sep : rest).push_back(v);
    // components of the rest (recursion re-marks; collect them first)
    const int tag2 = ++stamp;
    for (int v : rest) mark[v] = tag2;
    std::vector<std::vector<int>> comps;
    std::vector<char> done(nbr.size(), 0);
    for (int v : rest) {
      if (done[v]) continue;
      std::vector<int> c = bfs(v, tag2);
      for (int u : c) done[u] = 1;
      std::sort(c.begin(), c.end());
      comps.push_back(c);
    }
    for (auto& c : comps) order(c);
    for (int v : sep) out.push_back(v);
  }
};

struct Symbolic {
  std::vector<std::vector<int>> below;   // struct(K): I > K with L(I,K) != 0 (permuted indices)
  std::vector<int> level;
  int nlev = 0;
  long long ntiles = 0;
};

Symbolic symbolic(int nt, const uint8_t* adj, const std::vector<int>& perm) {
  Symbolic sy;
  std::vector<uint8_t> L((size_t)nt * nt, 0);
  for (int I = 0; I < nt; ++I)
    for (int J = 0; J < I; ++J) L[(size_t)I * nt + J] = adj ? adj[(size_t)perm[I] * nt + perm[J]] : 1;
  sy.below.resize(nt);
  sy.level.assign(nt, 0);
  for (int K = 0; K < nt; ++K) {
    std::vector<int>& s = sy.below[K];
    for (int I = K + 1; I < nt; ++I) if (L[(size_t)I * nt + K]) s.push_back(I);
    for (size_t a = 0; a < s.size(); ++a)
      for (size_t b = 0; b < a; ++b) L[(size_t)s[a] * nt + s[b]] = 1;
    if (!s.empty()) sy.level[s[0]] = std::max(sy.level[s[0]], sy.level[K] + 1);   // etree parent = min(struct)
    sy.ntiles += 1 + (long long)s.size();
  }
  for (int K = 0; K < nt; ++K) sy.nlev = std::max(sy.nlev, sy.level[K] + 1);
  return sy;
}

}  // namespace

struct CholPlan {
  int n = 0, nt = 0, nlev = 0;
  bool dense = true;
  std::vector<Level> lev;
  int symm_off = 0, nsymm = 0;
  int clear_off = 0, nclear = 0;   // {r0, h, c0, w} of every structure tile (fill included) at its physical lower position
  double flops = 0.0;              // FP64 flops of one factorisation + solve on this schedule (useful ones: h x w x nb extents)
  // deferred border updates (k_sp_update_partial / k_sp_update_reduce), run before level def_level
  int def_level = -1, def_part_off = 0, n_def_part = 0, def_src_off = 0, def_red_off = 0, n_def_red = 0;
  // sharded plan (chol_plan_create_sharded): levels [0, lev_split) factor the tile columns only THIS rank's tracks touch --
  // from this rank's partial sums, before the all-reduce --, levels [lev_split, nlev) the shared columns every rank factors
  // after it; shared_tiles = the (tile row, tile column) positions the all-reduce has to sum.  lev_split = 0 otherwise.
  int lev_split = 0;
  std::vector<int> shared_tiles;
  int* prog = nullptr;   // device
  double* scratch = nullptr;   // device: partial tiles of the deferred updates
  ~CholPlan() { if (prog) (void)hipFree(prog); if (scratch) (void)hipFree(scratch); }
};

namespace {
CholPlan* chol_plan_create_impl(int n, const uint8_t* adj_in, const uint8_t* tile_class);
}
CholPlan* chol_plan_create(int n, const uint8_t* adj) { return chol_plan_create_impl(n, adj, nullptr); }
// tile_class[nt]: 0 = shared (touched by the tracks of two or more ranks), 1 = private to THIS rank, 2 = private to another
// rank (not part of this rank's system).  adj: the union of the ranks' tile structures.  Null when the structure offers no
// level schedule (the caller then keeps the replicated plan; the decision has to be agreed between the ranks).
CholPlan* chol_plan_create_sharded(int n, const uint8_t* adj, const uint8_t* tile_class) {
  CholPlan* pl = chol_plan_create_impl(n, adj, tile_class);
  if (pl && pl->dense) { delete pl; return nullptr; }
  return pl;
}
namespace {
CholPlan* chol_plan_create_impl(int n, const uint8_t* adj_in, const uint8_t* tile_class) {
  CholPlan* pl = new CholPlan();
  pl->n = n;
  const int nt = (n + NB - 1) / NB;
  pl->nt = nt;
  if (n <= 0 || nt <= 2 || !adj_in || getenv("THEIA_HIP_DENSE_CHOLESKY")) return pl;
  // Sharded: eliminating a rank's private columns fills the shared tiles its tracks tie together -- on THAT rank; the sum of
  // the ranks' Schur complements has the union of those fills, so the shared part of every rank's structure gets the
  // clique of shared neighbours of every connected set of private tiles (anybody's).  Tiles of other ranks leave the graph.
  std::vector<uint8_t> adj_own;
  const uint8_t* adj = adj_in;
  if (tile_class) {
    adj_own.assign(adj_in, adj_in + (size_t)nt * nt);
    std::vector<int> comp(nt, -1);
    int ncomp = 0;
    for (int s0 = 0; s0 < nt; ++s0) {
      if (tile_class[s0] == 0 || comp[s0] >= 0) continue;
      std::vector<int> q{s0}, shared_nb;
      comp[s0] = ncomp;
      for (size_t i = 0; i < q.size(); ++i)
        for (int j = 0; j < nt; ++j) {
          if (j == q[i] || !adj_in[(size_t)q[i] * nt + j]) continue;
          if (tile_class[j] == 0) shared_nb.push_back(j);
          else if (comp[j] < 0) { comp[j] = ncomp; q.push_back(j); }
        }
      for (int a : shared_nb) for (int b : shared_nb) if (a != b) adj_own[(size_t)a * nt + b] = 1;
      ++ncomp;
    }
    for (int i = 0; i < nt; ++i)
      if (tile_class[i] == 2)
        for (int j = 0; j < nt; ++j) adj_own[(size_t)i * nt + j] = adj_own[(size_t)j * nt + i] = 0;
    adj = adj_own.data();
  }
  // tile graph without the (near) dense nodes, which are ordered last
  std::vector<int> deg(nt, 0);
  for (int i = 0; i < nt; ++i) for (int j = 0; j < nt; ++j) if (i != j && adj[(size_t)i * nt + j]) deg[i]++;
  std::vector<char> is_dense(nt, 0);
  for (int i = 0; i < nt; ++i) is_dense[i] = (!tile_class && deg[i] >= std::max(8, nt / 2)) ? 1 : 0;
  std::vector<std::vector<int>> nbr(nt);
  std::vector<int> sparse_nodes, dense_nodes;
  for (int i = 0; i < nt; ++i) {
    (is_dense[i] ? dense_nodes : sparse_nodes).push_back(i);
    if (is_dense[i]) continue;
    for (int j = 0; j < nt; ++j) if (i != j && !is_dense[j] && adj[(size_t)i * nt + j]) nbr[i].push_back(j);
  }
  std::vector<int> natural(nt);
  for (int i = 0; i < nt; ++i) natural[i] = i;
  std::vector<int> perm = natural;
  Symbolic best = symbolic(nt, adj, natural);
  if (tile_class) {
    // private columns first (nested dissection of their own graph), then the shared ones (of theirs, fill included: the
    // same order on every rank), then the tiles that are not this rank's
    std::vector<std::vector<int>> nb1(nt), nb0(nt);
    std::vector<int> mine, shared, other;
    for (int i = 0; i < nt; ++i) {
      (tile_class[i] == 1 ? mine : (tile_class[i] == 0 ? shared : other)).push_back(i);
      for (int j = 0; j < nt; ++j) {
        if (i == j || !adj[(size_t)i * nt + j]) continue;
        if (tile_class[i] == 1 && tile_class[j] == 1) nb1[i].push_back(j);
        if (tile_class[i] == 0 && tile_class[j] == 0) nb0[i].push_back(j);
      }
    }
    std::vector<int> cand;
    if (!mine.empty()) { Orderer o1(nb1); o1.order(mine); cand = o1.out; }
    if (!shared.empty()) { Orderer o0(nb0); o0.order(shared); cand.insert(cand.end(), o0.out.begin(), o0.out.end()); }
    cand.insert(cand.end(), other.begin(), other.end());
    if ((int)cand.size() != nt) return pl;
    perm = cand;
    best = symbolic(nt, adj, perm);
    // two phases: every shared column above every private one
    int lp = 0;
    for (int K = 0; K < nt; ++K) if (tile_class[perm[K]] == 1) lp = std::max(lp, best.level[K] + 1);
    for (int K = 0; K < nt; ++K) {
      if (tile_class[perm[K]] == 0) best.level[K] = std::max(best.level[K], lp);
      if (!best.below[K].empty()) best.level[best.below[K][0]] = std::max(best.level[best.below[K][0]], best.level[K] + 1);
    }
    best.nlev = 0;
    for (int K = 0; K < nt; ++K) if (tile_class[perm[K]] != 2) best.nlev = std::max(best.nlev, best.level[K] + 1);
    pl->lev_split = lp;
  } else if (!sparse_nodes.empty()) {
    Orderer od(nbr);
    od.order(sparse_nodes);
    std::vector<int> cand = od.out;
    for (int v : dense_nodes) cand.push_back(v);
    if ((int)cand.size() == nt) {
      Symbolic sc = symbolic(nt, adj, cand);
      if (sc.nlev < best.nlev) { best = sc; perm = cand; }
    }
  }
  if (best.ntiles >= (long long)nt * (nt + 1) / 2 || best.nlev >= nt) return pl;   // nothing to gain: dense path
  pl->dense = false;
  pl->nlev = best.nlev;

  auto r0 = [&](int I) { return I == nt ? n : perm[I] * NB; };                 // I == nt: the rhs row
  auto hh = [&](int I) { return I == nt ? 1 : std::min(NB, n - perm[I] * NB); };
  std::vector<int> prog;
  pl->lev.resize(best.nlev);
  // border = the dense nodes when they close the order: targets inside it collect their sources over all earlier levels
  int first_border = nt;
  if (!dense_nodes.empty() && !getenv("THEIA_HIP_NO_DEFERRED_BORDER")) {
    first_border = nt - (int)dense_nodes.size();
    for (size_t k = 0; k < dense_nodes.size(); ++k) if (perm[first_border + (int)k] != dense_nodes[k]) first_border = nt;   // natural order won
  }
  const int border_level = first_border < nt ? best.level[first_border] : best.nlev;
  std::map<std::pair<int, int>, std::vector<int>> deferred;
  for (int l = 0; l < best.nlev; ++l) {
    Level& lv = pl->lev[l];
    std::vector<int> ks;
    for (int K = 0; K < nt; ++K) if (best.level[K] == l && !(tile_class && tile_class[perm[K]] == 2)) ks.push_back(K);
    lv.potrf_off = (int)prog.size(); lv.npotrf = (int)ks.size();
    for (int K : ks) { prog.push_back(r0(K)); prog.push_back(hh(K)); prog.push_back(K);
      const double nb = hh(K); pl->flops += nb * nb * nb / 3.0 + nb * nb * nb / 3.0; }   // factor + triangular inverse
    lv.trsm_off = (int)prog.size();
    std::map<std::pair<int, int>, std::vector<int>> targets;
    for (int K : ks) {
      std::vector<int> s = best.below[K];
      s.push_back(nt);
      for (int I : s) { prog.push_back(r0(I)); prog.push_back(hh(I)); prog.push_back(r0(K)); prog.push_back(hh(K)); prog.push_back(K); lv.ntrsm++;
        pl->flops += 2.0 * hh(I) * (double)hh(K) * hh(K); }
      for (size_t a = 0; a < s.size(); ++a)
        for (size_t b = 0; b <= a; ++b) {
          if (s[b] == nt) continue;
          // (s[a] >= s[b]: both in the border, or the rhs row against a border tile -- the forward substitution of the
          // border rows is not read before the border's levels either)
          const bool defer = l < border_level && s[b] >= first_border;
          (defer ? deferred : targets)[{s[a], s[b]}].push_back(K);
        }
    }
    lv.upd_off = (int)prog.size(); lv.nupd = (int)targets.size();
    std::vector<int> srcs;
    for (auto& kv : targets) {
      const int I = kv.first.first, J = kv.first.second;
      prog.push_back(r0(I)); prog.push_back(hh(I)); prog.push_back(r0(J)); prog.push_back(hh(J));
      prog.push_back((int)srcs.size() / 2);
      for (int K : kv.second) { srcs.push_back(r0(K)); srcs.push_back(hh(K)); pl->flops += 2.0 * hh(I) * (double)hh(J) * hh(K); }
      prog.push_back((int)srcs.size() / 2);
      prog.push_back(I == J ? 1 : 0);
    }
    lv.upd_src_off = (int)prog.size();
    prog.insert(prog.end(), srcs.begin(), srcs.end());
    // backward substitution of this level's tiles
    lv.back_off = (int)prog.size(); lv.nback = (int)ks.size();
    std::vector<int> bsrc;
    for (int K : ks) {
      prog.push_back(r0(K)); prog.push_back(hh(K)); prog.push_back(K);
      prog.push_back((int)bsrc.size() / 2);
      for (int I : best.below[K]) { bsrc.push_back(r0(I)); bsrc.push_back(hh(I)); }
      prog.push_back((int)bsrc.size() / 2);
    }
    lv.back_src_off = (int)prog.size();
    prog.insert(prog.end(), bsrc.begin(), bsrc.end());
  }
  if (!deferred.empty()) {
    constexpr int kChunk = 6;   // sources per partial workgroup
    pl->def_level = border_level;
    std::vector<int> part, red, srcs;
    int slot = 0;
    for (auto& kv : deferred) {
      const int I = kv.first.first, J = kv.first.second;
      const int slot0 = slot;
      for (size_t c0 = 0; c0 < kv.second.size(); c0 += kChunk) {
        part.push_back(r0(I)); part.push_back(hh(I)); part.push_back(r0(J)); part.push_back(hh(J));
        part.push_back((int)srcs.size() / 2);
        for (size_t k = c0; k < std::min(kv.second.size(), c0 + kChunk); ++k) {
          const int K = kv.second[k];
          srcs.push_back(r0(K)); srcs.push_back(hh(K)); pl->flops += 2.0 * hh(I) * (double)hh(J) * hh(K);
        }
        part.push_back((int)srcs.size() / 2);
        part.push_back(slot++);
      }
      red.push_back(r0(I)); red.push_back(hh(I)); red.push_back(r0(J)); red.push_back(hh(J));
      red.push_back(slot0); red.push_back(slot - slot0); red.push_back(I == J ? 1 : 0);
    }
    pl->def_part_off = (int)prog.size(); pl->n_def_part = (int)part.size() / 7;
    prog.insert(prog.end(), part.begin(), part.end());
    pl->def_src_off = (int)prog.size();
    prog.insert(prog.end(), srcs.begin(), srcs.end());
    pl->def_red_off = (int)prog.size(); pl->n_def_red = (int)red.size() / 7;
    prog.insert(prog.end(), red.begin(), red.end());
    if (hipMalloc((void**)&pl->scratch, sizeof(double) * (size_t)slot * NB * NB) != hipSuccess) { pl->scratch = nullptr; pl->dense = true; pl->lev.clear(); return pl; }
  }
  // tiles of the factor structure that sit in the physical upper triangle
  pl->symm_off = (int)prog.size();
  for (int K = 0; K < nt; ++K)
    for (int I : best.below[K])
      if (perm[I] < perm[K]) { prog.push_back(r0(I)); prog.push_back(hh(I)); prog.push_back(r0(K)); prog.push_back(hh(K)); pl->nsymm++; }
  // every tile the assembly or the factorisation touches, at its physical lower-triangle position: the per-iteration
  // clear of the reduced system visits these instead of the whole n x n buffer (288 MB at 1000 cameras)
  pl->clear_off = (int)prog.size();
  {
    std::vector<char> seen((size_t)nt * nt, 0);
    auto add = [&](int I, int K) {
      const int pr = std::max(perm[I], perm[K]), pc = std::min(perm[I], perm[K]);
      if (seen[(size_t)pr * nt + pc]) return;
      seen[(size_t)pr * nt + pc] = 1;
      prog.push_back(pr * NB); prog.push_back(std::min(NB, n - pr * NB)); prog.push_back(pc * NB); prog.push_back(std::min(NB, n - pc * NB));
      pl->nclear++;
    };
    for (int K = 0; K < nt; ++K) { if (tile_class && tile_class[perm[K]] == 2) continue; add(K, K); for (int I : best.below[K]) add(I, K); }
    // what the all-reduce of a sharded solve sums: the shared columns of the factor structure, at the positions the
    // factorisation works on
    if (tile_class)
      for (int K = 0; K < nt; ++K) {
        if (tile_class[perm[K]] != 0) continue;
        pl->shared_tiles.push_back(perm[K]); pl->shared_tiles.push_back(perm[K]);
        for (int I : best.below[K]) { pl->shared_tiles.push_back(perm[I]); pl->shared_tiles.push_back(perm[K]); }
      }
  }
  if (getenv("THEIA_HIP_CREATE_TIMING")) {   // shape of the schedule
    fprintf(stderr, "theia_hip K3 plan: n = %d, %d tiles, %d levels, %lld factor tiles, %.1f MFLOP\n", n, nt, pl->nlev, best.ntiles, pl->flops * 1e-6);
    if (getenv("THEIA_HIP_K3_DEBUG")) {
      fprintf(stderr, "  degrees:"); for (int i = 0; i < nt; ++i) fprintf(stderr, " %d", deg[i]); fprintf(stderr, "\n  order (tile:level):");
      for (int I = 0; I < nt; ++I) fprintf(stderr, " %d:%d", perm[I], best.level[I]); fprintf(stderr, "\n");
      for (int i = 0; i < nt; ++i) { fprintf(stderr, "  nbr[%d]:", i); for (int j = 0; j < nt; ++j) if (i != j && adj[(size_t)i * nt + j]) fprintf(stderr, " %d", j); fprintf(stderr, "\n"); }
    }
    for (int l = 0; l < pl->nlev; ++l)
      fprintf(stderr, "  level %d: potrf %d, trsm %d, update targets %d\n", l, pl->lev[l].npotrf, pl->lev[l].ntrsm, pl->lev[l].nupd);
  }
  if (prog.empty()) prog.push_back(0);
  if (hipMalloc((void**)&pl->prog, sizeof(int) * prog.size()) != hipSuccess ||
      hipMemcpy(pl->prog, prog.data(), sizeof(int) * prog.size(), hipMemcpyHostToDevice) != hipSuccess) {
    pl->dense = true;   // dense schedule instead (same solution)
    pl->lev.clear();
  }
  return pl;
}
}  // namespace

namespace {
// blocks [0, nclear): one structure tile each; blocks beyond: the vector tail behind the matrix (rhs | colsq | gc | scalars),
// 4096 doubles per block -- the memset that used to follow as its own 4.7-us fill kernel
__global__ __launch_bounds__(256) void k_sp_clear(double* __restrict__ A, int lda, const int* __restrict__ items, int nclear,
                                                  double* __restrict__ tail, size_t tail_count) {
  if ((int)blockIdx.x >= nclear) {
    const size_t b0 = (size_t)(blockIdx.x - nclear) * 4096;
    for (size_t e = b0 + threadIdx.x; e < tail_count && e < b0 + 4096; e += 256) tail[e] = 0.0;
    return;
  }
  const int* it = items + 4 * blockIdx.x;
  const int r0 = it[0], h = it[1], c0 = it[2], w = it[3];
  for (int e = threadIdx.x; e < NB * NB; e += 256) {
    const int r = e >> 6, c = e & 63;
    if (r < h && c < w) A[(size_t)(r0 + r) * lda + c0 + c] = 0.0;
  }
}
}  // namespace

bool chol_plan_clear(const CholPlan* pl, double* A, int lda, hipStream_t st, double* tail, size_t tail_count) {
  if (!pl || pl->dense || pl->nclear == 0) return false;
  const int tb = tail ? (int)((tail_count + 4095) / 4096) : 0;
  k_sp_clear<<<pl->nclear + tb, 256, 0, st>>>(A, lda, pl->prog + pl->clear_off, pl->nclear, tail, tail_count);
  return true;
}

void chol_plan_destroy(CholPlan* pl) { delete pl; }
double chol_plan_flops(const CholPlan* pl) {
  if (!pl) return 0.0;
  if (pl->dense) { const double n = pl->n; return n * n * n / 3.0; }
  return pl->flops;
}
int chol_plan_levels(const CholPlan* pl) { return pl && !pl->dense ? pl->nlev : (pl ? pl->nt : 0); }

// phase 0: mirror + forward levels [0, lev_split) (a sharded plan's private columns; nothing but the mirror otherwise);
// phase 1: forward levels [lev_split, nlev), then the whole back-substitution.  chol_plan_solve = both.
void chol_plan_solve_phase(const CholPlan* pl, int phase, double* A, int lda, double* b, double* work, double* fail_flag, hipStream_t st) {
  const int n = pl->n;
  if (n <= 0) return;
  if (pl->dense) { if (phase == 1) dense_cholesky_solve(n, A, lda, b, work, fail_flag, st); return; }
  double* Linv = work;
  // back-substitution in place when the solution goes where the rhs row lives (the caller's layout): a tile
  // reads only its OWN y and the x of tiles solved by earlier launches
  const bool inplace = (b == A + (size_t)n * lda);
  double* x = inplace ? b : work + (size_t)pl->nt * NB * NB;
  const int* pg = pl->prog;
  if (phase == 0 && pl->nsymm) k_sp_symm<<<pl->nsymm, 256, 0, st>>>(A, lda, pg + pl->symm_off);
  const int l0 = phase == 0 ? 0 : pl->lev_split, l1 = phase == 0 ? pl->lev_split : (int)pl->lev.size();
  for (int li = l0; li < l1; ++li) {
    const Level& lv = pl->lev[li];
    if (li == pl->def_level && pl->n_def_part) {
      k_sp_update_partial<<<4 * pl->n_def_part, 256, 0, st>>>(A, lda, pg + pl->def_part_off, pg + pl->def_src_off, pl->scratch);
      k_sp_update_reduce<<<pl->n_def_red * 16, 256, 0, st>>>(A, lda, pg + pl->def_red_off, pl->scratch);
    }
    static const bool split = getenv("THEIA_HIP_K3_SPLIT_TRSM") != nullptr;   // development: the two launches
    if (split) {
      if (lv.npotrf) k_sp_potrf<<<lv.npotrf, 256, 0, st>>>(A, lda, pg + lv.potrf_off, Linv, fail_flag);
      if (lv.ntrsm) k_sp_trsm<<<lv.ntrsm, 256, 0, st>>>(A, lda, pg + lv.trsm_off, Linv);
    } else if (lv.npotrf + lv.ntrsm) {
      k_sp_potrf_trsm<<<lv.npotrf + 4 * lv.ntrsm, 256, 0, st>>>(A, lda, pg + lv.potrf_off, lv.npotrf, pg + lv.trsm_off, Linv, fail_flag);
    }
    if (lv.nupd) k_sp_update<<<4 * lv.nupd, 256, 0, st>>>(A, lda, pg + lv.upd_off, pg + lv.upd_src_off);
  }
  if (phase == 0) return;
  const double* y = A + (size_t)n * lda;
  for (int l = (int)pl->lev.size() - 1; l >= 0; --l) {
    const Level& lv = pl->lev[l];
    if (lv.nback) k_sp_back<<<lv.nback, 256, 0, st>>>(A, lda, pg + lv.back_off, pg + lv.back_src_off, Linv, y, x);
  }
  if (!inplace) (void)hipMemcpyAsync(b, x, sizeof(double) * n, hipMemcpyDeviceToDevice, st);
}

void chol_plan_solve(const CholPlan* pl, double* A, int lda, double* b, double* work, double* fail_flag, hipStream_t st) {
  chol_plan_solve_phase(pl, 0, A, lda, b, work, fail_flag, st);
  chol_plan_solve_phase(pl, 1, A, lda, b, work, fail_flag, st);
}
int chol_plan_split_level(const CholPlan* pl) { return pl && !pl->dense ? pl->lev_split : 0; }
// (tile row, tile column) pairs, two ints each, of the tiles a sharded plan's all-reduce has to carry
const std::vector<int>& chol_plan_shared_tiles(const CholPlan* pl) { return pl->shared_tiles; }

}  // namespace thip
