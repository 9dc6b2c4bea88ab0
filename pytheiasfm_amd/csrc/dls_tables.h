// DLS-PnP (Hesch & Roumeliotis, "A Direct Least-Squares (DLS) Method for PnP"; the reference: sfm/pose/dls_pnp.cc:67-200,
// sfm/pose/dls_impl.cc:52-754): index tables of the polynomial system, generated from the DEFINITIONS at library start-up
// (the reference ships them as 60 expanded coefficient formulas and a 1968-entry index list; nothing of either is
// transcribed here).
//
//   * rotation: Cayley-Gibbs-Rodrigues, C(s) = Cbar(s) / (1 + s.s), Cbar = (1 - s.s) I - 2 [s]x + 2 s s^T  (this is the
//     TRANSPOSE of the rotation of the quaternion (1, s): dls_pnp.cc:169-172), rbar = vec_row_major(Cbar): 9 quadratics
//   * cost J' = rbar^T D rbar = sum_ab D_ab (rbar_a rbar_b) (D = ls_cost_coefficients, dls_pnp.cc:111-118): a quartic
//     with 35 coefficients, table P; f_i = dJ'/ds_i: three cubics, 20 coefficients each (dls_impl.cc:62-338 lists the sums)
//   * Macaulay matrix of {f0 = u0 + u1 s1 + u2 s2 + u3 s3, f1, f2, f3} in degree 7 (120 monomials) in the REFERENCE'S
//     layout (dls_layout.h): the 27 monomials s1^a s2^b s3^c with a, b, c <= 2 carry the rows mu * f0, the other 93 rows are
//     multiples of f1 / f2 / f3 and the other 93 columns come in the generated table's order
//     (27 x 4 + 93 x 20 = 1968 non-zeros, the count of dls_impl.cc:340-754)
//   * the solver eliminates the 93 x 93 block over those rows / columns by a dense partial-pivot LU carrying the 27 reduced
//     columns as right-hand sides (dls_pnp.cc:143-146) -- on the device one workgroup of 192 threads per problem with the
//     augmented 93 x 120 matrix in REGISTERS (dls_device.h); the tables below say what every lane's registers start as.
#ifndef THEIA_HIP_DLS_TABLES_H_
#define THEIA_HIP_DLS_TABLES_H_

#include <cstdint>
#include <cstring>

#include "dls_layout.h"

namespace thip {
namespace dls {

constexpr int kReduced = 27;      // monomials with all exponents <= 2
constexpr int kBlock = 93;        // the other monomials of degree <= 7
constexpr int kAugCols = kBlock + kReduced;   // [block | right-hand sides]
constexpr int kJMono = 35;        // monomials of degree <= 4 (the cost quartic)
// register layout of the elimination: lane = (column group g = tid / 32, row group rg = tid % 32); the lane holds the rows
// 3 rg + {0, 1, 2} at the columns 6 i + g, i < 20 (cyclic over the column groups so that all groups shrink together)
constexpr int kColGroups = 6, kRowGroups = 32, kRowsPerLane = 3, kLocals = 20, kThreads = kColGroups * kRowGroups;
static_assert(kRowGroups * kRowsPerLane >= kBlock && kColGroups * kLocals == kAugCols, "register layout");

struct Tables {
  int8_t P[81][kJMono];              // coefficient of the mu-th monomial of degree <= 4 in rbar_a rbar_b, index 9 a + b
  uint8_t fsrc[60], fmul[60];        // coefficient m of f_{1+v} (index 20 v + m) = fmul * J'[fsrc]
  uint8_t init[kThreads][64];        // lane's starting registers, index 20 q + i: 1 + (index into the 60 coefficients) or 0
  uint8_t m00[kReduced][kReduced];   // M00[r][c] = u[m00 - 1], or 0
  uint8_t m01n[kReduced], m01j[kReduced][3], m01q[kReduced][3];   // M01 row r: its columns (ascending) and their u index
  uint8_t xslot[kBlock + 3];         // slot (0..26) of the solved rows the 27 x 27 result reads, 255 elsewhere
};

// monomials of degree <= maxdeg in the order the coefficient arrays use (degree, then s1 before s2 before s3)
inline int mono_list(int maxdeg, int out[][3]) {
  int n = 0;
  for (int d = 0; d <= maxdeg; ++d)
    for (int a = d; a >= 0; --a)
      for (int b = d - a; b >= 0; --b) { out[n][0] = a; out[n][1] = b; out[n][2] = d - a - b; ++n; }
  return n;
}

inline void build_tables(Tables* T) {
  std::memset(T, 0, sizeof(*T));
  int m3[20][3], m4[kJMono][3];
  mono_list(3, m3); mono_list(4, m4);
  auto find = [](int (*list)[3], int n, int a, int b, int c) { for (int i = 0; i < n; ++i) if (list[i][0] == a && list[i][1] == b && list[i][2] == c) return i; return -1; };
  // Cbar = (1 - s.s) I - 2 [s]x + 2 s s^T, entry (r, c), as an integer polynomial on the exponent grid [0, 2]^3
  int R[9][3][3][3];
  std::memset(R, 0, sizeof(R));
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      auto& P = R[3 * r + c];
      if (r == c) { P[0][0][0] += 1; for (int v = 0; v < 3; ++v) { int e[3] = {0, 0, 0}; e[v] = 2; P[e[0]][e[1]][e[2]] -= 1; } }
      { int e[3] = {0, 0, 0}; e[r] += 1; e[c] += 1; P[e[0]][e[1]][e[2]] += 2; }            // 2 s_r s_c
      if (r != c) {                                                                          // -2 [s]x(r, c)
        const int v = 3 - r - c;                       // the remaining axis
        const int sign = ((c - r + 3) % 3 == 1) ? -1 : 1;   // [s]x(r, c) = -s_v for (r, c) cyclic (0,1), (1,2), (2,0)
        int e[3] = {0, 0, 0}; e[v] = 1;
        P[e[0]][e[1]][e[2]] += -2 * sign;
      }
    }
  for (int a = 0; a < 9; ++a)
    for (int b = 0; b < 9; ++b) {
      int prod[5][5][5];
      std::memset(prod, 0, sizeof(prod));
      for (int i = 0; i < 27; ++i) for (int j = 0; j < 27; ++j) {
        const int x = (&R[a][0][0][0])[i], y = (&R[b][0][0][0])[j];
        if (x && y) prod[i / 9 + j / 9][(i / 3) % 3 + (j / 3) % 3][i % 3 + j % 3] += x * y;
      }
      for (int mu = 0; mu < kJMono; ++mu) T->P[9 * a + b][mu] = (int8_t)prod[m4[mu][0]][m4[mu][1]][m4[mu][2]];
    }
  for (int v = 0; v < 3; ++v)
    for (int m = 0; m < 20; ++m) {
      int e[3] = {m3[m][0], m3[m][1], m3[m][2]};
      e[v] += 1;
      T->fsrc[20 * v + m] = (uint8_t)find(m4, kJMono, e[0], e[1], e[2]);
      T->fmul[20 * v + m] = (uint8_t)e[v];
    }
  // columns of the augmented block: the reference's 93, then the 27 reduced monomials (the right-hand sides)
  int cexp[kAugCols][3];
  for (int c = 0; c < kBlock; ++c) for (int v = 0; v < 3; ++v) cexp[c][v] = dls_layout::kColMono[c][v];
  for (int j = 0; j < kReduced; ++j) { cexp[kBlock + j][0] = j / 9; cexp[kBlock + j][1] = (j / 3) % 3; cexp[kBlock + j][2] = j % 3; }
  for (int tid = 0; tid < kThreads; ++tid) {
    const int g = tid / kRowGroups, rg = tid % kRowGroups;
    for (int q = 0; q < kRowsPerLane; ++q)
      for (int i = 0; i < kLocals; ++i) {
        const int r = kRowsPerLane * rg + q, c = kColGroups * i + g;
        if (r >= kBlock) continue;
        const int e0 = cexp[c][0] - dls_layout::kRowMul[r][0], e1 = cexp[c][1] - dls_layout::kRowMul[r][1], e2 = cexp[c][2] - dls_layout::kRowMul[r][2];
        if (e0 < 0 || e1 < 0 || e2 < 0 || e0 + e1 + e2 > 3) continue;
        T->init[tid][kLocals * q + i] = (uint8_t)(1 + 20 * (dls_layout::kRowPoly[r] - 1) + find(m3, 20, e0, e1, e2));
      }
  }
  // the rows mu_j * f0: entries u[q] at the column of mu_j * {1, s1, s2, s3}
  std::memset(T->xslot, 255, sizeof(T->xslot));
  int nslot = 0;
  for (int j = 0; j < kReduced; ++j) {
    int cols[3], qs[3], n = 0;
    for (int q = 0; q < 4; ++q) {
      int e[3] = {j / 9, (j / 3) % 3, j % 3};
      if (q) e[q - 1] += 1;
      if (e[0] <= 2 && e[1] <= 2 && e[2] <= 2) { T->m00[j][9 * e[0] + 3 * e[1] + e[2]] = (uint8_t)(q + 1); continue; }
      int c = -1;
      for (int k = 0; k < kBlock; ++k) if (cexp[k][0] == e[0] && cexp[k][1] == e[1] && cexp[k][2] == e[2]) c = k;
      cols[n] = c; qs[n] = q; ++n;
    }
    for (int x = 0; x < n; ++x) for (int y = x + 1; y < n; ++y) if (cols[y] < cols[x]) { const int t = cols[x]; cols[x] = cols[y]; cols[y] = t; const int u = qs[x]; qs[x] = qs[y]; qs[y] = u; }
    T->m01n[j] = (uint8_t)n;
    for (int x = 0; x < n; ++x) {
      T->m01j[j][x] = (uint8_t)cols[x]; T->m01q[j][x] = (uint8_t)qs[x];
      if (T->xslot[cols[x]] == 255) T->xslot[cols[x]] = (uint8_t)nslot++;
    }
  }
}

// std::rand() of glibc (stdlib/random_r.c, TYPE_3: x^31 + x^3 + 1 additive feedback, seeded with 1 when srand was never
// called) -- the generator behind Eigen::Vector4d::Random() in dls_pnp.cc:134 (Eigen 3.4 MathFunctions.h
// random_default_impl<double>: x + (y - x) * double(std::rand()) / double(RAND_MAX) with x = -1, y = 1).
struct GlibcRand {
  int32_t r[34];
  int f, b;
  explicit GlibcRand(uint32_t seed = 1) {
    if (seed == 0) seed = 1;
    r[0] = (int32_t)seed;
    for (int i = 1; i < 31; ++i) {
      // 16807 * r[i-1] mod (2^31 - 1) without overflow (Schrage)
      const long hi = r[i - 1] / 127773, lo = r[i - 1] % 127773;
      long w = 16807 * lo - 2836 * hi;
      if (w < 0) w += 2147483647;
      r[i] = (int32_t)w;
    }
    f = 3; b = 0;
    for (int i = 0; i < 310; ++i) (void)next();
  }
  int32_t next() {
    const uint32_t v = (uint32_t)r[f] + (uint32_t)r[b];
    r[f] = (int32_t)v;
    const int32_t out = (int32_t)(v >> 1);
    if (++f >= 31) f = 0;
    if (++b >= 31) b = 0;
    return out;
  }
};
// the k-th coefficient of the k-th Vector4d::Random() draw, times 100 (dls_pnp.cc:134)
inline double macaulay_term_from_rand(int32_t r) { return 100.0 * (-1.0 + (2.0 * (double)r) / 2147483647.0); }

}  // namespace dls
}  // namespace thip
#endif
