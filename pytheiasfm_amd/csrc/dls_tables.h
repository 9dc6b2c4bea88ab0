// DLS-PnP (Hesch & Roumeliotis, "A Direct Least-Squares (DLS) Method for PnP"; the reference: sfm/pose/dls_pnp.cc:67-200,
// sfm/pose/dls_impl.cc:52-754): index tables of the polynomial system, generated from the DEFINITIONS at library start-up
// (the reference ships them as 60 expanded coefficient formulas and a 1968-entry index list; nothing of either is
// transcribed here).
//
//   * rotation: Cayley-Gibbs-Rodrigues, C(s) = Cbar(s) / (1 + s.s), Cbar = (1 - s.s) I - 2 [s]x + 2 s s^T  (this is the
//     TRANSPOSE of the rotation of the quaternion (1, s): dls_pnp.cc:169-172), rbar = vec_row_major(Cbar): 9 quadratics
//   * cost J' = rbar^T D rbar (D = ls_cost_coefficients, dls_pnp.cc:111-118), f_i = dJ'/ds_i = (d rbar/ds_i)^T (D + D^T) rbar:
//     three cubics, 20 coefficients each (dls_impl.cc:62-338 lists the expanded sums)
//   * Macaulay matrix of {f0 = u0 + u1 s1 + u2 s2 + u3 s3, f1, f2, f3} in degree 7 (120 monomials): the 27 monomials
//     s1^a s2^b s3^c with a, b, c <= 2 carry the rows mu * f0, every other monomial mu the row (mu / s_i^3) * f_i for the
//     first i with exponent >= 3 (27 x 4 + 93 x 20 = 1968 non-zeros, the count of dls_impl.cc:340-754)
//   * column order used here: the 27 reduced monomials (index 9a + 3b + c, so 1, s3, s2, s1 sit at 0, 1, 3, 9 as
//     dls_pnp.cc:152-154 reads them), then the other 93 by ascending degree 3..7.  A row of degree d touches columns of
//     degree d-3..d only, so the 93 x 93 block is block upper triangular in DESCENDING degree with diagonal blocks of
//     3, 9, 18, 27, 36 rows built from the cubic coefficients alone: partial-pivot LU of it never leaves a diagonal block.
#ifndef THEIA_HIP_DLS_TABLES_H_
#define THEIA_HIP_DLS_TABLES_H_

#include <cstdint>
#include <cstring>

namespace thip {
namespace dls {

constexpr int kReduced = 27;      // monomials with all exponents <= 2
constexpr int kRows = 93;         // the other monomials of degree <= 7
constexpr int kMono = 120;
constexpr int kMaxBlock = 36;     // degree-7 block

struct Tables {
  int8_t R[9][10];           // rbar_k = sum_m R[k][m] * mono2[m]
  int8_t dR[3][9][4];        // d rbar_k / d s_i = sum_q dR[i][k][q] * {1, s1, s2, s3}[q]
  int8_t div3[20][4];        // mono3[m3] / {1, s1, s2, s3}[q] as a mono2 index, or -1
  uint8_t row_poly[kRows];   // row r (monomial kReduced + r) is a multiple of f_{1 + row_poly[r]}
  uint8_t col_of[kRows][20]; // column of the row's term nu (the mono3 order of the f coefficients)
  uint8_t mul[kReduced][4];  // column of (reduced monomial j) * {1, s1, s2, s3}
  uint8_t blk_off[6];        // first ROW (0..93) of the degree 3, 4, 5, 6, 7 block, then 93
  uint8_t mono3_deg[20];
  uint8_t exps[kMono][3];
};

// monomials of degree <= 2 and <= 3 in the order the coefficient arrays use (degree, then s1 before s2 before s3)
inline int mono_list(int maxdeg, int out[][3]) {
  int n = 0;
  for (int d = 0; d <= maxdeg; ++d)
    for (int a = d; a >= 0; --a)
      for (int b = d - a; b >= 0; --b) { out[n][0] = a; out[n][1] = b; out[n][2] = d - a - b; ++n; }
  return n;
}

inline void build_tables(Tables* T) {
  std::memset(T, 0, sizeof(*T));
  int m2[10][3], m3[20][3];
  mono_list(2, m2); mono_list(3, m3);
  auto find2 = [&](int a, int b, int c) { for (int i = 0; i < 10; ++i) if (m2[i][0] == a && m2[i][1] == b && m2[i][2] == c) return i; return -1; };
  // Cbar = (1 - s.s) I - 2 [s]x + 2 s s^T, entry (r, c), as a polynomial over mono2
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      int8_t* P = T->R[3 * r + c];
      if (r == c) { P[find2(0, 0, 0)] += 1; for (int v = 0; v < 3; ++v) { int e[3] = {0, 0, 0}; e[v] = 2; P[find2(e[0], e[1], e[2])] -= 1; } }
      { int e[3] = {0, 0, 0}; e[r] += 1; e[c] += 1; P[find2(e[0], e[1], e[2])] += 2; }            // 2 s_r s_c
      if (r != c) {                                                                               // -2 [s]x(r, c)
        const int v = 3 - r - c;                       // the remaining axis
        const int sign = ((c - r + 3) % 3 == 1) ? -1 : 1;   // [s]x(r, c) = -s_v for (r, c) cyclic (0,1), (1,2), (2,0)
        int e[3] = {0, 0, 0}; e[v] = 1;
        P[find2(e[0], e[1], e[2])] += (int8_t)(-2 * sign);
      }
    }
  // derivatives: every rbar_k is at most quadratic, so d/ds_i lives on {1, s1, s2, s3}
  for (int i = 0; i < 3; ++i)
    for (int k = 0; k < 9; ++k)
      for (int m = 0; m < 10; ++m) {
        if (!T->R[k][m] || !m2[m][i]) continue;
        int e[3] = {m2[m][0], m2[m][1], m2[m][2]};
        const int pw = e[i]; e[i] -= 1;
        const int q = e[0] ? 1 : (e[1] ? 2 : (e[2] ? 3 : 0));
        T->dR[i][k][q] += (int8_t)(pw * T->R[k][m]);
      }
  for (int m = 0; m < 20; ++m) {
    T->mono3_deg[m] = (uint8_t)(m3[m][0] + m3[m][1] + m3[m][2]);
    for (int q = 0; q < 4; ++q) {
      int e[3] = {m3[m][0], m3[m][1], m3[m][2]};
      if (q) e[q - 1] -= 1;
      T->div3[m][q] = (int8_t)((e[0] < 0 || e[1] < 0 || e[2] < 0 || e[0] + e[1] + e[2] > 2) ? -1 : find2(e[0], e[1], e[2]));
    }
  }
  // column order
  static_assert(kReduced + kRows == kMono, "monomial count");
  int idx[8][8][8];
  std::memset(idx, -1, sizeof(idx));
  int n = 0;
  for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) for (int c = 0; c < 3; ++c) {
    n = 9 * a + 3 * b + c;
    idx[a][b][c] = n; T->exps[n][0] = (uint8_t)a; T->exps[n][1] = (uint8_t)b; T->exps[n][2] = (uint8_t)c;
  }
  n = kReduced;
  for (int d = 3; d <= 7; ++d) {
    T->blk_off[d - 3] = (uint8_t)(n - kReduced);
    for (int pass = 0; pass < 2; ++pass)     // the degree-7 monomials the 27 x 27 result reads ((3,2,2) and permutations) go last
      for (int a = d; a >= 0; --a)
        for (int b = d - a; b >= 0; --b) {
          const int c = d - a - b;
          if (a <= 2 && b <= 2 && c <= 2) continue;
          const bool needed = (d == 7) && ((a == 3) + (b == 3) + (c == 3) == 1) && a <= 3 && b <= 3 && c <= 3;
          if ((pass == 1) != needed) continue;
          idx[a][b][c] = n; T->exps[n][0] = (uint8_t)a; T->exps[n][1] = (uint8_t)b; T->exps[n][2] = (uint8_t)c;
          ++n;
        }
  }
  T->blk_off[5] = (uint8_t)(n - kReduced);
  for (int r = 0; r < kRows; ++r) {
    const uint8_t* e = T->exps[kReduced + r];
    const int i = e[0] >= 3 ? 0 : (e[1] >= 3 ? 1 : 2);
    T->row_poly[r] = (uint8_t)i;
    int s[3] = {e[0], e[1], e[2]};
    s[i] -= 3;
    for (int nu = 0; nu < 20; ++nu) T->col_of[r][nu] = (uint8_t)idx[s[0] + m3[nu][0]][s[1] + m3[nu][1]][s[2] + m3[nu][2]];
  }
  for (int j = 0; j < kReduced; ++j)
    for (int q = 0; q < 4; ++q) {
      int e[3] = {T->exps[j][0], T->exps[j][1], T->exps[j][2]};
      if (q) e[q - 1] += 1;
      T->mul[j][q] = (uint8_t)idx[e[0]][e[1]][e[2]];
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
