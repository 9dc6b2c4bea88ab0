// TEST INFRASTRUCTURE (CPU oracle) -- DLS-PnP, restated from the reference's formulation (sfm/pose/dls_pnp.cc:67-200):
// cost matrix -> Jacobian cubics -> dense 120 x 120 Macaulay matrix -> Schur complement through a dense partial-pivot
// LU of the 93 x 93 block (dls_pnp.cc:143-146) -> eigenvectors of the 27 x 27 multiplication matrix -> poses.
// The polynomial system is derived here by generic polynomial arithmetic on exponent grids (the cost quartic
// J' = rbar^T D rbar is expanded and differentiated) and every step is sequential.  The ROWS and COLUMNS of the 93 x 93
// block are the reference's (dls_layout.h, recovered from the structure of dls_impl.cc:340-754 by
// scripts/gen_dls_layout.py): the order decides the pivot sequence and every rounding of the elimination, so the LU
// below walks the same pivots as the reference's partialPivLu() (first maximum of the column, rows swapped) with a
// plain right-looking elimination (fused multiply-adds, as a -march=native build of the reference contracts them),
// a column-oriented back-substitution and M00 - M01 X accumulated in ascending column order.  Since round 4 the device
// follows exactly this route (csrc/dls_device.h), so DLS / gDLS hypotheses are bit-identical, not merely close.
// The reference's expanded coefficient formulas (dls_impl.cc:62-338) and index table (dls_impl.cc:340-754) are pinned
// through tests/golden/dls_reference_vectors.json (made by tests/golden/make_dls_golden.py from the reference text).
// Included by ransac_oracle.cpp inside its anonymous namespace (needs eig_general_t, quat_to_rot).
#include "dls_layout.h"

struct DlsPoly {   // polynomial in (s1, s2, s3), exponents 0..4 each
  double c[5][5][5];
  DlsPoly() { std::memset(c, 0, sizeof(c)); }
};
inline DlsPoly dls_mul(const DlsPoly& a, const DlsPoly& b) {
  DlsPoly o;
  for (int i = 0; i < 5; ++i) for (int j = 0; j < 5; ++j) for (int k = 0; k < 5; ++k) {
    if (a.c[i][j][k] == 0.0) continue;
    for (int p = 0; i + p < 5; ++p) for (int q = 0; j + q < 5; ++q) for (int r = 0; k + r < 5; ++r)
      if (b.c[p][q][r] != 0.0) o.c[i + p][j + q][k + r] += a.c[i][j][k] * b.c[p][q][r];
  }
  return o;
}
inline DlsPoly dls_diff(const DlsPoly& a, int var) {
  DlsPoly o;
  for (int i = 0; i < 5; ++i) for (int j = 0; j < 5; ++j) for (int k = 0; k < 5; ++k) {
    const int e[3] = {i, j, k};
    if (e[var] == 0) continue;
    int f[3] = {i, j, k}; f[var] -= 1;
    o.c[f[0]][f[1]][f[2]] += e[var] * a.c[i][j][k];
  }
  return o;
}

// rbar = vec (row-major) of (1 - s.s) I - 2 [s]x + 2 s s^T: the un-normalised TRANSPOSE of the rotation of the quaternion
// (1, s1, s2, s3), which is what "translation_factor * rot_vec" multiplies (dls_pnp.cc:169-172; rot_mat.data() is column-major)
inline void dls_rbar(DlsPoly r[9]) {
  auto mono = [](int a, int b, int c, double v) { DlsPoly p; p.c[a][b][c] = v; return p; };
  auto add = [](DlsPoly& a, const DlsPoly& b) { for (int i = 0; i < 125; ++i) (&a.c[0][0][0])[i] += (&b.c[0][0][0])[i]; };
  const DlsPoly s[3] = {mono(1, 0, 0, 1.0), mono(0, 1, 0, 1.0), mono(0, 0, 1, 1.0)};
  DlsPoly ss;   // s.s
  for (int v = 0; v < 3; ++v) add(ss, dls_mul(s[v], s[v]));
  // [s]x
  const double skew_sign[3][3] = {{0, -1, 1}, {1, 0, -1}, {-1, 1, 0}};
  const int skew_axis[3][3] = {{0, 2, 1}, {2, 0, 0}, {1, 0, 0}};
  for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) {
    DlsPoly p;
    if (a == b) { p.c[0][0][0] = 1.0; for (int i = 0; i < 125; ++i) (&p.c[0][0][0])[i] -= (&ss.c[0][0][0])[i]; }
    else { DlsPoly k = s[skew_axis[a][b]]; for (int i = 0; i < 125; ++i) (&p.c[0][0][0])[i] += -2.0 * skew_sign[a][b] * (&k.c[0][0][0])[i]; }
    DlsPoly outer = dls_mul(s[a], s[b]);
    for (int i = 0; i < 125; ++i) (&p.c[0][0][0])[i] += 2.0 * (&outer.c[0][0][0])[i];
    r[3 * a + b] = p;
  }
}

struct DlsMonomials {   // 120 monomials of degree <= 7: 27 reduced first (index 9a + 3b + c), then the reference's column order
  int e[120][3];
  int index[8][8][8];
  int row_poly[120], row_mul[120][3];   // row r = f_{row_poly[r]} * monomial row_mul[r] (f_0 = the random linear form)
  DlsMonomials() {
    std::memset(index, -1, sizeof(index));
    int n = 0;
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) for (int c = 0; c < 3; ++c) { e[n][0] = a; e[n][1] = b; e[n][2] = c; index[a][b][c] = n++; }
    for (int k = 0; k < thip::dls_layout::kBlock; ++k) {
      for (int v = 0; v < 3; ++v) e[n][v] = thip::dls_layout::kColMono[k][v];
      index[e[n][0]][e[n][1]][e[n][2]] = n; n++;
    }
    for (int r = 0; r < 120; ++r) {
      row_poly[r] = r < 27 ? 0 : thip::dls_layout::kRowPoly[r - 27];
      for (int v = 0; v < 3; ++v) row_mul[r][v] = r < 27 ? e[r][v] : thip::dls_layout::kRowMul[r - 27][v];
    }
  }
};

// glibc rand() (TYPE_3 additive feedback generator, default seed 1): Eigen's Vector4d::Random() of dls_pnp.cc:134 draws from it
struct DlsLibcRand {
  uint32_t st[31];
  int fp, rp;
  void seed(uint32_t s) {
    if (!s) s = 1;
    int32_t w = (int32_t)s;
    st[0] = (uint32_t)w;
    for (int i = 1; i < 31; ++i) {
      const int64_t t = (16807LL * (int64_t)w) % 2147483647LL;   // same value as Schrage's split form
      w = (int32_t)(t < 0 ? t + 2147483647LL : t);
      st[i] = (uint32_t)w;
    }
    fp = 3; rp = 0;
    for (int i = 0; i < 310; ++i) (void)draw();
  }
  int32_t draw() {
    st[fp] += st[rp];
    const int32_t out = (int32_t)((st[fp] >> 1) & 0x7fffffffu);
    fp = (fp + 1) % 31; rp = (rp + 1) % 31;
    return out;
  }
};
inline void dls_macaulay_terms(int call_index, double u[4]) {   // the 4 draws of the call_index-th DlsPnp call of a process
  DlsLibcRand g; g.seed(1);
  for (int i = 0; i < 4 * call_index; ++i) (void)g.draw();
  for (int k = 0; k < 4; ++k) u[k] = 100.0 * (-1.0 + (2.0 * (double)g.draw()) / 2147483647.0);
}

// 27 x 27 multiplication matrix of f0 = u0 + u1 s1 + u2 s2 + u3 s3 from the 9 x 9 cost matrix D (dls_pnp.cc:120-146)
inline bool dls_action_from_cost(const double* D, const double* u, double* fcoef_out, double* A, double* macaulay_out = nullptr) {
  // J' = rbar^T D rbar as a quartic, f_i = dJ'/ds_i
  DlsPoly rb[9];
  dls_rbar(rb);
  DlsPoly J;
  for (int a = 0; a < 9; ++a) for (int b = 0; b < 9; ++b) {
    const DlsPoly p = dls_mul(rb[a], rb[b]);
    for (int i = 0; i < 125; ++i) (&J.c[0][0][0])[i] += D[9 * a + b] * (&p.c[0][0][0])[i];
  }
  DlsPoly f[4];
  f[0].c[0][0][0] = u[0]; f[0].c[1][0][0] = u[1]; f[0].c[0][1][0] = u[2]; f[0].c[0][0][1] = u[3];
  for (int v = 0; v < 3; ++v) f[1 + v] = dls_diff(J, v);
  if (fcoef_out) for (int v = 0; v < 3; ++v) for (int i = 0; i < 125; ++i) fcoef_out[125 * v + i] = (&f[1 + v].c[0][0][0])[i];
  // Macaulay matrix in the reference's layout
  static const DlsMonomials mon;
  std::vector<double> M(120 * 120, 0.0);
  for (int row = 0; row < 120; ++row) {
    const int which = mon.row_poly[row];
    const int* sh = mon.row_mul[row];
    for (int a = 0; a < 4; ++a) for (int b = 0; a + b < 4; ++b) for (int c = 0; a + b + c < 4; ++c) {
      const double v = f[which].c[a][b][c];
      if (v != 0.0) M[(size_t)row * 120 + mon.index[sh[0] + a][sh[1] + b][sh[2] + c]] = v;
    }
  }
  if (macaulay_out) for (int i = 0; i < 14400; ++i) macaulay_out[i] = M[i];
  // Schur complement: M00 - M01 * lu(M11).solve(M10), dense partial-pivot LU of the 93 x 93 block with the 27 right-hand sides
  const int nb = 93, nr = 27, w = nb + nr;
  std::vector<double> Aug((size_t)nb * w);
  for (int r = 0; r < nb; ++r) {
    for (int c = 0; c < nb; ++c) Aug[(size_t)r * w + c] = M[(size_t)(27 + r) * 120 + 27 + c];
    for (int c = 0; c < nr; ++c) Aug[(size_t)r * w + nb + c] = M[(size_t)(27 + r) * 120 + c];
  }
  for (int k = 0; k < nb; ++k) {
    int p = k; double best = std::fabs(Aug[(size_t)k * w + k]);
    for (int r = k + 1; r < nb; ++r) if (std::fabs(Aug[(size_t)r * w + k]) > best) { best = std::fabs(Aug[(size_t)r * w + k]); p = r; }
    if (best == 0.0) return false;
    if (p != k) for (int c = 0; c < w; ++c) std::swap(Aug[(size_t)k * w + c], Aug[(size_t)p * w + c]);
    for (int r = k + 1; r < nb; ++r) {
      const double l = Aug[(size_t)r * w + k] / Aug[(size_t)k * w + k];
      if (l == 0.0) continue;
      for (int c = k + 1; c < w; ++c) Aug[(size_t)r * w + c] = __builtin_fma(-l, Aug[(size_t)k * w + c], Aug[(size_t)r * w + c]);
    }
  }
  // column-oriented back-substitution: x_k = rhs_k * (1 / u_kk), then retired from the rows above
  for (int k = nb - 1; k >= 0; --k) {
    const double inv_ukk = 1.0 / Aug[(size_t)k * w + k];   // Eigen's triangular solve with a matrix right-hand side multiplies by the inverted diagonal
    for (int c = 0; c < nr; ++c) Aug[(size_t)k * w + nb + c] = Aug[(size_t)k * w + nb + c] * inv_ukk;
    for (int i = 0; i < k; ++i) {
      const double u_ik = Aug[(size_t)i * w + k];
      if (u_ik == 0.0) continue;
      for (int c = 0; c < nr; ++c) Aug[(size_t)i * w + nb + c] = __builtin_fma(-u_ik, Aug[(size_t)k * w + nb + c], Aug[(size_t)i * w + nb + c]);
    }
  }
  for (int r = 0; r < 27; ++r) for (int c = 0; c < 27; ++c) {
    double s = M[(size_t)r * 120 + c];
    for (int j = 0; j < nb; ++j) { const double m01 = M[(size_t)r * 120 + 27 + j]; if (m01 != 0.0) s = __builtin_fma(-m01, Aug[(size_t)j * w + nb + c], s); }
    A[27 * r + c] = s;
  }
  return true;
}

// returns the number of solutions (<= 27); quats [w x y z], ts.  Optional outputs for the tests: fcoef (3 x 125 exponent
// grids of the Jacobian cubics), action (the 27 x 27 matrix).
inline int dls_pnp(int n, const double* feat, const double* world, const double u[4], double* quats, double* ts,
                   double* fcoef_out = nullptr, double* action_out = nullptr) {
  if (n < 3) return 0;
  std::vector<double> nn((size_t)n * 9);
  double Hinv[9] = {(double)n, 0, 0, 0, (double)n, 0, 0, 0, (double)n};
  for (int i = 0; i < n; ++i) {
    const double v[3] = {feat[2 * i], feat[2 * i + 1], 1.0};
    const double nrm = std::sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
    const double b[3] = {v[0] / nrm, v[1] / nrm, v[2] / nrm};
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { nn[(size_t)i * 9 + 3 * r + c] = b[r] * b[c]; Hinv[3 * r + c] -= b[r] * b[c]; }
  }
  // 3 x 3 inverse by cofactors (Eigen's fixed-size inverse)
  double Hm[9];
  {
    const double* a = Hinv;
    const double c00 = a[4] * a[8] - a[5] * a[7], c01 = a[5] * a[6] - a[3] * a[8], c02 = a[3] * a[7] - a[4] * a[6];
    const double det = (a[0] * c00 + a[1] * c01) + a[2] * c02;
    const double id = 1.0 / det;
    Hm[0] = c00 * id; Hm[1] = (a[2] * a[7] - a[1] * a[8]) * id; Hm[2] = (a[1] * a[5] - a[2] * a[4]) * id;
    Hm[3] = c01 * id; Hm[4] = (a[0] * a[8] - a[2] * a[6]) * id; Hm[5] = (a[2] * a[3] - a[0] * a[5]) * id;
    Hm[6] = c02 * id; Hm[7] = (a[1] * a[6] - a[0] * a[7]) * id; Hm[8] = (a[0] * a[4] - a[1] * a[3]) * id;
  }
  auto left_mult = [](const double* X, double* L) {   // R X = L vec(R), dls_impl.cc:52-58
    for (int i = 0; i < 27; ++i) L[i] = 0.0;
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) L[9 * r + 3 * r + c] = X[c];
  };
  double Tf[27];
  for (int i = 0; i < 27; ++i) Tf[i] = 0.0;
  for (int i = 0; i < n; ++i) {
    double L[27]; left_mult(world + 3 * i, L);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 9; ++c) {
      double s = 0.0;
      for (int k = 0; k < 3; ++k) s += (nn[(size_t)i * 9 + 3 * r + k] - (r == k ? 1.0 : 0.0)) * L[9 * k + c];
      Tf[9 * r + c] += s;
    }
  }
  {
    double t2[27];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 9; ++c) { double s = 0.0; for (int k = 0; k < 3; ++k) s += Hm[3 * r + k] * Tf[9 * k + c]; t2[9 * r + c] = s; }
    for (int i = 0; i < 27; ++i) Tf[i] = t2[i];
  }
  double D[81];
  for (int i = 0; i < 81; ++i) D[i] = 0.0;
  for (int i = 0; i < n; ++i) {
    double W[27]; left_mult(world + 3 * i, W);
    for (int k = 0; k < 27; ++k) W[k] += Tf[k];
    double PW[27];   // (I - n n^T) W
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 9; ++c) {
      double s = 0.0;
      for (int k = 0; k < 3; ++k) s += ((r == k ? 1.0 : 0.0) - nn[(size_t)i * 9 + 3 * r + k]) * W[9 * k + c];
      PW[9 * r + c] = s;
    }
    for (int a = 0; a < 9; ++a) for (int b = 0; b < 9; ++b) { double s = 0.0; for (int k = 0; k < 3; ++k) s += W[9 * k + a] * PW[9 * k + b]; D[9 * a + b] += s; }
  }
  double A[729], V[729], wr[27], wi[27];
  if (!dls_action_from_cost(D, u, fcoef_out, A)) return 0;
  if (action_out) for (int i = 0; i < 729; ++i) action_out[i] = A[i];
  if (!eig_general_t<27, true>(27, A, wr, wi, V)) return 0;
  int ns = 0;
  for (int i = 0; i < 27; ++i) {
    // eigenvector i as a complex vector; s1, s2, s3 = entries 9, 3, 1 over entry 0 (dls_pnp.cc:152-154)
    const int re_col = wi[i] < 0 ? i - 1 : i;
    const double sg = wi[i] < 0 ? -1.0 : 1.0;
    auto comp = [&](int row, double* re, double* im) { *re = V[27 * row + re_col]; *im = wi[i] == 0 ? 0.0 : sg * V[27 * row + re_col + 1]; };
    double d_re, d_im; comp(0, &d_re, &d_im);
    if (d_re == 0.0 && d_im == 0.0) continue;
    double sr[3], si[3];
    const int rows[3] = {9, 3, 1};
    for (int k = 0; k < 3; ++k) { double a, b; comp(rows[k], &a, &b); eig_cdiv(a, b, d_re, d_im, &sr[k], &si[k]); }
    const double kEps = 1e-6;
    if (!(std::fabs(si[0]) < kEps && std::fabs(si[1]) < kEps && std::fabs(si[2]) < kEps)) continue;
    // Quaterniond(1, s).inverse().normalized()
    double q[4] = {1.0, sr[0], sr[1], sr[2]};
    const double n2 = ((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3];
    double qi[4] = {q[0] / n2, -q[1] / n2, -q[2] / n2, -q[3] / n2};
    const double nq = std::sqrt(((qi[0] * qi[0] + qi[1] * qi[1]) + qi[2] * qi[2]) + qi[3] * qi[3]);
    double qs[4] = {qi[0] / nq, qi[1] / nq, qi[2] / nq, qi[3] / nq};   // soln_rotation
    // rot_mat = soln_rotation.inverse().toRotationMatrix(); translation = translation_factor * vec_colmajor(rot_mat)
    const double m2 = ((qs[0] * qs[0] + qs[1] * qs[1]) + qs[2] * qs[2]) + qs[3] * qs[3];
    const double qv[4] = {qs[0] / m2, -qs[1] / m2, -qs[2] / m2, -qs[3] / m2};
    double Rm[9]; quat_to_rot(qv, Rm);
    double t[3];
    for (int r = 0; r < 3; ++r) { double s = 0.0; for (int c = 0; c < 3; ++c) for (int k = 0; k < 3; ++k) s += Tf[9 * r + 3 * c + k] * Rm[3 * k + c]; t[r] = s; }
    // every point in front of the camera (soln_rotation * X + t).z >= 0
    double Rs[9]; quat_to_rot(qs, Rs);
    bool front = true;
    for (int j = 0; j < n && front; ++j) {
      const double* X = world + 3 * j;
      const double z = ((Rs[6] * X[0] + Rs[7] * X[1]) + Rs[8] * X[2]) + t[2];
      if (z < 0) front = false;
    }
    if (!front) continue;
    for (int k = 0; k < 4; ++k) quats[4 * ns + k] = qs[k];
    for (int k = 0; k < 3; ++k) ts[3 * ns + k] = t[k];
    ns++;
  }
  return ns;
}


// ------------------------------------------------------------------------------------------------ gDLS
// GdlsSimilarityTransform (sfm/transformation/gdls_similarity_transform.cc:67-228): the generalised-camera form of the same
// problem -- rays c_i + alpha_i x_i from camera positions c_i, unknown rotation, translation AND scale with
// s c_i + alpha_i x_i = R X_i + t.  Scale and translation are linear in vec(R) (:80-117: the 4 x 4 matrix H and the 4 x 9
// helper), the cost matrix is sum W^T (I - x x^T) W with W = L(X) - c scale_factor + translation_factor (:119-133), and from
// there on it is DLS: Jacobian cubics, Macaulay matrix with its own Vector4d::Random() equation, eigenvectors (:135-175).
inline bool gdls_inverse4(const double* a, double* inv) {   // Eigen's Matrix4d::inverse(): adjugate over determinant
  auto m3 = [&](int r0, int r1, int r2, int c0, int c1, int c2) {
    return a[4 * r0 + c0] * (a[4 * r1 + c1] * a[4 * r2 + c2] - a[4 * r1 + c2] * a[4 * r2 + c1]) -
           a[4 * r0 + c1] * (a[4 * r1 + c0] * a[4 * r2 + c2] - a[4 * r1 + c2] * a[4 * r2 + c0]) +
           a[4 * r0 + c2] * (a[4 * r1 + c0] * a[4 * r2 + c1] - a[4 * r1 + c1] * a[4 * r2 + c0]);
  };
  double cof[16];
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) {
      int rr[3], cc[3], k = 0, l = 0;
      for (int i = 0; i < 4; ++i) { if (i != r) rr[k++] = i; if (i != c) cc[l++] = i; }
      const double minor = m3(rr[0], rr[1], rr[2], cc[0], cc[1], cc[2]);
      cof[4 * r + c] = ((r + c) & 1) ? -minor : minor;
    }
  const double det = ((a[0] * cof[0] + a[1] * cof[1]) + a[2] * cof[2]) + a[3] * cof[3];
  if (det == 0.0) return false;
  for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) inv[4 * r + c] = cof[4 * c + r] / det;
  return true;
}

// origin / dir (unit) / world: n x 3 each.  Returns the number of solutions: quats [w x y z] (soln_rotation), ts, scales.
inline int gdls_similarity(int n, const double* origin, const double* dir, const double* world, const double u[4],
                           double* quats, double* ts, double* scales) {
  if (n < 4) return 0;
  auto left_mult = [](const double* X, double* L) {
    for (int i = 0; i < 27; ++i) L[i] = 0.0;
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) L[9 * r + 3 * r + c] = X[c];
  };
  double Hinv[16], sv[36];
  for (int i = 0; i < 16; ++i) Hinv[i] = 0.0;
  for (int i = 0; i < 36; ++i) sv[i] = 0.0;
  for (int i = 0; i < n; ++i) {
    const double* c = origin + 3 * i; const double* x = dir + 3 * i;
    const double cd = (c[0] * x[0] + c[1] * x[1]) + c[2] * x[2];
    Hinv[0] += ((c[0] * c[0] + c[1] * c[1]) + c[2] * c[2]) - cd * cd;
    for (int r = 0; r < 3; ++r) {
      const double t = -c[r] + cd * x[r];
      Hinv[4 * (r + 1)] += t; Hinv[r + 1] += t;
      for (int k = 0; k < 3; ++k) Hinv[4 * (r + 1) + k + 1] += (r == k ? 1.0 : 0.0) - x[r] * x[k];
    }
    double L[27]; left_mult(world + 3 * i, L);
    for (int cc = 0; cc < 9; ++cc) {
      double s0 = 0.0;
      for (int k = 0; k < 3; ++k) s0 += (c[k] - cd * x[k]) * L[9 * k + cc];
      sv[cc] += s0;
      for (int r = 0; r < 3; ++r) {
        double s1 = 0.0;
        for (int k = 0; k < 3; ++k) s1 += (x[r] * x[k] - (r == k ? 1.0 : 0.0)) * L[9 * k + cc];
        sv[9 * (r + 1) + cc] += s1;
      }
    }
  }
  double Hm[16];
  if (!gdls_inverse4(Hinv, Hm)) return 0;
  double sf[9], Tf[27];
  for (int cc = 0; cc < 9; ++cc)
    for (int r = 0; r < 4; ++r) {
      double s2 = 0.0;
      for (int k = 0; k < 4; ++k) s2 += Hm[4 * r + k] * sv[9 * k + cc];
      if (r == 0) sf[cc] = s2; else Tf[9 * (r - 1) + cc] = s2;
    }
  double D[81];
  for (int i = 0; i < 81; ++i) D[i] = 0.0;
  for (int i = 0; i < n; ++i) {
    const double* c = origin + 3 * i; const double* x = dir + 3 * i;
    double W[27]; left_mult(world + 3 * i, W);
    for (int r = 0; r < 3; ++r) for (int cc = 0; cc < 9; ++cc) W[9 * r + cc] += Tf[9 * r + cc] - c[r] * sf[cc];
    double PW[27];   // (I - x x^T) W
    for (int r = 0; r < 3; ++r) for (int cc = 0; cc < 9; ++cc) {
      double s2 = 0.0;
      for (int k = 0; k < 3; ++k) s2 += ((r == k ? 1.0 : 0.0) - x[r] * x[k]) * W[9 * k + cc];
      PW[9 * r + cc] = s2;
    }
    for (int a = 0; a < 9; ++a) for (int b = 0; b < 9; ++b) { double s2 = 0.0; for (int k = 0; k < 3; ++k) s2 += W[9 * k + a] * PW[9 * k + b]; D[9 * a + b] += s2; }
  }
  double A[729], V[729], wr[27], wi[27];
  if (!dls_action_from_cost(D, u, nullptr, A)) return 0;
  if (!eig_general_t<27, true>(27, A, wr, wi, V)) return 0;
  int ns = 0;
  for (int i = 0; i < 27; ++i) {
    const int re_col = wi[i] < 0 ? i - 1 : i;
    const double sg = wi[i] < 0 ? -1.0 : 1.0;
    auto comp = [&](int row, double* re, double* im) { *re = V[27 * row + re_col]; *im = wi[i] == 0 ? 0.0 : sg * V[27 * row + re_col + 1]; };
    double d_re, d_im; comp(0, &d_re, &d_im);
    if (d_re == 0.0 && d_im == 0.0) continue;
    double sr[3], si[3];
    const int rows[3] = {9, 3, 1};
    for (int k = 0; k < 3; ++k) { double a, b; comp(rows[k], &a, &b); eig_cdiv(a, b, d_re, d_im, &sr[k], &si[k]); }
    const double kEps = 1e-6;
    if (!(std::fabs(si[0]) < kEps && std::fabs(si[1]) < kEps && std::fabs(si[2]) < kEps)) continue;
    double q[4] = {1.0, sr[0], sr[1], sr[2]};
    const double n2 = ((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3];
    double qi[4] = {q[0] / n2, -q[1] / n2, -q[2] / n2, -q[3] / n2};
    const double nq = std::sqrt(((qi[0] * qi[0] + qi[1] * qi[1]) + qi[2] * qi[2]) + qi[3] * qi[3]);
    double qs[4] = {qi[0] / nq, qi[1] / nq, qi[2] / nq, qi[3] / nq};   // soln_rotation
    const double m2 = ((qs[0] * qs[0] + qs[1] * qs[1]) + qs[2] * qs[2]) + qs[3] * qs[3];
    const double qv[4] = {qs[0] / m2, -qs[1] / m2, -qs[2] / m2, -qs[3] / m2};
    double Rm[9]; quat_to_rot(qv, Rm);   // rot_mat = soln_rotation.inverse().toRotationMatrix(); rot_vec = its column-major data
    double t[3], sc = 0.0;
    for (int r = 0; r < 3; ++r) { double s2 = 0.0; for (int c = 0; c < 3; ++c) for (int k = 0; k < 3; ++k) s2 += Tf[9 * r + 3 * c + k] * Rm[3 * k + c]; t[r] = s2; }
    for (int c = 0; c < 3; ++c) for (int k = 0; k < 3; ++k) sc += sf[3 * c + k] * Rm[3 * k + c];
    // every point in front of its ray: (FromTwoVectors(x, e_z) * (R X + t - s c)).z = x . (R X + t - s c) >= 0  (:199-218)
    double Rs[9]; quat_to_rot(qs, Rs);
    bool front = true;
    for (int j = 0; j < n && front; ++j) {
      const double* X = world + 3 * j; const double* c = origin + 3 * j; const double* x = dir + 3 * j;
      double p[3];
      for (int r = 0; r < 3; ++r) p[r] = (((Rs[3 * r] * X[0] + Rs[3 * r + 1] * X[1]) + Rs[3 * r + 2] * X[2]) + t[r]) - sc * c[r];
      if ((x[0] * p[0] + x[1] * p[1]) + x[2] * p[2] < 0) front = false;
    }
    if (!front) continue;
    for (int k = 0; k < 4; ++k) quats[4 * ns + k] = qs[k];
    for (int k = 0; k < 3; ++k) ts[3 * ns + k] = t[k];
    scales[ns] = sc;
    ns++;
  }
  return ns;
}
