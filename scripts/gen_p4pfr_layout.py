"""Derives the polynomial system and the LAYOUT of the reference's 40 x 50 P4Pfr elimination template
(sfm/pose/four_point_focal_length_radial_distortion_helper.cc) and writes it as p4pfr_layout.h (identical copies for the oracle
and the device library).  Run in the build container only (/root/reference does not travel).

The reference file holds 327 generated coefficient formulas and three index lists; what the formulas are the coefficients OF is
not written anywhere in it.  This script works the system out from the geometry and proves, symbolically (sympy), that it is
the reference's:

  unknowns   a1 a2 a3 (coordinates in the 3-dimensional null space N of the linear constraints), k (radial distortion), w (= P33)
  rows of the projection matrix in those unknowns
      p1 = N[0:4] (a1 a2 a3 1)^T,  p2 = N[4:8] (a1 a2 a3 1)^T,
      (p3x p3y p3w) = D (a1 a2 a3 k a1 k a2 k a3 k w 1)^T,  p3z = w
  ten equations
      e0 = p2 . p3        e1 = p1 . p3        e2 = p1 . p2        e3 = |p1|^2 - |p2|^2        (3-vectors: the rows of K R are
                                                                                                  orthogonal, the first two equally long)
      e4 .. e8 = five cubic forms  sum_t c_t q_i q_j p3_l  over q = (p1x p1y p1z p2x p2y p2z) -- generators the reference's
                 template adds to the four above (they vanish on every scaled rotation with rows 1, 2 equally long); their
                 terms are READ OFF the constant and the w coefficient of each block of 50 reference formulas and then
                 checked on all 50
      e9 = (1 + k d0) - (U0 . (p3x p3y p3z) + p3w)                                             (the first point's own depth is 1)
  and the template's rows are monomial multiples of them: which multiple of which equation a row is and which monomial a
  column stands for is recovered from the three index lists (an entry (row, column) <- coefficient says
  monomial(column) = multiplier(row) * monomial(coefficient)) and checked for consistency over all 703 entries.

Every one of the 327 reference formulas is compared (expanded, exact) with the coefficient this derivation gives for the same
monomial; the header holds none of them -- at run time the coefficients come from polynomial arithmetic on N, D, d0, U0
(oracle/p4pfr_oracle.h, csrc/p4pfr_device.h).  Written out: the 50 column monomials, the (equation, multiplier) pair of the 40
rows, the cubic forms, the monomial product tables the run-time arithmetic indexes with, the action-matrix rows."""
import os
import re
import sys

import sympy as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = "/root/reference/src/theia/sfm/pose/four_point_focal_length_radial_distortion_helper.cc"


def parse_reference():
    src = open(SRC).read()
    body = src[src.index("const double t2"):src.index("// way too lazy")]
    stmts = [s.strip() for s in body.replace("\n", " ").split(";") if s.strip()]
    d = sp.symbols("d0:64")
    env, coeffs = {}, {}
    conv = lambda e: re.sub(r"data\((\d+)\)", lambda m: f"d[{m.group(1)}]", e)
    for s in stmts:
        if s.startswith("Eigen::Matrix"):
            continue
        m = re.match(r"const double (t\d+) = (.*)$", s)
        if m:
            env[m.group(1)] = eval(conv(m.group(2)), {"d": d, **env})
            continue
        m = re.match(r"coeffs\((\d+), 1\) = (.*)$", s)
        assert m, s[:80]
        coeffs[int(m.group(1))] = sp.expand(eval(conv(m.group(2)), {"d": d, **env}))
    assert sorted(coeffs) == list(range(1, 328))

    def arr(name):
        mm = re.search(name + r" = \{([^}]*)\}", src)
        return [int(x) for x in mm.group(1).replace("\n", " ").split(",") if x.strip()]
    R, C, K = arr("C_ind_r"), arr("C_ind_c"), arr("coeffs_ind_r")
    assert len(R) == len(C) == len(K) == 703
    rows = {}
    for r, c, k in zip(R, C, K):
        rows.setdefault(r, []).append((c, k + 1))          # coeffs1 drops the unused entry 0: list index k is coeffs(k + 1)
    am = arr("AM_ind")
    assert arr("b_ind_c") == list(range(30, 37)) and arr("b_ind_r") == list(range(7))
    return d, coeffs, rows, am


def main():
    d, coeffs, rows, am_ref = parse_reference()
    a1, a2, a3, k, w = U = sp.symbols("a1 a2 a3 k w")
    alpha = [a1, a2, a3, 1]
    tmp = [a1, a2, a3, k * a1, k * a2, k * a3, k, w, 1]
    N = lambda r, c: d[1 + c * 8 + r]                      # data(1 .. 32) = the 8 x 4 null-space matrix, column-major
    D = lambda r, c: d[33 + 3 * c + r]                     # data(33 .. 59) = the 3 x 9 matrix D, column-major
    p1 = [sum(N(r, c) * alpha[c] for c in range(4)) for r in range(3)]
    p2 = [sum(N(4 + r, c) * alpha[c] for c in range(4)) for r in range(3)]
    P3 = [sum(D(r, c) * tmp[c] for c in range(9)) for r in range(3)]
    p3 = [P3[0], P3[1], w]
    dot = lambda x, y: sum(x[i] * y[i] for i in range(3))
    eqs = {0: dot(p2, p3), 1: dot(p1, p3), 2: dot(p1, p2), 3: dot(p1, p1) - dot(p2, p2),
           9: (1 + k * d[60]) - (d[61] * p3[0] + d[62] * p3[1] + d[63] * p3[2] + P3[2])}
    first = {0: 1, 1: 25, 2: 49, 3: 59, 9: 319}
    count = {0: 24, 1: 24, 2: 10, 3: 10, 9: 9}
    mono_of = {}

    def match(e, expr, k0, n):
        P = sp.Poly(sp.expand(expr), *U)
        assert len(P.terms()) == n, (e, len(P.terms()))
        table = [(sp.expand(c), m) for m, c in P.terms()]
        for kk in range(k0, k0 + n):
            hit = [m for cc, m in table if sp.expand(cc - coeffs[kk]) == 0]
            assert len(hit) == 1, (e, kk, hit)
            mono_of[kk] = (e, hit[0])

    for e, ex in eqs.items():
        match(e, ex, first[e], count[e])
    q = p1 + p2
    nsym = {N(r, 3): (r if r < 4 else r - 1) for r in (0, 1, 2, 4, 5, 6)}   # constant column of N -> index into q
    forms = []
    for j in range(5):
        k0 = 69 + 50 * j
        const_syms = set(nsym) | {D(r, 8) for r in range(3)}
        w_syms = set(nsym) | {D(r, 7) for r in range(3)}
        const = [kk for kk in range(k0, k0 + 50) if coeffs[kk].free_symbols <= const_syms]
        assert len(const) == 1
        form = []
        for mon, cf in sp.Poly(coeffs[const[0]], *d).terms():
            idx = [i for i, e in enumerate(mon) for _ in range(e)]
            ns = [d[i] for i in idx if d[i] in nsym]
            ds = [i for i in idx if i >= 33]
            assert len(ns) == 2 and len(ds) == 1 and (ds[0] - 33) // 3 == 8 and (ds[0] - 33) % 3 < 2
            i0, i1 = sorted(nsym[s] for s in ns)
            form.append((i0, i1, (ds[0] - 33) % 3, int(cf)))
        wc = [kk for kk in range(k0, k0 + 50) if coeffs[kk].free_symbols <= w_syms and kk != const[0]
              and any(sp.Poly(t, *d).total_degree() == 2 for t in sp.Add.make_args(coeffs[kk]))]
        assert len(wc) == 1, wc
        for t in sp.Add.make_args(coeffs[wc[0]]):
            pt = sp.Poly(t, *d)
            if pt.total_degree() == 2:
                (mon, cf), = pt.terms()
                idx = [i for i, e in enumerate(mon) for _ in range(e)]
                i0, i1 = sorted(nsym[d[i]] for i in idx)
                form.append((i0, i1, 2, int(cf)))
        form.sort(key=lambda t: (t[2], t[0], t[1]))
        forms.append(form)
        match(4 + j, sum(cf * q[i] * q[jj] * p3[l] for i, jj, l, cf in form), k0, 50)
    assert len(mono_of) == 327
    # columns from the five rows that hold all 50 monomials with multiplier 1, then every row's multiplier
    col = {}
    for r in range(26, 31):
        for c, kk in rows[r]:
            assert col.setdefault(c, mono_of[kk][1]) == mono_of[kk][1]
    assert len(col) == 50 and len(set(col.values())) == 50
    row_eq, row_mul = {}, {}
    for r, ents in rows.items():
        for c, kk in ents:
            e, m = mono_of[kk]
            mu = tuple(x - y for x, y in zip(col[c], m))
            assert min(mu) >= 0 and row_mul.setdefault(r, mu) == mu and row_eq.setdefault(r, e) == e
        assert len(ents) == count.get(row_eq[r], 50)
    cols = [col[c] for c in range(50)]
    cidx = {m: c for c, m in enumerate(cols)}
    # action matrix: multiplication by a3 on the basis (columns 37 .. 49); columns 30 .. 36 are the reduced monomials
    am = []
    for c in range(37, 50):
        t = list(cols[c]); t[2] += 1
        tc = cidx[tuple(t)]
        assert tc >= 30
        am.append(tc - 30)                                  # row of RR = [reductions of 30 .. 36 ; identity on 37 .. 49]
    assert am == am_ref, (am, am_ref)
    one, ka, wa = cidx[(0, 0, 0, 0, 0)], cidx[(0, 0, 0, 1, 0)], cidx[(0, 0, 0, 0, 1)]
    sol_rows = [cidx[(1, 0, 0, 0, 0)] - 37, cidx[(0, 1, 0, 0, 0)] - 37, ka - 37, wa - 37]   # a1 a2 k w in the eigenvector
    assert one == 37 and sol_rows == [5, 7, 1, 3]           # helper.cc: V.row(5), V.row(7), V.row(1), V.row(3)
    # run-time product tables
    amono = [(1, 0, 0), (0, 1, 0), (0, 0, 1), (0, 0, 0)]                      # (a1 a2 a3 1)
    a2mono = sorted({tuple(x + y for x, y in zip(s, t)) for s in amono for t in amono}, key=lambda m: (-sum(m), [-v for v in m]))
    assert len(a2mono) == 10
    tmono = [(1, 0, 0, 0, 0), (0, 1, 0, 0, 0), (0, 0, 1, 0, 0), (1, 0, 0, 1, 0), (0, 1, 0, 1, 0), (0, 0, 1, 1, 0), (0, 0, 0, 1, 0),
             (0, 0, 0, 0, 1), (0, 0, 0, 0, 0)]                                # tmp
    mul_aa = [[a2mono.index(tuple(x + y for x, y in zip(s, t))) for t in amono] for s in amono]
    a_t = [[cidx[tuple(s[i] + t[i] for i in range(3)) + (t[3], t[4])] for t in tmono] for s in amono]
    a2_col = [cidx[m + (0, 0)] for m in a2mono]
    a2_t = [[cidx.get(tuple(s[i] + t[i] for i in range(3)) + (t[3], t[4]), -1) for t in tmono] for s in a2mono]
    assert all(v >= 0 for r in a2_t for v in r)
    t_col = [cidx[t] for t in tmono]
    row_src = []                                            # template entry (r, c) = equation coefficient row_src[r][c] (or none)
    support = {}
    for kk, (e, m) in mono_of.items():
        support.setdefault(e, set()).add(m)
    for r in range(40):
        line = []
        for c in range(50):
            m = tuple(x - y for x, y in zip(cols[c], row_mul[r]))
            line.append(cidx[m] if min(m) >= 0 and m in support[row_eq[r]] else -1)
        assert sum(v >= 0 for v in line) == len(rows[r])
        assert sorted(c for c, v in enumerate(line) if v >= 0) == sorted(c for c, _ in rows[r])
        row_src.append(line)

    def arr2(name, rowsv, typ="signed char"):
        body = ",\n    ".join("{" + ", ".join(f"{v}" for v in r) + "}" for r in rowsv)
        return f"constexpr {typ} {name}[{len(rowsv)}][{len(rowsv[0])}] = {{\n    {body}}};\n"

    def arr1(name, v, typ="signed char"):
        return f"constexpr {typ} {name}[{len(v)}] = {{" + ", ".join(str(x) for x in v) + "};\n"

    maxt = max(len(f) for f in forms)
    fpad = [[list(t) for t in f] + [[0, 0, 0, 0]] * (maxt - len(f)) for f in forms]
    out = ("// GENERATED by scripts/gen_p4pfr_layout.py -- do not edit.  The polynomial system and the layout of the reference's 40 x 50\n"
           "// P4Pfr elimination template (sfm/pose/four_point_focal_length_radial_distortion_helper.cc), derived from the geometry and\n"
           "// checked symbolically against all 327 reference formulas and all 703 template entries; see the script.\n"
           "//   unknowns (a1 a2 a3 k w); p1 = N[0:4] (a1 a2 a3 1), p2 = N[4:8] (a1 a2 a3 1), (p3x p3y p3w) = D tmp, p3z = w,\n"
           "//   tmp = (a1 a2 a3 k a1 k a2 k a3 k w 1); equations: 0 p2.p3, 1 p1.p3, 2 p1.p2, 3 |p1|^2 - |p2|^2, 4 .. 8 the cubic forms\n"
           "//   sum c q_i q_j p3_l (q = p1x p1y p1z p2x p2y p2z; l = x y z), 9 (1 + k d0) - (U0 . p3 + p3w).\n"
           "#pragma once\nnamespace thip {\nnamespace p4pfr_layout {\n"
           "constexpr int kRows = 40, kCols = 50, kElim = 37, kBasis = 13, kReduced = 7, kFirstReduced = 30;\n")
    out += "// exponents (a1 a2 a3 k w) of the column monomials\n" + arr2("kColMono", cols)
    out += "// row r = kRowMul[r] * equation kRowEq[r]\n" + arr1("kRowEq", [row_eq[r] for r in range(40)]) + arr2("kRowMul", [row_mul[r] for r in range(40)])
    out += "// template entry (r, c) = coefficient kRowSrc[r][c] of equation kRowEq[r] (a column index: the coefficient's monomial), -1 = zero\n"
    out += arr2("kRowSrc", row_src)
    out += f"// the cubic forms: (i, j, l, c) per term, kCubicTerms[f] terms\nconstexpr int kMaxCubicTerms = {maxt};\n"
    out += arr1("kCubicTerms", [len(f) for f in forms])
    out += "constexpr signed char kCubic[5][%d][4] = {\n    " % maxt + ",\n    ".join(
        "{" + ", ".join("{" + ", ".join(str(v) for v in t) + "}" for t in f) + "}" for f in fpad) + "};\n"
    out += "// monomial products: (a1 a2 a3 1) x (a1 a2 a3 1) -> the 10 quadratic monomials; those -> columns; (a1 a2 a3 1) x tmp -> columns;\n"
    out += "// quadratic x tmp -> columns; tmp -> columns\n"
    out += arr2("kMulAA", mul_aa) + arr1("kA2Col", a2_col) + arr2("kMulATmp", a_t) + arr2("kMulA2Tmp", a2_t) + arr1("kTmpCol", t_col)
    out += "// action matrix of a3 on the basis (columns 37 .. 49): row i = row kAmRow[i] of [reductions of columns 30 .. 36 ; identity]\n"
    out += arr1("kAmRow", am)
    out += "// eigenvector rows (basis positions) of 1, a1, a2, k, w; the eigenvalue is a3\n"
    out += "constexpr int kRowOne = 0, kRowA1 = %d, kRowA2 = %d, kRowK = %d, kRowW = %d;\n" % tuple(sol_rows)
    out += "}  // namespace p4pfr_layout\n}  // namespace thip\n"
    for rel in ("oracle/p4pfr_layout.h", "pytheiasfm_amd/csrc/p4pfr_layout.h"):
        with open(os.path.join(ROOT, rel), "w") as f:
            f.write(out)
    print("p4pfr_layout.h written: 10 equations, 40 rows, 50 columns; 327 formulas and 703 entries verified")


if __name__ == "__main__":
    sys.exit(main())
