#!/usr/bin/env python3
"""Elimination template of the P4Pf solver (absolute pose + unknown focal length from four 2D-3D correspondences:
Bujnak, Kukelova, Pajdla, "A general solution to the P4P problem for camera with unknown focal length", CVPR 2008; the
reference ships a generated 78 x 88 template: sfm/pose/four_point_focal_length_helper.cc:48-931).

Nothing of the reference's table is read here: the template is derived from the geometry.

Unknowns (x, y, z, w): the depths of points b, c, d relative to point a, and w = f^2.  With X_a = (a, f), X_b = x (b, f), ...
the four equations are ratios of the rigid point configuration's inner products against |X_a - X_d|^2:

    P1 = (X_a - X_b).(X_a - X_c) - k1 |X_a - X_d|^2      k1 = (g_ab + g_ac - g_bc) / (2 g_ad)
    P2 = |X_a - X_c|^2           - k2 |X_a - X_d|^2      k2 = g_ac / g_ad
    P3 = (X_a - X_b).(X_a - X_d) - k3 |X_a - X_d|^2      k3 = (g_ab + g_ad - g_bd) / (2 g_ad)
    P4 = (X_a - X_c).(X_a - X_d) - k4 |X_a - X_d|^2      k4 = (g_ac + g_ad - g_cd) / (2 g_ad)

(g_pq = squared distances of the world points).  The system has 10 solutions; the quotient-ring basis used is
B = [1, z, y, x, w, z^2, yz, xz, wz, <tenth>] with multiplication by z as the action (the solver reads x, y, z, w off the
eigenvectors).  The script works over GF(p) on a random instance:

  1. all multiples (monomial of degree <= D) * P_k, columns = monomials split into [others | targets = z * B not in B | B];
  2. checks that every target reduces onto B (row space of the multiples);
  3. prunes rows greedily, then columns, down to a square non-singular system  [others | targets] u = -[B] ;
  4. writes csrc/p4pf_tables.h: per row (polynomial, multiplier exponents), per column the monomial exponents.

Run:  python scripts/gen_p4pf_template.py  (deterministic; a few seconds).
"""
import itertools
import os
import sys

import numpy as np

P = 32003
NV = 4  # x, y, z, w


def inv(a):
    return pow(int(a) % P, P - 2, P)


def mono_mul(a, b):
    return tuple(i + j for i, j in zip(a, b))


def monos_upto(deg):
    out = []
    for d in range(deg + 1):
        for e in itertools.product(range(d + 1), repeat=NV):
            if sum(e) == d:
                out.append(e)
    return out


X, Y, Z, W = (1, 0, 0, 0), (0, 1, 0, 0), (0, 0, 1, 0), (0, 0, 0, 1)
ONE = (0, 0, 0, 0)


def mono(*vars_):
    m = ONE
    for v in vars_:
        m = mono_mul(m, v)
    return m


# The four polynomials, each a list of (monomial, coefficient) in the CANONICAL TERM ORDER the solvers fill their
# coefficient arrays in (oracle/p4pf_oracle.h p4pf_coefficients, csrc/p4pf_device.h).  v: k1..k4 and the inner products
# of the normalised image points (aa = a.a, ab = a.b, ...).
POLYS = [
    [(mono(W, X, Y), lambda v: 1), (mono(W, Z, Z), lambda v: -v["k1"]), (mono(W, X), lambda v: -1), (mono(W, Y), lambda v: -1),
     (mono(X, Y), lambda v: v["bc"]), (mono(W, Z), lambda v: 2 * v["k1"]), (mono(Z, Z), lambda v: -v["k1"] * v["dd"]),
     (W, lambda v: 1 - v["k1"]), (X, lambda v: -v["ab"]), (Y, lambda v: -v["ac"]), (Z, lambda v: 2 * v["k1"] * v["ad"]),
     (ONE, lambda v: v["aa"] - v["k1"] * v["aa"])],
    [(mono(W, Y, Y), lambda v: 1), (mono(W, Z, Z), lambda v: -v["k2"]), (mono(W, Y), lambda v: -2), (mono(Y, Y), lambda v: v["cc"]),
     (mono(W, Z), lambda v: 2 * v["k2"]), (mono(Z, Z), lambda v: -v["k2"] * v["dd"]), (W, lambda v: 1 - v["k2"]),
     (Y, lambda v: -2 * v["ac"]), (Z, lambda v: 2 * v["k2"] * v["ad"]), (ONE, lambda v: v["aa"] - v["k2"] * v["aa"])],
    [(mono(W, X, Z), lambda v: 1), (mono(W, Z, Z), lambda v: -v["k3"]), (mono(W, X), lambda v: -1), (mono(W, Z), lambda v: 2 * v["k3"] - 1),
     (mono(X, Z), lambda v: v["bd"]), (mono(Z, Z), lambda v: -v["k3"] * v["dd"]), (W, lambda v: 1 - v["k3"]),
     (X, lambda v: -v["ab"]), (Z, lambda v: 2 * v["k3"] * v["ad"] - v["ad"]), (ONE, lambda v: v["aa"] - v["k3"] * v["aa"])],
    [(mono(W, Y, Z), lambda v: 1), (mono(W, Z, Z), lambda v: -v["k4"]), (mono(W, Y), lambda v: -1), (mono(W, Z), lambda v: 2 * v["k4"] - 1),
     (mono(Y, Z), lambda v: v["cd"]), (mono(Z, Z), lambda v: -v["k4"] * v["dd"]), (W, lambda v: 1 - v["k4"]),
     (Y, lambda v: -v["ac"]), (Z, lambda v: 2 * v["k4"] * v["ad"] - v["ad"]), (ONE, lambda v: v["aa"] - v["k4"] * v["aa"])],
]


def poly_terms():
    return POLYS


def random_instance(rng):
    r = {n: int(rng.integers(1, P)) for n in ["a1", "a2", "b1", "b2", "c1", "c2", "d1", "d2", "gab", "gac", "gad", "gbc", "gbd", "gcd"]}
    d = lambda p, q: (r[p + "1"] * r[q + "1"] + r[p + "2"] * r[q + "2"]) % P
    half, gi = inv(2), inv(r["gad"])
    v = {"k1": (r["gab"] + r["gac"] - r["gbc"]) * half * gi % P, "k2": r["gac"] * gi % P,
         "k3": (r["gab"] + r["gad"] - r["gbd"]) * half * gi % P, "k4": (r["gac"] + r["gad"] - r["gcd"]) * half * gi % P}
    for p_ in "abcd":
        for q_ in "abcd":
            if p_ <= q_:
                v[p_ + q_] = d(p_, q_)
    return v


def instantiate(polys, v):
    return [{m: fn(v) % P for m, fn in p} for p in polys]


def rref_mod(M):
    """Row echelon form over GF(P), returns (R, pivot columns)."""
    M = M.copy() % P
    rows, cols = M.shape
    piv = []
    r = 0
    for c in range(cols):
        if r == rows:
            break
        nz = np.nonzero(M[r:, c])[0]
        if nz.size == 0:
            continue
        s = r + nz[0]
        if s != r:
            M[[r, s]] = M[[s, r]]
        M[r] = M[r] * inv(M[r, c]) % P
        f = M[:, c].copy()
        f[r] = 0
        idx = np.nonzero(f)[0]
        if idx.size:
            M[idx] = (M[idx] - np.outer(f[idx], M[r])) % P
        piv.append(c)
        r += 1
    return M, piv


def build(rows, inst, cols_index):
    M = np.zeros((len(rows), len(cols_index)), dtype=np.int64)
    for i, (k, mu) in enumerate(rows):
        for m, c in inst[k].items():
            M[i, cols_index[mono_mul(m, mu)]] = c
    return M


def reduces(rows, inst, order, n_other, n_target):
    """True when every target column has a row  e_target + (B columns only)  in the row space."""
    idx = {m: i for i, m in enumerate(order)}
    R, piv = rref_mod(build(rows, inst, idx))
    pivset = {c: i for i, c in enumerate(piv)}
    for t in range(n_other, n_other + n_target):
        if t not in pivset:
            return False
        row = R[pivset[t]]
        if np.any(row[:n_other]) or np.any(row[n_other:n_other + n_target][np.arange(n_target) != t - n_other]):
            return False
    return True


def main():
    rng = np.random.default_rng(20080624)
    polys = poly_terms()
    inst = instantiate(polys, random_instance(rng))
    inst2 = instantiate(polys, random_instance(rng))

    base9 = [ONE, Z, Y, X, W, mono_mul(Z, Z), mono_mul(Y, Z), mono_mul(X, Z), mono_mul(W, Z)]
    found = None
    for D in (2, 3, 4):
        mults = monos_upto(D)
        rows = [(k, mu) for k in range(4) for mu in mults]
        allm = set()
        for k, mu in rows:
            for m in inst[k]:
                allm.add(mono_mul(m, mu))
        for tenth in sorted((m for m in monos_upto(3) if m not in base9), key=lambda m: (sum(m), m)):
            B = base9 + [tenth]
            targets = [mono_mul(Z, b) for b in B if mono_mul(Z, b) not in B]
            if any(t not in allm for t in targets):
                continue
            others = sorted((m for m in allm if m not in B and m not in targets), key=lambda m: (-sum(m), m))
            order = others + targets + B
            if reduces(rows, inst, order, len(others), len(targets)) and reduces(rows, inst2, order, len(others), len(targets)):
                found = (D, B, targets, rows)
                break
        if found:
            break
    if not found:
        sys.exit("no template up to multiplier degree 4")
    D, B, targets, rows = found
    print("multiplier degree", D, "tenth basis monomial", B[9], "targets", targets, "rows", len(rows))

    def layout(rows_):
        allm_ = set()
        for k, mu in rows_:
            for m in inst[k]:
                allm_.add(mono_mul(m, mu))
        return sorted((m for m in allm_ if m not in B and m not in targets), key=lambda m: (-sum(m), m))

    # greedy row pruning, checked on two instances; several candidate orders, the smallest template wins
    def prune(order_key):
        keep_ = list(rows)
        for cand in sorted(rows, key=order_key):
            trial = [r for r in keep_ if r != cand]
            others_ = layout(trial)
            order_ = others_ + targets + B
            if reduces(trial, inst, order_, len(others_), len(targets)) and reduces(trial, inst2, order_, len(others_), len(targets)):
                keep_ = trial
        return keep_

    orders = [lambda r: (-sum(r[1]), r[0], r[1]), lambda r: (-sum(r[1]), -r[0], tuple(-e for e in r[1])),
              lambda r: (-sum(r[1]), -r[1][3], r[0], r[1]), lambda r: (-sum(r[1]), r[1][3], r[0], r[1])]
    keep = None
    for key in orders:
        k_ = prune(key)
        print("  pruning order -> rows", len(k_), "others", len(layout(k_)))
        if keep is None or len(k_) + len(layout(k_)) < len(keep) + len(layout(keep)):
            keep = k_
    others = layout(keep)
    keep = sorted(keep, key=lambda r: (r[0], sum(r[1]), r[1]))
    order = others + targets + B
    idx = {m: i for i, m in enumerate(order)}
    n = len(others) + len(targets)
    for ins in (inst, inst2):
        _, piv = rref_mod(build(keep, ins, idx)[:, :n].T.copy())
        assert len(piv) == len(keep), "pruned rows are dependent"
    print("after pruning: rows", len(keep), "others", len(others), "targets", len(targets))

    write_header(keep, others, targets, B)


def write_header(rows, others, targets, B):
    cols = others + targets + B
    n = len(others) + len(targets)
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    paths = [os.path.join(root, "pytheiasfm_amd", "csrc", "p4pf_tables.h"), os.path.join(root, "oracle", "p4pf_tables.h")]
    for path in paths:
      with open(path, "w") as f:
        f.write("// GENERATED by scripts/gen_p4pf_template.py -- do not edit.  Elimination template of the P4Pf solver: row r is the\n"
                "// polynomial kRowPoly[r] times the monomial x^e0 y^e1 z^e2 w^e3 of kRowMul[r]; column c is the monomial kColMono[c].\n"
                "// Columns: kOthers monomials to eliminate, kTargets = z * basis[5 + i], then the kBasis quotient-ring basis\n"
                "// [1, z, y, x, w, z^2, yz, xz, wz, wy].  kRows < kOthers + kTargets: the rows reduce the targets onto the basis\n"
                "// without determining every other monomial, so the solver works on the TRANSPOSED system (p4pf in ransac_device.h).\n"
                "#ifndef THEIA_HIP_P4PF_TABLES_H_\n#define THEIA_HIP_P4PF_TABLES_H_\n\n#include <cstdint>\n\nnamespace thip {\nnamespace p4pf {\n\n")
        f.write("constexpr int kRows = %d;\nconstexpr int kOthers = %d;\nconstexpr int kTargets = %d;\nconstexpr int kBasis = %d;\n"
                "constexpr int kElim = kOthers + kTargets;\nconstexpr int kCols = kElim + kBasis;\n\n" % (len(rows), len(others), len(targets), len(B)))
        idx = {m: i for i, m in enumerate(cols)}
        f.write("constexpr int kMaxTerms = 12;\n")
        f.write("// initialiser lists as macros: the device side instantiates the same tables in __constant__ memory\n")
        f.write("#define THIP_P4PF_POLY_TERMS {%s}\n" % ", ".join(str(len(p)) for p in POLYS))
        f.write("#define THIP_P4PF_ROW_COL {%s}\n"
                % ", ".join("{" + ", ".join(str(idx[mono_mul(m, mu)]) for m, _ in POLYS[k]) + "}" for k, mu in rows))
        f.write("#define THIP_P4PF_ROW_POLY {%s}\n" % ", ".join(str(k) for k, _ in rows))
        f.write("constexpr uint8_t kPolyTerms[4] = THIP_P4PF_POLY_TERMS;\n")
        f.write("constexpr uint8_t kRowCol[kRows][kMaxTerms] = THIP_P4PF_ROW_COL;   // column of term t of row r (canonical term order of the generator's POLYS)\n")
        f.write("constexpr uint8_t kRowPoly[kRows] = THIP_P4PF_ROW_POLY;\n")
        f.write("constexpr uint8_t kRowMul[kRows][4] = {%s};\n" % ", ".join("{%d, %d, %d, %d}" % mu for _, mu in rows))
        f.write("constexpr uint8_t kColMono[kCols][4] = {%s};\n" % ", ".join("{%d, %d, %d, %d}" % m for m in cols))
        f.write("\n}  // namespace p4pf\n}  // namespace thip\n\n#endif\n")
    print("wrote", os.path.normpath(path))


if __name__ == "__main__":
    main()
