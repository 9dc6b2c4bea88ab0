"""Derives the LAYOUT of the reference's 141 x 149 UPnP elimination template (the symmetric, 8-solution one:
sfm/pose/build_upnp_action_matrix_using_symmetry.cc) and writes it as upnp_layout.h (identical copies for the oracle and
the device library).  Run in the build container only (/root/reference does not travel).

The template's rows are monomial multiples of eight polynomial equations in the quaternion (q0, q1, q2, q3): the four
derivatives of the quartic UPnP cost (cubic + linear terms, 24 coefficients each: the 20 cubic monomials ordered by
(e3, e2, e1) ascending, then q0 .. q3) and q_i (|q|^2 - 1) = 0, after the reference has reduced the first seven of them
against their leading 7 x 7 block and cancelled column 10.  Which multiple of which equation a row is, and which monomial
a column stands for, has no closed form (the template came out of an automatic generator), so both are recovered here from
the STRUCTURE of SetUpTemplateMatrixUsingSymmetry (:83-2340): an assignment M2(r, c) = M1(i, j) says
monomial(column c) = multiplier(row r) * monomial(input column j); the relation propagates over all 2 233 assignments.
Only 141 (equation, multiplier) pairs and 149 column monomials are written -- none of the reference's statements.
The script also CHECKS, against the reference's text, the two rules the oracle and the device code are written from:
  * the support of an equation in the template: {i, 7, 8, 9, 11 .. 23} for i < 7, {10, 12, 15, 19, 23} for i = 7;
  * the input matrix of the cost derivatives (BuildActionMatrixUsingSymmetry :2349-2452): entry (i, m) is the sum over the
    monomials s_p of the UPnP rotation vector with d s_p / d q_i != 0 of (2 or 4) * A(q, p), where s_q * (d s_p / d q_i) ~ m,
    the terms ordered by q DESCENDING; the linear columns 2 b^T d s / d q_i."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = "/root/reference/src/theia/sfm/pose/build_upnp_action_matrix_using_symmetry.cc"

# UPnP rotation vector s (upnp.cc ComputeRotationVector): w^2 x^2 y^2 z^2 wx wy wz xy xz yz with (w, x, y, z) = q0 .. q3
S_MONO = [(2, 0, 0, 0), (0, 2, 0, 0), (0, 0, 2, 0), (0, 0, 0, 2), (1, 1, 0, 0), (1, 0, 1, 0), (1, 0, 0, 1), (0, 1, 1, 0), (0, 1, 0, 1), (0, 0, 1, 1)]


def input_monomials():
    cubic = sorted(((a, b, c, d) for a in range(4) for b in range(4) for c in range(4) for d in range(4) if a + b + c + d == 3),
                   key=lambda e: (e[3], e[2], e[1]))
    lin = [(1, 0, 0, 0), (0, 1, 0, 0), (0, 0, 1, 0), (0, 0, 0, 1)]
    return cubic + lin


def derivative_terms(i, m):
    """terms (coef, q, p) of input-matrix entry (i, column of cubic monomial m), ordered as the rule says"""
    out = []
    for p, sp in enumerate(S_MONO):
        if sp[i] == 0:
            continue
        d = list(sp); d[i] -= 1                      # d s_p / d q_i = sp[i] * (monomial d)
        for q, sq in enumerate(S_MONO):
            if tuple(x + y for x, y in zip(d, sq)) == m:
                out.append((2 * sp[i], q, p))
    out.sort(key=lambda t: -t[1])
    return out


def check_input_rule(text, mono):
    body = text[text.index("Matrix8d BuildActionMatrixUsingSymmetry"):]
    seen = 0
    for mm in re.finditer(r"M1\((\d), (\d+)\) = ([^;]*);", body):
        i, j, rhs = int(mm.group(1)), int(mm.group(2)), mm.group(3)
        if i >= 4:
            continue
        got = [(int(c), int(a), int(b)) for c, a, b in re.findall(r"(\d) \* M\((\d+), (\d+)\)", rhs)]
        lin = [(int(c), int(k)) for c, k in re.findall(r"(\d) \* C\[(\d+)\]", rhs)]
        if j < 20:
            assert got == derivative_terms(i, mono[j]) and not lin, (i, j, rhs, derivative_terms(i, mono[j]))
        else:
            k = j - 20                                  # coefficient of q_k in 2 b^T d s / d q_i
            want = [(2 * sp[i], p) for p, sp in enumerate(S_MONO) if sp[i] and tuple(x - (1 if t == i else 0) for t, x in enumerate(sp)) == mono[20 + k]]
            assert lin == want and not got, (i, j, rhs, want)
        seen += 1
    assert seen == 96
    fixed = {(int(a), int(b)): int(v) for a, b, v in re.findall(r"M1\(([4-7]), (\d+)\) = (-?1);", body)}
    for i in range(4):                                  # q_i (|q|^2 - 1)
        want = {}
        for k in range(4):
            e = [0, 0, 0, 0]; e[k] += 2; e[i] += 1
            want[(4 + i, mono.index(tuple(e)))] = 1
        want[(4 + i, 20 + i)] = -1
        assert {k: v for k, v in fixed.items() if k[0] == 4 + i} == want, i


def derive():
    text = open(SRC).read()
    mono = input_monomials()
    check_input_rule(text, mono)
    setup = text[text.index("void SetUpTemplateMatrixUsingSymmetry"):text.index("Matrix8d BuildActionMatrixUsingSymmetry")]
    ent = [(int(r), int(c), int(i), int(j)) for r, c, i, j in re.findall(r"M2\((\d+), (\d+)\) = M1\((\d+), (\d+)\);", setup)]
    rows = {}
    for r, c, i, j in ent:
        rows.setdefault(r, []).append((c, i, j))
    assert sorted(rows) == list(range(141))
    row_eq = {}
    for r, lst in rows.items():
        eqs = {i for _, i, _ in lst}
        assert len(eqs) == 1
        row_eq[r] = eqs.pop()
        sup = sorted(j for _, _, j in lst)
        want = [10, 12, 15, 19, 23] if row_eq[r] == 7 else sorted({row_eq[r], 7, 8, 9} | set(range(11, 24)))
        assert sup == want, (r, sup)
    # propagate monomial(col) = multiplier(row) + monomial(input column); the gauge (one common shift) is fixed afterwards
    row_mul, col_mono = {0: (0, 0, 0, 0)}, {}
    changed = True
    while changed:
        changed = False
        for r, lst in rows.items():
            if r not in row_mul:
                for c, _, j in lst:
                    if c in col_mono:
                        row_mul[r] = tuple(x - y for x, y in zip(col_mono[c], mono[j])); changed = True
                        break
            if r in row_mul:
                for c, _, j in lst:
                    e = tuple(x + y for x, y in zip(row_mul[r], mono[j]))
                    if c not in col_mono:
                        col_mono[c] = e; changed = True
                    assert col_mono[c] == e, (r, c)
    assert len(row_mul) == 141 and len(col_mono) == 149, (len(row_mul), len(col_mono))
    shift = tuple(-min(row_mul[r][k] for r in range(141)) for k in range(4))
    row_mul = {r: tuple(x + s for x, s in zip(e, shift)) for r, e in row_mul.items()}
    col_mono = {c: tuple(x + s for x, s in zip(e, shift)) for c, e in col_mono.items()}
    assert len(set(col_mono.values())) == 149 and all(min(e) >= 0 for e in col_mono.values())
    # the table is reproduced by the layout: every row places exactly the support of its equation
    back = {e: c for c, e in col_mono.items()}
    rebuilt = set()
    for r in range(141):
        sup = [10, 12, 15, 19, 23] if row_eq[r] == 7 else sorted({row_eq[r], 7, 8, 9} | set(range(11, 24)))
        for j in sup:
            rebuilt.add((r, back[tuple(x + y for x, y in zip(row_mul[r], mono[j]))], row_eq[r], j))
    assert rebuilt == set(ent) and len(ent) == len(rebuilt)
    return [row_eq[r] for r in range(141)], [row_mul[r] for r in range(141)], [col_mono[c] for c in range(149)], len(ent)


def emit(row_eq, row_mul, col_mono, nent):
    def arr(name, rowsv, width):
        body = ",\n    ".join(", ".join("{%s}" % ", ".join(str(v) for v in e) for e in rowsv[k:k + width]) for k in range(0, len(rowsv), width))
        return "constexpr signed char %s[%d][4] = {\n    %s};\n" % (name, len(rowsv), body)
    out = ["// GENERATED by scripts/gen_upnp_layout.py -- do not edit.  Layout of the reference's 141 x 149 UPnP template",
           "// (sfm/pose/build_upnp_action_matrix_using_symmetry.cc:83-2340): the equation and the monomial multiplier of every row, the",
           "// monomial of every column (exponents of q0 q1 q2 q3, one common gauge shift), recovered from the structure of the %d" % nent,
           "// assignments; see the script.  Template entry (r, c) = input(kRowEq[r], j) where kColMono[c] = kRowMul[r] + monomial(j), j in",
           "// the support of the equation: {i, 7, 8, 9, 11 .. 23} for equations 0 .. 6, {10, 12, 15, 19, 23} for equation 7.",
           "#pragma once", "namespace thip {", "namespace upnp_layout {",
           "constexpr int kRows = 141, kCols = 149, kEntries = %d;" % nent,
           "constexpr signed char kRowEq[141] = {%s};" % ", ".join(str(v) for v in row_eq),
           arr("kRowMul", row_mul, 8) + arr("kColMono", col_mono, 8) + "}  // namespace upnp_layout", "}  // namespace thip", ""]
    return "\n".join(out)


if __name__ == "__main__":
    txt = emit(*derive())
    for rel in ("oracle/upnp_layout.h", "pytheiasfm_amd/csrc/upnp_layout.h"):
        with open(os.path.join(ROOT, rel), "w") as f:
            f.write(txt)
    sys.stdout.write("wrote upnp_layout.h (%d bytes)\n" % len(txt))
