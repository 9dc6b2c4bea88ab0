"""Generates tests/golden/dls_reference_vectors.json from the reference's DLS-PnP tables, in THIS container only
(/root/reference does not travel).  The reference cannot be compiled here (Eigen / glog are not installed), so its
polynomial system is evaluated from its text: the 60 expanded Jacobian-coefficient sums and the 1968-entry
(index, value) list of CreateMacaulayMatrix (sfm/pose/dls_impl.cc:62-754) are parsed and run through numpy on random
inputs.  What is stored is data only:
  * per case: a random 9 x 9 cost matrix D, the terms u, the values of the three Jacobian cubics as (exponents, value)
    lists, the 27 eigenvalues of the Schur complement of the reference's Macaulay matrix (dls_pnp.cc:143-146) --
    invariant under the row / column order of the matrix, so any correct construction must reproduce them -- and, for
    the first two cases, the non-zero entries (row, column, value) of that matrix: the layout partialPivLu() walks.
Also writes tests/golden/libc_rand.json: the first values of the real glibc rand() (tests/golden/make_rand_golden.c)."""
import json
import os
import re
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/src/theia/sfm/pose/dls_impl.cc"


def parse_monomial(txt):
    txt = txt.strip()
    e = [0, 0, 0]
    if txt.startswith("constant"):
        return tuple(e)
    for tok in txt.split("*"):
        tok = tok.strip()
        m = re.fullmatch(r"s([123])(?:\^(\d))?", tok)
        assert m, txt
        e[int(m.group(1)) - 1] += int(m.group(2) or 1)
    return tuple(e)


def main():
    text = open(SRC).read()
    jac = text[text.index("void ExtractJacobianCoefficients"):text.index("MatrixXd CreateMacaulayMatrix")]
    mac = text[text.index("MatrixXd CreateMacaulayMatrix"):]
    polys = {}
    for m in re.finditer(r"f([123])_coeff\[(\d+)\]\s*=(.*?);\s*//([^\n]*)", jac, re.S):
        which, k, expr, comment = int(m.group(1)), int(m.group(2)), m.group(3), m.group(4)
        polys.setdefault(which, {})[k] = (re.sub(r"\s+", " ", expr), parse_monomial(comment))
    assert all(len(polys[w]) == 20 for w in (1, 2, 3)), {w: len(polys.get(w, {})) for w in (1, 2, 3)}
    idx = [int(v) for v in re.search(r"const int indices\[1968\] = \{(.*?)\};", mac, re.S).group(1).replace("\n", " ").split(",") if v.strip()]
    vals = [v.strip() for v in re.search(r"const double values\[1968\] = \{(.*?)\};", mac, re.S).group(1).replace("\n", " ").split(",") if v.strip()]
    assert len(idx) == 1968 and len(vals) == 1968

    rng = np.random.default_rng(20260928)
    cases = []
    for case in range(4):
        if case < 2:
            D = rng.normal(size=(9, 9))          # generic (also non-symmetric: the sums treat D(i,j) and D(j,i) apart)
        else:
            B = rng.normal(size=(12, 9)); D = B.T @ B    # symmetric positive semi-definite, like a real cost matrix
        u = 100.0 * rng.uniform(-1, 1, size=4)
        coef = {}
        for w in (1, 2, 3):
            coef[w] = [0.0] * 20
            for k, (expr, mono) in polys[w].items():
                py = re.sub(r"D\((\d), (\d)\)", r"D[\1,\2]", expr)
                coef[w][k] = float(eval(py, {"D": D}))
        env = {"a": coef[1], "b": coef[2], "c": coef[3], "u": list(u)}
        M = np.zeros(14400)
        for i, v in zip(idx, vals):
            M[i] = eval(v, env)
        M = M.reshape(120, 120).T        # macaulay_vec maps the column-major storage of the 120 x 120 matrix
        S = M[:27, :27] - M[:27, 27:] @ np.linalg.solve(M[27:, 27:], M[27:, :27])
        ev = np.linalg.eigvals(S)
        ev = sorted(ev, key=lambda z: (round(z.real, 6), z.imag))
        cases.append({
            "D": D.reshape(-1).tolist(), "u": u.tolist(),
            "f": {str(w): [[list(polys[w][k][1]), coef[w][k]] for k in range(20)] for w in (1, 2, 3)},
            "schur_eigenvalues": [[z.real, z.imag] for z in ev],
            "cond_M11": float(np.linalg.cond(M[27:, 27:])),
        })
        if case < 2:   # the matrix itself, entry by entry (row, column, value): pins the LAYOUT the elimination walks
            rr, cc = np.nonzero(M)
            cases[-1]["macaulay"] = [[int(r), int(c), float(M[r, c])] for r, c in zip(rr, cc)]
    json.dump({"source": "sfm/pose/dls_impl.cc:62-754 evaluated by tests/golden/make_dls_golden.py", "cases": cases},
              open(os.path.join(HERE, "dls_reference_vectors.json"), "w"), indent=0)
    exe = "/tmp/make_rand_golden"
    subprocess.check_call(["gcc", "-O1", "-o", exe, os.path.join(HERE, "make_rand_golden.c")])
    out = subprocess.check_output([exe]).decode().split()
    json.dump({"source": "glibc rand(), never seeded (tests/golden/make_rand_golden.c run in the build container)",
               "rand_max": int(out[0]), "values": [int(v) for v in out[1:]]},
              open(os.path.join(HERE, "libc_rand.json"), "w"))
    print("cases", len(cases), "cond", [c["cond_M11"] for c in cases])


if __name__ == "__main__":
    sys.exit(main())
