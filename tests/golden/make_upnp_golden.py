"""Reads the golden vector of the reference's own UPnP action-matrix test
(sfm/pose/build_upnp_action_matrix_using_symmetry_test.cc:49-84: cost parameters A (10 x 10), b (10) of
UpnpTests.MinimalNonCentralCameraPoseEstimation and the expected 8 x 8 action matrix, EXPECT_NEAR on the squared Frobenius
distance with 1e-6) and writes the numbers to tests/golden/upnp_action_matrix.json.  Run in the build container only
(/root/reference does not travel); data, not source text."""
import json
import os
import re

SRC = "/root/reference/src/theia/sfm/pose/build_upnp_action_matrix_using_symmetry_test.cc"


def numbers(block):
    out = []
    for tok in block.split(","):
        t = tok.replace(" ", "").replace("\n", "")
        if not t:
            continue
        m = re.fullmatch(r"(0*)-(\d[\d.]*)", t)          # clang-format left "0 - 2.17812" for -2.17812
        out.append(-float(m.group(2)) if m and m.group(1) else float(t))
    return out


def main():
    text = open(SRC).read()
    a = numbers(re.search(r"a_matrix <<(.*?);", text, re.S).group(1))
    b = numbers(re.search(r"b_vector <<(.*?);", text, re.S).group(1))
    act = numbers(re.search(r"action_matrix <<(.*?);", text, re.S).group(1))
    assert len(a) == 100 and len(b) == 10 and len(act) == 64
    out = {"source": "build_upnp_action_matrix_using_symmetry_test.cc:49-84", "tolerance_squared_frobenius": 1e-6,
           "a_matrix_row_major": a, "b_vector": b, "action_matrix_row_major": act}
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "upnp_action_matrix.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("wrote upnp_action_matrix.json")


if __name__ == "__main__":
    main()
