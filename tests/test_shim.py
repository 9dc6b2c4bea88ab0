"""The compiled C++ host side of the boundary (shim/: BundleAdjuster with the reference's registration calls, the RANSAC
batch front end) builds against include/theia_hip.h and links libtheia_hip.so; the GPU test runs it."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "shim", "_build", "shim_test")


def _build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "shim")])
    assert os.path.exists(EXE)


def test_shim_builds_and_links_against_the_c_abi():
    _build()
    out = subprocess.check_output(["ldd", EXE]).decode()
    assert "libtheia_hip.so" in out and "not found" not in out.split("libtheia_hip.so")[1].split("\n")[0]


@pytest.mark.gpu
def test_shim_runs_bundle_adjustment_and_ransac_on_the_device():
    _build()
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    for line in ("ok bundle adjustment", "ok relative pose batch", "ok calibrated absolute pose batches", "ok rigid transformation batch", "ok radial-distortion absolute pose batch", "ok fundamental matrix batch",
                 "ok view batch", "ok track covariances"):
        assert line in r.stdout, r.stdout
