"""ctypes loader of the CPU oracle (oracle/_build/liboracle.so) -- TEST SIDE ONLY.

The oracle shares the plain-C struct layout of include/theia_hip.h, so the
ctypes classes of pytheiasfm_amd._capi describe both.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from pytheiasfm_amd import _capi as capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "_build", "liboracle.so")

_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB):
        build()
    L = C.CDLL(LIB)
    dp = capi.c_double_p
    L.oracle_reprojection_error.argtypes = [C.c_int, dp, dp, dp, dp, dp, dp, dp, dp, dp]
    L.oracle_ba_options_default.argtypes = [C.POINTER(capi.BaOptions)]
    L.oracle_ba_evaluate.argtypes = [C.POINTER(capi.BaProblem), C.POINTER(capi.BaOptions), dp, dp, dp, dp]
    L.oracle_ba_reduced_system.argtypes = [C.POINTER(capi.BaProblem), C.POINTER(capi.BaOptions), C.c_double,
                                           capi.c_int32_p, dp, dp, C.c_int64]
    L.oracle_ba_solve.argtypes = [C.POINTER(capi.BaProblem), C.POINTER(capi.BaOptions), C.POINTER(capi.BaSummary)]
    L.oracle_loss_evaluate.argtypes = [C.c_int, C.c_double, C.c_double, dp]
    L.oracle_sphere_plus.argtypes = [dp, dp, dp]
    L.oracle_sphere_plus_jacobian.argtypes = [dp, dp]
    _lib = L
    return L


def default_options():
    o = capi.BaOptions()
    load().oracle_ba_options_default(C.byref(o))
    return o


def reprojection_error(model, ext, intr, X, uv, sqrt_info=None):
    L = load()
    f8 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    ext, X, uv = f8(ext), f8(X), f8(uv)
    k = np.zeros(10); k[: len(intr)] = intr
    res = np.zeros(2); Je = np.zeros((2, 6)); Ji = np.zeros((2, len(intr))); Jp = np.zeros((2, 4))
    si = None if sqrt_info is None else f8(sqrt_info)
    ok = L.oracle_reprojection_error(model, capi.ptr(ext, C.c_double), capi.ptr(k, C.c_double),
                                     capi.ptr(X, C.c_double), capi.ptr(uv, C.c_double),
                                     capi.ptr(si, C.c_double), capi.ptr(res, C.c_double),
                                     capi.ptr(Je, C.c_double), capi.ptr(Ji, C.c_double), capi.ptr(Jp, C.c_double))
    return ok, res, Je, Ji, Jp


def evaluate(problem, options):
    L = load()
    pd = 3 if options.use_homogeneous_point_parametrization else 4
    n = problem.obs_uv.shape[0]
    cost = C.c_double(0)
    r = np.zeros((n, 2)); jc = np.zeros((n, 2, 6)); jp = np.zeros((n, 2, pd))
    st = problem.as_struct()
    ok = L.oracle_ba_evaluate(C.byref(st), C.byref(options), C.byref(cost), capi.ptr(r, C.c_double),
                              capi.ptr(jc, C.c_double), capi.ptr(jp, C.c_double))
    return ok, cost.value, r, jc, jp


def reduced_system(problem, options, radius):
    L = load()
    ncam = problem.cam_ext.shape[0]
    cap = (6 * ncam) ** 2
    S = np.zeros(cap); rhs = np.zeros(6 * ncam); n = C.c_int32(0)
    st = problem.as_struct()
    rc = L.oracle_ba_reduced_system(C.byref(st), C.byref(options), radius, C.byref(n), capi.ptr(S, C.c_double),
                                    capi.ptr(rhs, C.c_double), cap)
    assert rc == 0, rc
    n = n.value
    return S[: n * n].reshape(n, n).copy(), rhs[:n].copy()


def solve(problem, options, trace_capacity=256):
    """Runs the oracle LM loop; problem parameters are updated in place."""
    L = load()
    s = capi.BaSummary()
    tr = capi.Trace(trace_capacity)
    tr.attach(s)
    st = problem.as_struct()
    rc = L.oracle_ba_solve(C.byref(st), C.byref(options), C.byref(s))
    assert rc == 0, rc
    tr.finish(s)
    return s, tr
