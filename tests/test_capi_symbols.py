"""CPU: the C-ABI library loads and exports every symbol include/theia_hip.h
declares (no compute calls here)."""
import ctypes
import os
import re

from pytheiasfm_amd import _capi as capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "theia_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(theia_(?:hip_\w+|ba_options_default|ransac_params_default))\s*\(", src)
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    lib = capi.lib()
    declared = header_functions()
    assert len(declared) >= 18
    for name in declared:
        assert hasattr(lib, name), f"libtheia_hip.so does not export {name}"
    assert sorted(capi.EXPORTED_SYMBOLS) == declared


def test_version_and_error_strings():
    lib = capi.lib()
    assert b"gfx950" in lib.theia_hip_version()
    assert isinstance(lib.theia_hip_last_error(), bytes)


def test_struct_layouts_match_header_sizes():
    # plain-C layout: 3 int32 + flags + int64 + 12 pointers
    assert ctypes.sizeof(capi.BaProblem) == 4 * 4 + 8 + 23 * 8   # + 7 camera-prior pointers + obs_kind + 3 inverse-depth pointers
    assert ctypes.sizeof(capi.BaOptions) == 10 * 4 + 7 * 8
    assert ctypes.sizeof(capi.RansacParams) == 3 * 8 + 8 * 4
    assert ctypes.sizeof(capi.BaSummary) == 4 * 4 + 4 * 8 + 2 * 4 + 5 * 8 + 3 * 8 + 8 + 2 * 4


def test_defaults_match_reference_structs():
    from pytheiasfm_amd import ba
    o = ba.default_options()
    # bundle_adjustment.h:87-167
    assert o.loss_function_type == 0 and o.robust_loss_width == 2.0
    assert o.max_num_iterations == 100 and o.use_homogeneous_point_parametrization == 1
    assert o.function_tolerance == 1e-6 and o.gradient_tolerance == 1e-10 and o.parameter_tolerance == 1e-8
    assert o.max_trust_region_radius == 1e12 and o.use_inner_iterations == 1
    from pytheiasfm_amd import ransac
    ransac._sig()
    p = capi.RansacParams()
    capi.lib().theia_ransac_params_default(ctypes.byref(p))
    # sample_consensus_estimator.h:59-68
    assert p.error_thresh == -1 and p.failure_probability == 0.01 and p.min_iterations == 100
    assert p.max_iterations == 2 ** 31 - 1 and p.use_mle == 0 and p.use_lo == 0 and p.lo_start_iterations == 50
    assert p.ransac_type == 0
