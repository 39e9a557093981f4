"""CPU-side checks of the native library: it builds for gfx950, loads without a
GPU, and exports exactly the symbols include/lss_hip.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "lss_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lss_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def built():
    from lss2_multimodal_nu_amd import build_native
    return build_native.build(verbose=False)


def test_library_exports_every_declared_symbol(built):
    lib = ctypes.CDLL(built)
    syms = header_symbols()
    assert len(syms) >= 14
    for s in syms:
        assert hasattr(lib, s), "missing export " + s


def test_binding_table_matches_header(built):
    from lss2_multimodal_nu_amd import _native
    assert sorted(_native.SIGNATURES) == header_symbols()
    L = _native.lib()
    assert L.lss_abi_version() == 1
    assert b"NULL" in L.lss_error_string(-1)


def test_argument_checks_without_gpu(built):
    """Argument validation happens before any HIP call, so it is testable here."""
    from lss2_multimodal_nu_amd import _native
    L = _native.lib()
    assert L.lss_points_to_voxels(None, None, None, None, None, None, None, 1, 1, 1, 1, 1, 1, 1, 1, None, None, None, None) == -1
    assert L.lss_conv2d_packed_weight_bytes(4, 128, 1, 1, 1) == 4 * 128 * 2
    one = ctypes.c_void_p(16)
    assert L.lss_lift_splat_fwd(one, one, one, 1, 1, 1, 1, 1, 48, 1, 1, 1, one, 0, None) == -2  # C not 64/128
    assert L.lss_lift_splat_fwd(one, one, one, 1, 1, 1, 1, 1, 64, 1, 1, 1, one, 7, None) == -3  # bad layout
    assert L.lss_depthnet_softmax_fwd(one, one, one, 1, 100, 4, 4, 4, one, one, 0, None) == -2  # Cin % 64


def test_exact_index_kernel_has_no_contracted_fma():
    """geom_bucket.hip must be built with -ffp-contract=off."""
    from lss2_multimodal_nu_amd import build_native
    assert "-ffp-contract=off" in build_native.SOURCES["geom_bucket.hip"]


def test_no_oracle_import_in_product():
    pkg = os.path.join(ROOT, "lss2_multimodal_nu_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py"):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", txt, flags=re.M), f
