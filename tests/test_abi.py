"""
The C-ABI library: builds for gfx950 without a GPU, loads, and exports every
symbol include/mixemt_hip.h declares.  No compute calls here (CPU box).
"""
import ctypes
import os
import re

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def lib_path():
    from mixemt_amd import build
    return build.build()


def _declared(headers=("mixemt_hip.h", "mixemt_hip_tuning.h")):
    names = set()
    for header in headers:
        text = open(os.path.join(ROOT, "include", header)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names.update(re.findall(r"\b(mxm_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


def test_header_declares_the_boundary():
    names = _declared()
    for must in ("mxm_build_em_matrix", "mxm_em_iter", "mxm_m_finalize", "mxm_em_loop",
                 "mxm_em_step", "mxm_workspace_bytes", "mxm_last_error", "mxm_version"):
        assert must in names


def test_boundary_header_holds_no_tuning_knobs():
    """Process-wide knobs and measurement hooks live in mixemt_hip_tuning.h, not in the boundary."""
    boundary = _declared(("mixemt_hip.h",))
    assert not [n for n in boundary if n.startswith("mxm_set_") or n.startswith("mxm_diag_")]
    assert "mxm_set_timing_events" in _declared(("mixemt_hip_tuning.h",))


def test_library_exports_every_declared_symbol(lib_path):
    lib = ctypes.CDLL(lib_path)
    for name in _declared():
        assert hasattr(lib, name), "missing export: %s" % name


def test_binding_table_matches_header(lib_path):
    from mixemt_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared()
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "mixemt_hip.h")).read()
    declared = int(re.search(r"#define\s+MXM_VERSION\s+(\d+)", header).group(1))
    assert lib.mxm_version() == declared == _lib.ABI_VERSION          # header, library and binding agree
    assert ctypes.sizeof(_lib.EmState) == 24
    assert lib.mxm_linear_supported(5408) == 1 and lib.mxm_linear_supported(3) == 0
    assert lib.mxm_linear_supported(8192) == 1 and lib.mxm_linear_supported(8193) == 0
    assert lib.mxm_workspace_bytes(1000000, 5408, 1) >= 1024 * 5408 * 8
    # the quad dictionary's never-overflows size counts the 64 KB pieces the encoder hands out (six of the largest records
    # fit one; 5120 may be open at the end), not the records alone
    for rows in (1, 7, 1000000):
        assert lib.mxm_quad_bytes(rows, 5408) >= (-(-rows // 6) + 5120) * 65536 > rows * (2048 + 256 * 32)
    assert lib.mxm_quad_bytes(0, 5408) == 0


def test_binding_refuses_a_library_of_another_abi_version(lib_path, monkeypatch):
    """mxm_version() is what a binding has to detect an incompatible library with: load() compares."""
    from mixemt_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "ABI_VERSION", _lib.ABI_VERSION + 1)
    with pytest.raises(_lib.MixemtHipError, match="ABI version"):
        _lib.load()


def test_tuning_knobs_reset_and_take_their_ranges(lib_path):
    """Setters validate their argument; mxm_reset_tuning restores the shipped defaults (host-side state only)."""
    from mixemt_amd import _lib
    lib = _lib.load()
    assert lib.mxm_set_batch_tile(7) < 0 and lib.mxm_set_batch_tile(2) == 0
    assert lib.mxm_set_quad_encoder(2) < 0 and lib.mxm_set_quad_encoder(0) == 0 and lib.mxm_set_quad_encoder(1) == 0
    assert lib.mxm_restart_tile(5408) == 2
    assert lib.mxm_reset_tuning() == 0
    assert lib.mxm_restart_tile(5408) == 4
    buf = ctypes.create_string_buffer(96)
    assert lib.mxm_describe_stream_kernel(5408, 1, buf, len(buf)) == 0
    assert buf.value == b"em_iter_wide_kernel<512, 6, 1, 3, 1>"


def test_code_object_is_gfx950(lib_path):
    blob = open(lib_path, "rb").read()
    assert b"gfx950" in blob
    assert b"em_iter_wide_kernel" in blob and b"build_em_matrix_kernel" in blob


def test_product_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import numpy
    from conftest import em_args
    from mixemt_amd import em
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        em.run_em(numpy.zeros((4, 3)), numpy.ones(4), em_args())
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        em.converged(numpy.zeros(3), numpy.zeros(3))
