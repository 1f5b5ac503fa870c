"""The C-ABI shared library builds for gfx950 without a GPU, loads, and exports every symbol that
include/tdeed_hip.h declares (no compute calls here)."""
import os
import re

import pytest

from helpers import ROOT


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from tdeed_amd import _lib
    return _lib.load()


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "tdeed_hip.h")).read()
    declared = set(re.findall(r"\b(tdeed_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 25
    from tdeed_amd import _lib
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for name in declared:
        assert getattr(lib, name) is not None


def test_abi_version_and_error_string(lib):
    assert lib.tdeed_abi_version() == 1
    assert isinstance(lib.tdeed_last_error(), bytes)


def test_argument_validation_without_gpu(lib):
    """Argument checks run before any launch, so they are testable on the CPU box: errors are loud."""
    from tdeed_amd._lib import call, HipCallError
    with pytest.raises(HipCallError, match="null pointer"):
        call("tdeed_gemm_fwd", None, 8, None, 0, 0, None, 0, 8, 8, 8, None, 8, None, None, None, 0, 0, None, 8,
             1, 0, 0, 0, 0, None, None, 0, 0, 0, 0, None)
    with pytest.raises(HipCallError, match="multiples of 8"):
        call("tdeed_gemm_fwd", 1 << 20, 12, None, 0, 0, None, 0, 8, 12, 8, 1 << 20, 12, None, None, None, 0, 0,
             1 << 20, 8, 1, 0, 0, 0, 0, None, None, 0, 0, 0, 0, None)
    with pytest.raises(HipCallError, match="group width"):
        call("tdeed_gconv3x3_fwd", 1 << 20, 1, 8, 8, 24, 12, 1, 1 << 20, None, 1 << 20, 1 << 20, 1 << 20, 1 << 20, None, None, None, 1, 0, None)


def test_missing_library_is_loud(monkeypatch, tmp_path):
    from tdeed_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.HipLibraryMissing):
        _lib.load()
