"""CPU-side checks of the C ABI: the library builds for gfx950, loads, and exports every symbol
include/s3grl.h declares.  No compute calls (no GPU here)."""
import re
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge

    ge.build()
    from s3grl_amd import _native

    return _native.lib()


def test_exports_every_declared_symbol(lib):
    from s3grl_amd import _native

    header = (REPO / "include" / "s3grl.h").read_text()
    declared = set(re.findall(r"\b(s3grl_[a-z_]+)\s*\(", header))
    assert declared == set(_native.SYMBOLS)
    for name in declared:
        assert getattr(lib, name) is not None


def test_abi_version_and_status_strings(lib):
    assert lib.s3grl_abi_version() == 6
    assert lib.s3grl_status_string(0) == b"ok"
    assert lib.s3grl_status_string(2) == b"not implemented"


def test_status_maps_to_reference_exceptions():
    from s3grl_amd import _native as N

    with pytest.raises(NotImplementedError):
        N.check(N.ERR_NOT_IMPLEMENTED, "x")
    with pytest.raises(AssertionError):
        N.check(N.ERR_NO_FEATURES, "x")
    with pytest.raises(ValueError):
        N.check(N.ERR_INVALID_ARGUMENT, "x")


def test_engine_refuses_to_run_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from s3grl_amd.engine import Engine

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Engine()


def test_product_never_imports_oracle():
    for p in (REPO / "s3grl_amd").rglob("*.py"):
        txt = p.read_text()
        assert "import oracle" not in txt and "from oracle" not in txt, p
