"""CPU tier: the C-ABI library loads and exports every symbol the header declares.
No compute call is made (there is no GPU here)."""
import os
import re

import pytest

from conftest import ROOT
from picasso_amd import _lib


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "picasso_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pmi_[a-z0-9_]+)\s*\(", text)))


def test_library_is_built():
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"


def test_every_declared_symbol_is_exported_and_bound():
    lib = _lib.load()
    names = _header_symbols()
    assert len(names) >= 25
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/picasso_hip.h but not exported"
        assert name in _lib.SYMBOLS, f"{name} has no ctypes signature in picasso_amd/_lib.py"
    assert set(_lib.SYMBOLS) == set(names)


def test_version_and_error_string():
    lib = _lib.load()
    assert lib.pmi_version() >= 100
    assert isinstance(_lib.last_error(), str)


def test_no_gpu_fails_loudly(monkeypatch):
    """Without a device the product path raises; it never falls back to a CPU path."""
    import numpy as np
    from picasso_amd import backend
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(_lib.HipBackendError, match="no HIP device"):
        backend.gaussmle_arrays(np.zeros((1, 7, 7), np.float32), 1e-3, 10)
    with pytest.raises(_lib.HipBackendError, match="no HIP device"):
        backend.identify_arrays(np.zeros((1, 32, 32), np.uint16), 100, 7)


def test_product_package_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under picasso_amd/ may reference it."""
    pkg = os.path.join(ROOT, "picasso_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "import oracle" not in src and "from oracle" not in src, f
                assert "picasso_oracle" not in src and "orc_" not in src, f
