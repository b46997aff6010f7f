"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/frameino_hip.h declares, the ctypes
table covers all of them, and argument validation fails loudly (no GPU needed: validation happens before any launch)."""
import ctypes
import os

import pytest

from frameino_amd import _lib


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _lib.load()


def test_header_symbols_are_exported_and_bound(lib):
    declared = _lib.declared_symbols()
    assert len(declared) >= 14
    assert sorted(_lib.SIGNATURES) == declared, "ctypes table and header disagree"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/frameino_hip.h but not exported"


def test_version_and_error_channel(lib):
    assert lib.fino_version() == 100
    # bad dtype -> FINO_ERR_ARG before anything touches a device
    rc = lib.fino_gemm(1, 1, 0, 1, 8, 8, 8, 8, 8, 8, 0, 0, 0, 0, 0, 0, 7, 0)
    assert rc == -1 and b"dtype" in lib.fino_last_error()
    rc = lib.fino_gemm(16, 16, 0, 16, 8, 12, 8, 8, 8, 16, 0, 0, 0, 0, 0, 0, 0, 0)
    assert rc == -1 and b"multiples of 8" in lib.fino_last_error()
    rc = lib.fino_attn_fwd(16, 16, 16, 16, 1, 1, 8, 8, 96, *([8] * 12), ctypes.c_float(1.0), 0, 0)
    assert rc == -3 and b"head_dim" in lib.fino_last_error()
    rc = lib.fino_adaln_modulate(16, 16, 4, 12, 16, 16, 16, 16, 0, 0, ctypes.c_float(1e-6), 0, 0)
    assert rc == -1


def test_product_path_has_no_cpu_fallback():
    import torch
    from frameino_amd import ops
    x = torch.zeros(4, 64, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.layernorm(x)


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load(str(tmp_path / "nope.so"))
