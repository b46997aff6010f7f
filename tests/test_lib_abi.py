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
    assert lib.fino_version() == 103 == _lib.ABI_VERSION
    # bad dtype -> FINO_ERR_ARG before anything touches a device
    rc = lib.fino_gemm(1, 1, 0, 1, 8, 8, 8, 8, 8, 8, 0, 0, 0, 0, 0, 0, 7, 0)
    assert rc == -1 and b"dtype" in lib.fino_last_error()
    rc = lib.fino_gemm(16, 16, 0, 16, 8, 12, 8, 8, 8, 16, 0, 0, 0, 0, 0, 0, 0, 0)
    assert rc == -1 and b"multiples of 8" in lib.fino_last_error()
    rc = lib.fino_attn_fwd(16, 16, 16, 16, 1, 1, 8, 8, 96, *([8] * 12), ctypes.c_float(1.0), 0, 0)
    assert rc == -3 and b"head_dim" in lib.fino_last_error()
    rc = lib.fino_adaln_modulate(16, 16, 4, 12, 16, 16, 16, 16, 0, 0, ctypes.c_float(1e-6), 0, 0)
    assert rc == -1


def test_argument_validation_of_the_later_entry_points(lib):
    f1 = ctypes.c_float(1.0)
    # workspace must be 16-byte aligned
    rc = lib.fino_attn_fwd_ws(16, 16, 16, 16, 1, 1, 8, 8, 128, *([8] * 12), f1, 0, 24, 1024, 0)
    assert rc == -1 and b"workspace" in lib.fino_last_error()
    assert lib.fino_attn_workspace_bytes(1, 24, 12320, 12320, 96) == 0        # unsupported head_dim: nothing to split
    assert lib.fino_attn_workspace_bytes(2, 24, 12320, 12320, 128) > 0        # the bench shape has a partial last round
    # MXFP8: K (= cols) must be a multiple of 128
    assert lib.fino_mxfp8_scale_bytes(256, 100) == 0 and lib.fino_mxfp8_scale_bytes(300, 256) == 2 * 2 * 1024
    rc = lib.fino_quantize_mxfp8(16, 16, 16, 4, 100, 104, 0, 0)
    assert rc == -1 and b"multiple of 128" in lib.fino_last_error()
    rc = lib.fino_gemm_mxfp8(16, 16, 16, 16, 0, 16, 8, 8, 64, 8, 0, 0, 0, 0, 0, 0, 0, 0)
    assert rc == -1 and b"multiple of 128" in lib.fino_last_error()
    # UniPC step: null history buffers
    rc = lib.fino_cfg_unipc_step(16, 0, 16, 0, 16, 16, 4, 3, 4, 8, 8, 16, 0, 0)
    assert rc == -1 and b"null" in lib.fino_last_error()
    # K-blocked A: the kernel turns a K-tile index into its block by a reciprocal multiply, exact only while
    # kt * (a_tpb * ceil(65536 / a_tpb) - 65536) < 65536 -- a shape beyond that is refused, not computed wrongly:
    # a_tpb = 255 (a_block_k = 16320), K = 2 blocks -> largest K-tile index 509, 509 * 254 >= 65536
    blk = 255 * 64
    rc = lib.fino_gemm_blocked_a(16, 16, 0, 16, 8, 256, 2 * blk, blk, 8 * blk, 1, 0, blk, 2 * blk, 256, 16, 256, 16, 0, 0, 0, 0, 0)
    assert rc == -3 and b"exact range" in lib.fino_last_error(), lib.fino_last_error()
    # ... a non-power-of-two block inside the range passes validation (M = 0: nothing is launched): a_tpb = 3, K = 3072
    rc = lib.fino_gemm_blocked_a(16, 16, 0, 16, 0, 256, 3072, 192, 8 * 192, 1, 0, 192, 3072, 256, 16, 256, 16, 0, 0, 0, 0, 0)
    assert rc == 0, lib.fino_last_error()
    # per-call tile height: 0 (planned) or 2 .. 8
    rc = lib.fino_gemm_split_n(16, 16, 0, 16, 8, 256, 64, 64, 64, 256, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 9, 0)
    assert rc == -1 and b"tile_m" in lib.fino_last_error()
    # trajectory builder: even tap count
    rc = lib.fino_traj_blur_quantize(16, 16, 16, 16, 44, 3, 8, 8, 0)
    assert rc == -1 and b"odd" in lib.fino_last_error()


def test_product_path_has_no_cpu_fallback():
    import torch
    from frameino_amd import ops
    x = torch.zeros(4, 64, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.layernorm(x)


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load(str(tmp_path / "nope.so"))


def test_an_experiment_build_is_refused_as_the_product_library(tmp_path, monkeypatch):
    """a library compiled with -DFINO_EXPERIMENT (wrong-result timing switches) reports a negative version
    and `_lib.load()` refuses it unless FINO_ALLOW_EXPERIMENT=1; the switches themselves do not compile without the macro."""
    import subprocess
    src = tmp_path / "fake.c"
    src.write_text(f"int fino_version(void) {{ return -{_lib.ABI_VERSION}; }}\n")
    so = tmp_path / "libfake_experiment.so"
    subprocess.run(["gcc", "-shared", "-fPIC", "-o", str(so), str(src)], check=True)
    monkeypatch.delenv("FINO_ALLOW_EXPERIMENT", raising=False)
    with pytest.raises(RuntimeError, match="EXPERIMENT build"):
        _lib.load(str(so))
    monkeypatch.setenv("FINO_ALLOW_EXPERIMENT", "1")
    with pytest.raises(AttributeError):              # accepted, then fails on the first symbol the stand-in lacks
        _lib.load(str(so))
    # a library of another ABI version (a stale build found through FINO_LIB_PATH) is refused with a clear message
    src.write_text("int fino_version(void) { return 100; }\n")
    stale = tmp_path / "libfake_stale.so"
    subprocess.run(["gcc", "-shared", "-fPIC", "-o", str(stale), str(src)], check=True)
    with pytest.raises(RuntimeError, match="binds ABI version"):
        _lib.load(str(stale))
    # the version follows the macro (the switches and their guard live in tools/debug/experiments.patch since round 5: next test)
    csrc = os.path.join(os.path.dirname(_lib.HEADER_PATH), "..", "frameino_amd", "csrc")
    assert "return -FINO_VERSION" in open(os.path.join(csrc, "fino_api.cpp")).read()


def test_product_sources_carry_no_experiment_switch_and_the_patch_applies(tmp_path):
    """the wrong-result timing switches (PD_X_* / PW_X_* / F8_X_* / FR_X_* / GP_X_* / W4_X_* ...) are not in
    frameino_amd/csrc any more -- they live in tools/debug/experiments.patch, which must keep applying to the product sources
    (tools/debug/mkvar.sh --experiments builds the variant libraries from the patched scratch copy)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, "tools", "debug", "strip_experiments.py")
    p = subprocess.run([sys.executable, tool, "--check"], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr
    dst = str(tmp_path / "csrc_exp")
    p = subprocess.run([sys.executable, tool, "--apply", dst], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr
    import re
    common = open(os.path.join(dst, "fino_common.h")).read()
    guard = common[common.index("#if (defined("):common.index("#error")]
    used = set()
    for f in os.listdir(dst):
        if f.endswith((".hip", ".h", ".cpp")) and f != "fino_common.h":
            used |= set(re.findall(r"\b((?:W4|F8|FR|PD|PW|GP)_X_[A-Z]+|FINO_GEMM_DESYNC_EXP)\b", open(os.path.join(dst, f)).read()))
    assert len(used) > 20 and all(f"defined({u})" in guard for u in used), sorted(u for u in used if f"defined({u})" not in guard)
    patched = open(os.path.join(dst, "fino_attention.hip")).read()
    assert "PD_X_NODMA" in patched and "PD_X_NODMA" not in open(os.path.join(root, "frameino_amd", "csrc", "fino_attention.hip")).read()
    for f in os.listdir(os.path.join(root, "frameino_amd", "csrc")):
        if f.endswith((".hip", ".h", ".cpp")):
            assert "wrong results" not in open(os.path.join(root, "frameino_amd", "csrc", f)).read().lower(), f
