"""The 4-wave attention kernel keeps O^T in the physical registers a128..a255 and touches them only through the asm of
csrc/fino_attention_w4_regs.h.  The compiler is told they are clobbered there, not that they are live in between: this
test compiles the file to ISA (hipcc cross-compiles without a GPU) and checks that NO compiler-generated instruction
names one of them, that nothing is spilled to scratch inside the tile loop, and that the generated header is current."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "frameino_amd", "csrc")


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs hipcc")
def test_compiler_never_touches_the_accumulator_registers(tmp_path):
    out = tmp_path / "w4.s"
    subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-honor-nans", "-S", "--cuda-device-only",
                    "-I", os.path.join(ROOT, "include"), "-x", "hip", os.path.join(CSRC, "fino_attention_w4.hip"),
                    "-o", str(out)], check=True, capture_output=True, timeout=900)
    reserved = re.compile(r"\ba(12[89]|1[3-9]\d|2[0-4]\d|25[0-5])\b|\ba\[(12[89]|1[3-9]\d|2[0-4]\d|25[0-5]):")
    in_asm, kernels, bad, mfma_in_asm = False, 0, [], 0
    for ln in out.read_text().splitlines():
        s = ln.strip()
        if s.startswith(";;#ASMSTART"):
            in_asm = True
        elif s.startswith(";;#ASMEND"):
            in_asm = False
        elif s.startswith(".amdhsa_kernel") and "attn_w4_kernel" in s:
            kernels += 1
        elif s.startswith(";") or not s:
            pass
        elif in_asm:
            mfma_in_asm += "v_mfma" in s
        elif reserved.search(s.split(";")[0]):
            bad.append(s)
    assert kernels >= 8                       # {bf16, fp16} x {long, short KV symbol} x {plain, fold}
    assert mfma_in_asm > 1000 and not bad, bad[:5]


def test_generated_register_header_is_current(tmp_path):
    have = open(os.path.join(CSRC, "fino_attention_w4_regs.h")).read()
    gen = os.path.join(ROOT, "tools", "gen_attn_w4_regs.py")
    src = open(gen).read().replace('os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "frameino_amd", "csrc",\n                    "fino_attention_w4_regs.h")',
                                   repr(str(tmp_path / "regs.h")))
    (tmp_path / "gen.py").write_text(src)
    subprocess.run([sys.executable, str(tmp_path / "gen.py")], check=True, capture_output=True)
    assert (tmp_path / "regs.h").read_text() == have
