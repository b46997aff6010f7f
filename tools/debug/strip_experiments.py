#!/usr/bin/env python3
"""Keeps the wrong-result timing experiments OUT of the product sources.

The drop-one / fixed-operand / no-barrier switches behind the measurements DESIGN.md cites (`PD_X_*`, `PW_X_*`, `F8_X_*`,
`FR_X_*`, `GP_X_*`, `W4_X_*`, `FINO_GEMM_DESYNC_EXP`) used to live as `#ifdef` branches inside the translation units the product
library is built from.  Round 5: `frameino_amd/csrc/*` holds only code that can ship; the branches live in ONE patch,
`tools/debug/experiments.patch`, which `tools/debug/mkvar.sh --experiments` applies to a scratch copy of csrc/ before it
builds a variant library.

    python tools/debug/strip_experiments.py --make-patch     # (maintainer) sources that still carry the branches ->
                                                             #   stripped sources in place + tools/debug/experiments.patch
    python tools/debug/strip_experiments.py --check          # CI: the product sources carry no experiment macro
    python tools/debug/strip_experiments.py --apply <dir>    # copy csrc/ to <dir> and apply the patch there

Stripping = a partial preprocessor pass: every conditional whose controlling expression mentions ONLY experiment macros is
resolved with those macros undefined (the taken branch is kept verbatim, the directive lines and the dead branches go); all
other conditionals are left untouched."""
import argparse
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "frameino_amd", "csrc")
PATCH = os.path.join(ROOT, "tools", "debug", "experiments.patch")
EXP = re.compile(r"\b(?:PD_X_\w+|PW_X_\w+|F8_X_\w+|FR_X_\w+|GP_X_\w+|W4_X_\w+|FINO_GEMM_DESYNC_EXP)\b")
# ... and, only for resolving conditionals (fino_common.h's guard "an experiment switch without -DFINO_EXPERIMENT is an error" goes
# into the patch with the switches; fino_api.cpp keeps `#ifdef FINO_EXPERIMENT` -> negative fino_version(): that is product code)
EXP_EVAL = re.compile(EXP.pattern[:-3] + r"|FINO_EXPERIMENT)\b")
IDENT = re.compile(r"\b[A-Za-z_]\w*\b")
FILES = ["fino_attention.hip", "fino_attention_fp8.hip", "fino_attention_w4.hip", "fino_gemm.hip", "fino_gemm_common.h",
         "fino_common.h"]


def _cond_value(expr):
    """value of a preprocessor expression when every experiment macro is undefined, or None if it mentions anything else"""
    e = re.sub(r"/\*.*?\*/", " ", expr)
    e = re.sub(r"//.*$", " ", e)
    e = e.replace("\\\n", " ").replace("\n", " ")
    names = [m for m in IDENT.findall(e) if m != "defined"]
    if not names or not all(EXP_EVAL.fullmatch(n) for n in names):
        return None
    e = re.sub(r"defined\s*\(\s*\w+\s*\)|defined\s+\w+", "0", e)
    e = EXP_EVAL.sub("0", e)                                  # an undefined macro evaluates to 0 in #if
    e = e.replace("&&", " and ").replace("||", " or ").replace("!", " not ").replace(" not =", "!=")
    if not re.fullmatch(r"[\s0-9()<>=!andort]+", e):
        return None
    return bool(eval(e))                                  # noqa: S307  (digits, parentheses and boolean operators only)


def strip_text(text):
    lines = text.split("\n")
    out, stack = [], []          # stack entries: dict(kind="exp"|"other", taken=bool, done=bool, parent_live=bool)
    i = 0
    while i < len(lines):
        line = lines[i]
        full, j = line, i
        while full.rstrip().endswith("\\") and j + 1 < len(lines):        # continuation lines of a directive
            j += 1
            full = full.rstrip()[:-1] + "\n" + lines[j]
        m = re.match(r"\s*#\s*(ifdef|ifndef|if|elif|else|endif)\b(.*)", full, re.S)
        live = all(s["taken"] for s in stack if s["kind"] == "exp")
        if not m:
            if live:
                out.append(line)
            i += 1
            continue
        d, rest = m.group(1), m.group(2)
        if d in ("ifdef", "ifndef", "if"):
            if d == "if":
                val = _cond_value(rest)
            else:
                name = rest.strip().split()[0] if rest.strip() else ""
                val = (d == "ifndef") if EXP.fullmatch(name) else None
            if val is None:
                stack.append(dict(kind="other", taken=True))
                if live:
                    out.extend(lines[i:j + 1])
            else:
                stack.append(dict(kind="exp", taken=val, done=val))
        elif d == "elif":
            top = stack[-1]
            if top["kind"] == "other":
                if live:
                    out.extend(lines[i:j + 1])
            else:
                val = _cond_value(rest)
                if val is None:
                    raise SystemExit(f"#elif mixing experiment and product macros: {full!r}")
                top["taken"] = (not top["done"]) and val
                top["done"] = top["done"] or val
        elif d == "else":
            top = stack[-1]
            if top["kind"] == "other":
                if live:
                    out.extend(lines[i:j + 1])
            else:
                top["taken"] = not top["done"]
                top["done"] = True
        else:                                              # endif
            top = stack.pop()
            if top["kind"] == "other" and all(s["taken"] for s in stack if s["kind"] == "exp"):
                out.extend(lines[i:j + 1])
        i = j + 1
    if stack:
        raise SystemExit("unbalanced conditionals")
    return "\n".join(out)


def leftovers(text):
    return sorted(set(EXP.findall(text)))


def make_patch():
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        a, b = os.path.join(tmp, "a"), os.path.join(tmp, "b")          # a = stripped (product), b = with experiments
        os.makedirs(a), os.makedirs(b)
        for f in FILES:
            src = open(os.path.join(CSRC, f)).read()
            stripped = strip_text(src)
            if leftovers(stripped):
                # comments may still name a switch: that is documentation of the patch, reported, not fatal
                print(f"{f}: still mentions {leftovers(stripped)}", file=sys.stderr)
            open(os.path.join(a, f), "w").write(stripped)
            open(os.path.join(b, f), "w").write(src)
        p = subprocess.run(["diff", "-u", "-r", "a", "b"], cwd=tmp, capture_output=True, text=True)
        if p.returncode not in (0, 1):
            raise SystemExit(p.stderr)
        open(PATCH, "w").write(p.stdout)
        for f in FILES:
            shutil.copy(os.path.join(a, f), os.path.join(CSRC, f))
    print(f"wrote {PATCH} ({len(p.stdout.splitlines())} lines); product sources stripped in place")


def check():
    bad = {}
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".h", ".cpp")):
            text = open(os.path.join(CSRC, f)).read()
            hits = [ln for ln in text.split("\n") if re.match(r"\s*#", ln) and EXP.search(ln)]
            if hits or "wrong results" in text.lower():
                bad[f] = hits[:3] or ["'wrong results' in a comment"]
    if bad:
        raise SystemExit(f"experiment switches in the product sources: {bad}")
    print("product sources carry no experiment switch")


def apply(dst):
    if os.path.exists(dst):
        shutil.rmtree(dst)
    shutil.copytree(CSRC, dst, ignore=shutil.ignore_patterns("build", ".pytest_cache"))
    p = subprocess.run(["patch", "-p1", "-s", "-i", PATCH], cwd=dst, capture_output=True, text=True)
    if p.returncode != 0:
        raise SystemExit(f"tools/debug/experiments.patch no longer applies to frameino_amd/csrc (re-base it: the hunks are small):\n"
                         f"{p.stdout}\n{p.stderr}")
    print(dst)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--make-patch", action="store_true")
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--apply", metavar="DIR")
    a = ap.parse_args()
    if a.make_patch:
        make_patch()
    elif a.apply:
        apply(a.apply)
    else:
        check()
