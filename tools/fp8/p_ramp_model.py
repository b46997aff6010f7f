#!/usr/bin/env python3
"""CPU model of the two ways the fp8 attention kernels turn a softmax weight into its e4m3 operand byte:
exp2 then round to e4m3 (FINO_FP8_P_EXP2) against the byte written directly as rne(8 (s - m) + 56 - c) (FINO_FP8_P_RAMP), P's
share of the attention output error alone (q, k, v exact), for several logit spreads and offsets c.  numpy only.

    python tools/fp8/p_ramp_model.py
"""
import numpy as np


def e4m3_decode(b):
    b = b.astype(np.int64)
    e, m = b >> 3, b & 7
    return np.where(e == 0, m / 8 * 2.0 ** -6, 2.0 ** (e - 7) * (1 + m / 8))


TAB = e4m3_decode(np.arange(0, 127))          # 0 .. 0x7e: every finite non-negative e4m3 value, ascending


def e4m3_round(v):
    idx = np.clip(np.searchsorted(TAB, v), 1, len(TAB) - 1)
    lo, hi = TAB[idx - 1], TAB[idx]
    return np.where(v - lo <= hi - v, idx - 1, idx)


def rel_rms(a, ref):
    return np.sqrt(((a - ref) ** 2).mean() / (ref ** 2).mean())


def main():
    offsets = (0.25, 0.344, 0.4375, 0.5, 0.5625)
    print("logit std | exp2 + e4m3 rounding | ramp, c = " + ", ".join(f"{c}" for c in offsets))
    for scale in (0.5, 1, 2, 4, 8):
        rng = np.random.default_rng(1)
        L, rows = 8192, 128
        s = rng.standard_normal((rows, L)) * scale
        v = rng.standard_normal((L, 64))
        # m: a whole number of octaves, stale by 0 .. 2 (deferred rescale): the maximum lands at 2^6 .. 2^8
        m = np.rint(s.max(1, keepdims=True) - 6) - rng.integers(0, 3, size=(rows, 1))
        x = s - m
        p = 2.0 ** x
        ref = (p @ v) / p.sum(1, keepdims=True)
        pe = TAB[e4m3_round(p)]
        out = [rel_rms((pe @ v) / pe.sum(1, keepdims=True), ref)]
        for c in offsets:
            pf = TAB[np.clip(np.rint(8 * x + 56 - c), 0, 126).astype(np.int64)]
            out.append(rel_rms((pf @ v) / pf.sum(1, keepdims=True), ref))
        print(f"{scale:9} | " + " ".join(f"{o:.4f}" for o in out))


if __name__ == "__main__":
    main()
