"""Torch-tensor front end of the C ABI (include/frameino_hip.h).

PyTorch is plumbing here: it owns device memory and the HIP stream; every op below forwards raw
pointers + strides to libframeino_hip.so on torch's *current* stream (so the calls are capturable in a
torch.cuda.CUDAGraph == hipGraph).  No op has a torch/CPU fallback.
"""
import os

import torch

from . import _lib

BF16, F16 = 0, 1
EPI_NONE, EPI_GELU_TANH, EPI_RESIDUAL, EPI_GATED_RESIDUAL, EPI_GATED_RESIDUAL_STAGED = 0, 1, 2, 3, 4
EPI_F32, EPI_F32_RESIDUAL = 5, 6           # fp32 output of a split-bf16 product (gemm_f32)


class KernelTimer:
    """HIP-event timing of named kernel launches on the stream they are launched on (used by bench.py for the
    live `roofline.achieved` figure).  `with KernelTimer({"attn_self"}) as kt: ...; kt.summary()`."""

    active = None

    def __init__(self, names):
        self.names = set(names)
        self.events = {n: [] for n in self.names}
        self.flops = {}
        self.bytes = {}

    def __enter__(self):
        KernelTimer.active = self
        return self

    def __exit__(self, *exc):
        KernelTimer.active = None

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for n, evs in self.events.items():
            ms = [s.elapsed_time(e) for s, e in evs]
            if ms:
                out[n] = {"launches": len(ms), "avg_us": 1e3 * sum(ms) / len(ms), "total_ms": sum(ms)}
        return out


def _timed(name):
    kt = KernelTimer.active
    if kt is None or name not in kt.names:
        return None
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    kt.events[name].append((s, e))
    s.record()
    return e


def _timed_gemm(epilogue, flops, nbytes=0.0):
    """a GEMM launch under KernelTimer: counted under "gemm_epi<epilogue>" (bench.py's roofline.secondary rows: one per epilogue
    class of the block GEMMs) when that name is asked for, else under "gemm".  Returns a closure to call after the launch."""
    kt = KernelTimer.active
    if kt is None:
        return None
    name = f"gemm_epi{int(epilogue)}"
    if name not in kt.names:
        name = "gemm"
        if name not in kt.names:
            return None
    ev = _timed(name)

    def done():
        ev.record()
        kt.flops[name] = kt.flops.get(name, 0.0) + flops
        kt.bytes[name] = kt.bytes.get(name, 0.0) + nbytes
    return done


def _dt(t):
    if t.dtype == torch.bfloat16:
        return BF16
    if t.dtype == torch.float16:
        return F16
    raise TypeError(f"frameino_amd: activations must be bf16 or fp16, got {t.dtype}")


F32 = 2          # FINO_F32: fp32 STORAGE (the Wan VAE's rearrangement kernels in its fp32-faithful mode); never an MFMA operand type


def _dt3(t):
    return F32 if t.dtype == torch.float32 else _dt(t)


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return 0 if t is None else t.data_ptr()


def _rows2d(t):
    """View [..., D] with contiguous last dim and uniform row stride as (rows, D, ld)."""
    if not t.is_cuda:
        raise RuntimeError("frameino_amd ops need CUDA(HIP) tensors; there is no CPU path")
    if t.stride(-1) != 1:
        raise ValueError("last dimension must be contiguous")
    t2 = t if t.dim() == 2 else t.reshape(-1, t.shape[-1]) if t.is_contiguous() else None
    if t2 is None:
        if t.dim() == 3 and t.shape[0] == 1:
            t2 = t[0]
        else:
            raise ValueError("need a 2-D row-strided view")
    return t2, t2.shape[0], t2.shape[1], t2.stride(0)


def adaln_modulate(x, shift, scale, sel=None, eps=1e-6, out=None):
    """y = T(LN(x)*(1+scale[sel])+shift[sel]).  shift/scale: fp32 [R, D] views of one table (same row stride)."""
    x2, rows, dim, ldx = _rows2d(x)
    out = torch.empty_like(x2) if out is None else out
    o2, _, _, ldy = _rows2d(out)
    assert shift.dtype == torch.float32 and scale.dtype == torch.float32
    ms = shift.stride(0) if shift.dim() == 2 else 0
    if scale.dim() == 2:
        assert scale.stride(0) == ms
    _lib.check(_lib.lib().fino_adaln_modulate(_p(x2), _p(o2), rows, dim, ldx, ldy, _p(shift), _p(scale), ms, _p(sel),
                                             eps, _dt(x), _stream()), "fino_adaln_modulate")
    return out.view(x.shape) if out.shape != x.shape and out.numel() == x.numel() else out


def layernorm(x, weight=None, bias=None, eps=1e-5, out=None):
    x2, rows, dim, ldx = _rows2d(x)
    out = torch.empty_like(x2) if out is None else out
    o2, _, _, ldy = _rows2d(out)
    for t in (weight, bias):
        assert t is None or t.dtype == torch.float32
    _lib.check(_lib.lib().fino_layernorm(_p(x2), _p(o2), rows, dim, ldx, ldy, _p(weight), _p(bias), eps, _dt(x),
                                        _stream()), "fino_layernorm")
    return out.view(x.shape) if out.shape != x.shape and out.numel() == x.numel() else out


def layernorm_zero(x, weight, bias, shift, scale, sel=None, eps=1e-5, out=None):
    """y = T(T(T(LN(x)*w+b) * T(1+scale[sel])) + shift[sel]) -- CogVideoXLayerNormZero / AdaLayerNorm in T."""
    x2, rows, dim, ldx = _rows2d(x)
    out = torch.empty_like(x2) if out is None else out
    o2, _, _, ldy = _rows2d(out)
    for t in (weight, bias, shift, scale):
        assert t is None or t.dtype == torch.float32
    ms = shift.stride(0) if shift.dim() == 2 else 0
    _lib.check(_lib.lib().fino_layernorm_zero(_p(x2), _p(o2), rows, dim, ldx, ldy, _p(weight), _p(bias), _p(shift),
                                             _p(scale), ms, _p(sel), eps, _dt(x), _stream()), "fino_layernorm_zero")
    return out.view(x.shape) if out.shape != x.shape and out.numel() == x.numel() else out


def gated_residual(x, y, gate=None, sel=None, out=None, staged=False):
    """out = T(float(x) + float(y)*gate[sel]) (gate None => T(x+y)); staged: T(x + T(y*gate))."""
    x2, rows, dim, ldx = _rows2d(x)
    y2, _, _, ldy = _rows2d(y)
    out = torch.empty_like(x2) if out is None else out
    o2, _, _, ldo = _rows2d(out)
    ms = gate.stride(0) if (gate is not None and gate.dim() == 2) else 0
    fn = _lib.lib().fino_gated_residual_staged if staged else _lib.lib().fino_gated_residual
    _lib.check(fn(_p(x2), _p(y2), _p(o2), rows, dim, ldx, ldy, ldo, _p(gate), ms, _p(sel), _dt(x), _stream()),
               "fino_gated_residual")
    return out.view(x.shape) if out.shape != x.shape and out.numel() == x.numel() else out


SCALE_FOLDED = -1.0     # FINO_ATTN_SCALE_FOLDED: q already carries softmax_scale * log2(e) (rmsnorm_rope_(out_scale=...))
LOG2E = 1.4426950408889634


def rmsnorm_rope_(x, weight, eps, cos=None, sin=None, head_dim=0, out_scale=1.0):
    """In place on the row-strided view x [rows, D] (e.g. the q or k column block of a fused QKV buffer).
    out_scale: multiplies the fp32 result before its one rounding (q for attention(scale=SCALE_FOLDED))."""
    x2, rows, dim, ldx = _rows2d(x)
    if cos is not None:
        assert cos.dtype == torch.float32 and cos.is_contiguous() and cos.shape == (rows, head_dim // 2)
        assert sin.dtype == torch.float32 and sin.is_contiguous() and sin.shape == cos.shape
    _lib.check(_lib.lib().fino_rmsnorm_rope_scaled(_p(x2), rows, dim, ldx, _p(weight), eps, _p(cos), _p(sin), head_dim,
                                                  float(out_scale), _dt(x), _stream()), "fino_rmsnorm_rope")
    return x


def qkv_rmsnorm_rope_(qkv, dim, q_weight, q_eps, k_weight, k_eps, cos, sin, head_dim, q_out_scale=1.0, out=None,
                      head_off=None, head_ld=None):
    """q (columns [0, dim)) and k ([dim, 2 dim)) of the fused projection qkv [rows, 3 dim] in ONE launch: the arithmetic of
    rmsnorm_rope_(q, ..., out_scale=q_out_scale) and rmsnorm_rope_(k, ...).  With `out` (flat buffer) + head_off [3 heads]
    + head_ld [heads] (int64 device tables) nothing changes in place and q, k and v are scattered by head (see
    rmsnorm_rope_scatter)."""
    x2, rows, width, ldx = _rows2d(qkv)
    assert width == 3 * dim
    assert cos.dtype == torch.float32 and cos.is_contiguous() and cos.shape == (rows, head_dim // 2) and sin.shape == cos.shape
    if out is not None:
        assert head_off.dtype == torch.int64 and head_off.numel() == 3 * (dim // head_dim) and head_off.is_contiguous()
        assert head_ld.dtype == torch.int64 and head_ld.numel() == dim // head_dim and out.is_contiguous()
    _lib.check(_lib.lib().fino_qkv_rmsnorm_rope(_p(x2), rows, dim, ldx, _p(q_weight), q_eps, _p(k_weight), k_eps, _p(cos),
                                                _p(sin), head_dim, float(q_out_scale), _p(out), _p(head_off), _p(head_ld),
                                                _dt(qkv), _stream()), "fino_qkv_rmsnorm_rope")
    return qkv if out is None else out


def rmsnorm_rope_scatter(x, weight, eps, cos, sin, head_dim, out, head_off, head_ld, out_scale=1.0):
    """rmsnorm_rope_ out of place: head hd of row r of x [rows, D] (row-strided view) is written to the flat buffer `out`
    at element head_off[hd] + r * head_ld[hd] (int64 device tables).  weight = cos = None: a scattering copy."""
    x2, rows, dim, ldx = _rows2d(x)
    assert head_off.dtype == torch.int64 and head_ld.dtype == torch.int64 and head_off.is_cuda and head_ld.is_cuda
    assert head_off.numel() == dim // head_dim == head_ld.numel() and out.is_contiguous() and out.dtype == x.dtype
    if cos is not None:
        assert cos.dtype == torch.float32 and cos.is_contiguous() and cos.shape == (rows, head_dim // 2)
        assert sin.dtype == torch.float32 and sin.is_contiguous() and sin.shape == cos.shape
    _lib.check(_lib.lib().fino_rmsnorm_rope_scatter(_p(x2), rows, dim, ldx, _p(weight), eps, _p(cos), _p(sin), head_dim,
                                                    float(out_scale), _p(out), _p(head_off), _p(head_ld), _dt(x),
                                                    _stream()), "fino_rmsnorm_rope_scatter")
    return out


def headnorm_rope_(x, heads, head_dim, weight, bias, eps, cos=None, sin=None, rope_row0=0, out_scale=1.0):
    """In place on x [B, rows, heads*head_dim] (row-strided): per-head LayerNorm + RoPE on rows >= rope_row0.
    out_scale: multiplies the result before it is stored (q for attention(scale=SCALE_FOLDED))."""
    assert x.dim() == 3 and x.stride(2) == 1
    b, rows, _ = x.shape
    _lib.check(_lib.lib().fino_headnorm_rope_scaled(_p(x), b, rows, heads, head_dim, x.stride(1), x.stride(0), _p(weight),
                                                   _p(bias), eps, _p(cos), _p(sin), rope_row0, float(out_scale), _dt(x),
                                                   _stream()), "fino_headnorm_rope")
    return x


SPLIT_ATTENTION_TAIL = os.environ.get("FINO_ATTN_SPLIT", "1") != "0"      # A/B knob; the split is the default
_attn_ws = {}      # device index -> fp32 workspace of the attention tail split (caller-owned per the C ABI; grown on demand)


def _attention_workspace(b, heads, lq, lk, dh, device):
    need = _lib.lib().fino_attn_workspace_bytes(b, heads, lq, lk, dh)
    if need <= 0:
        return None, 0
    key = (device.index, torch.cuda.current_stream().cuda_stream)       # concurrent streams must not share partials
    ws = _attn_ws.get(key)
    if ws is None or ws.numel() * 4 < need:
        ws = _attn_ws[key] = torch.empty(need // 4, dtype=torch.float32, device=device)
    return ws, need


def attention(q, k, v, heads, out=None, scale=None):
    """q [B, Lq, H*Dh] (row-strided view), k/v [B, Lk, H*Dh] -> o [B, Lq, H*Dh].  Non-causal SDPA."""
    assert q.dim() == 3 and k.dim() == 3 and v.dim() == 3
    b, lq, hd = q.shape
    lk = k.shape[1]
    dh = hd // heads
    for t in (q, k, v):
        assert t.stride(2) == 1 and t.is_cuda
    if out is None:
        out = torch.empty((b, lq, hd), dtype=q.dtype, device=q.device)
    scale = dh ** -0.5 if scale is None else scale
    ev = _timed("attn_self" if lk > 1024 else "attn_cross")      # same rule as the kernel symbols (VAR 0 / 1)
    ws, ws_bytes = _attention_workspace(b, heads, lq, lk, dh, q.device) if SPLIT_ATTENTION_TAIL else (None, 0)
    _lib.check(_lib.lib().fino_attn_fwd_ws(_p(q), _p(k), _p(v), _p(out), b, heads, lq, lk, dh,
                                          q.stride(0), q.stride(1), dh, k.stride(0), k.stride(1), dh,
                                          v.stride(0), v.stride(1), dh, out.stride(0), out.stride(1), dh,
                                          float(scale), _dt(q), _p(ws), ws_bytes, _stream()), "fino_attn_fwd_ws")
    if ev is not None:
        ev.record()
        kt_ = KernelTimer.active
        nm = "attn_self" if lk > 1024 else "attn_cross"
        kt_.flops[nm] = kt_.flops.get(nm, 0.0) + 4.0 * b * lq * lk * hd      # 4.Lq.Lk.(H.Dh) per batch element
        # algorithmic bytes of the launch: Q read + O written (Lq rows), K and V read (Lk rows), 2 bytes per element
        kt_.bytes[nm] = kt_.bytes.get(nm, 0.0) + 2.0 * b * hd * (2 * lq + 2 * lk)
    return out


def attention_tail_supported(b, heads, lq, lk, dh):
    return bool(_lib.lib().fino_attn_tail_supported(int(b), int(heads), int(lq), int(lk), int(dh)))


def attention_tail(q, k, v, heads, lk_b, tail_mult, out=None, scale=None):
    """attention() over key sequences whose tail is ONE row repeated (a zero-padded prompt): batch element i attends to its first
    lk_b[i] rows of k / v [B, Lk, H*Dh]; the last of them stands for tail_mult[i] identical keys (its logit gets + ln mult), rows
    from lk_b[i] on are ignored.  Equals attention() on the expanded sequences up to the rounding of one weight
    (fino_attn_fwd_tail: the walking kernel; head_dim 128, B <= 4, Lk > 64 -- attention_tail_supported)."""
    import ctypes
    assert q.dim() == 3 and k.dim() == 3 and v.dim() == 3
    b, lq, hd = q.shape
    lk = k.shape[1]
    dh = hd // heads
    for t in (q, k, v):
        assert t.stride(2) == 1 and t.is_cuda
    assert len(lk_b) == b and len(tail_mult) == b
    if out is None:
        out = torch.empty((b, lq, hd), dtype=q.dtype, device=q.device)
    scale = dh ** -0.5 if scale is None else scale
    lk_arr = (ctypes.c_int * b)(*[int(x) for x in lk_b])
    mult_arr = (ctypes.c_float * b)(*[float(x) for x in tail_mult])
    ev = _timed("attn_cross")
    _lib.check(_lib.lib().fino_attn_fwd_tail(_p(q), _p(k), _p(v), _p(out), b, heads, lq, lk, dh,
                                            q.stride(0), q.stride(1), dh, k.stride(0), k.stride(1), dh,
                                            v.stride(0), v.stride(1), dh, out.stride(0), out.stride(1), dh,
                                            float(scale), _dt(q), ctypes.cast(lk_arr, ctypes.c_void_p),
                                            ctypes.cast(mult_arr, ctypes.c_void_p), _stream()), "fino_attn_fwd_tail")
    if ev is not None:
        ev.record()
        kt_ = KernelTimer.active
        kt_.flops["attn_cross"] = kt_.flops.get("attn_cross", 0.0) + 4.0 * lq * hd * sum(int(x) for x in lk_b)
    return out


def attention_probs_supported(b, heads, lq, lk, dh):
    return bool(_lib.lib().fino_attn_probs_supported(int(b), int(heads), int(lq), int(lk), int(dh)))


def row_rrms(x, eps, out=None):
    """rrms[row] = 1 / sqrt(mean(x[row]^2) + eps), fp32 [rows]: the statistic of rmsnorm_rope_ alone, bit for bit (fino_row_rrms)"""
    x2, rows, dim, ldx = _rows2d(x)
    if out is None:
        out = torch.empty(rows, dtype=torch.float32, device=x.device)
    assert out.dtype == torch.float32 and out.is_contiguous() and out.numel() >= rows
    _lib.check(_lib.lib().fino_row_rrms(_p(x2), rows, dim, ldx, float(eps), _p(out), _dt(x), _stream()), "fino_row_rrms")
    return out


def attention_probs(q, k, heads, lk_b, tail_mult, kp, out=None, scale=None, q_rrms=None, q_weight=None):
    """P = softmax(scale q.K^T) per head over a short key sequence: q [B, Lq, H*128], k [B, Lk <= 128, H*128] (row-strided views)
    -> p [B, Lq, H*kp] (kp a multiple of 8 >= every lk_b; columns from lk_b[i] on are zeros).  lk_b / tail_mult as attention_tail.
    The A operand of the re-associated out-projection P.(V W_o^T) (fino_attn_probs).  q_rrms [B, Lq] fp32 + q_weight [H*128]: q is
    the RAW projection, RMS-normalised while it is loaded (rmsnorm_rope_'s arithmetic and rounding points, without its pass)."""
    import ctypes
    assert q.dim() == 3 and k.dim() == 3
    b, lq, hd = q.shape
    lk = k.shape[1]
    dh = hd // heads
    for t in (q, k):
        assert t.stride(2) == 1 and t.is_cuda
    if out is None:
        out = torch.empty((b, lq, heads * kp), dtype=q.dtype, device=q.device)
    assert out.shape == (b, lq, heads * kp) and out.stride(2) == 1
    scale = dh ** -0.5 if scale is None else scale
    lk_arr = (ctypes.c_int * b)(*[int(x) for x in lk_b])
    mult_arr = (ctypes.c_float * b)(*[float(x) for x in tail_mult])
    ev = _timed("attn_cross")
    _lib.check(_lib.lib().fino_attn_probs(_p(q), _p(k), _p(out), b, heads, lq, lk, dh, q.stride(0), q.stride(1), dh,
                                         k.stride(0), k.stride(1), dh, int(kp), out.stride(0), out.stride(1), float(scale),
                                         _dt(q), ctypes.cast(lk_arr, ctypes.c_void_p), ctypes.cast(mult_arr, ctypes.c_void_p),
                                         _p(q_rrms), q_rrms.stride(0) if q_rrms is not None else 0, _p(q_weight), _stream()),
               "fino_attn_probs")
    if ev is not None:
        ev.record()
        kt_ = KernelTimer.active
        kt_.flops["attn_cross"] = kt_.flops.get("attn_cross", 0.0) + 2.0 * lq * hd * sum(int(x) for x in lk_b)
    return out


_attn_fp8_ws = {}     # (device index, stream) -> byte workspace of the quantised K / V images of attention_fp8


# how a softmax weight becomes its e4m3 operand byte (include/frameino_hip.h: FINO_FP8_P_*): exp2 + rounding on the
# transcendental / conversion units, or the byte written directly as rne(8 (s - m) + 55.5) -- exp2 with a piecewise-linear
# mantissa, a third of the vector-instruction time of a kernel that is bound by exactly those instructions
FP8_P_EXP2, FP8_P_RAMP = 0, 1
FP8_P_MODES = {"exp2": FP8_P_EXP2, "ramp": FP8_P_RAMP}
_p_env = os.environ.get("FINO_FP8_P_MODE", "ramp")
if _p_env not in FP8_P_MODES:
    raise ValueError(f"FINO_FP8_P_MODE={_p_env!r}: expected one of {' | '.join(FP8_P_MODES)} (how the opt-in fp8 attention turns a "
                     f"softmax weight into its e4m3 byte; 'ramp' -- a piecewise-linear exp2, the default since round 4 -- adds "
                     f"~0.3e-2 to the 5.5e-2 rel-RMS of the fp8 operands, 'exp2' is the exact form)")
FP8_P_DEFAULT = FP8_P_MODES[_p_env]


def attention_fp8(q, k, v, heads, out=None, scale=None, p_mode=None):
    """attention() with fp8 (e4m3) matrix operands, head_dim 64 or 128: K / V quantised per call (block-scaled, V transposed), Q
    and P in registers, both products on the block-scaled fp8 MFMA, softmax and accumulation in fp32 (fino_attn_fwd_fp8).
    p_mode: "exp2" | "ramp" (or the FP8_P_* integers); None = FP8_P_DEFAULT."""
    p_mode = FP8_P_DEFAULT if p_mode is None else FP8_P_MODES.get(p_mode, p_mode)
    assert q.dim() == 3 and k.dim() == 3 and v.dim() == 3
    b, lq, hd = q.shape
    lk = k.shape[1]
    dh = hd // heads
    for t in (q, k, v):
        assert t.stride(2) == 1 and t.is_cuda
    if out is None:
        out = torch.empty((b, lq, hd), dtype=q.dtype, device=q.device)
    scale = dh ** -0.5 if scale is None else scale
    need = _lib.lib().fino_attn_fp8_kv_bytes(b, heads, lk, dh)
    if need <= 0:
        raise RuntimeError(f"attention_fp8: head_dim {dh} is not supported (64 or 128)")
    key = (q.device.index, torch.cuda.current_stream().cuda_stream)
    ws = _attn_fp8_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = _attn_fp8_ws[key] = torch.empty(need, dtype=torch.uint8, device=q.device)
    ev = _timed("attn_self" if lk > 1024 else "attn_cross")
    _lib.check(_lib.lib().fino_attn_fwd_fp8(_p(q), _p(k), _p(v), _p(out), b, heads, lq, lk, dh, q.stride(0), q.stride(1),
                                           k.stride(0), k.stride(1), v.stride(0), v.stride(1), out.stride(0), out.stride(1),
                                           float(scale), _dt(q), int(p_mode), _p(ws), need, _stream()), "fino_attn_fwd_fp8")
    if ev is not None:
        ev.record()
        kt_ = KernelTimer.active
        nm = "attn_self" if lk > 1024 else "attn_cross"
        kt_.flops[nm] = kt_.flops.get(nm, 0.0) + 4.0 * b * lq * lk * hd
    return out


def attention_partial_floats(batch, heads, lq, head_dim):
    return _lib.lib().fino_attn_partial_bytes(batch, heads, lq, head_dim) // 4


def attention_partial(q, k, v, heads, out=None, scale=None):
    """Attention of q [B, Lq, H*Dh] over ONE key range k/v [B, Lk, H*Dh]: -> fp32 partial buffer (unnormalised O, m, l per
    (head, q-block)) for `attention_merge`."""
    assert q.dim() == 3 and k.dim() == 3 and v.dim() == 3
    b, lq, hd = q.shape
    lk = k.shape[1]
    dh = hd // heads
    for t in (q, k, v):
        assert t.stride(2) == 1 and t.is_cuda
    need = _lib.lib().fino_attn_partial_bytes(b, heads, lq, dh)
    if out is None or out.numel() * 4 < need:
        out = torch.empty(need // 4, dtype=torch.float32, device=q.device)
    scale = dh ** -0.5 if scale is None else scale
    _lib.check(_lib.lib().fino_attn_partial(_p(q), _p(k), _p(v), b, heads, lq, lk, dh, q.stride(0), q.stride(1), dh,
                                           k.stride(0), k.stride(1), dh, v.stride(0), v.stride(1), dh, float(scale),
                                           _dt(q), _p(out), need, _stream()), "fino_attn_partial")
    return out


def attention_merge(parts, batch, lq, heads, head_dim, dtype, out=None):
    """1..3 partials of `attention_partial` (same queries, disjoint key ranges) -> o [B, Lq, H*Dh]."""
    assert 1 <= len(parts) <= 3
    if out is None:
        out = torch.empty((batch, lq, heads * head_dim), dtype=dtype, device=parts[0].device)
    ps = list(parts) + [None] * (3 - len(parts))
    _lib.check(_lib.lib().fino_attn_merge(_p(out), batch, heads, lq, head_dim, out.stride(0), out.stride(1), head_dim,
                                         _p(ps[0]), _p(ps[1]), _p(ps[2]), _dt(out), _stream()), "fino_attn_merge")
    return out


def gemm_plan(m, n):
    """(rows run as 256-row tiles, tile height of the remaining rows or 0) -- fino_gemm's tiling on this device"""
    import ctypes
    r, t = ctypes.c_int64(), ctypes.c_int()
    _lib.check(_lib.lib().fino_gemm_plan(m, n, ctypes.byref(r), ctypes.byref(t)), "fino_gemm_plan")
    return r.value, t.value


def gemm(a, w, bias=None, epilogue=EPI_NONE, residual=None, gate=None, sel=None, out=None, out2=None, split=0, tile_m=0):
    """C = epilogue(A.W^T + bias).  a [M, K] row-strided, w [N, K] (nn.Linear weight).
    `out2`, `split`: columns [split, N) of the product go to out2 [M, N - split], columns [0, split) to out [M, split]
    (fino_gemm_split_n: the fused q | k | v projection of a token shard, k | v landing in the all-gather's send buffer).
    `tile_m`: this call's tile height (0 = planned, 8 = 256-row tiles only, 2 .. 7 = 32 x tile_m rows); results do not
    depend on it."""
    a2, m, k, lda = _rows2d(a)
    assert w.dim() == 2 and w.stride(1) == 1 and w.shape[1] == k and w.dtype == a.dtype
    n = w.shape[0]
    if out is None:
        out = torch.empty((m, split or n), dtype=a.dtype, device=a.device)
    o2, _, _, ldc = _rows2d(out)
    if out2 is not None:
        assert 0 < split < n and residual is None
        c2, _, _, ldc2 = _rows2d(out2)
        ev = _timed("gemm")
        _lib.check(_lib.lib().fino_gemm_split_n(_p(a2), _p(w), _p(bias), _p(o2), m, n, k, lda, w.stride(0), ldc, epilogue,
                                               0, 0, 0, 0, 0, _dt(a), _p(c2), ldc2, split, tile_m, _stream()),
                   "fino_gemm_split_n")
        if ev is not None:
            ev.record()
            KernelTimer.active.flops["gemm"] = KernelTimer.active.flops.get("gemm", 0.0) + 2.0 * m * n * k
        return out, out2
    r2, ldr = (None, 0)
    if residual is not None:
        r2, _, _, ldr = _rows2d(residual)
    ms = gate.stride(0) if (gate is not None and gate.dim() == 2) else 0
    if bias is not None:
        assert bias.dtype == a.dtype and bias.is_contiguous()
    # algorithmic bytes: A [M, K] and W [N, K] read, C [M, N] written (+ the residual read), 2 bytes per element
    done = _timed_gemm(epilogue, 2.0 * m * n * k, 2.0 * (m * k + n * k + m * n * (2 if residual is not None else 1)))
    if tile_m:
        _lib.check(_lib.lib().fino_gemm_split_n(_p(a2), _p(w), _p(bias), _p(o2), m, n, k, lda, w.stride(0), ldc, epilogue,
                                               _p(r2), ldr, _p(gate), ms, _p(sel), _dt(a), None, 0, 0, tile_m, _stream()),
                   "fino_gemm_split_n")
    else:
        _lib.check(_lib.lib().fino_gemm(_p(a2), _p(w), _p(bias), _p(o2), m, n, k, lda, w.stride(0), ldc, epilogue, _p(r2),
                                       ldr, _p(gate), ms, _p(sel), _dt(a), _stream()), "fino_gemm")
    if done is not None:
        done()
    return out


def gemm_blocked_a(a_blocks, rows, w, bias, residual, gate, sel, out, tile_m=0):
    """out = residual + gate[sel] * (A.W^T + bias) with A given as K blocks: a_blocks [nblk, rows_pad, blk_k], or
    [groups, peers, rows_pad, blk_k] with K block j * groups + g at a_blocks[g, j] (last dims row-strided):
    A[i, b * blk_k + c] = block b [i, c] for i < rows.  fino_gemm_blocked_a: the out-projection reading the heads exchange's
    return buffers as they arrived."""
    if a_blocks.dim() == 3:
        a_blocks = a_blocks[None]
    assert a_blocks.dim() == 4 and a_blocks.stride(3) == 1 and w.dtype == a_blocks.dtype and w.stride(1) == 1
    groups, peers, _, bk = a_blocks.shape
    n, k = w.shape
    assert k == groups * peers * bk and residual.shape == (rows, n) and out.shape == (rows, n)
    r2, _, _, ldr = _rows2d(residual)
    o2, _, _, ldc = _rows2d(out)
    ms = gate.stride(0) if gate.dim() == 2 else 0
    ev = _timed("gemm")
    _lib.check(_lib.lib().fino_gemm_blocked_a(_p(a_blocks), _p(w), _p(bias), _p(o2), rows, n, k, bk, a_blocks.stride(1),
                                              groups, a_blocks.stride(0), a_blocks.stride(2), w.stride(0), ldc, _p(r2),
                                              ldr, _p(gate), ms, _p(sel), _dt(a_blocks), tile_m, _stream()),
               "fino_gemm_blocked_a")
    if ev is not None:
        ev.record()
        KernelTimer.active.flops["gemm"] = KernelTimer.active.flops.get("gemm", 0.0) + 2.0 * rows * n * k
    return out


def quantize_mxfp8(x, out=None):
    """x [rows, cols] bf16|fp16 (cols % 128 == 0) -> (q uint8 [rows, cols], scales uint8 [...]) in the layout
    fino_gemm_mxfp8 consumes (OCP e4m3 elements, one e8m0 scale per 32 K-elements)."""
    x2, rows, cols, ldx = _rows2d(x)
    nbytes = _lib.lib().fino_mxfp8_scale_bytes(rows, cols)
    if nbytes <= 0:
        raise ValueError(f"quantize_mxfp8: cols={cols} must be a positive multiple of 128")
    if out is None:
        q = torch.empty((rows, cols), dtype=torch.uint8, device=x.device)
        s = torch.zeros(nbytes, dtype=torch.uint8, device=x.device)
    else:
        q, s = out
    _lib.check(_lib.lib().fino_quantize_mxfp8(_p(x2), _p(q), _p(s), rows, cols, ldx, _dt(x), _stream()),
               "fino_quantize_mxfp8")
    return q, s


def ln_mxfp8(mode, x, weight=None, bias=None, shift=None, scale=None, sel=None, eps=1e-6, out=None):
    """adaln_modulate (mode 0) / layernorm (1) / layernorm_zero (2) with the result as MXFP8 activations: -> (q, scales) as
    quantize_mxfp8 would make them from the T-rounded output, in one pass (fino_ln_mxfp8)."""
    x2, rows, dim, ldx = _rows2d(x)
    nbytes = _lib.lib().fino_mxfp8_scale_bytes(rows, dim)
    if nbytes <= 0:
        raise ValueError(f"ln_mxfp8: dim={dim} must be a positive multiple of 128")
    if out is None:
        q = torch.empty((rows, dim), dtype=torch.uint8, device=x.device)
        s = torch.zeros(nbytes, dtype=torch.uint8, device=x.device)
    else:
        q, s = out
    for t in (weight, bias, shift, scale):
        assert t is None or t.dtype == torch.float32
    ms = shift.stride(0) if (shift is not None and shift.dim() == 2) else 0
    _lib.check(_lib.lib().fino_ln_mxfp8(mode, _p(x2), _p(q), _p(s), rows, dim, ldx, _p(weight), _p(bias), _p(shift),
                                       _p(scale), ms, _p(sel), eps, _dt(x), _stream()), "fino_ln_mxfp8")
    return q, s


def gemm_mxfp8(aq, a_scales, wq, w_scales, bias=None, epilogue=EPI_NONE, residual=None, gate=None, sel=None, out=None,
               out_dtype=torch.bfloat16):
    """C = epilogue(dequant(aq).dequant(wq)^T + bias): aq [M, K], wq [N, K] uint8 (e4m3) + their MX scales."""
    m, k = aq.shape
    n = wq.shape[0]
    assert wq.shape[1] == k and aq.dtype == torch.uint8 and wq.dtype == torch.uint8 and aq.is_contiguous() \
        and wq.is_contiguous()
    if out is None:
        out = torch.empty((m, n), dtype=out_dtype, device=aq.device)
    o2, _, _, ldc = _rows2d(out)
    r2, ldr = (None, 0)
    if residual is not None:
        r2, _, _, ldr = _rows2d(residual)
    ms = gate.stride(0) if (gate is not None and gate.dim() == 2) else 0
    ev = _timed("gemm")
    _lib.check(_lib.lib().fino_gemm_mxfp8(_p(aq), _p(a_scales), _p(wq), _p(w_scales), _p(bias), _p(o2), m, n, k, ldc,
                                         epilogue, _p(r2), ldr, _p(gate), ms, _p(sel), _dt(out), _stream()),
               "fino_gemm_mxfp8")
    if ev is not None:
        ev.record()
        KernelTimer.active.flops["gemm"] = KernelTimer.active.flops.get("gemm", 0.0) + 2.0 * m * n * k
    return out


def gemm_mxfp8_q(aq, a_scales, wq, w_scales, bias, epilogue=EPI_NONE, out=None):
    """MXFP8 GEMM whose result leaves quantised: -> (q uint8 [M, N], scales) ready to be the next MXFP8 GEMM's A."""
    m, k = aq.shape
    n = wq.shape[0]
    assert bias is not None and wq.shape[1] == k
    if out is None:
        q = torch.empty((m, n), dtype=torch.uint8, device=aq.device)
        s = torch.zeros(_lib.lib().fino_mxfp8_scale_bytes(m, n), dtype=torch.uint8, device=aq.device)
    else:
        q, s = out
    ev = _timed("gemm")
    _lib.check(_lib.lib().fino_gemm_mxfp8_q(_p(aq), _p(a_scales), _p(wq), _p(w_scales), _p(bias), _p(q), _p(s), m, n, k,
                                           epilogue, _dt(bias), _stream()), "fino_gemm_mxfp8_q")
    if ev is not None:
        ev.record()
        KernelTimer.active.flops["gemm"] = KernelTimer.active.flops.get("gemm", 0.0) + 2.0 * m * n * k
    return q, s


def skinny_linear(x, w, b=None, silu_input=False):
    """fp32 y = W.(silu?)(x) + b for M <= 16 rows; w fp32 or bf16/fp16."""
    assert x.dtype == torch.float32 and x.dim() == 2 and x.is_contiguous() and w.is_contiguous()
    m, k = x.shape
    n = w.shape[0]
    wd = -1 if w.dtype == torch.float32 else _dt(w)
    if b is not None:
        assert b.dtype == w.dtype
    y = torch.empty((m, n), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().fino_skinny_linear(_p(x), _p(w), _p(b), _p(y), m, n, k, wd, int(silu_input), _stream()),
               "fino_skinny_linear")
    return y


def patchify(x, patch, out=None):
    """x [C, F, H, W] -> a [L, C*pt*ph*pw]"""
    assert x.dim() == 4 and x.is_contiguous()
    c, f, h, w = x.shape
    pt, ph, pw = patch
    L = (f // pt) * (h // ph) * (w // pw)
    if out is None:
        out = torch.empty((L, c * pt * ph * pw), dtype=x.dtype, device=x.device)
    _lib.check(_lib.lib().fino_patchify(_p(x), _p(out), c, f, h, w, pt, ph, pw, out.stride(0), _dt(x), _stream()),
               "fino_patchify")
    return out


def unpatchify(y, cout, frames, height, width, patch, out=None):
    """y [L, pt*ph*pw*cout] -> [cout, F, H, W]"""
    pt, ph, pw = patch
    if out is None:
        out = torch.empty((cout, frames, height, width), dtype=y.dtype, device=y.device)
    _lib.check(_lib.lib().fino_unpatchify(_p(y), _p(out), cout, frames, height, width, pt, ph, pw, y.stride(0), _dt(y),
                                         _stream()), "fino_unpatchify")
    return out


def wan_model_input(lat, cond, id_lat, traj, dtype, out=None):
    """lat [C,Fg,H,W], cond [C,1,H,W], id_lat [C,n_id,H,W]|None, traj [C,Fg+n_id,H,W] (fp32) -> [2C,Fg+n_id,H,W]."""
    c, fg, h, w = lat.shape
    nid = 0 if id_lat is None else id_lat.shape[1]
    for t in (lat, cond, id_lat, traj):
        assert t is None or (t.dtype == torch.float32 and t.is_contiguous())
    assert traj.shape == (c, fg + nid, h, w) and cond.shape == (c, 1, h, w)
    if out is None:
        out = torch.empty((2 * c, fg + nid, h, w), dtype=dtype, device=lat.device)
    _lib.check(_lib.lib().fino_wan_model_input(_p(lat), _p(cond), _p(id_lat), _p(traj), _p(out), c, fg, nid, h, w,
                                              _dt(out), _stream()), "fino_wan_model_input")
    return out


def cfg_euler_step_(lat, cond_pred, uncond_pred, guidance, dt_dev, round_out=True):
    """In place on lat [C,Fg,H,W] fp32; preds [C,Ft,H,W] of the model dtype; dt_dev: 1-element fp32 device tensor."""
    c, fg, h, w = lat.shape
    ft = cond_pred.shape[1]
    assert lat.dtype == torch.float32 and lat.is_contiguous() and cond_pred.is_contiguous()
    assert dt_dev.dtype == torch.float32 and dt_dev.is_cuda
    _lib.check(_lib.lib().fino_cfg_euler_step(_p(cond_pred), _p(uncond_pred), _p(lat), c, fg, ft, h, w, float(guidance),
                                             _p(dt_dev), int(round_out), _dt(cond_pred), _stream()),
               "fino_cfg_euler_step")
    return lat


def cfg_unipc_step_(x, last, m0, m1, cond_pred, uncond_pred, coef_dev):
    """UniPC multistep update fused with CFG; x/last/m0/m1 fp32 [C,Fg,H,W] in place, coef_dev fp32[10] (device)."""
    c, fg, h, w = x.shape
    ft = cond_pred.shape[1]
    for t in (x, last, m0, m1):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.shape == x.shape
    assert coef_dev.dtype == torch.float32 and coef_dev.numel() >= 10 and cond_pred.is_contiguous()
    _lib.check(_lib.lib().fino_cfg_unipc_step(_p(cond_pred), _p(uncond_pred), _p(x), _p(last), _p(m0), _p(m1), c, fg,
                                             ft, h, w, _p(coef_dev), _dt(cond_pred), _stream()), "fino_cfg_unipc_step")
    return x


def cfg_vpred_step_(lat, pred, coef_dev, has_uncond=True):
    """lat [Fg, C, H, W] of T in place; pred [2|1, Ft, C, H, W] of T; coef_dev fp32[5] = {sa, sb, ca, cb, g}."""
    assert lat.is_contiguous() and pred.is_contiguous() and coef_dev.dtype == torch.float32 and lat.dtype == pred.dtype
    _lib.check(_lib.lib().fino_cfg_vpred_step(_p(pred), _p(lat), lat.numel(), pred[0].numel(), _p(coef_dev),
                                             int(has_uncond), _dt(lat), _stream()), "fino_cfg_vpred_step")
    return lat


def cfg_dpm_step_(lat, pred, x0_old, noise, coef_dev, has_uncond=True):
    """CogVideoXDPMScheduler step + CFG: lat [Fg, C, H, W] of T in place, pred [2|1, Ft, C, H, W] of T, x0_old fp32 like
    lat (in/out), noise of T like lat, coef_dev fp32[9] = {sa, sb, m1, m2, m3, m4, mn, g, use_old}."""
    assert lat.is_contiguous() and pred.is_contiguous() and noise.is_contiguous() and x0_old.is_contiguous()
    assert coef_dev.dtype == torch.float32 and coef_dev.numel() >= 9 and lat.dtype == pred.dtype == noise.dtype
    assert x0_old.dtype == torch.float32 and x0_old.numel() == lat.numel() == noise.numel()
    _lib.check(_lib.lib().fino_cfg_dpm_step(_p(pred), _p(lat), _p(x0_old), _p(noise), lat.numel(), pred[0].numel(),
                                           _p(coef_dev), int(has_uncond), _dt(lat), _stream()), "fino_cfg_dpm_step")
    return lat


# ------------------------------------------------------------------------------------------------ Wan VAE (channels-last)
_ZERO_PAGE = {}


def zero_page(device):
    z = _ZERO_PAGE.get(str(device))
    if z is None:
        z = _ZERO_PAGE[str(device)] = torch.zeros(256, dtype=torch.uint8, device=device)
    return z


def conv3d_cl(x, w2, bias, kernel, stride=(1, 1, 1), pad=(0, 0, 0), out_thw=None, upsample2x=False, residual=None,
              out=None):
    """x [T,H,W,Cin_pad] channels-last; w2 [Cout_pad, kt*kh*kw*Cin_pad] (tap-major); pad = FRONT pads (t,h,w)."""
    assert x.dim() == 4 and x.is_contiguous() and w2.is_contiguous()
    t, h, w, cin = x.shape
    kt, kh, kw = kernel
    cout = w2.shape[0]
    assert w2.shape[1] == kt * kh * kw * cin, (w2.shape, kernel, cin)
    if out_thw is None:
        up = 2 if upsample2x else 1
        out_thw = ((t + pad[0] - kt) // stride[0] + 1, (h * up + 2 * pad[1] - kh) // stride[1] + 1,
                   (w * up + 2 * pad[2] - kw) // stride[2] + 1)
    to, ho, wo = out_thw
    if out is None:
        out = torch.empty((to, ho, wo, cout), dtype=x.dtype, device=x.device)
    ev = _timed("conv3d")
    _lib.check(_lib.lib().fino_conv3d(_p(x), _p(w2), _p(bias), _p(out), t, h, w, cin, to, ho, wo, cout, kt, kh, kw,
                                     stride[0], stride[1], stride[2], pad[0], pad[1], pad[2], int(upsample2x),
                                     EPI_RESIDUAL if residual is not None else EPI_NONE, _p(residual),
                                     _p(zero_page(x.device)), _dt(x), _stream()), "fino_conv3d")
    if ev is not None:
        ev.record()
        kt_ = KernelTimer.active
        kt_.flops["conv3d"] = kt_.flops.get("conv3d", 0.0) + 2.0 * to * ho * wo * cout * w2.shape[1]
    return out


def rmsnorm_silu_cl(x, gamma, c_valid, silu=True, out=None):
    assert x.is_contiguous() and gamma.dtype == torch.float32
    cpad = x.shape[-1]
    rows = x.numel() // cpad
    out = torch.empty_like(x) if out is None else out
    _lib.check(_lib.lib().fino_rmsnorm_silu_cl(_p(x), _p(out), rows, c_valid, cpad, _p(gamma), int(silu), _dt(x),
                                              _stream()), "fino_rmsnorm_silu_cl")
    return out


def softmax_rows_(s, n, scale):
    assert s.dim() == 2 and s.stride(1) == 1
    _lib.check(_lib.lib().fino_softmax_rows(_p(s), s.shape[0], n, s.stride(0), float(scale), _dt3(s), _stream()),
               "fino_softmax_rows")
    return s


def dup_up3d_add(main, x, c_in, c_out, factor_t, factor_s):
    t, h, w, cin_pad = x.shape
    out = torch.empty_like(main)
    assert main.shape == (1 + (t - 1) * factor_t, h * factor_s, w * factor_s, main.shape[3])
    _lib.check(_lib.lib().fino_dup_up3d_add(_p(main), _p(x), _p(out), t, h, w, c_in, cin_pad, c_out, main.shape[3],
                                           factor_t, factor_s, _dt3(x), _stream()), "fino_dup_up3d_add")
    return out


def avg_down3d_add(main, x, c_in, c_out, factor_t, factor_s):
    t, h, w, cin_pad = x.shape
    out = torch.empty_like(main)
    _lib.check(_lib.lib().fino_avg_down3d_add(_p(main), _p(x), _p(out), t, h, w, c_in, cin_pad, c_out, main.shape[3],
                                             factor_t, factor_s, _dt3(x), _stream()), "fino_avg_down3d_add")
    return out


def vae_unpatchify_clamp(y, channels, patch):
    t, h, w, cpad = y.shape
    out = torch.empty((channels, t, h * patch, w * patch), dtype=torch.float32, device=y.device)
    _lib.check(_lib.lib().fino_vae_unpatchify_clamp(_p(y), _p(out), t, h, w, cpad, channels, patch, _dt3(y), _stream()),
               "fino_vae_unpatchify_clamp")
    return out


def vae_patchify(x, c_pad, patch, dtype):
    c, t, hp, wp = x.shape
    assert x.dtype == torch.float32 and x.is_contiguous()
    y = torch.empty((t, hp // patch, wp // patch, c_pad), dtype=dtype, device=x.device)
    _lib.check(_lib.lib().fino_vae_patchify(_p(x), _p(y), t, hp // patch, wp // patch, c_pad, c, patch, _dt3(y),
                                           _stream()), "fino_vae_patchify")
    return y


# ---- the Wan VAE computing like fp32 on the bf16 matrix pipe: split-bf16 products (include/frameino_hip.h, FINO_VERSION 103) ----
# an fp32 value = hi + mid + lo (three bf16 planes); a product keeps the terms >= 2^-16 relative
SPLIT_PRODUCTS = {2: ((0, 0), (0, 1), (1, 0)),                                    # (A plane, W plane) per product, in K order
                  3: ((0, 0), (0, 1), (0, 2), (1, 0), (1, 1), (2, 0))}


def _pack_planes(planes):
    v = 0
    for i, q in enumerate(planes):
        v |= int(q) << (2 * i)
    return v


def split_bf16(x, layout, nplanes=3, out=None):
    """x fp32 [..., C] (C % 8 == 0) -> bf16 [..., nseg * C] of split planes:
      layout "planes": the planes side by side (the A operand of conv3d_split_cl), "A" / "W": one plane per product of
      SPLIT_PRODUCTS[nplanes] -- the operands of gemm(..., EPI_F32) -- A side / W side."""
    assert x.dtype == torch.float32 and x.stride(-1) == 1
    x2 = x if x.dim() == 2 else x.reshape(-1, x.shape[-1])
    rows, cols = x2.shape
    prod = SPLIT_PRODUCTS[nplanes]
    planes = tuple(range(nplanes)) if layout == "planes" else tuple(p[0 if layout == "A" else 1] for p in prod)
    nseg = len(planes)
    if out is None:
        out = torch.empty(tuple(x.shape[:-1]) + (nseg * cols,), dtype=torch.bfloat16, device=x.device)
    assert out.is_contiguous() and out.numel() == rows * nseg * cols
    _lib.check(_lib.lib().fino_split_bf16(_p(x2), _p(out), rows, cols, x2.stride(0), nseg * cols, nseg, _pack_planes(planes),
                                         _stream()), "fino_split_bf16")
    return out


def rmsnorm_silu_cl_f32(x, gamma, c_valid, silu=True, split=None, nplanes=3):
    """WanRMS_norm [+ SiLU] on fp32 channels-last activations, in fp32; split=None -> fp32 out, "planes" | "A" | "W" -> the split
    planes of split_bf16 in the same pass"""
    assert x.is_contiguous() and x.dtype == torch.float32 and gamma.dtype == torch.float32
    cpad = x.shape[-1]
    rows = x.numel() // cpad
    if split is None:
        planes, out = (), torch.empty_like(x)
    else:
        prod = SPLIT_PRODUCTS[nplanes]
        planes = tuple(range(nplanes)) if split == "planes" else tuple(p[0 if split == "A" else 1] for p in prod)
        out = torch.empty(tuple(x.shape[:-1]) + (len(planes) * cpad,), dtype=torch.bfloat16, device=x.device)
    _lib.check(_lib.lib().fino_rmsnorm_silu_cl_f32(_p(x), _p(out), rows, c_valid, cpad, _p(gamma), int(silu), len(planes),
                                                  _pack_planes(planes), _stream()), "fino_rmsnorm_silu_cl_f32")
    return out


def conv3d_split_cl(xp, w6, bias, kernel, nplanes, stride=(1, 1, 1), pad=(0, 0, 0), out_thw=None, upsample2x=False,
                    residual=None):
    """conv3d_cl on split operands: xp bf16 [T, H, W, nplanes * Cin_pad] (split_bf16 layout "planes"), w6 bf16 [Cout_pad,
    taps * products * Cin_pad] (W planes of the products per tap), bias fp32 [Cout_pad], residual fp32 -> fp32 [To, Ho, Wo, Cout_pad]"""
    assert xp.dim() == 4 and xp.is_contiguous() and w6.is_contiguous() and xp.dtype == torch.bfloat16 and w6.dtype == torch.bfloat16
    t, h, w, cw = xp.shape
    cin = cw // nplanes
    kt, kh, kw = kernel
    cout = w6.shape[0]
    nprod = len(SPLIT_PRODUCTS[nplanes])
    assert cin * nplanes == cw and w6.shape[1] == kt * kh * kw * nprod * cin, (xp.shape, w6.shape, kernel)
    assert bias is None or bias.dtype == torch.float32
    if out_thw is None:
        up = 2 if upsample2x else 1
        out_thw = ((t + pad[0] - kt) // stride[0] + 1, (h * up + 2 * pad[1] - kh) // stride[1] + 1,
                   (w * up + 2 * pad[2] - kw) // stride[2] + 1)
    to, ho, wo = out_thw
    out = torch.empty((to, ho, wo, cout), dtype=torch.float32, device=xp.device)
    if residual is not None:
        assert residual.dtype == torch.float32 and residual.is_contiguous() and residual.shape == out.shape
    ev = _timed("conv3d")
    _lib.check(_lib.lib().fino_conv3d_split(_p(xp), _p(w6), _p(bias), _p(out), t, h, w, cin, nplanes, to, ho, wo, cout, kt, kh, kw,
                                           stride[0], stride[1], stride[2], pad[0], pad[1], pad[2], int(upsample2x),
                                           EPI_RESIDUAL if residual is not None else EPI_NONE, _p(residual),
                                           _p(zero_page(xp.device)), _stream()), "fino_conv3d_split")
    if ev is not None:
        ev.record()
        kt_ = KernelTimer.active
        kt_.flops["conv3d"] = kt_.flops.get("conv3d", 0.0) + 2.0 * to * ho * wo * cout * w6.shape[1]
    return out


def gemm_f32(a_split, w_split, bias=None, residual=None, out=None):
    """fp32 C = A.W^T + bias [+ R] from operands already expanded by split_bf16 ("A" / "W" layouts: one plane per product):
    a_split [M, products * K], w_split [N, products * K] bf16; bias fp32 [N], residual / out fp32 [M, N]"""
    a2, m, k, lda = _rows2d(a_split)
    assert w_split.dim() == 2 and w_split.stride(1) == 1 and w_split.shape[1] == k and a_split.dtype == w_split.dtype == torch.bfloat16
    n = w_split.shape[0]
    if out is None:
        out = torch.empty((m, n), dtype=torch.float32, device=a_split.device)
    assert out.dtype == torch.float32 and out.stride(-1) == 1
    o2 = out if out.dim() == 2 else out.view(-1, out.shape[-1])
    r2, ldr = None, 0
    if residual is not None:
        assert residual.dtype == torch.float32 and residual.stride(-1) == 1
        r2 = residual if residual.dim() == 2 else residual.view(-1, residual.shape[-1])
        ldr = r2.stride(0)
    assert bias is None or (bias.dtype == torch.float32 and bias.is_contiguous())
    _lib.check(_lib.lib().fino_gemm(_p(a2), _p(w_split), _p(bias), _p(o2), m, n, k, lda, w_split.stride(0), o2.stride(0),
                                   EPI_F32_RESIDUAL if residual is not None else EPI_F32, _p(r2), ldr, None, 0, None, BF16,
                                   _stream()), "fino_gemm")
    return out


# ------------------------------------------------------------------------------------------------ CogVideoX VAE
def vae_blend_tiles_(a, b, extent, axis):
    """diffusers' blend_v (axis 0) / blend_h (axis 1) in place on the channels-last tile b [T, Hb, Wb, Cpad] from its upper / left
    neighbour a (fino_vae_blend_tiles)"""
    assert a.dim() == 4 and b.dim() == 4 and a.is_contiguous() and b.is_contiguous() and a.dtype == b.dtype
    assert a.shape[0] == b.shape[0] and a.shape[3] == b.shape[3]
    _lib.check(_lib.lib().fino_vae_blend_tiles(_p(a), _p(b), a.shape[0], a.shape[1], a.shape[2], b.shape[1], b.shape[2],
                                              a.shape[3], int(extent), int(axis), _dt(a), _stream()), "fino_vae_blend_tiles")
    return b


_GN_WS = {}


def groupnorm_cl(x, c_valid, groups, gamma, beta, eps=1e-6, mod=None, silu=False, out=None):
    """x [T, H, W, Cpad] channels-last -> T(GroupNorm(x)) [ * scale[z] + shift[z] ] [ -> silu ].  mod = (scale, shift),
    each [Tz, hz, wz, Cpad] of x's dtype (CogVideoXSpatialNorm3D's conv_y / conv_b at latent resolution)."""
    assert x.dim() == 4 and x.is_contiguous() and gamma.dtype == torch.float32 and beta.dtype == torch.float32
    t, h, w, cp = x.shape
    out = torch.empty_like(x) if out is None else out
    need = _lib.lib().fino_groupnorm_workspace_bytes(cp)
    key = (x.device.index, torch.cuda.current_stream().cuda_stream, cp)
    ws = _GN_WS.get(key)
    if ws is None:
        ws = _GN_WS[key] = torch.empty(need // 4, dtype=torch.float32, device=x.device)
    ms = mh = None
    tz = hz = wz = 0
    if mod is not None:
        ms, mh = mod
        assert ms.shape == mh.shape and ms.shape[3] == cp and ms.is_contiguous() and mh.is_contiguous() \
            and ms.dtype == x.dtype
        tz, hz, wz = ms.shape[:3]
    _lib.check(_lib.lib().fino_groupnorm_cl(_p(x), _p(out), t, h, w, c_valid, cp, groups, _p(gamma), _p(beta), eps,
                                           _p(ms), _p(mh), tz, hz, wz, int(silu), _p(ws), need, _dt(x), _stream()),
               "fino_groupnorm_cl")
    return out


def avg_pool_time2(x):
    t, h, w, cp = x.shape
    assert x.is_contiguous() and t > 1
    out = torch.empty(((t & 1) + (t - (t & 1)) // 2, h, w, cp), dtype=x.dtype, device=x.device)
    _lib.check(_lib.lib().fino_avg_pool_time2(_p(x), _p(out), t, h, w, cp, _dt(x), _stream()), "fino_avg_pool_time2")
    return out
