"""Seeded random-init instances of the two DiT mirrors, generated ON the device (no checkpoints offline; a 5B-parameter
model initialised on the host and copied would take minutes).  Used by bench.py, tools/ and the full-size parity tests."""
import torch


def random_wan_model(cfg, device, seed=0, dtype=torch.bfloat16):
    """Wan2.2-5B-shaped `WanTransformer3DModel`: N(0, 0.02^2) weights, norm gains ~1, the reference's fp32 islands
    (architecture/transformer_wan.py:393) kept fp32."""
    from .transformer_wan import WanTransformer3DModel
    torch.manual_seed(seed)
    with torch.device("meta"):
        m = WanTransformer3DModel(**cfg)
    m = m.to_empty(device=device)
    g = torch.Generator(device=device).manual_seed(seed)
    keep = WanTransformer3DModel._keep_in_fp32_modules
    with torch.no_grad():
        for name, p in m.named_parameters():
            if name.endswith("norm_q.weight") or name.endswith("norm_k.weight") or name.endswith("norm2.weight"):
                t = 1.0 + 0.05 * torch.randn(p.shape, generator=g, device=device)
            elif "scale_shift_table" in name:
                t = torch.randn(p.shape, generator=g, device=device) / p.shape[-1] ** 0.5
            else:
                t = 0.02 * torch.randn(p.shape, generator=g, device=device)
            p.data = t.to(torch.float32 if any(k in name for k in keep) else dtype)
    m.reset_caches()
    return m.eval()


def random_cog_model(cfg, device, seed=0, dtype=torch.bfloat16):
    """CogVideoX-5B-shaped `CogVideoXTransformer3DModel` (all parameters and the learned positional table in `dtype`:
    the reference has no fp32 islands on this backbone)."""
    from .cogvideox_transformer_3d import CogVideoXTransformer3DModel
    with torch.device("meta"):
        m = CogVideoXTransformer3DModel(**cfg)
    m = m.to_empty(device=device)
    g = torch.Generator(device=device).manual_seed(seed)
    with torch.no_grad():
        for name, p in m.named_parameters():
            t = 0.02 * torch.randn(p.shape, generator=g, device=device)
            if name.endswith("norm.weight") or "norm_q.weight" in name or "norm_k.weight" in name \
                    or name.startswith("norm_final.weight"):
                t = 1.0 + t
            p.data = t.to(dtype)
        for name, b in m.named_buffers():
            b.data = (0.02 * torch.randn(b.shape, generator=g, device=device)).to(dtype if b.is_floating_point() else b.dtype)
    m.reset_caches()
    return m.eval()
