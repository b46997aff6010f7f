"""INTEGRATION.md section 2's claim, on the reference's OWN classes: `MI355WanAttnProcessor` / `MI355CogVideoXAttnProcessor`
installed with `set_processor` on the reference's vendored `Attention` (architecture/attention_processor.py:50-820, the
class diffusers ships) are called through its `forward` (kwarg filtering by signature, :556-600) and return what
`WanAttnProcessor2_0` / `CogVideoXAttnProcessor2_0` return.

Runs only in the build container (needs /root/reference; diffusers itself is absent, so the import goes through the
stand-in under tools/golden/diffusers_stub -- utilities only: the Attention container is the reference's).  No GPU here:
the processors' kernel front end (`frameino_amd.ops`) is swapped for tests/cpu_ops.py, so what is checked is the
PROTOCOL -- attribute names read off `attn`, argument passing, output shapes / tuple order -- and the processor body's
arithmetic structure; the kernels themselves are the -m gpu tests' business."""
import os
import sys

import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference tree (build container only)")


@pytest.fixture()
def ref_modules(monkeypatch):
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    monkeypatch.syspath_prepend(REF)
    monkeypatch.syspath_prepend(os.path.join(here, "tools", "golden", "diffusers_stub"))
    monkeypatch.chdir(REF)
    from tests import cpu_ops
    import frameino_amd.attention_processor as ap
    monkeypatch.setattr(ap, "ops", cpu_ops)
    cpu_ops.headnorm_rope_ = _headnorm_rope_
    import importlib
    mods = {n: importlib.import_module(n) for n in ("architecture.attention_processor", "architecture.transformer_wan",
                                                    "architecture.embeddings")}
    yield ap, mods
    for n in list(sys.modules):
        if n.startswith(("architecture", "diffusers", "pipelines")):
            sys.modules.pop(n, None)


def _headnorm_rope_(x, heads, head_dim, weight, bias, eps, cos=None, sin=None, rope_row0=0):
    """CPU contract of fino_headnorm_rope: per-head LayerNorm + adjacent-pair RoPE on rows >= rope_row0, in place."""
    import torch.nn.functional as F
    b, rows, _ = x.shape
    y = x.reshape(b, rows, heads, head_dim).float()
    if weight is not None:
        y = F.layer_norm(y, (head_dim,), weight.float(), bias.float(), eps)
    if cos is not None:
        v = y[:, rope_row0:]
        xr, xi = v.reshape(*v.shape[:-1], -1, 2).unbind(-1)
        rot = torch.stack([-xi, xr], dim=-1).flatten(3)
        y[:, rope_row0:] = v * cos[None, :, None, :] + rot * sin[None, :, None, :]
    x.copy_(y.reshape(b, rows, heads * head_dim).to(x.dtype))
    return x


def test_wan_processor_on_the_references_attention(ref_modules):
    ap, mods = ref_modules
    RefAttention = mods["architecture.attention_processor"].Attention
    tw = mods["architecture.transformer_wan"]
    torch.manual_seed(0)
    heads, dh = 2, 24
    d = heads * dh
    for cross in (False, True):
        attn = RefAttention(query_dim=d, heads=heads, kv_heads=heads, dim_head=dh, qk_norm="rms_norm_across_heads",
                            eps=1e-6, bias=True, cross_attention_dim=None, out_bias=True,
                            processor=tw.WanAttnProcessor2_0()).eval()
        with torch.no_grad():
            for p in attn.parameters():
                p.copy_(torch.randn_like(p) * (0.2 if p.ndim > 1 else 0.1) + (1.0 if p.ndim == 1 and p.numel() == d and
                                                                              "norm" in str(p.shape) else 0.0))
        x = torch.randn(2, 3 * 2 * 4, d)
        ctx = torch.randn(2, 7, d) if cross else None
        rot = None if cross else tw.WanRotaryPosEmbed(dh, (1, 2, 2), 64)(torch.zeros(1, 1, 3, 4, 8))
        with torch.no_grad():
            ref = attn(x, encoder_hidden_states=ctx, rotary_emb=rot)
            assert isinstance(attn.get_processor(), tw.WanAttnProcessor2_0)
            attn.set_processor(ap.MI355WanAttnProcessor())                  # the plugin call a maintainer makes
            assert isinstance(attn.get_processor(), ap.MI355WanAttnProcessor)
            out = attn(x, encoder_hidden_states=ctx, rotary_emb=rot, some_unknown_kwarg=1)     # filtered by signature
        torch.testing.assert_close(out, ref, atol=2e-5, rtol=2e-5)


def test_cogvideox_processors_on_the_references_attention(ref_modules):
    ap, mods = ref_modules
    A = mods["architecture.attention_processor"]
    emb = mods["architecture.embeddings"]
    torch.manual_seed(1)
    heads, dh = 2, 16
    d = heads * dh
    attn = A.Attention(query_dim=d, dim_head=dh, heads=heads, qk_norm="layer_norm", eps=1e-6, bias=True, out_bias=True,
                       processor=A.CogVideoXAttnProcessor2_0()).eval()
    with torch.no_grad():
        for p in attn.parameters():
            p.copy_(torch.randn_like(p) * 0.2 + (1.0 if p.ndim == 1 and p.numel() == dh else 0.0))
    txt, vid = torch.randn(2, 5, d), torch.randn(2, 3 * 4 * 4, d)
    cos, sin = emb.get_3d_rotary_pos_embed(dh, ((0, 0), (4, 4)), (4, 4), 3)
    with torch.no_grad():
        rh, re = attn(vid, encoder_hidden_states=txt, image_rotary_emb=(cos, sin))
        attn.set_processor(ap.MI355CogVideoXAttnProcessor())
        oh, oe = attn(vid, encoder_hidden_states=txt, image_rotary_emb=(cos, sin))
        torch.testing.assert_close(oh, rh, atol=2e-5, rtol=2e-5)
        torch.testing.assert_close(oe, re, atol=2e-5, rtol=2e-5)
        attn.fuse_projections()                                             # the reference's own fuse (:769-820)
        attn.set_processor(ap.MI355FusedCogVideoXAttnProcessor())
        fh, fe = attn(vid, encoder_hidden_states=txt, image_rotary_emb=(cos, sin))
        torch.testing.assert_close(fh, rh, atol=2e-5, rtol=2e-5)
        torch.testing.assert_close(fe, re, atol=2e-5, rtol=2e-5)


def test_reference_tiling_raises_on_the_wan22_vae_and_the_mirror_says_so(ref_modules):
    """frameino_amd/autoencoder_kl_wan.py::enable_tiling returns the un-tiled result because the REFERENCE's tiled paths
    (architecture/autoencoder_kl_wan.py:1270-1397) produce none on the Wan2.2 VAE (patch_size=2, is_residual=True): tiled_encode
    feeds un-patchified tiles to a 12-channel conv_in, tiled_decode drops first_chunk.  Both are shown on the reference's own
    class here; the mirror's one-time warning is checked next to it."""
    import importlib
    import warnings
    ref = importlib.import_module("architecture.autoencoder_kl_wan")
    cfg = dict(base_dim=8, decoder_base_dim=16, z_dim=4, dim_mult=[1, 2, 4, 4], num_res_blocks=1, attn_scales=[],
               temperal_downsample=[False, True, True], dropout=0.0, latents_mean=[0.0] * 4, latents_std=[1.0] * 4,
               is_residual=True, in_channels=12, out_channels=12, patch_size=2, scale_factor_temporal=4, scale_factor_spatial=16)
    torch.manual_seed(0)
    vae = ref.AutoencoderKLWan(**cfg).eval()
    vae.enable_tiling(tile_sample_min_height=32, tile_sample_min_width=32, tile_sample_stride_height=24,
                      tile_sample_stride_width=24)
    with torch.no_grad():
        with pytest.raises(RuntimeError, match="12 channels"):
            vae.encode(torch.rand(1, 3, 5, 64, 96))
        with pytest.raises(RuntimeError, match="must match the size"):
            vae.decode(torch.randn(1, 4, 2, 8, 12))
        vae.disable_tiling()                                   # ... and both work un-tiled
        assert vae.decode(torch.randn(1, 4, 2, 4, 6), return_dict=False)[0].shape == (1, 3, 5, 64, 96)
    from frameino_amd.autoencoder_kl_wan import AutoencoderKLWan
    AutoencoderKLWan._warned_tiling = False
    mine = AutoencoderKLWan(**cfg)
    with pytest.warns(RuntimeWarning, match="UN-TILED"):
        mine.enable_tiling(tile_sample_min_height=32)
    assert mine.use_tiling and mine.tile_sample_min_height == 32 and mine.decode_chunk_frames == 8
    with warnings.catch_warnings():
        warnings.simplefilter("error")                         # once per process
        mine.enable_tiling()
    mine.disable_tiling()
    assert not mine.use_tiling and mine.decode_chunk_frames == "auto"
