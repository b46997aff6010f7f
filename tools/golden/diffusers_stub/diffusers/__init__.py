"""Minimal stand-in for the `diffusers` package (NOT installed in this image).

Purpose: let the reference's own model files under /root/reference/architecture import
UNMODIFIED in this container so that tools/golden/make_golden.py can run them on CPU and
record golden input/output vectors.  This is test tooling only: nothing under
frameino_amd/ imports it, and it never travels as "the reference".

The arithmetic-bearing classes here (Attention, FeedForward, RMSNorm, FP32LayerNorm,
CogVideoXLayerNormZero, AdaLayerNorm, DiagonalGaussianDistribution) are restated from the
published semantics of huggingface/diffusers (~v0.35, unpinned by the reference:
requirements.txt:12) -- parity for those third-party pieces is therefore "unpinned"
(see DESIGN.md).  Everything the reference vendors itself (embeddings, processors, blocks,
VAE) runs from the reference's own source.
"""
__version__ = "0.35.0.stub"


def __getattr__(name):
    """`from diffusers import AutoencoderKLCogVideoX, CogVideoXDPMScheduler` (train_code/train_cogvideox_motion_FrameINO.py
    :45-48, imported by the reference's CogVideoX pipeline at call time)"""
    if name == "AutoencoderKLCogVideoX":
        from .models import AutoencoderKLCogVideoX
        return AutoencoderKLCogVideoX
    if name in ("CogVideoXDPMScheduler", "CogVideoXDDIMScheduler", "FlowMatchEulerDiscreteScheduler",
                "UniPCMultistepScheduler"):
        from . import schedulers
        return getattr(schedulers, name)
    raise AttributeError(name)
