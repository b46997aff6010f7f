"""Loading HF-diffusers-format checkpoints (the layout `from_pretrained` reads at /root/reference/app.py:156-161):
`<dir>/config.json` + `diffusion_pytorch_model*.safetensors` (optionally sharded with an index json).
Key names are the module paths of the reference model files, which the mirrors in this package keep.
"""
import glob
import json
import os

import torch

_IGNORED_CFG = ("_class_name", "_diffusers_version", "_name_or_path", "_use_default_values")


def read_config(path):
    with open(os.path.join(path, "config.json")) as f:
        cfg = json.load(f)
    return {k: v for k, v in cfg.items() if k not in _IGNORED_CFG}


def read_state_dict(path):
    from safetensors.torch import load_file
    files = sorted(glob.glob(os.path.join(path, "*.safetensors")))
    if not files:
        raise FileNotFoundError(f"no *.safetensors under {path}")
    idx = glob.glob(os.path.join(path, "*.safetensors.index.json"))
    if idx:
        with open(idx[0]) as f:
            files = sorted({os.path.join(path, v) for v in json.load(f)["weight_map"].values()})
    sd = {}
    for fn in files:
        sd.update(load_file(fn))
    return sd


def _accepted(cls, cfg):
    import inspect
    params = inspect.signature(cls.__init__).parameters
    return {k: v for k, v in cfg.items() if k in params}


def load_wan_transformer(path, torch_dtype=torch.bfloat16, device="cuda"):
    """WanTransformer3DModel.from_pretrained equivalent (fp32 islands of transformer_wan.py:393 are kept fp32)."""
    from .transformer_wan import WanTransformer3DModel
    cfg = read_config(path)
    m = WanTransformer3DModel(**_accepted(WanTransformer3DModel, cfg)).to(device)
    sd = {k: v for k, v in read_state_dict(path).items() if "norm_added_q" not in k}      # :394 keys to ignore
    return m.load_reference_state_dict(sd, dtype=torch_dtype).eval()


def load_wan_vae(path, torch_dtype=torch.bfloat16, device="cuda"):
    from .autoencoder_kl_wan import AutoencoderKLWan
    cfg = read_config(path)
    vae = AutoencoderKLWan(**_accepted(AutoencoderKLWan, cfg)).to(device)
    return vae.load_reference_state_dict(read_state_dict(path), dtype=torch_dtype)


def load_cogvideox_transformer(path, torch_dtype=torch.bfloat16, device="cuda", **overrides):
    """`use_FrameIn=True` is passed at load time by the reference (train_cogvideox_motion_FrameINO.py:682-686)."""
    from .cogvideox_transformer_3d import CogVideoXTransformer3DModel
    cfg = dict(read_config(path), **overrides)
    m = CogVideoXTransformer3DModel(**_accepted(CogVideoXTransformer3DModel, cfg)).to(device)
    return m.load_reference_state_dict(read_state_dict(path), dtype=torch_dtype).eval()
