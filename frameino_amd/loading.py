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


def config_report(cfg, assumed, cls=None, name="config.json"):
    """What a checkpoint's config.json says against the hyper-parameters this package ASSUMES offline (configs.py: the
    released folders could not be read in the build environment) and against the constructor of the
    mirror: -> list of human-readable lines, one per key that differs, is unknown to the mirror, or is missing from the
    file.  The checkpoint always wins; the report is what a first run on real weights should print and eyeball."""
    import inspect
    lines = []
    norm = lambda v: list(v) if isinstance(v, (list, tuple)) else v      # noqa: E731
    for k in sorted(set(cfg) | set(assumed)):
        if k in cfg and k in assumed and norm(cfg[k]) != norm(assumed[k]):
            lines.append(f"{name}: {k} = {cfg[k]!r} (this package assumed {assumed[k]!r} offline)")
        elif k in assumed and k not in cfg:
            lines.append(f"{name}: {k} absent, constructor default / assumed value {assumed[k]!r} is used")
    if cls is not None:
        params = inspect.signature(cls.__init__).parameters
        for k in sorted(cfg):
            if k not in params:
                lines.append(f"{name}: {k} = {cfg[k]!r} is not a parameter of {cls.__name__} and is ignored")
    return lines


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


def load_cogvideox_vae(path, torch_dtype=torch.bfloat16, device="cuda"):
    """AutoencoderKLCogVideoX.from_pretrained equivalent (`<repo>/vae` of zai-org/CogVideoX-5b-I2V, app.py:150-151)."""
    from .autoencoder_kl_cogvideox import AutoencoderKLCogVideoX
    cfg = read_config(path)
    vae = AutoencoderKLCogVideoX(**_accepted(AutoencoderKLCogVideoX, cfg)).to(device)
    return vae.load_reference_state_dict(read_state_dict(path), dtype=torch_dtype)


def load_scheduler(path):
    """`<repo>/scheduler/scheduler_config.json` -> the matching sampler of frameino_amd.schedulers (by `_class_name`:
    the released Wan2.2 folder names UniPCMultistepScheduler, CogVideoX-5b-I2V CogVideoXDPMScheduler /
    CogVideoXDDIMScheduler; FlowMatchEulerDiscreteScheduler is what the reference's training configures)."""
    from . import schedulers
    with open(os.path.join(path, "scheduler_config.json")) as f:
        cfg = json.load(f)
    name = cfg.get("_class_name", "")
    cls = getattr(schedulers, name, None)
    if cls is None:
        raise NotImplementedError(f"scheduler {name!r}: the built samplers are FlowMatchEulerDiscreteScheduler, "
                                  f"UniPCMultistepScheduler, CogVideoXDDIMScheduler, CogVideoXDPMScheduler")
    return cls(**{k: v for k, v in cfg.items() if not k.startswith("_")})
