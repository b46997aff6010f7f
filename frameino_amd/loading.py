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


def load_wan_transformer(path, torch_dtype=torch.bfloat16, device="cuda", cls=None, **overrides):
    """WanTransformer3DModel.from_pretrained equivalent (fp32 islands of transformer_wan.py:393 are kept fp32)."""
    from .transformer_wan import WanTransformer3DModel
    WanTransformer3DModel = cls or WanTransformer3DModel
    cfg = dict(read_config(path), **overrides)
    m = WanTransformer3DModel(**_accepted(WanTransformer3DModel, cfg)).to(device)
    sd = {k: v for k, v in read_state_dict(path).items() if "norm_added_q" not in k}      # :394 keys to ignore
    return m.load_reference_state_dict(sd, dtype=torch_dtype).eval()


def load_wan_vae(path, torch_dtype=torch.bfloat16, device="cuda", cls=None, **overrides):
    """AutoencoderKLWan.from_pretrained equivalent.  `torch_dtype=torch.float32` (what the reference app asks for,
    app.py:157) is accepted: fp32 master weights are kept, `.dtype` reports fp32 so that the pipeline's casts behave as in
    the reference, and the convolutions compute in bf16 with fp32 accumulation (there is no fp32 MFMA conv path)."""
    from .autoencoder_kl_wan import AutoencoderKLWan
    AutoencoderKLWan = cls or AutoencoderKLWan
    cfg = dict(read_config(path), **overrides)
    vae = AutoencoderKLWan(**_accepted(AutoencoderKLWan, cfg)).to(device)
    return vae.load_reference_state_dict(read_state_dict(path), dtype=torch_dtype)


def load_cogvideox_transformer(path, torch_dtype=torch.bfloat16, device="cuda", cls=None, **overrides):
    """`use_FrameIn=True` is passed at load time by the reference (train_cogvideox_motion_FrameINO.py:682-686)."""
    from .cogvideox_transformer_3d import CogVideoXTransformer3DModel
    CogVideoXTransformer3DModel = cls or CogVideoXTransformer3DModel
    cfg = dict(read_config(path), **overrides)
    m = CogVideoXTransformer3DModel(**_accepted(CogVideoXTransformer3DModel, cfg)).to(device)
    return m.load_reference_state_dict(read_state_dict(path), dtype=torch_dtype).eval()


def load_cogvideox_vae(path, torch_dtype=torch.bfloat16, device="cuda", cls=None, **overrides):
    """AutoencoderKLCogVideoX.from_pretrained equivalent (`<repo>/vae` of zai-org/CogVideoX-5b-I2V, app.py:150-151)."""
    from .autoencoder_kl_cogvideox import AutoencoderKLCogVideoX
    AutoencoderKLCogVideoX = cls or AutoencoderKLCogVideoX
    cfg = dict(read_config(path), **overrides)
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


# ----------------------------------------------------------------------------------------------- from_pretrained surface
_HUB_KWARGS = ("cache_dir", "force_download", "local_files_only", "revision", "token", "use_safetensors", "variant",
               "low_cpu_mem_usage", "proxies", "device_map", "offload_folder", "use_auth_token", "resume_download")


def _local_folder(path, subfolder=None):
    """Checkpoints are LOCAL folders here (no hub access on the boxes this package runs on): a hub id such as
    "Wan-AI/Wan2.2-TI2V-5B-Diffusers" must have been downloaded to a directory first."""
    full = os.path.join(str(path), subfolder) if subfolder else str(path)
    if not os.path.isdir(full):
        raise OSError(f"{full!r} is not a local folder.  frameino_amd loads diffusers-format folders from disk only; "
                      f"download the repository first (`huggingface-cli download {path} --local-dir <dir>`) and pass <dir>.")
    return full


class FromPretrainedMixin:
    """`X.from_pretrained(path, subfolder=None, torch_dtype=None, **config_overrides)` -- the loader call of the
    reference's entry points (/root/reference/app.py:156-157, test_code/run_cogvideox_FrameIn_mass_evaluation.py:92-94,
    train_code/train_cogvideox_motion_FrameINO.py:682-686 with `use_FrameIn=`), for local diffusers-format folders.
    Like diffusers, the model is built on the host (move it with `.to("cuda")` or `pipe.to("cuda")`); `torch_dtype=None`
    means fp32 there and bf16 here -- the kernels compute in bf16 / fp16 only, fp32 islands are kept whatever is asked."""
    _loader_name = None

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, subfolder=None, torch_dtype=None, device="cpu", **kwargs):
        for k in _HUB_KWARGS:
            kwargs.pop(k, None)
        path = _local_folder(pretrained_model_name_or_path, subfolder)
        loader = globals()[cls._loader_name]
        import inspect
        extra = {k: v for k, v in kwargs.items() if k in inspect.signature(cls.__init__).parameters}
        unknown = sorted(set(kwargs) - set(extra))
        if unknown:
            raise TypeError(f"{cls.__name__}.from_pretrained: unexpected keyword arguments {unknown}")
        return loader(path, torch_dtype=torch_dtype or torch.bfloat16, device=device, cls=cls, **extra)


def _pipeline_from_pretrained(cls, path, components, torch_dtype=None, **given):
    """`Pipeline.from_pretrained(base_folder, transformer=..., vae=..., text_encoder=..., torch_dtype=...)`
    (/root/reference/app.py:161, test_code/run_cogvideox_FrameIn_mass_evaluation.py:101-107): every component that is not
    handed in is loaded from its sub-folder of the base folder when that sub-folder exists (text encoder / tokenizer
    through `transformers`), the scheduler by the class its scheduler_config.json names."""
    for k in _HUB_KWARGS:
        given.pop(k, None)
    base = _local_folder(path)
    dt = torch_dtype or torch.bfloat16
    parts = {}
    for name, load in components.items():
        if given.get(name) is not None:
            parts[name] = given.pop(name)
        elif os.path.isdir(os.path.join(base, name)):
            parts[name] = load(os.path.join(base, name), dt)
        else:
            given.pop(name, None)
            parts[name] = None
    index = {}
    if os.path.exists(os.path.join(base, "model_index.json")):
        with open(os.path.join(base, "model_index.json")) as f:
            index = json.load(f)
    return parts, index, given


def _load_text_encoder(folder, dt):
    import transformers
    with open(os.path.join(folder, "config.json")) as f:
        arch = (json.load(f).get("architectures") or ["T5EncoderModel"])[0]
    return getattr(transformers, arch).from_pretrained(folder, torch_dtype=dt).eval()


def _load_tokenizer(folder, dt):
    import transformers
    return transformers.AutoTokenizer.from_pretrained(folder)
