import torch


def maybe_allow_in_graph(cls):
    return cls


def is_torch_version(*args, **kwargs):
    return True


def randn_tensor(shape, generator=None, device=None, dtype=None, layout=None):
    # diffusers semantics: sample on the generator's device (CPU by default), then move.
    gen_device = generator.device if generator is not None else (device or torch.device("cpu"))
    t = torch.randn(shape, generator=generator, device=gen_device, dtype=dtype)
    return t.to(device) if device is not None else t


def __getattr__(name):      # names only the reference's training script imports (never called on the denoising path)
    if name.startswith("__"):
        raise AttributeError(name)
    from .._inert import Inert
    return Inert
