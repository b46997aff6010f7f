import torch
from torch import nn


class ModelMixin(nn.Module):
    @property
    def dtype(self):
        # diffusers' get_parameter_dtype: the first floating-point parameter OUTSIDE `_keep_in_fp32_modules` (a model loaded with
        # torch_dtype=float16 reports float16 although its fp32 islands come first in named_parameters())
        keep = getattr(self, "_keep_in_fp32_modules", None) or []
        last = None
        for n, p in self.named_parameters():
            last = p.dtype
            if any(k in n for k in keep):
                continue
            if p.is_floating_point():
                return p.dtype
        return last if last is not None else torch.float32

    @property
    def device(self):
        for p in self.parameters():
            return p.device
        return torch.device("cpu")
