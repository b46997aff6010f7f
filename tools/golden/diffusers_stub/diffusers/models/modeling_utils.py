import torch
from torch import nn


class ModelMixin(nn.Module):
    @property
    def dtype(self):
        for p in self.parameters():
            return p.dtype
        return torch.float32

    @property
    def device(self):
        for p in self.parameters():
            return p.device
        return torch.device("cpu")
