import torch.nn.functional as F
from torch import nn


class FP32SiLU(nn.Module):
    def forward(self, x):
        return F.silu(x.float(), inplace=False).to(x.dtype)


_ACT = {"swish": nn.SiLU, "silu": nn.SiLU, "mish": nn.Mish, "gelu": nn.GELU, "relu": nn.ReLU}


def get_activation(act_fn):
    return _ACT[act_fn.lower()]()


class GELU(nn.Module):
    """diffusers.models.activations.GELU: Linear then gelu (optionally tanh-approximate)."""

    def __init__(self, dim_in, dim_out, approximate="none", bias=True):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out, bias=bias)
        self.approximate = approximate

    def forward(self, hidden_states):
        hidden_states = self.proj(hidden_states)
        return F.gelu(hidden_states, approximate=self.approximate)
