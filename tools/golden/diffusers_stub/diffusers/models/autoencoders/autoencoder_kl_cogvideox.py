"""Stand-in for diffusers' AutoencoderKLCogVideoX (third-party: no source under /root/reference, not installable
offline).  The arithmetic is the builder's restatement in /root/repo/oracle/cog_vae.py (ONE restatement, used here and by
the tests: parity for this class is UNPINNED); this file only gives it the interface the reference
pipeline calls: .encode(x).latent_dist.sample(generator) / .mode(), .decode(z).sample, .config.*, .dtype,
enable_slicing() / enable_tiling()."""
import os
import sys

import torch
from torch import nn

from ...configuration_utils import ConfigMixin, register_to_config
from ..modeling_utils import ModelMixin
from .vae import DecoderOutput, DiagonalGaussianDistribution

_REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), *[".."] * 6))
if _REPO not in sys.path:
    sys.path.append(_REPO)


class _EncoderOutput:
    def __init__(self, latent_dist):
        self.latent_dist = latent_dist


class AutoencoderKLCogVideoX(ModelMixin, ConfigMixin):
    @register_to_config
    def __init__(self, in_channels=3, out_channels=3, block_out_channels=(128, 256, 256, 512), latent_channels=16,
                 layers_per_block=3, norm_eps=1e-6, norm_num_groups=32, temporal_compression_ratio=4,
                 scaling_factor=0.7, invert_scale_latents=False):
        super().__init__()
        from oracle import cog_vae as V
        self._V = V
        self._cfg = dict(in_channels=in_channels, out_channels=out_channels, block_out_channels=tuple(block_out_channels),
                         latent_channels=latent_channels, layers_per_block=layers_per_block, norm_eps=norm_eps,
                         norm_num_groups=norm_num_groups, temporal_compression_ratio=temporal_compression_ratio,
                         scaling_factor=scaling_factor, invert_scale_latents=invert_scale_latents)
        self.params = nn.ParameterDict()
        self._names = {}

    def load_flat_state_dict(self, sd):
        for k, v in sd.items():
            key = k.replace(".", "__")
            self._names[key] = k
            self.params[key] = nn.Parameter(v.clone(), requires_grad=False)

    def _sd(self):
        return {self._names[k]: v.data for k, v in self.params.items()}

    def enable_slicing(self):
        pass

    def enable_tiling(self):
        pass

    def encode(self, x, return_dict=True):
        moments = self._V.encode_moments(self._sd(), self._cfg, x)
        return _EncoderOutput(DiagonalGaussianDistribution(moments))

    def decode(self, z, return_dict=True):
        return DecoderOutput(sample=self._V.decode(self._sd(), self._cfg, z))
