from dataclasses import dataclass

import torch


@dataclass
class DecoderOutput:
    sample: torch.Tensor
    commit_loss: object = None


class DiagonalGaussianDistribution:
    def __init__(self, parameters, deterministic=False):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.deterministic = deterministic
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)

    def sample(self, generator=None):
        noise = torch.randn(self.mean.shape, generator=generator, dtype=self.parameters.dtype)
        return self.mean + self.std * noise.to(self.parameters.device)

    def mode(self):
        return self.mean
