import contextlib

import torch

from ..configuration_utils import ConfigMixin


class _Bar:
    def update(self, n=1):
        pass


class DiffusionPipeline(ConfigMixin):
    def register_modules(self, **kwargs):
        for k, v in kwargs.items():
            setattr(self, k, v)

    @property
    def _execution_device(self):
        return torch.device("cpu")

    @contextlib.contextmanager
    def progress_bar(self, iterable=None, total=None):
        yield _Bar()

    def maybe_free_model_hooks(self):
        pass
