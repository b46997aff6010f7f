import functools
import inspect


class FrozenDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


def register_to_config(init):
    @functools.wraps(init)
    def inner(self, *args, **kwargs):
        sig = inspect.signature(init)
        bound = sig.bind(self, *args, **kwargs)
        bound.apply_defaults()
        cfg = {k: v for k, v in bound.arguments.items() if k != "self"}
        init(self, *args, **kwargs)
        self._internal_dict = FrozenDict(cfg)

    return inner


class ConfigMixin:
    config_name = None

    @property
    def config(self):
        return self._internal_dict

    def register_to_config(self, **kwargs):
        d = dict(getattr(self, "_internal_dict", {}))
        d.update(kwargs)
        self._internal_dict = FrozenDict(d)
