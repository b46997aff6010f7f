import logging as _logging

USE_PEFT_BACKEND = False


class _Logging:
    @staticmethod
    def get_logger(name):
        return _logging.getLogger(name)


logging = _Logging()


def scale_lora_layers(model, weight):
    pass


def unscale_lora_layers(model, weight=None):
    pass


def deprecate(*args, **kwargs):
    pass


def is_torch_version(*args, **kwargs):
    return True


def is_torch_xla_available():
    return False


def replace_example_docstring(doc):
    def deco(fn):
        return fn

    return deco


def is_ftfy_available():
    import importlib.util
    import sys
    return "ftfy" in sys.modules or importlib.util.find_spec("ftfy") is not None


def __getattr__(name):      # names only the reference's training script imports (never called on the denoising path)
    if name.startswith("__"):
        raise AttributeError(name)
    from .._inert import Inert
    return Inert
