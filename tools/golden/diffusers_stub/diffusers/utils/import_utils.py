def is_torch_npu_available():
    return False


def is_torch_xla_version(*a, **k):
    return False


def is_xformers_available():
    return False
