class FromOriginalModelMixin:
    pass


class PeftAdapterMixin:
    pass


class WanLoraLoaderMixin:
    pass


class CogVideoXLoraLoaderMixin:
    pass
