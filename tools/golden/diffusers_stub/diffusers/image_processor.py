from typing import Any

PipelineImageInput = Any


class IPAdapterMaskProcessor:
    pass
