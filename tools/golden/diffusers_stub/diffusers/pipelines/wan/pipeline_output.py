from dataclasses import dataclass


@dataclass
class WanPipelineOutput:
    frames: object
