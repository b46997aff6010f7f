from dataclasses import dataclass


@dataclass
class CogVideoXPipelineOutput:
    frames: object
