"""Inert stand-in for names of diffusers that the reference's TRAINING script imports at module level
(train_code/train_cogvideox_motion_FrameINO.py:45-60) and the denoising path never calls: the reference's CogVideoX
pipeline imports that script at call time for one helper (pipelines/pipeline_cogvideox_i2v_motion_FrameINO.py:816)."""


class Inert:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return self

    def __bool__(self):
        return False

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return Inert()
