"""Stand-ins importable as `diffusers.models.X` by the reference's CogVideoX pipeline (type hints + the VAE it calls)."""
from .autoencoders.autoencoder_kl_cogvideox import AutoencoderKLCogVideoX  # noqa: F401


class CogVideoXTransformer3DModel:          # type hint only: the pipeline is handed the reference's own architecture/ class
    pass
