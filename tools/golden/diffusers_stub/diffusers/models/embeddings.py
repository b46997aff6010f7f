"""diffusers.models.embeddings -> forwards to the reference's own vendored copy
(architecture/embeddings.py), so the code exercised is the reference's."""
import importlib


def __getattr__(name):
    mod = importlib.import_module("architecture.embeddings")
    return getattr(mod, name)
