"""diffusers.models.attention_processor -> the reference's own VENDORED copy (architecture/attention_processor.py:
`Attention` :50-820 and every processor), so that the attention container the goldens run through is the reference's
code, not a builder re-typing of it.

`Attention.__init__` imports `.normalization` relatively (architecture/attention_processor.py:141), a module the
reference tree does not contain (it exists in diffusers only): `architecture.normalization` is provided here as a shim
over the stand-in normalisation classes (FP32LayerNorm / RMSNorm -- third-party arithmetic, restated, unpinned)."""
import importlib
import sys
import types


def _normalization_shim():
    if "architecture.normalization" in sys.modules:
        return
    from . import normalization as N
    m = types.ModuleType("architecture.normalization")
    m.FP32LayerNorm, m.RMSNorm, m.AdaLayerNorm = N.FP32LayerNorm, N.RMSNorm, N.AdaLayerNorm

    class _Absent:                                        # qk_norm variants FrameINO never selects
        def __init__(self, *a, **k):
            raise NotImplementedError("not on the FrameINO path")

    m.LpNorm = m.MochiRMSNorm = _Absent
    m.get_normalization = lambda *a, **k: _Absent()
    sys.modules["architecture.normalization"] = m


def __getattr__(name):
    _normalization_shim()
    return getattr(importlib.import_module("architecture.attention_processor"), name)
