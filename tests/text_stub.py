"""Tiny stand-ins for the text side of the pipelines' `prompt=` route -- test infrastructure, shared by
tools/golden/make_golden.py (which runs the REFERENCE pipelines with them) and the `-m gpu` tests (which run the HIP
pipelines with the same objects): the text encoder is the real `transformers` class (UMT5EncoderModel for Wan,
T5EncoderModel for CogVideoX) at toy width with weights stored in the fixture; the tokenizer is a character-level
stand-in with the call surface the pipelines use (reference pipelines/pipeline_wan_i2v_motion_FrameINO.py:221-231,
pipelines/pipeline_cogvideox_i2v_motion_FrameINO.py:240-256).  The real tokenizers are sentencepiece models shipped with
the checkpoints: not available offline, and nothing of the path depends on WHICH ids a prompt maps to.
"""
from types import SimpleNamespace

import torch

VOCAB = 64
PAD, EOS = 0, 1


class CharTokenizer:
    """ids = 2 + (ord(c) mod 62) per character, EOS appended, padded with 0 (T5 convention: pad 0, eos 1)."""

    pad_token_id, eos_token_id = PAD, EOS

    def _ids(self, text, add_special_tokens):
        ids = [2 + (ord(c) % (VOCAB - 2)) for c in text]
        return ids + [EOS] if add_special_tokens else ids

    def __call__(self, text, padding=False, max_length=None, truncation=False, add_special_tokens=True,
                 return_attention_mask=True, return_tensors=None):
        text = [text] if isinstance(text, str) else list(text)
        rows = [self._ids(t, add_special_tokens) for t in text]
        if truncation and max_length is not None:
            rows = [r[:max_length - 1] + [EOS] if (len(r) > max_length and add_special_tokens) else r[:max_length]
                    for r in rows]
        if padding == "max_length":
            width = max_length
        elif padding in ("longest", True):
            width = max(len(r) for r in rows)
        else:
            width = None
        if width is not None:
            mask = [[1] * len(r) + [0] * (width - len(r)) for r in rows]
            rows = [r + [PAD] * (width - len(r)) for r in rows]
        else:
            mask = [[1] * len(r) for r in rows]
        if return_tensors == "pt":
            rows, mask = torch.tensor(rows, dtype=torch.long), torch.tensor(mask, dtype=torch.long)
        return SimpleNamespace(input_ids=rows, attention_mask=mask)

    def batch_decode(self, ids, **kw):
        return ["".join(chr(int(i)) if 32 <= int(i) < 127 else "?" for i in row) for row in ids]


TEXT_CFG = dict(vocab_size=VOCAB, d_model=16, d_kv=8, d_ff=32, num_layers=2, num_heads=2,
                relative_attention_num_buckets=8, relative_attention_max_distance=16, dropout_rate=0.0,
                feed_forward_proj="gated-gelu")


def tiny_text_encoder(kind, state_dict=None, seed=0):
    """kind "umt5" (Wan2.2: UMT5EncoderModel) or "t5" (CogVideoX: T5EncoderModel), d_model 16 = the tiny DiTs' text_dim.
    `state_dict`: the fixture's weights (the generator calls it without and stores what the seeded init produced)."""
    from transformers import T5Config, T5EncoderModel, UMT5Config, UMT5EncoderModel
    torch.manual_seed(seed)
    m = (UMT5EncoderModel(UMT5Config(**TEXT_CFG)) if kind == "umt5" else T5EncoderModel(T5Config(**TEXT_CFG))).eval()
    if state_dict is None:
        g = torch.Generator().manual_seed(seed + 1)
        with torch.no_grad():
            for n, p in m.named_parameters():          # seeded re-init: well-scaled, every parameter away from its default
                p.copy_(torch.randn(p.shape, generator=g) * (0.5 if p.ndim > 1 else 0.2) + (1.0 if "norm" in n else 0.0))
    else:
        m.load_state_dict(state_dict)
    return m
