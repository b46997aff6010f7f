"""HF-diffusers-format folders (config.json + safetensors shards) round-trip into the mirrors by key (CPU only:
loading does not touch the HIP library)."""
import json

import torch

from tests.conftest import load_golden
from tests.parity import model_cfg


def test_wan_transformer_folder_roundtrip(tmp_path):
    from safetensors.torch import save_file
    from frameino_amd import loading
    cfg, sd, _ = load_golden("wan_dit_tiny")
    mc = model_cfg(cfg)
    (tmp_path / "config.json").write_text(json.dumps(dict(mc, _class_name="WanTransformer3DModel",
                                                          _diffusers_version="0.35.0", patch_size=list(mc["patch_size"]))))
    keys = sorted(sd)
    half = len(keys) // 2
    shards = {"diffusion_pytorch_model-00001-of-00002.safetensors": {k: sd[k].contiguous() for k in keys[:half]},
              "diffusion_pytorch_model-00002-of-00002.safetensors": {k: sd[k].contiguous() for k in keys[half:]}}
    weight_map = {}
    for fn, part in shards.items():
        save_file(part, str(tmp_path / fn))
        weight_map.update({k: fn for k in part})
    (tmp_path / "diffusion_pytorch_model.safetensors.index.json").write_text(json.dumps({"weight_map": weight_map}))
    m = loading.load_wan_transformer(str(tmp_path), torch_dtype=torch.bfloat16, device="cpu")
    own = m.state_dict()
    assert set(own) == set(sd)
    for k, v in own.items():
        keep32 = any(s in k for s in m._keep_in_fp32_modules)
        assert v.dtype == (torch.float32 if keep32 else torch.bfloat16), k
        assert torch.equal(v.float(), sd[k].to(v.dtype).float())


def test_wan_vae_folder_roundtrip(tmp_path):
    from safetensors.torch import save_file
    from frameino_amd import loading
    cfg, sd, _ = load_golden("wan_vae_tiny")
    c = {k: (list(v) if isinstance(v, (list, tuple)) else v) for k, v in cfg.items()}
    c["is_residual"] = bool(c["is_residual"])
    (tmp_path / "config.json").write_text(json.dumps(dict(c, _class_name="AutoencoderKLWan")))
    save_file({k: v.contiguous() for k, v in sd.items()}, str(tmp_path / "diffusion_pytorch_model.safetensors"))
    vae = loading.load_wan_vae(str(tmp_path), device="cpu")
    assert vae.config.z_dim == 4 and set(vae.state_dict()) == set(sd)
