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


def test_config_report_names_every_difference():
    from frameino_amd.configs import WAN22_5B_CFG
    from frameino_amd.loading import config_report
    from frameino_amd.transformer_wan import WanTransformer3DModel
    cfg = dict(WAN22_5B_CFG, ffn_dim=13824, patch_size=[1, 2, 2], pos_embed_seq_len=None, brand_new_key=3)
    del cfg["rope_max_seq_len"]
    lines = config_report(cfg, WAN22_5B_CFG, WanTransformer3DModel, "transformer/config.json")
    text = "\n".join(lines)
    assert "ffn_dim = 13824 (this package assumed 14336 offline)" in text
    assert "rope_max_seq_len absent" in text and "brand_new_key = 3 is not a parameter" in text
    assert "patch_size" not in text                      # list vs tuple is not a difference
    assert config_report(dict(WAN22_5B_CFG), WAN22_5B_CFG, WanTransformer3DModel) == []


def test_scheduler_and_cog_vae_folders(tmp_path):
    from safetensors.torch import save_file
    from frameino_amd import loading
    from frameino_amd.schedulers import CogVideoXDPMScheduler, UniPCMultistepScheduler
    from oracle import cog_vae as V
    (tmp_path / "scheduler_config.json").write_text(json.dumps(
        {"_class_name": "UniPCMultistepScheduler", "_diffusers_version": "0.35.0", "flow_shift": 5.0,
         "prediction_type": "flow_prediction", "use_flow_sigmas": True, "solver_order": 2, "num_train_timesteps": 1000}))
    assert isinstance(loading.load_scheduler(str(tmp_path)), UniPCMultistepScheduler)
    (tmp_path / "scheduler_config.json").write_text(json.dumps(
        {"_class_name": "CogVideoXDPMScheduler", "snr_shift_scale": 1.0, "timestep_spacing": "trailing",
         "prediction_type": "v_prediction", "rescale_betas_zero_snr": True}))
    assert isinstance(loading.load_scheduler(str(tmp_path)), CogVideoXDPMScheduler)
    cfg = dict(V.COGVIDEOX_VAE_CFG, block_out_channels=[16, 32, 32, 64], norm_num_groups=8, latent_channels=4,
               layers_per_block=2)
    sd = V.cog_vae_random_state_dict(cfg, 3)
    d = tmp_path / "vae"
    d.mkdir()
    (d / "config.json").write_text(json.dumps(dict(cfg, _class_name="AutoencoderKLCogVideoX", sample_height=480)))
    save_file({k: v.contiguous() for k, v in sd.items()}, str(d / "diffusion_pytorch_model.safetensors"))
    vae = loading.load_cogvideox_vae(str(d), device="cpu")
    assert vae.config.latent_channels == 4 and set(vae.state_dict()) == set(sd)
