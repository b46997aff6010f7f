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


# ---------------------------------------------------------------------------------------------- the loader call sites
def _write_model(folder, cfg, sd, class_name):
    from safetensors.torch import save_file
    folder.mkdir(parents=True, exist_ok=True)
    c = {k: (list(v) if isinstance(v, (list, tuple)) else v) for k, v in cfg.items()}
    (folder / "config.json").write_text(json.dumps(dict(c, _class_name=class_name, _diffusers_version="0.35.0")))
    save_file({k: v.contiguous() for k, v in sd.items()}, str(folder / "diffusion_pytorch_model.safetensors"))


def _tiny_t5(folder, umt5=False):
    """a randomly initialised 1-layer (U)MT5 / T5 encoder saved in the transformers layout (no tokenizer: its
    sentencepiece model cannot be made offline -- the pipelines accept pre-computed prompt embeddings)"""
    import transformers
    cls, ccls = (transformers.UMT5EncoderModel, transformers.UMT5Config) if umt5 else \
        (transformers.T5EncoderModel, transformers.T5Config)
    m = cls(ccls(vocab_size=64, d_model=16, d_kv=8, d_ff=32, num_layers=1, num_heads=2))
    m.save_pretrained(str(folder))


def test_app_py_loader_lines_with_only_the_imports_changed(tmp_path):
    """/root/reference/app.py:156-163, line for line, against synthetic local folders:
        transformer = WanTransformer3DModel.from_pretrained(transformer_ckpt_path, torch_dtype=torch.float16)
        vae = AutoencoderKLWan.from_pretrained(base_model_id, subfolder="vae", torch_dtype=torch.float32)
        pipe = WanImageToVideoPipeline.from_pretrained(base_model_id, transformer=transformer, vae=vae, torch_dtype=torch.bfloat16)
        pipe.to("cuda"); pipe.enable_model_cpu_offload()"""
    from frameino_amd.autoencoder_kl_wan import AutoencoderKLWan
    from frameino_amd.pipeline_wan_i2v_motion_frameino import WanImageToVideoPipeline
    from frameino_amd.schedulers import UniPCMultistepScheduler
    from frameino_amd.transformer_wan import WanTransformer3DModel
    cfg, sd, _ = load_golden("wan_dit_tiny")
    mc = model_cfg(cfg)
    transformer_ckpt_path = tmp_path / "FrameINO_Wan2.2_5B_Stage2_MotionINO_v1.6"
    _write_model(transformer_ckpt_path, dict(mc, patch_size=list(mc["patch_size"])), sd, "WanTransformer3DModel")
    base_model_id = tmp_path / "Wan2.2-TI2V-5B-Diffusers"
    vcfg, vsd, _ = load_golden("wan_vae_tiny")
    vcfg = dict(vcfg, is_residual=bool(vcfg["is_residual"]))
    _write_model(base_model_id / "vae", vcfg, vsd, "AutoencoderKLWan")
    (base_model_id / "scheduler").mkdir()
    (base_model_id / "scheduler" / "scheduler_config.json").write_text(json.dumps(
        {"_class_name": "UniPCMultistepScheduler", "flow_shift": 5.0, "prediction_type": "flow_prediction",
         "use_flow_sigmas": True, "solver_order": 2, "num_train_timesteps": 1000}))
    _tiny_t5(base_model_id / "text_encoder", umt5=True)
    (base_model_id / "model_index.json").write_text(json.dumps(
        {"_class_name": "WanImageToVideoPipeline", "expand_timesteps": True, "boundary_ratio": None}))
    transformer_ckpt_path, base_model_id = str(transformer_ckpt_path), str(base_model_id)

    transformer = WanTransformer3DModel.from_pretrained(transformer_ckpt_path, torch_dtype=torch.float16)
    vae = AutoencoderKLWan.from_pretrained(base_model_id, subfolder="vae", torch_dtype=torch.float32)
    pipe = WanImageToVideoPipeline.from_pretrained(base_model_id, transformer=transformer, vae=vae, torch_dtype=torch.bfloat16)
    pipe.to("cpu")                                     # "cuda" in the app; there is no GPU in this container
    pipe.enable_model_cpu_offload()

    assert pipe.transformer is transformer and pipe.vae is vae
    assert transformer.dtype == torch.float16 and transformer.blocks[0].scale_shift_table.dtype == torch.float32
    assert vae.dtype == torch.float32 and vae._dtype == torch.bfloat16          # fp32 interface, bf16 convolutions
    assert isinstance(pipe.scheduler, UniPCMultistepScheduler) and pipe.config.expand_timesteps is True
    assert type(pipe.text_encoder).__name__ == "UMT5EncoderModel" and pipe.text_encoder.dtype == torch.bfloat16
    assert pipe.tokenizer is None                                                # no tokenizer/ sub-folder here
    # the VAE memory switches of diffusers exist (architecture/autoencoder_kl_wan.py:1084-1133)
    vae.enable_slicing(); vae.enable_tiling(); assert vae.use_tiling and vae.decode_chunk_frames == 8
    vae.disable_tiling(); vae.disable_slicing(); assert not vae.use_tiling and vae.decode_chunk_frames == "auto"
    # a hub id that was not downloaded is an explicit error, not a hang on a socket
    import pytest
    with pytest.raises(OSError, match="not a local folder"):
        WanTransformer3DModel.from_pretrained("uva-cv-lab/FrameINO_Wan2.2_5B_Stage2_MotionINO_v1.6")
    with pytest.raises(TypeError, match="unexpected keyword"):
        WanTransformer3DModel.from_pretrained(transformer_ckpt_path, not_a_config_key=1)


def test_cogvideox_mass_evaluation_loader_lines_with_only_the_imports_changed(tmp_path):
    """/root/reference/test_code/run_cogvideox_FrameIn_mass_evaluation.py:92-108, line for line:
        transformer = CogVideoXTransformer3DModel.from_pretrained(transformer_ckpt_path, torch_dtype=torch.float16)
        text_encoder = T5EncoderModel.from_pretrained(base_model_id, subfolder="text_encoder", torch_dtype=torch.float16)
        vae = AutoencoderKLCogVideoX.from_pretrained(base_model_id, subfolder="vae", torch_dtype=torch.float16)
        vae.enable_slicing(); vae.enable_tiling()
        pipe = CogVideoXImageToVideoPipeline.from_pretrained(base_model_id, text_encoder=, transformer=, vae=, torch_dtype=)
        pipe.enable_model_cpu_offload()"""
    from transformers import T5EncoderModel
    from frameino_amd.autoencoder_kl_cogvideox import AutoencoderKLCogVideoX
    from frameino_amd.cogvideox_transformer_3d import CogVideoXTransformer3DModel
    from frameino_amd.pipeline_cogvideox_i2v_motion_frameino import CogVideoXImageToVideoPipeline
    from frameino_amd.schedulers import CogVideoXDPMScheduler
    from tests.test_oracle_golden import cog_pipe_fixture
    dit_cfg, dit_sd, vae_cfg, vae_sd, _ = cog_pipe_fixture(load_golden)
    transformer_ckpt_path = tmp_path / "FrameINO_CogVideoX_Stage2_MotionINO_v1.0"
    stored = {k: v for k, v in dit_cfg.items() if k != "use_FrameIn"}           # the released config predates the flag
    _write_model(transformer_ckpt_path, stored, dit_sd, "CogVideoXTransformer3DModel")
    base_model_id = tmp_path / "CogVideoX-5b-I2V"
    _write_model(base_model_id / "vae", vae_cfg, vae_sd, "AutoencoderKLCogVideoX")
    _tiny_t5(base_model_id / "text_encoder")
    (base_model_id / "scheduler").mkdir()
    (base_model_id / "scheduler" / "scheduler_config.json").write_text(json.dumps(
        {"_class_name": "CogVideoXDPMScheduler", "snr_shift_scale": 1.0, "timestep_spacing": "trailing",
         "prediction_type": "v_prediction", "rescale_betas_zero_snr": True}))
    transformer_ckpt_path, base_model_id = str(transformer_ckpt_path), str(base_model_id)

    transformer = CogVideoXTransformer3DModel.from_pretrained(transformer_ckpt_path, torch_dtype=torch.float16,
                                                              use_FrameIn=True)      # train_code :682-686 passes it at load
    text_encoder = T5EncoderModel.from_pretrained(base_model_id, subfolder="text_encoder", torch_dtype=torch.float16)
    vae = AutoencoderKLCogVideoX.from_pretrained(base_model_id, subfolder="vae", torch_dtype=torch.float16)
    vae.enable_slicing()
    vae.enable_tiling()
    pipe = CogVideoXImageToVideoPipeline.from_pretrained(base_model_id, text_encoder=text_encoder, transformer=transformer,
                                                         vae=vae, torch_dtype=torch.float16)
    pipe.enable_model_cpu_offload()

    assert pipe.transformer is transformer and pipe.vae is vae and pipe.text_encoder is text_encoder
    assert transformer.dtype == torch.float16 and transformer.config.use_FrameIn is True
    assert vae.dtype == torch.float16 and vae.use_slicing and vae.use_tiling
    assert isinstance(pipe.scheduler, CogVideoXDPMScheduler)
    assert set(transformer.state_dict()) >= set(dit_sd)
    assert pipe.vae_scaling_factor_image == vae_cfg["scaling_factor"]
