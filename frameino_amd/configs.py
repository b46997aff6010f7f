"""Architecture hyper-parameters of the released checkpoints the path is built for (from the HF
configs; the code defaults at architecture/transformer_wan.py:398-416 are the 14B values).  A checkpoint folder's
own config.json always wins (frameino_amd/loading.py); these are what bench.py / the examples instantiate offline."""

# Wan2.2-TI2V-5B FrameINO: 48 latent channels + 48 trajectory channels in, 48 out, 24 heads x 128, 30 layers
WAN22_5B_CFG = dict(
    patch_size=(1, 2, 2), num_attention_heads=24, attention_head_dim=128, in_channels=96, out_channels=48,
    text_dim=4096, freq_dim=256, ffn_dim=14336, num_layers=30, cross_attn_norm=True, eps=1e-6,
    rope_max_seq_len=1024,
)

# Wan2.2 VAE (z = 48, 16x spatial / 4x temporal compression, residual down/up blocks, 2x2 pixel patchify)
WAN22_VAE_CFG = dict(
    base_dim=160, decoder_base_dim=256, z_dim=48, dim_mult=[1, 2, 4, 4], num_res_blocks=2,
    temperal_downsample=[False, True, True], is_residual=True, in_channels=12, out_channels=12, patch_size=2,
    scale_factor_temporal=4, scale_factor_spatial=16,
)

# CogVideoX-5B FrameINO (stage 2): [noisy | first frame | trajectory] x 16 channels in, 48 heads x 64, 42 layers
COGVIDEOX_5B_FRAMEINO_CFG = dict(
    num_attention_heads=48, attention_head_dim=64, in_channels=48, out_channels=16, flip_sin_to_cos=True,
    freq_shift=0, time_embed_dim=512, text_embed_dim=4096, num_layers=42, sample_width=90, sample_height=60,
    sample_frames=49, patch_size=2, temporal_compression_ratio=4, max_text_seq_length=226,
    norm_elementwise_affine=True, norm_eps=1e-5, use_rotary_positional_embeddings=True,
    use_learned_positional_embeddings=True, use_FrameIn=True,
)
