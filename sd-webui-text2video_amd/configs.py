"""Model hyper-parameter sets of the released checkpoints (plain dicts; no arithmetic).

MODELSCOPE_UNET are the public `configuration.json` values of damo-vilab/modelscope-damo-text-to-video-synthesis
consumed at reference scripts/modelscope/t2v_pipeline.py:76-94 (ZeroScope reuses them); VAE_DDCONFIG is the literal
dict at t2v_pipeline.py:116-127; LVDM_UNET / LVDM_SCHEDULE are scripts/videocrafter/base_t2v/model_config.yaml:4-46
(its first stage uses VAE_DDCONFIG as well, :52-66)."""

MODELSCOPE_UNET = dict(
    in_dim=4, dim=320, y_dim=768, context_dim=1024, out_dim=4, dim_mult=[1, 2, 4, 4],
    num_heads=8, head_dim=64, num_res_blocks=2, attn_scales=[1, 0.5, 0.25], dropout=0.1,
    temporal_attention=True)

VAE_DDCONFIG = dict(
    double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128,
    ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0)

SCALE_FACTOR = 0.18215          # t2v_pipeline.py:297
SCHEDULE = dict(num_timesteps=1000, init_beta=0.00085, last_beta=0.0120)   # t2v_pipeline.py:107-111

LVDM_UNET = dict(
    image_size=32, in_channels=4, out_channels=4, model_channels=320, attention_resolutions=[4, 2, 1], num_res_blocks=2,
    channel_mult=[1, 2, 4, 4], num_heads=8, transformer_depth=1, context_dim=768, use_checkpoint=False, legacy=False,
    kernel_size_t=1, padding_t=0, temporal_length=16, use_relative_position=True)

LVDM_SCHEDULE = dict(timesteps=1000, linear_start=0.00085, linear_end=0.012)
