"""TEST INFRASTRUCTURE — model hyper-parameter sets used by the oracle, tests and bench.

MODELSCOPE_UNET are the public `configuration.json` values of
damo-vilab/modelscope-damo-text-to-video-synthesis consumed at
reference scripts/modelscope/t2v_pipeline.py:76-94; VAE_DDCONFIG is the literal dict at
t2v_pipeline.py:116-127.  The TINY_* sets keep the reference topology (same block list,
same attention sites, same state-dict key set) at 1/25 of the parameters so CPU tests run
in seconds.  context_dim stays 1024: the reference hard-codes 1024 for the decoder's
SpatialTransformers (t2v_model.py:293).
"""

MODELSCOPE_UNET = dict(
    in_dim=4, dim=320, y_dim=768, context_dim=1024, out_dim=4, dim_mult=[1, 2, 4, 4],
    num_heads=8, head_dim=64, num_res_blocks=2, attn_scales=[1, 0.5, 0.25], dropout=0.1,
    temporal_attention=True)

TINY_UNET = dict(
    in_dim=4, dim=64, y_dim=768, context_dim=1024, out_dim=4, dim_mult=[1, 2, 4, 4],
    num_heads=2, head_dim=64, num_res_blocks=2, attn_scales=[1, 0.5, 0.25], dropout=0.1,
    temporal_attention=True)

VAE_DDCONFIG = dict(
    double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128,
    ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0)

TINY_VAE_DDCONFIG = dict(
    double_z=True, z_channels=4, resolution=64, in_channels=3, out_ch=3, ch=64,
    ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0)

SCALE_FACTOR = 0.18215          # t2v_pipeline.py:297
SCHEDULE = dict(num_timesteps=1000, init_beta=0.00085, last_beta=0.0120)   # t2v_pipeline.py:107-111


# VideoCrafter base text-to-video model: reference scripts/videocrafter/base_t2v/model_config.yaml:21-46 (UNet),
# :48-66 (first stage = the same AutoencoderKL ddconfig as ModelScope), linear schedule 0.00085..0.012.
LVDM_UNET = dict(
    image_size=32, in_channels=4, out_channels=4, model_channels=320, attention_resolutions=[4, 2, 1], num_res_blocks=2,
    channel_mult=[1, 2, 4, 4], num_heads=8, transformer_depth=1, context_dim=768, use_checkpoint=False, legacy=False,
    kernel_size_t=1, padding_t=0, temporal_length=16, use_relative_position=True)

# Smallest topology that still exercises head_dim 40 and 80, a down / up level and a skip 1x1 conv.
TINY_LVDM_UNET = dict(
    image_size=8, in_channels=4, out_channels=4, model_channels=320, attention_resolutions=[1, 2], num_res_blocks=1,
    channel_mult=[1, 2], num_heads=8, transformer_depth=1, context_dim=768, use_checkpoint=False, legacy=False,
    kernel_size_t=1, padding_t=0, temporal_length=16, use_relative_position=True)

LVDM_SCHEDULE = dict(timesteps=1000, linear_start=0.00085, linear_end=0.012)
