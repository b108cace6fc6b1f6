"""TEST INFRASTRUCTURE — model hyper-parameter sets used by the oracle and the tests.

The released-model dicts (MODELSCOPE_UNET, VAE_DDCONFIG, LVDM_UNET, schedules) live in the product package
(sd-webui-text2video_amd/configs.py) and are re-exported here.  The TINY_* sets keep the reference topology (same block list,
same attention sites, same state-dict key set) at 1/25 of the parameters so CPU tests run
in seconds.  context_dim stays 1024: the reference hard-codes 1024 for the decoder's
SpatialTransformers (t2v_model.py:293).
"""
from sd_webui_text2video_amd.configs import (LVDM_SCHEDULE, LVDM_UNET, MODELSCOPE_UNET, SCALE_FACTOR, SCHEDULE,  # noqa: F401
                                             VAE_DDCONFIG)


TINY_UNET = dict(
    in_dim=4, dim=64, y_dim=768, context_dim=1024, out_dim=4, dim_mult=[1, 2, 4, 4],
    num_heads=2, head_dim=64, num_res_blocks=2, attn_scales=[1, 0.5, 0.25], dropout=0.1,
    temporal_attention=True)

TINY_VAE_DDCONFIG = dict(
    double_z=True, z_channels=4, resolution=64, in_channels=3, out_ch=3, ch=64,
    ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0)

# Smallest topology that still exercises head_dim 40 and 80, a down / up level and a skip 1x1 conv.
TINY_LVDM_UNET = dict(
    image_size=8, in_channels=4, out_channels=4, model_channels=320, attention_resolutions=[1, 2], num_res_blocks=1,
    channel_mult=[1, 2], num_heads=8, transformer_depth=1, context_dim=768, use_checkpoint=False, legacy=False,
    kernel_size_t=1, padding_t=0, temporal_length=16, use_relative_position=True)

