"""MI355X-native ModelScope / ZeroScope text-to-video denoising hot path
(3-D UNet sampling loop + VAE decode) behind the reference's own Python entry points.

Import as `sd_webui_text2video_amd` (see ../sd_webui_text2video_amd.py)."""
__version__ = "0.1.0"
