"""TEST INFRASTRUCTURE — not part of the product path.

Imports the *real* reference implementation (kabachuha/sd-webui-text2video, mounted
read-only at /root/reference) inside this container by stubbing the packages it needs
but which are not installed (AUTOMATIC1111 `modules.*`, Stability `ldm.*`, `omegaconf`).
Nothing from the reference is copied: the reference classes are imported where they lie.

Only `tests/golden/make_golden.py` and `tests/test_oracle_pin.py` use this module, to
  (a) generate the committed golden fixtures under tests/golden/, and
  (b) pin `oracle/torch_port.py` (the travelling CPU restatement) against the reference.
/root/reference does not exist on the GPU box, so nothing that runs there may import this.

Stub list follows SURVEY.md §8(c) / Appendix E.
"""
from __future__ import annotations

import importlib
import importlib.util
import os
import sys
import types

import torch

REFERENCE_ROOT = os.environ.get("T2V_REFERENCE_ROOT", "/root/reference")
_SCRIPTS = os.path.join(REFERENCE_ROOT, "scripts")
_BOOTSTRAPPED = None


def reference_available() -> bool:
    return os.path.isdir(_SCRIPTS)


class _Opts(types.SimpleNamespace):
    def __getattr__(self, k):  # opts.<anything> -> None
        return None


def _mod(name: str, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _load_file(path: str, name: str):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def bootstrap(sdp_attention: bool = True):
    """Return a namespace with the reference's own classes (imported unmodified).

    .t2v      -> scripts/modelscope/t2v_model.py      (UNetSD, AutoencoderKL, beta_schedule, _i)
    .samplers -> scripts/samplers/samplers_common.py  (Txt2VideoSampler, available_samplers)
    .ae       -> videocrafter/.../autoencoder_modules.py (Decoder twin of ldm's)
    .state    -> the stubbed webui `shared.state`
    """
    global _BOOTSTRAPPED
    if _BOOTSTRAPPED is not None:
        return _BOOTSTRAPPED
    if not reference_available():
        raise RuntimeError(f"reference not mounted at {REFERENCE_ROOT}")
    if _SCRIPTS not in sys.path:
        sys.path.insert(0, _SCRIPTS)

    state = types.SimpleNamespace(interrupted=False, skipped=False, sampling_step=0,
                                  sampling_steps=0, job_count=0, job_no=0, job="")
    shared = _mod("modules.shared", opts=_Opts(data={}),
                  cmd_opts=types.SimpleNamespace(opt_sdp_attention=sdp_attention),
                  state=state, device=torch.device("cpu"), xformers_available=False)
    pp = _mod("modules.prompt_parser", reconstruct_cond_batch=lambda c, i: c)
    interrupted = type("InterruptedException", (BaseException,), {})
    sc = _mod("modules.sd_samplers_common", InterruptedException=interrupted)
    ho = _mod("modules.sd_hijack_optimizations", get_xformers_flash_attention_op=lambda q, k, v: None)
    _mod("modules", shared=shared, prompt_parser=pp, sd_samplers_common=sc, sd_hijack_optimizations=ho)
    _mod("modules.paths", models_path="/nonexistent")
    for n in ("ldm", "ldm.modules", "ldm.modules.diffusionmodules", "ldm.modules.distributions"):
        _mod(n)
    _mod("ldm.util", instantiate_from_config=None)
    if "omegaconf" not in sys.modules:
        _mod("omegaconf")
        _mod("omegaconf.listconfig", ListConfig=type("ListConfig", (list,), {}))

    vc = os.path.join(_SCRIPTS, "videocrafter/lvdm/models/modules/")
    ae = _load_file(vc + "autoencoder_modules.py", "_ref_vc_autoencoder_modules")
    di = _load_file(vc + "distributions.py", "_ref_vc_distributions")
    vu = importlib.import_module("videocrafter.lvdm.models.modules.util")
    _mod("ldm.modules.diffusionmodules.model", Decoder=ae.Decoder, Encoder=ae.Encoder)
    _mod("ldm.modules.diffusionmodules.util",
         make_beta_schedule=vu.make_beta_schedule, make_ddim_timesteps=vu.make_ddim_timesteps,
         make_ddim_sampling_parameters=vu.make_ddim_sampling_parameters,
         extract_into_tensor=vu.extract_into_tensor,
         noise_like=lambda shape, device, repeat=False: torch.randn(shape, device=device))
    _mod("ldm.modules.distributions.distributions",
         DiagonalGaussianDistribution=di.DiagonalGaussianDistribution)

    t2v = importlib.import_module("modelscope.t2v_model")
    smp = importlib.import_module("samplers.samplers_common")
    _BOOTSTRAPPED = types.SimpleNamespace(t2v=t2v, samplers=smp, ae=ae, state=state,
                                          shared=shared, InterruptedException=interrupted)
    return _BOOTSTRAPPED


def build_reference_unet(cfg: dict):
    """Instantiate the reference UNetSD (t2v_model.py:98) from a `oracle.configs` dict."""
    ref = bootstrap()
    unet = ref.t2v.UNetSD(
        in_dim=cfg["in_dim"], dim=cfg["dim"], y_dim=cfg.get("y_dim", 768),
        context_dim=cfg["context_dim"], out_dim=cfg["out_dim"], dim_mult=list(cfg["dim_mult"]),
        num_heads=cfg["num_heads"], head_dim=cfg["head_dim"], num_res_blocks=cfg["num_res_blocks"],
        attn_scales=list(cfg["attn_scales"]), dropout=cfg.get("dropout", 0.1),
        temporal_attention=cfg.get("temporal_attention", True)).eval()
    betas = ref.t2v.beta_schedule("linear_sd", 1000, init_beta=0.00085, last_beta=0.0120)
    unet.register_schedule(given_betas=betas.numpy())
    return unet, betas


def build_reference_vae(ddconfig: dict, embed_dim: int = 4):
    ref = bootstrap()
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):  # the Decoder ctor prints its z shape
        vae = ref.t2v.AutoencoderKL(dict(ddconfig), embed_dim, None).eval()
    return vae


def bootstrap_pipeline():
    """The reference's outer entry point module scripts/modelscope/t2v_pipeline.py (TextToVideoSynthesis.infer,
    tensor2vid, postprocess_video), imported unmodified.  Extra stubs: cv2 (only `cvtColor(img, COLOR_RGB2BGR)` is
    called, postprocess_video :431), open_clip and the webui modules clip_hardcode.py imports at module top
    (the text encoder is outside the hot path: callers install a stand-in `preprocess`)."""
    ref = bootstrap()
    if getattr(ref, "pipeline", None) is not None:
        return ref.pipeline
    import contextlib
    import numpy as np
    if "cv2" not in sys.modules:
        _mod("cv2", COLOR_RGB2BGR=4, cvtColor=lambda img, code: np.ascontiguousarray(img[:, :, ::-1]))
    if "open_clip" not in sys.modules:
        _mod("open_clip", tokenizer=types.SimpleNamespace(_tokenizer=types.SimpleNamespace(encoder={}, decoder={})))
    mods = sys.modules["modules"]
    dev = _mod("modules.devices", autocast=contextlib.nullcontext, torch_gc=lambda: None, device=torch.device("cpu"))
    gp = _mod("modules.generation_parameters_copypaste", quote=lambda v: str(v))
    hj = _mod("modules.sd_hijack", model_hijack=types.SimpleNamespace())
    ti = _mod("modules.textual_inversion")
    ti.textual_inversion = _mod("modules.textual_inversion.textual_inversion", Embedding=object,
                                EmbeddingDatabase=lambda *a, **k: types.SimpleNamespace())
    mods.devices, mods.generation_parameters_copypaste, mods.sd_hijack, mods.textual_inversion = dev, gp, hj, ti
    ref.pipeline = importlib.import_module("modelscope.t2v_pipeline")
    return ref.pipeline
