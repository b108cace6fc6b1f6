"""CPU: the C-ABI library builds, loads and exports every symbol include/t2v_hip.h declares;
the Python mirror of the header constants is consistent."""
import ctypes
import os
import re

from sd_webui_text2video_amd import _lib as L

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
HEADER = open(os.path.join(ROOT, "include", "t2v_hip.h")).read()


def test_header_constants_match_binding():
    defs = dict(re.findall(r"#define\s+(T2V_\w+)\s+\(?(-?\d+)\)?", HEADER))
    enums = dict(re.findall(r"(T2V_\w+)\s*=\s*(\d+)", HEADER))
    allc = {**{k: int(v) for k, v in defs.items()}, **{k: int(v) for k, v in enums.items()}}
    expect = {
        "T2V_ABI_VERSION": L.ABI_VERSION, "T2V_OP_GEMM": L.OP_GEMM, "T2V_OP_GROUPNORM": L.OP_GROUPNORM,
        "T2V_OP_LAYERNORM": L.OP_LAYERNORM, "T2V_OP_ATTENTION": L.OP_ATTENTION, "T2V_OP_SOFTMAX": L.OP_SOFTMAX,
        "T2V_OP_NCTHW_TO_CL": L.OP_NCTHW_TO_CL, "T2V_OP_CL_TO_NCTHW": L.OP_CL_TO_NCTHW,
        "T2V_OP_TIME_EMBED": L.OP_TIME_EMBED, "T2V_OP_COPY2D": L.OP_COPY2D, "T2V_OP_DDIM_STEP": L.OP_DDIM_STEP,
        "T2V_OP_MEMSET": L.OP_MEMSET, "T2V_OP_LINCOMB": L.OP_LINCOMB, "T2V_OP_RELPOS_ATTN": L.OP_RELPOS_ATTN, "T2V_GATHER_PLAIN": L.GATHER_PLAIN, "T2V_GATHER_CONV3X3": L.GATHER_CONV3X3,
        "T2V_GATHER_TCONV3": L.GATHER_TCONV3, "T2V_GATHER_CONV3X3_C8": L.GATHER_CONV3X3_C8,
        "T2V_EPI_NONE": L.EPI_NONE, "T2V_EPI_GEGLU": L.EPI_GEGLU, "T2V_F16": L.F16, "T2V_F32": L.F32,
        "T2V_EXT_SLOTS": L.EXT_SLOTS, "T2V_EXT_X": L.EXT_X, "T2V_EXT_T": L.EXT_T, "T2V_EXT_CTX": L.EXT_CTX,
        "T2V_EXT_OUT": L.EXT_OUT, "T2V_EXT_XT": L.EXT_XT, "T2V_EXT_XT_OUT": L.EXT_XT_OUT,
        "T2V_EXT_NOISE": L.EXT_NOISE, "T2V_EXT_EPS": L.EXT_EPS, "T2V_OP_NI": L.OP_NI, "T2V_OP_NF": L.OP_NF,
        "T2V_OP_NP": L.OP_NP, "T2V_GN_ROWS_PER_BLOCK": L.GN_ROWS_PER_BLOCK,
    }
    for k, v in expect.items():
        assert allc[k] == v, (k, allc[k], v)
    assert ctypes.sizeof(L.T2VOp) == 8 + 4 * L.OP_NI + 4 * L.OP_NF + 8 * L.OP_NP


def test_library_exports_every_declared_symbol(built_lib):
    declared = set(re.findall(r"\b(t2v_\w+)\s*\(", HEADER))
    declared -= {"t2v_plan_run)"}
    assert set(L.EXPORTS) <= declared, set(L.EXPORTS) - declared
    for name in declared:
        assert hasattr(built_lib, name), f"{name} not exported by libt2v_hip.so"
    assert built_lib.t2v_abi_version() == L.ABI_VERSION


def test_validation_rejects_bad_programs_without_gpu(built_lib):
    """Argument validation runs before any HIP call, so it is testable on a CPU-only host."""
    op = (L.T2VOp * 1)()
    op[0].kind = 99
    h = ctypes.c_void_p()
    assert built_lib.t2v_plan_create(op, 1, ctypes.byref(h)) == -1
    assert b"unknown op kind" in built_lib.t2v_last_error()
    op[0].kind = L.OP_GEMM
    op[0].i[0], op[0].i[1], op[0].i[2] = 128, 6, 64          # N not a multiple of 4
    assert built_lib.t2v_plan_create(op, 1, ctypes.byref(h)) == -1
    assert b"multiple of 4" in built_lib.t2v_last_error()


def test_product_path_has_no_cpu_fallback():
    import pytest
    import torch
    from oracle import configs
    from sd_webui_text2video_amd import unet
    net = unet.UNetSD(**configs.TINY_UNET, init_weights=False)
    with pytest.raises(L.T2VError):
        net(torch.zeros(1, 4, 2, 8, 8), torch.tensor([1]), torch.zeros(1, 7, 1024))
