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
        "T2V_EPI_NONE": L.EPI_NONE, "T2V_EPI_GEGLU": L.EPI_GEGLU, "T2V_EPI_TATTN": L.EPI_TATTN, "T2V_F16": L.F16, "T2V_F32": L.F32,
        "T2V_EXT_SLOTS": L.EXT_SLOTS, "T2V_EXT_X": L.EXT_X, "T2V_EXT_T": L.EXT_T, "T2V_EXT_CTX": L.EXT_CTX,
        "T2V_EXT_OUT": L.EXT_OUT, "T2V_EXT_XT": L.EXT_XT, "T2V_EXT_XT_OUT": L.EXT_XT_OUT,
        "T2V_EXT_NOISE": L.EXT_NOISE, "T2V_EXT_EPS": L.EXT_EPS, "T2V_OP_NI": L.OP_NI, "T2V_OP_NF": L.OP_NF,
        "T2V_OP_NP": L.OP_NP, "T2V_GN_ROWS_PER_BLOCK": L.GN_ROWS_PER_BLOCK, "T2V_SYNC_INTS": L.SYNC_INTS, "T2V_SYNC_BARRIER_INTS": L.SYNC_BARRIER_INTS,
        "T2V_OP_EMBED_ROWS": L.OP_EMBED_ROWS, "T2V_OP_TO_UINT8": L.OP_TO_UINT8, "T2V_OP_ALLGATHER": L.OP_ALLGATHER,
        "T2V_OP_HALO_EXCHANGE": L.OP_HALO_EXCHANGE, "T2V_OP_RESHARD_ROWS": L.OP_RESHARD_ROWS, "T2V_OP_ALLTOALL": L.OP_ALLTOALL,
        "T2V_OP_STATS_HALO": L.OP_STATS_HALO, "T2V_GN_PART_BYTES": L.GN_PART_BYTES,
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


def test_validation_covers_every_op_kind_without_gpu(built_lib):
    """`t2v_plan_create` / `t2v_run_ops` validate the records of EVERY op kind (not only GEMM) before any HIP call;
    each malformed record below is refused with T2V_ERR_BAD_ARG and a message naming the problem."""
    h = ctypes.c_void_p()
    ptr = 0x1000                                                      # any non-null, non-slot value: nothing is dereferenced

    def refused(kind, i=(), f=(), p=(), needle=b""):
        op = (L.T2VOp * 1)()
        op[0].kind = kind
        for k, v in enumerate(i):
            op[0].i[k] = v
        for k, v in enumerate(f):
            op[0].f[k] = v
        for k, v in enumerate(p):
            op[0].p[k] = v
        rc = built_lib.t2v_plan_create(op, 1, ctypes.byref(h))
        msg = built_lib.t2v_last_error()
        assert rc == -1 and needle in msg, (kind, rc, msg)

    refused(L.OP_GROUPNORM, i=(1, 16, 330, 336, 32, 1, 1, 336), p=(ptr,) * 5, needle=b"GroupNorm")          # C % groups != 0
    refused(L.OP_LAYERNORM, i=(8, 4096, 4096, 4096), p=(ptr,) * 4, needle=b"LayerNorm")                      # C > 2048
    refused(L.OP_ATTENTION, i=(4, 4, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 48), f=(1.0,), p=(ptr,) * 4, needle=b"head_dim")
    refused(L.OP_ATTENTION, i=(4, 5, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 64, 1), f=(1.0,), p=(ptr,) * 4, needle=b"causal")
    refused(L.OP_RELPOS_ATTN, i=(40, 40, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 64, 4), p=(ptr,) * 6, needle=b"relative-position")
    refused(L.OP_RELPOS_ATTN, i=(4, 16, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 64, 4, 14), p=(ptr,) * 6, needle=b"inside the keys")
    refused(L.OP_SOFTMAX, i=(4, 64, 32, 64), p=(ptr, ptr), needle=b"softmax")                                # ld < cols
    refused(L.OP_COPY2D, i=(4, 6, 8, 8), p=(ptr, ptr), needle=b"cols % 4")
    refused(L.OP_DDIM_STEP, i=(4, 0), p=(ptr, ptr, 0, ptr), needle=b"DDIM")                                  # inner == 0 (division by zero on the device)
    refused(L.OP_DDIM_STEP, i=(6, 64, 0, 0, 0, 0, 4), p=(ptr, ptr, 0, ptr), needle=b"DDIM")                  # C % channels-per-sample
    refused(L.OP_LINCOMB, i=(16, 7), p=(ptr,) * 7, needle=b"lincomb")
    refused(L.OP_TIME_EMBED, i=(2, 321), p=(ptr,) * 3, needle=b"time-embedding")
    refused(L.OP_TO_UINT8, i=(1, 3, 0, 8, 8), p=(ptr, ptr), needle=b"uint8")
    refused(L.OP_ALLGATHER, i=(512, 0, 2, 2), p=(ptr,), needle=b"all-gather")                                # part >= nparts
    refused(L.OP_HALO_EXCHANGE, i=(512, 0, 0, -1, -1), p=(ptr,), needle=b"halo")
    refused(L.OP_STATS_HALO, i=(512, 0, 2, 1, 4096, 0, 3, 0, 2), p=(ptr, ptr), needle=b"statistics + halo")   # next rank outside the communicator
    refused(L.OP_STATS_HALO, i=(512, 0, 2, 1, 4096, 0, 3, 0, -1), p=(ptr,), needle=b"statistics + halo")      # no raw buffer
    refused(L.OP_RESHARD_ROWS, i=(8, 64, 0, 4, 4, 64, 64, 0), p=(ptr, ptr), needle=b"resharding")            # chunk of 0 rows
    refused(L.OP_ALLTOALL, i=(512, 0, 4, 1, 3, 4, 0), p=(ptr, ptr), needle=b"all-to-all")                    # last slice longer than the others
    refused(L.OP_GEMM, i=(256, 640, 64, 64, 64, 640, 0, 0, 1, 640, 0, 0, 0, 0, 0, 0, 0, 1, 0, 1, 0, 0, 8), p=(ptr, ptr, 0, ptr, 0, ptr, 0, ptr),
            needle=b"fused LayerNorm")                                                                        # N != 320 on the LN-fused form
    # and a well-formed record of each collective kind is accepted (nothing runs at plan creation)
    ok = (L.T2VOp * 3)()
    ok[0].kind, ok[1].kind, ok[2].kind = L.OP_ALLGATHER, L.OP_ALLTOALL, L.OP_STATS_HALO
    for k, v in enumerate((512, 0, 4, 1, 4096, 0, 3, 0, 2)):
        ok[2].i[k] = v
    ok[2].p[0], ok[2].p[1] = ptr, ptr
    ok[0].i[0], ok[0].i[2], ok[0].i[3], ok[0].p[0] = 512, 4, 1, ptr
    for k, v in enumerate((1024, 0, 4, 1, 32, 29, 1)):
        ok[1].i[k] = v
    ok[1].p[0], ok[1].p[1] = ptr, ptr
    assert built_lib.t2v_plan_create(ok, 3, ctypes.byref(h)) == 0 and built_lib.t2v_plan_num_ops(h) == 3
    built_lib.t2v_plan_destroy(h)


def test_product_path_has_no_cpu_fallback():
    import pytest
    import torch
    from oracle import configs
    from sd_webui_text2video_amd import unet
    net = unet.UNetSD(**configs.TINY_UNET, init_weights=False)
    with pytest.raises(L.T2VError):
        net(torch.zeros(1, 4, 2, 8, 8), torch.tensor([1]), torch.zeros(1, 7, 1024))


def test_missing_rccl_is_an_error_code_not_a_crash(built_lib):
    """with an unloadable librccl the communicator entry points return T2V_ERR_COMM and a message (the loader
    used to call dlerror() twice and build a std::string from the NULL the second call returns -> SIGSEGV in rank 0 while the
    other ranks hung in the id broadcast).  Forced here with T2V_RCCL_SONAME; runs in a child process because the library
    resolves RCCL once per process."""
    import subprocess
    import sys
    code = (
        "import ctypes, sys\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "from sd_webui_text2video_amd import _lib as L\n"
        "lib = L.load()\n"
        "buf = ctypes.create_string_buffer(128)\n"
        "rc = lib.t2v_comm_unique_id(buf)\n"
        "h = ctypes.c_void_p()\n"
        "rc2 = lib.t2v_comm_create(buf, 2, 0, ctypes.byref(h))\n"
        "print('RC', rc, rc2, lib.t2v_last_error().decode())\n")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                         env={**os.environ, "T2V_RCCL_SONAME": "/nonexistent/librccl.so.1"})
    assert out.returncode == 0, out.stderr[-2000:]
    line = next(ln for ln in out.stdout.splitlines() if ln.startswith("RC "))
    assert line.startswith("RC -5 -5 ") and "cannot dlopen librccl" in line and "/nonexistent/librccl.so.1" in line


def test_validation_of_round3_records_without_gpu(built_lib):
    """T2V_EPI_TATTN (fused QKV + temporal attention) records are validated before any launch: tile 10 and only tile 10, head-major
    N = 192 * heads, 2 <= F <= 32, pixels * F <= 192, M = samples * ceil(HW / pixels) * 192, no bias / residual."""
    h = ctypes.c_void_p()
    ptr = 0x1000

    def make(**kw):
        op = (L.T2VOp * 1)()
        op[0].kind = L.OP_GEMM
        i = dict({0: 2 * 8 * 192, 1: 5 * 192, 2: 320, 3: 320, 4: 320, 5: 320, 7: L.GATHER_PLAIN, 8: 24, 9: 64, 10: 8, 16: L.EPI_TATTN, 17: L.F16, 19: 1, 22: 10})
        i.update(kw.get("i", {}))
        for k, v in i.items():
            op[0].i[k] = v
        op[0].f[1] = kw.get("scale", 0.125)
        pp = {0: ptr, 1: ptr, 5: ptr}
        pp.update(kw.get("p", {}))
        for k, v in pp.items():
            op[0].p[k] = v
        return op

    assert built_lib.t2v_plan_create(make(), 1, ctypes.byref(h)) == 0
    built_lib.t2v_plan_destroy(h)
    for bad, needle in ((dict(i={22: 8}), b"tile 10"), (dict(i={16: 0, 8: 0}), b"tile 10"), (dict(i={8: 40}), b"F <= 32"), (dict(i={10: 9}), b"pixels * F"),
                        (dict(i={0: 192 * 15}), b"samples * ceil"), (dict(i={1: 5 * 192 + 64}), b"N = 192"), (dict(p={2: ptr}), b"no bias"),
                        (dict(scale=0.0), b"positive scale"), (dict(i={17: L.F32}), b"fp16 out"), (dict(i={5: 256}), b"ldc")):
        rc = built_lib.t2v_plan_create(make(**bad), 1, ctypes.byref(h))
        msg = built_lib.t2v_last_error()
        assert rc == -1 and needle in msg, (bad, rc, msg)
    # copy2d: the low-order output exists for fp32 -> fp16 casts only
    op = (L.T2VOp * 1)()
    op[0].kind = L.OP_COPY2D
    for k, v in enumerate((4, 8, 8, 8, L.F16, L.F16)):
        op[0].i[k] = v
    op[0].p[0], op[0].p[1], op[0].p[2] = ptr, ptr, ptr
    assert built_lib.t2v_plan_create(op, 1, ctypes.byref(h)) == -1 and b"low-order" in built_lib.t2v_last_error()
