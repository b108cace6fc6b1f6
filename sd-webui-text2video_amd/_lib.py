"""ctypes binding of libt2v_hip.so (include/t2v_hip.h).  Loading is strict: if the shared
library is missing or cannot be loaded, every product entry point raises — there is no CPU
or PyTorch fallback on the product path."""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("T2V_LIB_PATH") or os.path.join(_HERE, "libt2v_hip.so")   # T2V_LIB_PATH: A/B builds of the same ABI (tools/build_variant.py)

# ---- run-time switches ---------------------------------------------------------------------
# SUPPORTED, always honoured:
#   T2V_LIB_PATH      another build of the same ABI            T2V_DEVICE_CUS   override the compute-unit count the tile policy plans for
#   T2V_COLLECTIVES   auto | lib | host (parallel.py)          T2V_RCCL_SONAME  the RCCL library to dlopen (csrc/comm.hip)
#   T2V_GN_EPI=0 / T2V_GN_COOP=0   no in-launch statistics exchange / no single-pass cooperative GroupNorm: the setting for a GPU that is
#                     SHARED with other work (the fused norms need every workgroup of a launch co-resident)
#   T2V_PRECISE       0 | 1 | r3 | all: which fp16 operand classes are split hi + lo (unet.py)
#   T2V_EXCHANGE      records | barrier (csrc/norm.hip)
# Everything else that starts with T2V_ is the A/B switch of a measured experiment and is read ONLY when
# T2V_EXPERIMENTAL=1 (tests/conftest.py and the tools/ scripts set it): a deployment cannot land on an untested combination by accident.
EXPERIMENTAL = os.environ.get("T2V_EXPERIMENTAL", "0") == "1"


def knob(name: str, default):
    """Value of an experiment switch: the environment's only under T2V_EXPERIMENTAL=1, else `default`."""
    if os.environ.get("T2V_EXPERIMENTAL", "0") == "1":
        return os.environ.get(name, default)
    return default


# ---- mirrors of include/t2v_hip.h (checked against the header by tests/test_abi.py) -------
ABI_VERSION = 8
OP_GEMM, OP_GROUPNORM, OP_LAYERNORM, OP_ATTENTION, OP_SOFTMAX = 1, 2, 3, 4, 5
OP_NCTHW_TO_CL, OP_CL_TO_NCTHW, OP_TIME_EMBED, OP_COPY2D, OP_DDIM_STEP, OP_MEMSET = 6, 7, 8, 9, 10, 11
OP_LINCOMB = 12
OP_RELPOS_ATTN = 13
OP_EMBED_ROWS = 14
OP_TO_UINT8, OP_ALLGATHER, OP_HALO_EXCHANGE = 15, 16, 17
OP_RESHARD_ROWS, OP_ALLTOALL, OP_STATS_HALO = 18, 19, 20
GATHER_PLAIN, GATHER_CONV3X3, GATHER_TCONV3, GATHER_CONV3X3_C8 = 0, 1, 2, 3
EPI_NONE, EPI_GEGLU, EPI_TATTN, EPI_STATS, EPI_GN, EPI_XATTN = 0, 1, 2, 3, 4, 5
GN_PIECES = 36                  # T2V_GN_PIECES
F16, F32 = 0, 1
EXT_SLOTS = 16
EXT_X, EXT_T, EXT_CTX, EXT_OUT, EXT_XT, EXT_XT_OUT, EXT_NOISE, EXT_EPS = 1, 2, 3, 4, 5, 6, 7, 8
OP_NI, OP_NF, OP_NP = 32, 8, 12
GN_ROWS_PER_BLOCK = 64           # T2V_GN_ROWS_PER_BLOCK
SYNC_INTS = 4096                 # T2V_SYNC_INTS
SYNC_BARRIER_INTS = 512          # T2V_SYNC_BARRIER_INTS
GN_PART_BYTES = 2 << 20          # T2V_GN_PART_BYTES

EXPORTS = [
    "t2v_abi_version", "t2v_last_error", "t2v_device_info", "t2v_run_ops", "t2v_plan_create",
    "t2v_plan_num_ops", "t2v_plan_run", "t2v_plan_run_timed", "t2v_plan_destroy",
    "t2v_unet_forward", "t2v_vae_decode", "t2v_ddim_step",
    "t2v_comm_unique_id", "t2v_comm_create", "t2v_comm_size", "t2v_comm_destroy", "t2v_plan_set_comm", "t2v_comm_all_gather",
    "t2v_comm_window_create", "t2v_comm_window_open", "t2v_comm_counters", "t2v_comm_window_kind",
    "t2v_async_status", "t2v_sync_reset", "t2v_debug_poison_exchange",
]


class T2VOp(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int32), ("tag", ctypes.c_int32),
                ("i", ctypes.c_int32 * OP_NI), ("f", ctypes.c_float * OP_NF),
                ("p", ctypes.c_uint64 * OP_NP)]


class T2VError(RuntimeError):
    pass


_lib = None


def load():
    """Load libt2v_hip.so (once).  Raises T2VError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise T2VError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    # torch ships its own libamdhip64; import it FIRST so that this library binds to the same
    # HIP runtime instance (two runtimes in one process => "no ROCm-capable device").
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    vp, u64p = ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)
    opp = ctypes.POINTER(T2VOp)
    lib.t2v_abi_version.restype = ctypes.c_int
    lib.t2v_last_error.restype = ctypes.c_char_p
    lib.t2v_device_info.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int), u64p]
    lib.t2v_run_ops.argtypes = [opp, ctypes.c_int, u64p, ctypes.c_int, vp]
    lib.t2v_plan_create.argtypes = [opp, ctypes.c_int, ctypes.POINTER(vp)]
    lib.t2v_plan_num_ops.argtypes = [vp]
    lib.t2v_plan_run.argtypes = [vp, u64p, ctypes.c_int, vp]
    lib.t2v_plan_run_timed.argtypes = [vp, u64p, ctypes.c_int, vp, ctypes.POINTER(ctypes.c_float)]
    lib.t2v_plan_destroy.argtypes = [vp]
    lib.t2v_plan_destroy.restype = None
    lib.t2v_unet_forward.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.t2v_vae_decode.argtypes = [vp, vp, vp, vp]
    lib.t2v_ddim_step.argtypes = [vp, vp, vp, vp, vp, ctypes.POINTER(ctypes.c_float), vp]
    lib.t2v_comm_unique_id.argtypes = [ctypes.c_char_p]
    lib.t2v_comm_create.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(vp)]
    lib.t2v_comm_size.argtypes = [vp]
    lib.t2v_comm_destroy.argtypes = [vp]
    lib.t2v_comm_destroy.restype = None
    lib.t2v_plan_set_comm.argtypes = [vp, vp]
    lib.t2v_comm_all_gather.argtypes = [vp, vp, ctypes.c_uint64, vp]
    lib.t2v_comm_window_create.argtypes = [vp, ctypes.c_uint64, ctypes.c_char_p]
    lib.t2v_comm_window_open.argtypes = [vp, ctypes.c_char_p]
    lib.t2v_comm_counters.argtypes = [vp, u64p]
    lib.t2v_comm_counters.restype = None
    lib.t2v_comm_window_kind.argtypes = [vp]
    lib.t2v_comm_window_kind.restype = ctypes.c_char_p
    lib.t2v_async_status.restype = ctypes.c_int
    lib.t2v_sync_reset.argtypes = [vp, vp]
    lib.t2v_debug_poison_exchange.argtypes = [ctypes.c_int]
    lib.t2v_debug_poison_exchange.restype = None
    if lib.t2v_abi_version() != ABI_VERSION:
        raise T2VError(f"libt2v_hip.so ABI {lib.t2v_abi_version()} != binding {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


_exchange_disabled = False


def exchange_disabled() -> bool:
    """True once the library has reported an asynchronous fault (T2V_ERR_ASYNC: a workgroup of a fused-norm launch gave up waiting for the
    others — the device is shared with a client that holds compute units) or refused a co-resident launch (T2V_ERR_RESIDENCY: the
    lowering's occupancy table does not hold on this device).  The library then refuses the launches that rely on a
    co-resident grid; programs are lowered WITHOUT norms fused into GEMM epilogues from then on (program.Program.gn_epilogue), and cached
    programs that have them are lowered again (unet.UNetSD.forward)."""
    return _exchange_disabled


def check(rc: int):
    global _exchange_disabled
    if rc != 0:
        msg = load().t2v_last_error()
        if rc in (-6, -7) and not (msg or b"").startswith(b"peer exchange"):
            # T2V_ERR_ASYNC / T2V_ERR_RESIDENCY: lower without fused norms from now on (exchange_disabled).  (A peer-window wait that gave
            # up is T2V_ERR_ASYNC too: the library itself sends the exchanges through RCCL from then on, the programs stay as they are.)
            _exchange_disabled = True
        raise T2VError(f"libt2v_hip error {rc}: {msg.decode() if msg else '?'}")


def async_status():
    """Raise T2VError if a kernel of an earlier run raised an asynchronous fault (include/t2v_hip.h: t2v_async_status) —
    call after synchronising at the end of a job (a video, a decode): the job's results are invalid if this raises."""
    check(load().t2v_async_status())


def device_info():
    lib = load()
    name = ctypes.create_string_buffer(64)
    cus, mem = ctypes.c_int(0), ctypes.c_uint64(0)
    check(lib.t2v_device_info(name, 64, ctypes.byref(cus), ctypes.byref(mem)))
    return name.value.decode(), cus.value, mem.value
