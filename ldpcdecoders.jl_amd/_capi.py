"""ctypes binding of libldpc_mi355x.so (include/ldpc_mi355x.h).

This is the same C ABI the Julia `ccall` shim binds.  There is
no fallback of any kind: a missing library or a missing gfx950 device raises.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# LDPC_MI355X_LIB selects another build of the same library (tuning experiments, installed copies)
LIB_PATH = os.environ.get("LDPC_MI355X_LIB") or os.path.join(CSRC, "libldpc_mi355x.so")
# The experiments build (-DLDPC_EXPERIMENTS): the same code plus the environment knobs and the fault injection that
# tests and tools use to force a code path.  The product library above reads no environment variable.
EXP_LIB_PATH = os.environ.get("LDPC_MI355X_EXP_LIB") or os.path.join(CSRC, "libldpc_mi355x_exp.so")

LDPC_OK = 0
STATUS_NAMES = {
    0: "LDPC_OK",
    1: "LDPC_ERR_INVALID_ARGUMENT",
    2: "LDPC_ERR_NO_DEVICE",
    3: "LDPC_ERR_HIP",
    4: "LDPC_ERR_OUT_OF_MEMORY",
    5: "LDPC_ERR_UNSUPPORTED",
}

# every symbol include/ldpc_mi355x.h declares
EXPORTED_SYMBOLS = (
    "ldpc_abi_version",
    "ldpc_build_target",
    "ldpc_last_error",
    "ldpc_device_count",
    "ldpc_trim_memory",
    "ldpc_set_wait_limit_ms",
    "ldpc_get_wait_limit_ms",
    "ldpc_bp_create",
    "ldpc_bp_destroy",
    "ldpc_bp_get_info",
    "ldpc_bp_decode_batch",
    "ldpc_bp_decode_batch_device",
    "ldpc_bp_last_status",
    "ldpc_bp_last_timing",
    "ldpc_bp_call_timing",
    "ldpc_bp_call_phase_ticks",
    "ldpc_bp_create_multi",
    "ldpc_bp_destroy_multi",
    "ldpc_bp_multi_handle",
    "ldpc_bp_decode_batch_multi",
    "ldpc_bp_decode_batch_multi_device",
    "ldpc_bp_multi_last_status",
    "ldpc_bp_multi_get_info",
    "ldpc_osd_create",
    "ldpc_osd_destroy",
    "ldpc_osd_postprocess_batch",
    "ldpc_bpots_create",
    "ldpc_bpots_destroy",
    "ldpc_bpots_kernel",
    "ldpc_bpots_decode_batch",
    "ldpc_bpots_decode_batch_device",
)
# ... and include/ldpc_mi355x_debug.h (test hooks, not part of the boundary)
DEBUG_SYMBOLS = ("ldpc_debug_team_rows", "ldpc_debug_team_plan", "ldpc_debug_div_check", "ldpc_debug_process_state",
                 "ldpc_debug_adopt_process_state", "ldpc_debug_team_irr", "ldpc_debug_llr_check")

MULTI_MAX_DEVICES = 16
EXCHANGE_AUTO, EXCHANGE_COPY, EXCHANGE_RCCL, EXCHANGE_NONE = 0, 1, 2, 3


class LdpcError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {message}")
        self.status = status
        self.message = message


class BPInfo(ctypes.Structure):
    _fields_ = [
        ("s", ctypes.c_int64), ("n", ctypes.c_int64), ("nnz", ctypes.c_int64),
        ("max_iters", ctypes.c_int64), ("per", ctypes.c_double),
        ("max_check_degree", ctypes.c_int32), ("max_bit_degree", ctypes.c_int32),
        ("device", ctypes.c_int32), ("tile_syndromes", ctypes.c_int32),
        ("waves_per_tile", ctypes.c_int32), ("resident_tiles", ctypes.c_int32),
        ("workspace_bytes", ctypes.c_int64),
        ("last_kernel", ctypes.c_int32), ("last_team_size", ctypes.c_int32),
        ("last_lds_rows", ctypes.c_int32), ("last_rows_on_chip", ctypes.c_int32), ("reserved_info", ctypes.c_int32 * 2),
    ]


class BPMultiInfo(ctypes.Structure):
    _fields_ = [
        ("ndev", ctypes.c_int32), ("exchange", ctypes.c_int32), ("devices", ctypes.c_int32 * MULTI_MAX_DEVICES),
        ("scatter_ms", ctypes.c_double), ("root_decode_ms", ctypes.c_double), ("gather_ms", ctypes.c_double),
        ("decode_ms_max", ctypes.c_double),
        ("scatter_bytes_per_peer", ctypes.c_int64), ("gather_bytes_per_peer", ctypes.c_int64),
    ]


class BPOptions(ctypes.Structure):
    _fields_ = [
        ("device", ctypes.c_int32), ("waves_per_tile", ctypes.c_int32),
        ("resident_tiles", ctypes.c_int32), ("kernel_variant", ctypes.c_int32),
        ("defer_threshold", ctypes.c_int32), ("llr_exact", ctypes.c_int32), ("reserved", ctypes.c_int32 * 10),
    ]


def build(force: bool = False) -> str:
    """Compile the HIP library for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in ("ldpc_mi355x.hip", "ldpc_multi.hip", "host_env.hpp", "host_wait.hpp", "pick_tile.hip", "pick_lds.hip", "pick_node.hip", "pick_team.hip", "pickers.hpp",
                                             "ldpc_bpots.hip", "osd_host.cpp", "bp_kernels.hpp", "bp_lds_kernels.hpp", "bp_node_kernels.hpp", "bp_team_kernels.hpp", "latency_mode.hpp",
                                             "bpots_kernels.hpp", "portable_math.h", "Makefile")]
    srcs.append(os.path.join(_HERE, "..", "include", "ldpc_mi355x.h"))
    srcs.append(os.path.join(_HERE, "..", "include", "ldpc_mi355x_debug.h"))
    for target in (LIB_PATH, EXP_LIB_PATH):
        stale = (not os.path.exists(target)) or any(
            os.path.getmtime(f) > os.path.getmtime(target) for f in srcs if os.path.exists(f))
        if force or stale:
            subprocess.check_call(["make", "-s", "-C", CSRC, "all"])
            break
    return LIB_PATH


_LIBS = {}


def _one_hip_runtime() -> None:
    """Keep ONE HIP runtime in the process.  PyTorch-ROCm bundles its own libamdhip64.so.7;
    the library links against the same SONAME, so whichever copy is mapped first serves both.
    When torch is installed it must be that copy, or torch tensors / streams handed to
    ldpc_bp_decode_batch_device would belong to a different runtime instance.  Without torch
    (plain ctypes or the Julia shim) the system ROCm runtime is used."""
    if os.environ.get("LDPC_MI355X_NO_TORCH_PRELOAD"):
        return
    try:
        import torch  # noqa: F401
    except Exception:
        pass


def lib(experiments: bool = False) -> ctypes.CDLL:
    """Load the HIP library; raises (never falls back) if it is not built.  experiments=True: the build with the
    environment knobs and the fault injection (tests, tools); both builds may live in one process."""
    if experiments in _LIBS:
        return _LIBS[experiments]
    path = EXP_LIB_PATH if experiments else LIB_PATH
    if not os.path.exists(path):
        raise LdpcError(2, f"{path} is not built; run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "or `make -C ldpcdecoders.jl_amd/csrc`. There is no CPU fallback.")
    _one_hip_runtime()
    L = ctypes.CDLL(path)
    vp, i64, i32, f64 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_double
    L.ldpc_abi_version.restype = i32
    L.ldpc_build_target.restype = ctypes.c_char_p
    L.ldpc_last_error.restype = ctypes.c_char_p
    L.ldpc_device_count.restype = i32
    L.ldpc_trim_memory.restype = i32
    L.ldpc_set_wait_limit_ms.restype = i32
    L.ldpc_set_wait_limit_ms.argtypes = [i64]
    L.ldpc_get_wait_limit_ms.restype = i64
    L.ldpc_debug_team_plan.restype = i32
    L.ldpc_debug_team_plan.argtypes = [i64, i64, i64, i32, i32, ctypes.POINTER(i32 * 6)]
    L.ldpc_debug_div_check.restype = i32
    L.ldpc_debug_div_check.argtypes = [i64, vp, vp, vp, vp]
    L.ldpc_debug_llr_check.restype = i32
    L.ldpc_debug_llr_check.argtypes = [i64, vp, vp, vp]
    L.ldpc_debug_team_irr.restype = i32
    L.ldpc_debug_team_irr.argtypes = [i64, i64, vp, vp, i32, i32, i32, ctypes.POINTER(i32 * 2), vp, vp, vp, vp, vp]
    L.ldpc_debug_team_rows.restype = i32
    L.ldpc_debug_team_rows.argtypes = [i64, i64, vp, vp, i32, i32, i32, ctypes.POINTER(i32 * 2), ctypes.POINTER(i32 * 5), vp, vp, vp, vp]
    L.ldpc_bp_create_multi.restype = i32
    L.ldpc_bp_create_multi.argtypes = [i32, ctypes.POINTER(i32), i32, i64, i64, i64, vp, vp, f64, i64, ctypes.POINTER(BPOptions), ctypes.POINTER(vp)]
    L.ldpc_bp_destroy_multi.restype = i32
    L.ldpc_bp_destroy_multi.argtypes = [vp]
    L.ldpc_bp_multi_handle.restype = vp
    L.ldpc_bp_multi_handle.argtypes = [vp, i32]
    L.ldpc_bp_decode_batch_multi.restype = i32
    L.ldpc_bp_decode_batch_multi.argtypes = [vp, i64, vp, vp, vp, vp, vp]
    L.ldpc_bp_decode_batch_multi_device.restype = i32
    L.ldpc_bp_decode_batch_multi_device.argtypes = [vp, i64, vp, vp, vp, vp, vp, vp]
    L.ldpc_bp_multi_last_status.restype = i32
    L.ldpc_bp_multi_last_status.argtypes = [vp]
    L.ldpc_bp_multi_get_info.restype = i32
    L.ldpc_bp_multi_get_info.argtypes = [vp, ctypes.POINTER(BPMultiInfo)]
    L.ldpc_bp_create.restype = i32
    L.ldpc_bp_create.argtypes = [i64, i64, i64, vp, vp, f64, i64, ctypes.POINTER(BPOptions), ctypes.POINTER(vp)]
    L.ldpc_bp_destroy.restype = i32
    L.ldpc_bp_destroy.argtypes = [vp]
    L.ldpc_bp_get_info.restype = i32
    L.ldpc_bp_get_info.argtypes = [vp, ctypes.POINTER(BPInfo)]
    L.ldpc_bp_decode_batch.restype = i32
    L.ldpc_bp_decode_batch.argtypes = [vp, i64, vp, vp, vp, vp, vp]
    L.ldpc_bp_decode_batch_device.restype = i32
    L.ldpc_bp_decode_batch_device.argtypes = [vp, i64, vp, vp, vp, vp, vp, vp]
    L.ldpc_bp_last_status.restype = i32
    L.ldpc_bp_last_status.argtypes = [vp]
    L.ldpc_bp_last_timing.restype = i32
    L.ldpc_bp_last_timing.argtypes = [vp, ctypes.POINTER(f64), ctypes.POINTER(f64), ctypes.POINTER(i64)]
    L.ldpc_bp_call_timing.restype = i32
    L.ldpc_bp_call_timing.argtypes = [vp, i32, ctypes.POINTER(f64), ctypes.POINTER(f64), ctypes.POINTER(i64)]
    L.ldpc_bp_call_phase_ticks.restype = i32
    L.ldpc_bp_call_phase_ticks.argtypes = [vp, i32, ctypes.POINTER(ctypes.c_uint64 * 3)]
    L.ldpc_osd_create.restype = i32
    L.ldpc_osd_create.argtypes = [i64, i64, i64, vp, vp, i64, ctypes.POINTER(vp)]
    L.ldpc_osd_destroy.restype = i32
    L.ldpc_osd_destroy.argtypes = [vp]
    L.ldpc_osd_postprocess_batch.restype = i32
    L.ldpc_osd_postprocess_batch.argtypes = [vp, i64, vp, vp, vp, vp, i32]
    L.ldpc_bpots_create.restype = i32
    L.ldpc_bpots_create.argtypes = [i64, i64, i64, vp, vp, f64, i64, i64, f64, i32, ctypes.POINTER(vp)]
    L.ldpc_bpots_destroy.restype = i32
    L.ldpc_bpots_destroy.argtypes = [vp]
    L.ldpc_bpots_kernel.restype = i32
    L.ldpc_bpots_kernel.argtypes = [vp]
    L.ldpc_bpots_decode_batch.restype = i32
    L.ldpc_bpots_decode_batch.argtypes = [vp, i64, vp, vp, vp, vp]
    L.ldpc_bpots_decode_batch_device.restype = i32
    L.ldpc_bpots_decode_batch_device.argtypes = [vp, i64, vp, vp, vp, vp, vp]
    L.ldpc_debug_process_state.restype = vp
    L.ldpc_debug_adopt_process_state.restype = i32
    L.ldpc_debug_adopt_process_state.argtypes = [vp]
    # both builds in one process: the one loaded second orders its team grids through the first one's per-device events,
    # so that no two team grids of the process share a device (include/ldpc_mi355x_debug.h)
    other = _LIBS.get(not experiments)
    if other is not None and os.path.realpath(path) != os.path.realpath(EXP_LIB_PATH if not experiments else LIB_PATH):
        check(L.ldpc_debug_adopt_process_state(other.ldpc_debug_process_state()), L)
    _LIBS[experiments] = L
    return L


def knobs_in_env() -> bool:
    """Is any of the library's experiment knobs (LDPC_TEAM_*, LDPC_DEFER_*, ... DESIGN.md "Environment knobs") set?"""
    return any(k.startswith("LDPC_") and not k.startswith("LDPC_MI355X_") for k in os.environ)


def lib_for(experiments=None) -> ctypes.CDLL:
    """The library a new decoder binds: the product build, unless the caller asks for the experiments build or has
    set one of its knobs in the environment (only that build reads them)."""
    return lib(knobs_in_env() if experiments is None else bool(experiments))


def check(status: int, L: ctypes.CDLL = None) -> None:
    if status != LDPC_OK:
        raise LdpcError(status, (L or lib()).ldpc_last_error().decode("utf-8", "replace"))
