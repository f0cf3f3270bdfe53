"""ctypes loader for the HIP engine (lib/liblorastencil_hip.so, C ABI of include/lorastencil.h).

There is no Python or CPU implementation behind this module: if the shared library has not been built
(``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C lorastencil_amd/csrc``) loading fails
loudly, and without a HIP device every compute entry point returns ``LORA_ENODEVICE`` / a HIP error, which
``check()`` raises as ``LoraError``.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# The one engine library, in-tree.  (Development tools that A/B two builds assign LIB_PATH before the first lib() call;
# no environment variable can swap the engine under the product.)
LIB_PATH = os.path.join(_HERE, "lib", "liblorastencil_hip.so")

LORA_OK = 0
LORA_EINVAL = -1
LORA_EUNSUPPORTED = -2
LORA_EHIP = -3
LORA_ENOMEM = -4
LORA_ENODEVICE = -5

F64 = 0
BF16 = 1
VARIANT_AUTO, VARIANT_DIRECT, VARIANT_MFMA = 0, 1, 2
BC_REFERENCE, BC_DIRICHLET, BC_PERIODIC = 0, 1, 2

_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int)
_vp = ctypes.c_void_p


class LoraError(RuntimeError):
    def __init__(self, status: int, where: str, detail: str = ""):
        self.status = status
        super().__init__(f"{where}: status {status} ({detail})")


class RunInfo(ctypes.Structure):
    _fields_ = [
        ("sweep_seconds", ctypes.c_double),
        ("total_seconds", ctypes.c_double),
        ("gstencils", ctypes.c_double),
        ("gstencils_refconv", ctypes.c_double),
        ("hbm_gbs", ctypes.c_double),
        ("variant", ctypes.c_int),
        ("steps_per_launch", ctypes.c_int),
    ]


class RunProfile(ctypes.Structure):
    _fields_ = [
        ("fused_launches", ctypes.c_int),
        ("apps_per_fused_launch", ctypes.c_int),
        ("two_launches", ctypes.c_int),
        ("single_launches", ctypes.c_int),
        ("fused_ms", ctypes.c_float),
        ("two_ms", ctypes.c_float),
        ("single_ms", ctypes.c_float),
    ]


_comm_begin_t = ctypes.CFUNCTYPE(ctypes.c_int, _vp)
_comm_xfer_t = ctypes.CFUNCTYPE(ctypes.c_int, _vp, _vp, ctypes.c_size_t, ctypes.c_int, _vp)


class SlabComm(ctypes.Structure):
    """lora_slab_comm: the neighbour-exchange callbacks of the C++ slab driver."""
    _fields_ = [("ctx", _vp), ("group_begin", _comm_begin_t), ("send", _comm_xfer_t), ("recv", _comm_xfer_t),
                ("group_end", _comm_begin_t)]


class SlabDesc(ctypes.Structure):
    _fields_ = [("shape", ctypes.c_int), ("dtype", ctypes.c_int), ("global_dims", ctypes.c_int * 3), ("params", _dp),
                ("weights", _dp), ("rank", ctypes.c_int), ("nranks", ctypes.c_int), ("device", ctypes.c_int),
                ("exchange_every", ctypes.c_int), ("boundary", ctypes.c_int), ("flags", ctypes.c_int),
                ("options", ctypes.c_char_p)]


class SlabInfo(ctypes.Structure):
    _fields_ = [("begin", ctypes.c_int), ("end", ctypes.c_int), ("ghost", ctypes.c_int), ("ghost_top", ctypes.c_int),
                ("ghost_bottom", ctypes.c_int), ("apps_per_launch", ctypes.c_int), ("exchange_every", ctypes.c_int),
                ("steps_done", ctypes.c_int), ("local_dims", ctypes.c_int * 3), ("launches", ctypes.c_long),
                ("exchanges", ctypes.c_long), ("local_bytes", ctypes.c_size_t)]


SLAB_NO_OVERLAP, SLAB_NO_DEFER, SLAB_NO_FUSION, SLAB_RING_OF_ONE, SLAB_OVERLAP = 1, 2, 4, 8, 16


class BlockDesc(ctypes.Structure):
    _fields_ = [("shape", ctypes.c_int), ("dtype", ctypes.c_int), ("global_dims", ctypes.c_int * 3), ("grid", ctypes.c_int * 2),
                ("coords", ctypes.c_int * 2), ("params", _dp), ("weights", _dp), ("device", ctypes.c_int),
                ("exchange_every", ctypes.c_int), ("flags", ctypes.c_int), ("options", ctypes.c_char_p)]


class BlockInfo(ctypes.Structure):
    _fields_ = [("own_begin", ctypes.c_int * 2), ("own_end", ctypes.c_int * 2), ("ghost", ctypes.c_int), ("ghost_lo", ctypes.c_int * 2),
                ("ghost_hi", ctypes.c_int * 2), ("apps_per_launch", ctypes.c_int), ("exchange_every", ctypes.c_int),
                ("steps_done", ctypes.c_int), ("local_dims", ctypes.c_int * 3), ("launches", ctypes.c_long),
                ("exchanges", ctypes.c_long), ("local_bytes", ctypes.c_size_t), ("bytes_per_refresh", ctypes.c_size_t)]


class Rng(ctypes.Structure):
    _fields_ = [("r", ctypes.c_int32 * 34), ("pos", ctypes.c_int32)]


# name -> (restype, argtypes): every symbol include/lorastencil.h declares
SIGNATURES = {
    "lora_shape_ntaps": (ctypes.c_int, [ctypes.c_int]),
    "lora_shape_ndim": (ctypes.c_int, [ctypes.c_int]),
    "lora_shape_from_name": (ctypes.c_int, [ctypes.c_char_p]),
    "lora_shape_info_name": (ctypes.c_char_p, [ctypes.c_int]),
    "lora_padded_count": (ctypes.c_size_t, [ctypes.c_int, _ip]),
    "lora_shape_gstencil_factor": (ctypes.c_int, [ctypes.c_int]),
    "lora_strerror": (ctypes.c_char_p, [ctypes.c_int]),
    "lora_last_error": (ctypes.c_char_p, []),
    "lora_device_count": (ctypes.c_int, []),
    "lora_gpu_1d1r": (ctypes.c_int, [_dp, _dp, _dp, ctypes.c_int, ctypes.c_int]),
    "lora_gpu_1d2r": (ctypes.c_int, [_dp, _dp, _dp, ctypes.c_int, ctypes.c_int]),
    "lora_gpu_star_2d1r": (ctypes.c_int, [_dp, _dp, _dp, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "lora_gpu_star_2d3r": (ctypes.c_int, [_dp, _dp, _dp, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "lora_gpu_box_2d3r": (ctypes.c_int, [_dp, _dp, _dp, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "lora_gpu_box_3d1r": (ctypes.c_int, [_dp, _dp, _dp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "lora_gpu_star_3d1r": (ctypes.c_int, [_dp, _dp, _dp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "lora_run_host": (ctypes.c_int, [ctypes.c_int, _dp, _dp, _dp, ctypes.c_int, _ip, ctypes.c_int,
                                     ctypes.POINTER(RunInfo)]),
    "lora_run_host_dtype": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, _vp, _vp, _dp, ctypes.c_int, _ip, ctypes.c_int,
                                           ctypes.POINTER(RunInfo)]),
    "lora_last_run_info": (ctypes.c_int, [ctypes.POINTER(RunInfo)]),
    "lora_f64_to_bf16": (None, [_dp, ctypes.POINTER(ctypes.c_uint16), ctypes.c_size_t]),
    "lora_bf16_to_f64": (None, [ctypes.POINTER(ctypes.c_uint16), _dp, ctypes.c_size_t]),
    "lora_plan_create": (ctypes.c_int, [ctypes.POINTER(_vp), ctypes.c_int, ctypes.c_int, _ip, _dp]),
    "lora_plan_set_weights": (ctypes.c_int, [_vp, _dp, ctypes.c_int]),
    "lora_plan_get_weights": (ctypes.c_int, [_vp, _dp, ctypes.c_int]),
    "lora_plan_set_variant": (ctypes.c_int, [_vp, ctypes.c_int]),
    "lora_plan_set_boundary": (ctypes.c_int, [_vp, ctypes.c_int]),
    "lora_set_default_boundary": (ctypes.c_int, [ctypes.c_int]),
    "lora_set_default_normalize": (ctypes.c_int, [ctypes.c_int]),
    "lora_plan_set_option": (ctypes.c_int, [_vp, ctypes.c_char_p, ctypes.c_int]),
    "lora_plan_get_option": (ctypes.c_int, [_vp, ctypes.c_char_p, _ip]),
    "lora_plan_kernel_signature": (ctypes.c_char_p, [_vp]),
    "lora_slab_comm_rccl": (ctypes.c_int, [ctypes.POINTER(SlabComm), _vp]),
    "lora_slab_comm_loopback": (ctypes.c_int, [ctypes.POINTER(SlabComm), ctypes.c_int]),
    "lora_slab_create": (ctypes.c_int, [ctypes.POINTER(_vp), ctypes.POINTER(SlabDesc), ctypes.POINTER(SlabComm)]),
    "lora_slab_destroy": (None, [_vp]),
    "lora_slab_info": (ctypes.c_int, [_vp, ctypes.POINTER(SlabInfo)]),
    "lora_slab_load": (ctypes.c_int, [_vp, _vp]),
    "lora_slab_load_device": (ctypes.c_int, [_vp, _vp]),
    "lora_slab_refresh_ghosts": (ctypes.c_int, [_vp]),
    "lora_slab_run": (ctypes.c_int, [_vp, ctypes.c_int]),
    "lora_slab_sync": (ctypes.c_int, [_vp]),
    "lora_slab_store": (ctypes.c_int, [_vp, _vp]),
    "lora_slab_buffer": (_vp, [_vp, ctypes.c_int]),
    "lora_slab_stream": (_vp, [_vp]),
    "lora_slab_plan": (_vp, [_vp]),
    "lora_slab_run_many": (ctypes.c_int, [ctypes.POINTER(_vp), ctypes.c_int, ctypes.c_int]),
    "lora_slab_refresh_ghosts_many": (ctypes.c_int, [ctypes.POINTER(_vp), ctypes.c_int]),
    "lora_block_create": (ctypes.c_int, [ctypes.POINTER(_vp), ctypes.POINTER(BlockDesc), ctypes.POINTER(SlabComm)]),
    "lora_block_destroy": (None, [_vp]),
    "lora_block_info": (ctypes.c_int, [_vp, ctypes.POINTER(BlockInfo)]),
    "lora_block_load": (ctypes.c_int, [_vp, _vp]),
    "lora_block_run": (ctypes.c_int, [_vp, ctypes.c_int]),
    "lora_block_run_many": (ctypes.c_int, [ctypes.POINTER(_vp), ctypes.c_int, ctypes.c_int]),
    "lora_block_sync": (ctypes.c_int, [_vp]),
    "lora_block_store": (ctypes.c_int, [_vp, _vp]),
    "lora_block_buffer": (_vp, [_vp, ctypes.c_int]),
    "lora_block_stream": (_vp, [_vp]),
    "lora_block_plan": (_vp, [_vp]),
    "lora_run_host_blocks": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, _vp, _vp, _dp, ctypes.c_int, ctypes.POINTER(ctypes.c_int),
                                            ctypes.POINTER(ctypes.c_int), ctypes.c_int, _vp]),
    "lora_run_host_multi": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, _vp, _vp, _dp, ctypes.c_int, _ip, ctypes.c_int,
                                           ctypes.c_int, ctypes.POINTER(RunInfo)]),
    "lora_plan_run_profiled": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, _vp, ctypes.POINTER(RunProfile)]),
    "lora_plan_padded_bytes": (ctypes.c_size_t, [_vp]),
    "lora_plan_kernel_name": (ctypes.c_char_p, [_vp]),
    "lora_plan_step": (ctypes.c_int, [_vp, _vp, _vp, _vp]),
    "lora_plan_step_region": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, ctypes.c_int, _vp]),
    "lora_plan_region_granularity": (ctypes.c_int, [_vp]),
    "lora_plan_step2": (ctypes.c_int, [_vp, _vp, _vp, _vp]),
    "lora_plan_halo": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, _vp]),
    "lora_copy_block_f64": (ctypes.c_int, [_vp, ctypes.c_long, _vp, ctypes.c_long, ctypes.c_long, ctypes.c_long, _vp]),
    "lora_plan_stepk": (ctypes.c_int, [_vp, _vp, _vp, _vp]),
    "lora_plan_step2_region": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, ctypes.c_int, _vp]),
    "lora_plan_stepk_region": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, ctypes.c_int, _vp]),
    "lora_plan_stepn_region": (ctypes.c_int, [_vp, ctypes.c_int, _vp, _vp, ctypes.c_int, ctypes.c_int, _vp]),
    "lora_plan_prepare_run": (ctypes.c_int, [_vp, ctypes.c_int]),
    "lora_plan_stepn_region2": (ctypes.c_int, [_vp, ctypes.c_int, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp]),
    "lora_debug_span_cover": (ctypes.c_int, [ctypes.c_int] * 6 + [ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    "lora_plan_run": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, _vp]),
    "lora_plan_destroy": (None, [_vp]),
    "lora_default_params": (ctypes.c_int, [ctypes.c_int, _dp]),
    "lora_effective_weights": (ctypes.c_int, [ctypes.c_int, _dp, _dp]),
    "lora_factorize_7x7": (ctypes.c_int, [_dp, _dp, _dp, _dp]),
    "lora_svd_7x7": (ctypes.c_int, [_dp, _dp, _dp, _dp]),
    "lora_separable_3x3x3": (ctypes.c_int, [_dp, ctypes.POINTER(ctypes.c_float)]),
    "lora_rng_seed": (None, [ctypes.POINTER(Rng), ctypes.c_uint]),
    "lora_rng_next": (ctypes.c_int, [ctypes.POINTER(Rng)]),
    "lora_fill_rand": (None, [_dp, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(Rng)]),
}

_lib = None


def lib() -> ctypes.CDLL:
    """The loaded engine; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build the HIP engine first (python -c 'import __graft_entry__ as g; "
                f"g.build()' or make -C lorastencil_amd/csrc).  lorastencil_amd has no CPU fallback."
            )
        # PyTorch-ROCm ships its own libamdhip64; if this library pulled in the system one first and torch loaded
        # its copy afterwards, the process would hold two HIP runtimes and the second sees no device.  Import torch
        # first wherever it is installed, so that both bind to the same runtime.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError here = header / library mismatch
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(status: int, where: str) -> None:
    if status != LORA_OK:
        L = lib()
        detail = L.lora_strerror(status).decode()
        last = L.lora_last_error().decode()
        raise LoraError(status, where, f"{detail}; {last}" if last else detail)
