"""lorastencil_amd -- MI355X-native low-rank stencil engine (drop-in for the hot path of zondie17/LoRAStencil).

The compute lives in ``lib/liblorastencil_hip.so`` (hand-written HIP for gfx950 behind the C ABI of
``include/lorastencil.h``).  This package is the thin host-side mirror of the reference's operator interface:

* ``ops``   -- the seven ``gpu_*`` operators, device-resident ``Plan``s, params tables, factor precompute, fill;
* ``slab``  -- multi-GPU slab decomposition with halo exchange over ``torch.distributed`` (RCCL).

No CPU fallback exists: importing works anywhere the library is built, computing needs a HIP device.
"""
from . import _lib
from .ops import (  # noqa: F401
    SHAPES,
    GlibcRand,
    LoraError,
    Plan,
    default_params,
    device_count,
    effective_weights,
    factorize_7x7,
    from_bf16,
    gpu_1d1r,
    gpu_1d2r,
    gpu_box_2d3r,
    gpu_box_3d1r,
    gpu_star_2d1r,
    gpu_star_2d3r,
    gpu_star_3d1r,
    interior,
    padded_shape,
    reference_input,
    run_host,
    separable_3x3x3,
    svd_7x7,
    to_bf16,
)

VARIANT_AUTO, VARIANT_DIRECT, VARIANT_MFMA = _lib.VARIANT_AUTO, _lib.VARIANT_DIRECT, _lib.VARIANT_MFMA
__all__ = [n for n in dir() if not n.startswith("_")]
