"""ctypes binding of the C ABI in include/mof.h (libmof_hip.so).

This is plumbing for tests, bench.py and the multi-GPU driver; the product is the
shared library. It fails loudly: a missing library raises ImportError-like
``MofLibraryError`` and there is no Python/CPU compute fallback anywhere.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MOF_LIB_PATH: diagnostic override used by the same-box A/B scripts (tools/ab_variants.sh, tools/ab_commit.sh) to load a
# variant built under /tmp without touching the product library
LIB_PATH = os.environ.get("MOF_LIB_PATH") or os.path.join(_HERE, "libmof_hip.so")

MOF_OK = 0
MOF_ERR_BAD_ARG = -1
MOF_ERR_BUSY = -2
MOF_ERR_HIP = -3
MOF_ERR_NOT_INIT = -4
MOF_ERR_UNSUPPORTED = -5
MOF_ERR_NO_DEVICE = -6
MOF_ERR_NO_MEMORY = -7


class MofLibraryError(RuntimeError):
    pass


class MofError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"mof error {code}: {message}")
        self.code = code


class FftConfig(C.Structure):
    _fields_ = [("frame_width", C.c_int), ("frame_height", C.c_int), ("patch_size", C.c_int),
                ("grid_x", C.c_int), ("grid_y", C.c_int), ("origin_x", C.c_int), ("origin_y", C.c_int),
                ("stride_x", C.c_int), ("stride_y", C.c_int), ("max_px_speed", C.c_double), ("device", C.c_int),
                ("peak_model", C.c_int), ("search_radius", C.c_int)]


class SrConfig(C.Structure):
    _fields_ = [("resolution", C.c_int), ("magnitude", C.c_double), ("device", C.c_int), ("logpolar_variant", C.c_int),
                ("batch_chunk", C.c_int), ("pipeline_lanes", C.c_int)]


class BmConfig(C.Structure):
    _fields_ = [("frame_width", C.c_int), ("frame_height", C.c_int), ("block_size", C.c_int),
                ("step_size", C.c_int), ("scan_radius", C.c_int), ("grid_x", C.c_int), ("grid_y", C.c_int),
                ("low_contrast_rule", C.c_int), ("device", C.c_int)]


# every symbol include/mof.h declares: (name, restype, argtypes)
_VP, _SZ, _I = C.c_void_p, C.c_size_t, C.c_int
SYMBOLS = {
    "mof_version": (C.c_char_p, []),
    "mof_last_error": (C.c_char_p, []),
    "mof_device_count": (_I, []),
    "mof_purge_deferred": (_I, []),
    "mof_deferred_count": (_I, []),
    "mof_fft_release_graphs": (_I, [_VP]),
    "mof_fft_graph_pinned": (_I, [_VP]),
    "mof_bm_release_graphs": (_I, [_VP]),
    "mof_bm_graph_pinned": (_I, [_VP]),
    "mof_sr_release_graphs": (_I, [_VP]),
    "mof_sr_graph_pinned": (_I, [_VP]),
    "mof_fft_config_reference": (_I, [C.POINTER(FftConfig), _I, _I, C.c_double]),
    "mof_fft_create": (_I, [C.POINTER(FftConfig), C.POINTER(_VP)]),
    "mof_fft_kernel_variant": (C.c_char_p, [_VP]),
    "mof_fft_destroy": (None, [_VP]),
    "mof_fft_set_prev": (_I, [_VP, _VP, _SZ]),
    "mof_fft_reset": (_I, [_VP]),
    "mof_fft_process": (_I, [_VP, _VP, _SZ, _VP, C.POINTER(_I)]),
    "mof_fft_long_range_patches": (_I, [_VP]),
    "mof_fft_process_long_range": (_I, [_VP, _VP, _SZ, _VP, C.POINTER(_I)]),
    "mof_fft_process_long_range_batch_device": (_I, [_VP, _VP, _SZ, _VP, _SZ, _SZ, _I, _VP, _VP]),
    "mof_fft_process_batch_device": (_I, [_VP, _VP, _SZ, _VP, _SZ, _SZ, _I, _VP, _VP]),
    "mof_fft_process_sequence_device": (_I, [_VP, _VP, _SZ, _SZ, _I, _VP, _VP]),
    "mof_fft_process_sequence_device_bgr": (_I, [_VP, _VP, _SZ, _SZ, _I, _VP, _VP]),
    "mof_fft_process_batch_device_bgr": (_I, [_VP, _VP, _SZ, _VP, _SZ, _SZ, _I, _VP, _VP]),
    "mof_fft_process_batch_host": (_I, [_VP, _VP, _SZ, _VP, _SZ, _SZ, _I, _VP]),
    "mof_host_alloc": (_I, [_SZ, C.POINTER(_VP)]),
    "mof_host_free": (_I, [_VP]),
    "mof_host_register": (_I, [_VP, _SZ]),
    "mof_host_unregister": (_I, [_VP]),
    "mof_fft_sync": (_I, [_VP]),
    "mof_shard_slab_pairs": (_I, [_I, _I]),
    "mof_shard_partition": (_I, [_I, _I, _I, C.POINTER(_I), C.POINTER(_I)]),
    "mof_shard_fft_create": (_I, [C.POINTER(FftConfig), C.POINTER(_I), _I, C.POINTER(_VP)]),
    "mof_shard_fft_destroy": (None, [_VP]),
    "mof_shard_fft_devices": (_I, [_VP]),
    "mof_shard_fft_stream": (_VP, [_VP, _I]),
    "mof_shard_fft_process_batch_device": (_I, [_VP, C.POINTER(_VP), _SZ, C.POINTER(_VP), _SZ, _SZ, _I, C.POINTER(_VP), _I]),
    "mof_shard_fft_sync": (_I, [_VP]),
    "mof_shard_fft_init_gather": (_I, [_VP]),
    "mof_shard_fft_gather_ready": (_I, [_VP]),
    "mof_shard_fft_gather_ranks": (_I, [_VP]),
    "mof_shard_bm_create": (_I, [C.POINTER(BmConfig), C.POINTER(_I), _I, C.POINTER(_VP)]),
    "mof_shard_bm_destroy": (None, [_VP]),
    "mof_shard_bm_devices": (_I, [_VP]),
    "mof_shard_bm_stream": (_VP, [_VP, _I]),
    "mof_shard_bm_init_gather": (_I, [_VP]),
    "mof_shard_bm_gather_ready": (_I, [_VP]),
    "mof_shard_bm_gather_ranks": (_I, [_VP]),
    "mof_shard_bm_slab_bytes": (_SZ, [_VP, _I]),
    "mof_shard_bm_locate": (_I, [_VP, _I, _I, C.POINTER(_SZ), C.POINTER(_SZ), C.POINTER(_SZ)]),
    "mof_shard_bm_process_batch_device": (_I, [_VP, C.POINTER(_VP), _SZ, C.POINTER(_VP), _SZ, _SZ, _I, C.POINTER(_VP), _I]),
    "mof_shard_bm_sync": (_I, [_VP]),
    "mof_bm_config_block_method": (_I, [C.POINTER(BmConfig), _I, _I, _I]),
    "mof_bm_config_fast_spaced": (_I, [C.POINTER(BmConfig), _I, _I, _I, _I, _I]),
    "mof_bm_create": (_I, [C.POINTER(BmConfig), C.POINTER(_VP)]),
    "mof_bm_destroy": (None, [_VP]),
    "mof_bm_set_prev": (_I, [_VP, _VP, _SZ]),
    "mof_bm_reset": (_I, [_VP]),
    "mof_bm_process": (_I, [_VP, _VP, _SZ, _VP, _VP, _VP]),
    "mof_bm_refine": (_I, [_VP, _I, _I, _I, _I, _VP]),
    "mof_bm_process_batch_device": (_I, [_VP, _VP, _SZ, _VP, _SZ, _SZ, _I, _VP, _VP, _VP, _VP]),
    "mof_bm_process_batch_device_bgr": (_I, [_VP, _VP, _SZ, _VP, _SZ, _SZ, _I, _VP, _VP, _VP, _VP]),
    "mof_bm_process_batch_host": (_I, [_VP, _VP, _SZ, _VP, _SZ, _SZ, _I, _VP, _VP, _VP]),
    "mof_bm_sync": (_I, [_VP]),
    "mof_sr_create": (_I, [C.POINTER(SrConfig), C.POINTER(_VP)]),
    "mof_sr_destroy": (None, [_VP]),
    "mof_sr_reset": (_I, [_VP]),
    "mof_sr_reserve": (_I, [_VP, _I]),
    "mof_sr_process": (_I, [_VP, _VP, _SZ, _VP]),
    "mof_sr_process_batch_device": (_I, [_VP, _VP, _SZ, _VP, _SZ, _SZ, _I, _VP, _VP]),
    "mof_sr_process_sequence_device": (_I, [_VP, _VP, _SZ, _SZ, _I, _VP, _VP, C.POINTER(_I)]),
    "mof_sr_process_sequence_host": (_I, [_VP, _VP, _SZ, _SZ, _I, _VP, C.POINTER(_I)]),
    "mof_sr_logpolar_batch_device": (_I, [_VP, _VP, _SZ, _SZ, _I, _I, _VP, _VP]),
    "mof_geom_layout_reference": (_I, [_VP, _I, _I]),
    "mof_geom_undistort_points": (_I, [_VP, C.c_double, _VP, _I, _VP]),
    "mof_geom_find_homography": (_I, [_VP, _VP, _I, _VP, _VP, C.POINTER(_I)]),
    "mof_geom_decompose_homography": (_I, [_VP, _VP, _VP, _VP, C.POINTER(_I)]),
    "mof_geom_get_rt": (_I, [_VP, _VP, _VP, _VP, _I, _VP, C.POINTER(_I), _VP, _VP]),
    "mof_geom_get_2dt": (_I, [_VP, _VP, _VP, _VP, _VP, C.POINTER(_I)]),
    "mof_geom_get_rt_batch_device": (_I, [_VP, _VP, _VP, _VP, _I, _I, _VP, _VP]),
    "mof_geom_get_2dt_batch_device": (_I, [_VP, _VP, _VP, _VP, _I, _VP, _VP]),
}

_lib = None


def load():
    """Load libmof_hip.so (built by __graft_entry__.build() / csrc/Makefile). Never falls back."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MofLibraryError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        # ROCm PyTorch wheels bundle their own libamdhip64. If this library pulled in /opt/rocm's copy first, the
        # process would hold two HIP runtimes and the second one to initialise sees no device. Letting torch load
        # first (when it is installed) makes the order of imports irrelevant for every Python user of the ABI.
        if not os.environ.get("MOF_NO_TORCH_PRELOAD"):
            try:
                import torch  # noqa: F401
            except Exception:  # torch is optional plumbing
                pass
        try:
            lib = C.CDLL(LIB_PATH)
        except OSError as exc:  # pragma: no cover - depends on the host
            raise MofLibraryError(f"cannot load {LIB_PATH}: {exc}") from exc
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)  # AttributeError if the library does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc: int) -> None:
    if rc != MOF_OK:
        raise MofError(rc, load().mof_last_error().decode("utf-8", "replace"))
