"""ctypes binding of libgeosplat_hip.so (the C-ABI declared in include/geosplat_hip.h).

There is NO fallback: if the HIP library is missing the import of any op raises.  PyTorch is only used for
device memory (``tensor.data_ptr()``) and the current HIP stream.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgeosplat_hip.so")

GS_MAX_LEVELS = 16


class GsEnv(C.Structure):
    _fields_ = [("lut", C.c_void_p), ("lut_res", C.c_int), ("base", C.c_void_p), ("base_res", C.c_int),
                ("num_levels", C.c_int), ("levels", C.c_void_p * GS_MAX_LEVELS), ("res", C.c_int * GS_MAX_LEVELS),
                ("min_roughness", C.c_float), ("max_roughness", C.c_float)]


class GsTailView(C.Structure):
    _fields_ = [("viewmat", C.c_void_p), ("K", C.c_void_p), ("cam_pos", C.c_void_p), ("vis_records", C.c_void_p), ("v_packed", C.c_void_p),
                ("packed_index", C.c_void_p), ("W", C.c_int), ("H", C.c_int)]


class GsTileLevel(C.Structure):
    _fields_ = [("R", C.c_int), ("n_mirrors", C.c_int), ("margin", C.c_int), ("bw", C.c_int), ("nb", C.c_int), ("tile_begin", C.c_int),
                ("tile_end", C.c_int), ("reserved", C.c_int), ("src", C.c_void_p), ("scale", C.c_void_p), ("out_scale", C.c_void_p),
                ("bounds", C.c_void_p), ("tiles", C.c_void_p), ("segments", C.c_void_p), ("row_begin", C.c_void_p), ("row_counts", C.c_void_p),
                ("desc", C.c_void_p), ("weights", C.c_void_p), ("dst", C.c_void_p), ("lds_bytes", C.c_size_t)]


class GsEnvGrad(C.Structure):
    _fields_ = [("base", C.c_void_p), ("levels", C.c_void_p * GS_MAX_LEVELS)]


# every symbol include/geosplat_hip.h declares (tests/test_abi.py checks the list against the header)
SYMBOLS = [
    "gs_last_error", "gs_version", "gs_project_ws_bytes", "gs_project_fwd", "gs_project_fwd_vis", "gs_isect_emit", "gs_sort_ws_bytes",
    "gs_isect_sort", "gs_isect_bin_ws_bytes", "gs_isect_bin", "gs_isect_offsets", "gs_raster_ws_bytes", "gs_raster_fwd", "gs_raster_prepare", "gs_raster_prepare_vis", "gs_raster_composite", "gs_raster_grad_stride", "gs_raster_bwd", "gs_raster_bwd_acc", "gs_selftest_rcp", "gs_selftest_exp", "gs_selftest_cube_edges", "gs_isect_bin_cap", "gs_isect_offsets_cap", "gs_isect_bin_tiles_cap", "gs_isect_offsets_tiles_cap", "gs_raster_prepare_vis_cap", "gs_raster_composite_cap", "gs_raster_bwd_cap", "gs_raster_bwd_acc_cap", "gs_raster_composite_tone", "gs_raster_bwd_tone_acc", "gs_raster_log_ws_bytes", "gs_raster_composite_tone_log", "gs_raster_bwd_tone_log_acc", "gs_project_bwd_cap", "gs_project_bwd", "gs_shade_fwd",
    "gs_shade_bwd_ws_bytes", "gs_shade_bwd", "gs_tonemap_fwd", "gs_tonemap_bwd", "gs_tonemap_fwd3", "gs_tonemap_bwd3", "gs_cubemap_mip_fwd", "gs_cubemap_mip_chain_fwd", "gs_cube_sample_linear",
    "gs_cubemap_mip_bwd", "gs_diffuse_cubemap_fwd", "gs_diffuse_cubemap_bwd", "gs_specular_bounds", "gs_specular_bounds_ws_bytes", "gs_specular_bounds_fast", "gs_cube_dir_table",
    "gs_specular_cubemap_fwd", "gs_specular_cubemap_bwd", "gs_specular_tiles_count", "gs_specular_tiles_fill",
    "gs_specular_tiles_check", "gs_specular_tiles_apply", "gs_specular_tiles_apply_multi", "gs_mgadapter_fwd", "gs_mgadapter_bwd", "gs_vertex_normals_fwd",
    "gs_vertex_normals_bwd", "gs_photo_loss_ws_bytes", "gs_photo_loss", "gs_hashgrid_fwd", "gs_hashgrid_bwd_ws_bytes", "gs_hashgrid_bwd", "gs_hashgrid_bwd_fixed_ws_bytes", "gs_hashgrid_bwd_fixed", "gs_mlp_wgrad_ws_bytes", "gs_mlp_wgrad",
    "gs_flexicubes_ws_bytes", "gs_flexicubes_count", "gs_flexicubes_fwd", "gs_flexicubes_bwd", "gs_flexicubes_entropy_fwd",
    "gs_flexicubes_entropy_bwd", "gs_front_ws_bytes", "gs_front_fwd", "gs_isect_bin_front_ws_bytes", "gs_isect_bin_front", "gs_tail_bwd", "gs_tail_bwd_multi", "gs_tail_bwd_multi_parts", "gs_tail_priv_ws_bytes", "gs_tail_priv_reduce", "gs_activation_chain",
]

_lib: Optional[C.CDLL] = None


class GeoSplatHipError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load the HIP library; raises (never falls back) if it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GeoSplatHipError(
                f"{LIB_PATH} not found: build it with `python -m geosplatting_amd.build` "
                "(hipcc --offload-arch=gfx950).  geosplatting_amd has no CPU / PyTorch fallback.")
        l = C.CDLL(LIB_PATH)
        l.gs_last_error.restype = C.c_char_p
        l.gs_project_ws_bytes.restype = C.c_size_t
        l.gs_project_ws_bytes.argtypes = [C.c_int]
        l.gs_sort_ws_bytes.restype = C.c_size_t
        l.gs_sort_ws_bytes.argtypes = [C.c_int64, C.c_int, C.c_int]
        l.gs_isect_bin_ws_bytes.restype = C.c_size_t
        l.gs_isect_bin_ws_bytes.argtypes = [C.c_int, C.c_int64, C.c_int, C.c_int]
        l.gs_shade_bwd_ws_bytes.restype = C.c_size_t
        l.gs_shade_bwd_ws_bytes.argtypes = [C.c_void_p, C.c_int]
        l.gs_hashgrid_bwd_ws_bytes.restype = C.c_size_t
        l.gs_hashgrid_bwd_ws_bytes.argtypes = [C.c_int, C.c_int, C.c_int]
        l.gs_hashgrid_bwd_fixed_ws_bytes.restype = C.c_size_t
        l.gs_hashgrid_bwd_fixed_ws_bytes.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
        l.gs_mlp_wgrad_ws_bytes.restype = C.c_size_t
        l.gs_mlp_wgrad_ws_bytes.argtypes = [C.c_int64]
        l.gs_flexicubes_ws_bytes.restype = C.c_size_t
        l.gs_flexicubes_ws_bytes.argtypes = [C.c_int, C.c_int, C.c_int]
        l.gs_photo_loss_ws_bytes.restype = C.c_size_t
        l.gs_photo_loss_ws_bytes.argtypes = [C.c_int, C.c_int]
        l.gs_specular_bounds_ws_bytes.restype = C.c_size_t
        l.gs_specular_bounds_ws_bytes.argtypes = [C.c_int]
        l.gs_front_ws_bytes.restype = C.c_size_t
        l.gs_front_ws_bytes.argtypes = [C.c_int]
        l.gs_isect_bin_front_ws_bytes.restype = C.c_size_t
        l.gs_isect_bin_front_ws_bytes.argtypes = [C.c_int, C.c_int64, C.c_int, C.c_int]
        l.gs_tail_priv_ws_bytes.restype = C.c_size_t
        l.gs_tail_priv_ws_bytes.argtypes = [C.c_void_p, C.c_int]
        l.gs_raster_log_ws_bytes.restype = C.c_size_t
        l.gs_raster_log_ws_bytes.argtypes = [C.c_int64, C.c_int, C.c_int, C.c_int]
        l.gs_raster_ws_bytes.restype = C.c_size_t
        l.gs_raster_ws_bytes.argtypes = [C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int]
        _lib = l
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise GeoSplatHipError(f"{what} failed ({rc}): {lib().gs_last_error().decode()}")


def ptr(t: Optional[torch.Tensor]) -> Optional[C.c_void_p]:
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "device-resident contiguous tensors only"
    return C.c_void_p(t.data_ptr())


def stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def f32(x: float) -> C.c_float:
    return C.c_float(float(x))


def i64(x: int) -> C.c_int64:
    return C.c_int64(int(x))


def require_cuda(*tensors: torch.Tensor) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise GeoSplatHipError("geosplatting_amd ops run on the GPU only (tensor on %s); there is no CPU path" % t.device)


_shared_streams = {}


def shared_stream(device: torch.device, role: str) -> "torch.cuda.Stream":
    """ONE HIP stream per (device, role) for the whole process -- "front0", "front1", "tail", "comm": the step engine, the call shape
    (viewbatch) and the gradient buckets all take theirs from here.  HIP maps its streams onto four hardware queues; with a private
    set per object a process that holds an engine AND renders through the call shape had eight streams, two per queue."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), role)
    st = _shared_streams.get(key)
    if st is None:
        st = _shared_streams[key] = torch.cuda.Stream(device=device)
    return st
