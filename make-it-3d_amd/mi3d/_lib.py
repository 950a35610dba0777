"""ctypes binding of libmi3d.so (the C ABI declared in include/mi3d.h).

PyTorch is plumbing here: it owns device memory and streams; every kernel is reached through the
plain-pointer C ABI with `tensor.data_ptr()` and the current HIP stream.  There is NO fallback: if the
shared library is missing or a launch fails this module raises.
"""
import ctypes as C
import os

import torch

_CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "csrc")
LIB_PATH = os.environ.get("MI3D_LIB", os.path.join(_CSRC, "libmi3d.so"))  # MI3D_LIB: development builds only

_lib = None

vp, u32, i32, f32 = C.c_void_p, C.c_uint32, C.c_int32, C.c_float

_SIGNATURES = {
    # Part 1 ------------------------------------------------------------------------------------------
    "mi3d_near_far_from_aabb": [vp, vp, vp, u32, f32, vp, vp, vp],
    "mi3d_sph_from_ray": [vp, vp, f32, u32, vp, vp],
    "mi3d_morton3D": [vp, u32, vp, vp],
    "mi3d_morton3D_invert": [vp, u32, vp, vp],
    "mi3d_packbits": [vp, u32, f32, vp, vp],
    "mi3d_march_rays_train": [vp, vp, vp, f32, f32, u32, u32, u32, u32, u32, vp, vp, vp, vp, vp, vp, vp, vp, vp],
    "mi3d_march_zero_tail": [vp, u32, u32, vp, vp, vp, vp],
    "mi3d_composite_rays_train_forward": [vp, vp, vp, vp, u32, u32, f32, vp, vp, vp, vp],
    "mi3d_composite_rays_train_backward": [vp, vp, vp, vp, vp, vp, vp, vp, u32, u32, f32, vp, vp, vp],
    "mi3d_composite_sdf_rays_train_forward": [vp, vp, vp, vp, u32, u32, f32, vp, vp, vp, vp],
    "mi3d_composite_sdf_rays_train_backward": [vp, vp, vp, vp, vp, vp, vp, vp, u32, u32, f32, vp, vp, vp],
    "mi3d_march_rays": [u32, u32, vp, vp, vp, vp, f32, f32, u32, u32, u32, vp, vp, vp, vp, vp, vp, vp, vp],
    "mi3d_composite_rays": [u32, u32, f32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp],
    "mi3d_composite_sdf_rays": [u32, u32, f32, vp, vp, vp, vp, vp, vp, vp, vp, vp],
    "mi3d_infer_begin": [vp, vp, u32, u32, vp],
    "mi3d_march_rays_ctl": [vp, u32, vp, vp, vp, vp, f32, f32, u32, u32, u32, vp, vp, vp, vp, vp, vp, vp],
    "mi3d_composite_rays_ctl": [vp, u32, f32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp],
    "mi3d_compact_alive_ctl": [vp, vp, vp, u32, u32, u32, vp],
    "mi3d_infer_begin2": [vp, vp, u32, u32, u32, u32, vp],
    "mi3d_march_rays_compact_ctl": [vp, u32, vp, vp, vp, vp, f32, f32, u32, u32, u32, vp, vp, u32, vp, vp, vp, vp, vp, vp, vp],
    "mi3d_composite_rays_compact_ctl": [vp, u32, f32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp],
    "mi3d_compact_alive_ctl2": [vp, vp, vp, u32, u32, vp],
    # Part 2 ------------------------------------------------------------------------------------------
    "mi3d_hashgrid_forward": [vp, u32, vp, u32, u32, f32, u32, vp, vp],
    "mi3d_hashgrid_backward": [vp, u32, vp, u32, u32, f32, u32, vp, vp],
    # Part 3 ------------------------------------------------------------------------------------------
    "mi3d_grid_encode_points": [vp, vp, u32, vp, vp, u32, u32, f32, vp, u32, u32, f32, u32, vp, vp],
    "mi3d_grid_scatter_points": [vp, vp, u32, vp, vp, u32, u32, f32, vp, u32, u32, f32, u32, f32, vp, vp],
    "mi3d_grid_encode_points_planes": [vp, vp, u32, vp, u32, u32, f32, vp, u32, u32, f32, u32, f32, vp, i32, vp],
    "mi3d_grid_scatter_binned": [vp, vp, u32, vp, u32, u32, f32, vp, i32, u32, u32, f32, u32, f32, vp, C.c_size_t, vp, vp],
    "mi3d_grid_scatter_binned_plus": [vp, vp, u32, vp, u32, u32, f32, vp, vp, i32, u32, u32, f32, u32, f32, vp, C.c_size_t,
                                      vp, vp],
    "mi3d_grid_encode_points_planes_counted": [vp, vp, u32, vp, vp, u32, u32, f32, vp, u32, u32, f32, u32, f32, vp, i32, vp],
    "mi3d_grid_encode_plan": [u32, f32, f32, u32, u32, f32, u32, vp, vp],           # host-side queries
    "mi3d_grid_scatter_plan": [u32, u32, f32, f32, u32, u32, f32, u32, C.c_size_t, vp],
    "mi3d_grid_level_routes": [u32, u32, f32, u32, i32, vp],
    # Part 4 ------------------------------------------------------------------------------------------
    "mi3d_mlp_supported": [u32, u32, u32, u32],
    "mi3d_mlp_forward": [vp, u32, i32, u32, vp, vp, vp, vp, vp, vp, u32, u32, u32, i32, vp, vp],
    "mi3d_mlp_forward_counted": [vp, u32, i32, u32, vp, u32, vp, vp, vp, vp, vp, vp, u32, u32, u32, i32, vp, vp],
    "mi3d_mlp_backward": [vp, u32, i32, vp, u32, vp, vp, vp, vp, vp, vp, u32, u32, u32, i32, vp, u32, vp, vp, vp, vp, vp, vp, vp],
    # Part 5 ------------------------------------------------------------------------------------------
    "mi3d_field_head_forward": [vp, vp, vp, u32, vp, u32, f32, f32, f32, f32, vp, vp, vp, vp, vp],
    "mi3d_field_head_forward_counted": [vp, vp, vp, u32, vp, vp, u32, f32, f32, f32, f32, vp, vp, vp, vp, vp],
    "mi3d_field_head_backward": [vp, vp, vp, u32, vp, u32, u32, f32, f32, f32, f32, vp, vp, vp, vp, vp, vp],
    # Part 6 ------------------------------------------------------------------------------------------
    "mi3d_sumsq_accumulate": [vp, C.c_size_t, vp, vp],
    "mi3d_adan_step": [vp, vp, vp, vp, vp, vp, C.c_size_t, vp, f32, f32, i32, f32, f32, f32, f32, f32, f32, f32, f32, f32,
                       i32, vp],
    # Part 7 ------------------------------------------------------------------------------------------
    "mi3d_points_rasterize": [vp, u32, u32, u32, f32, u32, vp, C.c_size_t, vp, vp, vp, vp],
    "mi3d_points_composite_forward": [vp, vp, u32, u32, u32, vp, u32, C.c_double, vp, vp],
    "mi3d_points_composite_backward": [vp, vp, u32, u32, u32, vp, u32, C.c_double, vp, vp],
}


class Mi3dError(RuntimeError):
    pass


def lib():
    """Load libmi3d.so once. Fails loudly - there is no CPU or eager fallback for the hot path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise Mi3dError(f"{LIB_PATH} not found - build it first: python -c 'import __graft_entry__ as g; g.build()'")
        _lib = C.CDLL(LIB_PATH)
        for name, args in _SIGNATURES.items():
            fn = getattr(_lib, name)
            fn.argtypes = args
            fn.restype = C.c_int
        _lib.mi3d_abi_version.restype = C.c_int
        _lib.mi3d_last_error_string.restype = C.c_char_p
        _lib.mi3d_last_error_string.argtypes = [C.c_int]
        _lib.mi3d_hashgrid_levels.restype = u32
        _lib.mi3d_hashgrid_levels.argtypes = [u32, u32, f32, u32, vp, vp, vp]
        _lib.mi3d_grid_scatter_binned_workspace.restype = C.c_size_t
        _lib.mi3d_grid_scatter_binned_workspace.argtypes = [u32, u32, f32, f32, u32, u32, f32, u32]
        _lib.mi3d_points_rasterize_workspace.restype = C.c_size_t
        _lib.mi3d_points_rasterize_workspace.argtypes = [u32, u32, u32, f32]
    return _lib


def declare(name, argtypes, restype=C.c_int):
    fn = getattr(lib(), name)
    fn.argtypes, fn.restype = argtypes, restype
    return fn


def call(name, *args):
    err = getattr(lib(), name)(*args)
    if err != 0:
        raise Mi3dError(f"{name} failed: hipError {err} ({lib().mi3d_last_error_string(err).decode()})")


def launch(name, t, *args):
    """`name(*args, stream)` under a device guard for tensor t: the kernel goes to torch's current stream of the
    device t (and therefore every pointer in args) lives on, whatever torch.cuda.current_device() says."""
    with torch.cuda.device(t.device):
        call(name, *args, stream(t))


def stream(t=None):
    """The HIP stream the C ABI launches on: torch's current stream of `t`'s device (of the current device without t)."""
    dev = t.device if isinstance(t, torch.Tensor) and t.is_cuda else None
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def on(t):
    """Device guard for a launch whose pointers live on `t`'s device (a model on cuda:1 without set_device)."""
    return torch.cuda.device(t.device)


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def dev_f32(t, name, shape_last=None):
    """Validate a device fp32 contiguous tensor (the C ABI takes raw pointers: no silent copies of bad inputs)."""
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise Mi3dError(f"{name} must live on the GPU (got {t.device})")
    if t.dtype != torch.float32:
        raise Mi3dError(f"{name} must be float32 (got {t.dtype})")
    if not t.is_contiguous():
        raise Mi3dError(f"{name} must be contiguous")
    if shape_last is not None and (t.dim() == 0 or t.shape[-1] != shape_last):
        raise Mi3dError(f"{name} must have last dimension {shape_last} (got {tuple(t.shape)})")
    return t


def dev_typed(t, name, dtype):
    if not t.is_cuda or t.dtype != dtype or not t.is_contiguous():
        raise Mi3dError(f"{name} must be a contiguous {dtype} GPU tensor (got {t.dtype} on {t.device})")
    return t
