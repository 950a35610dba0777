"""`tinycudann` - the slice of the tiny-cuda-nn torch bindings the reference uses
(/root/reference/nerf/network_tcnn.py:10,54-65,107,197): `tcnn.Encoding(n_input_dims=3,
encoding_config={"otype": "HashGrid", ...}, dtype=torch.float32)`, an nn.Module with ONE flat fp32
parameter tensor named `params` (state_dict key `encoder.params`), `forward(x [n,3] in [0,1]) ->
[n, n_levels*2]`, differentiable w.r.t. `params`.  Backed by csrc/hashgrid.hip (gfx950).

tiny-cuda-nn itself is an un-vendored, un-pinned dependency of the reference: the arithmetic here
follows its published grid.h (see oracle/hashgrid_ref.c) - PARITY UNPINNED.
"""
import ctypes as C

import numpy as np
import torch
import torch.nn as nn
from torch.autograd import Function

from mi3d import _lib as L
from mi3d.grid_ops import GridParameter

__all__ = ["Encoding", "Network", "NetworkWithInputEncoding"]


class _hashgrid(Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x, params, cfg):
        x = L.dev_f32(x.contiguous().view(-1, 3), "x", 3)
        params = L.dev_f32(params, "params")
        n = x.shape[0]
        out = torch.empty(n, cfg["n_levels"] * 2, dtype=torch.float32, device=x.device)
        L.launch("mi3d_hashgrid_forward", x, L.ptr(x), n, L.ptr(params), cfg["n_levels"], cfg["base_resolution"],
               cfg["per_level_scale"], cfg["log2_hashmap_size"], L.ptr(out))
        ctx.save_for_backward(x)
        ctx.cfg, ctx.n_params = cfg, params.numel()
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dout):
        (x,) = ctx.saved_tensors
        cfg = ctx.cfg
        dout = L.dev_f32(dout.float().contiguous(), "dout")
        grad = torch.zeros(ctx.n_params, dtype=torch.float32, device=x.device)
        L.launch("mi3d_hashgrid_backward", x, L.ptr(x), x.shape[0], L.ptr(dout), cfg["n_levels"], cfg["base_resolution"],
               cfg["per_level_scale"], cfg["log2_hashmap_size"], L.ptr(grad))
        return None, grad, None  # no gradient w.r.t. the input positions (the reference never asks for it)


def grid_levels(n_levels, base_resolution, per_level_scale, log2_hashmap_size):
    """(total entries, offsets[n_levels+1], resolutions, scales) of a HashGrid - host side, via the C ABI."""
    offs = (C.c_uint32 * (n_levels + 1))()
    res = (C.c_uint32 * n_levels)()
    scl = (C.c_float * n_levels)()
    total = L.lib().mi3d_hashgrid_levels(n_levels, base_resolution, per_level_scale, log2_hashmap_size, offs, res,
                                         scl)
    return int(total), np.array(offs, np.uint32), np.array(res, np.uint32), np.array(scl, np.float32)


class Encoding(nn.Module):
    def __init__(self, n_input_dims, encoding_config, seed=1337, dtype=None):
        super().__init__()
        if n_input_dims != 3:
            raise NotImplementedError("only 3-D inputs (the reference's use)")
        if encoding_config.get("otype", "HashGrid") not in ("HashGrid", "Grid"):
            raise NotImplementedError(f"encoding otype {encoding_config.get('otype')!r}: only HashGrid is built")
        if encoding_config.get("n_features_per_level", 2) != 2:
            raise NotImplementedError("n_features_per_level must be 2")
        if dtype not in (None, torch.float32):
            raise NotImplementedError("dtype=torch.float32 only (network_tcnn.py:64)")
        self.n_input_dims = n_input_dims
        self.encoding_config = dict(encoding_config)
        self.dtype = torch.float32
        self.seed = seed
        self.cfg = dict(n_levels=int(encoding_config.get("n_levels", 16)),
                        base_resolution=int(encoding_config.get("base_resolution", 16)),
                        per_level_scale=float(np.float32(encoding_config.get("per_level_scale", 2.0))),
                        log2_hashmap_size=int(encoding_config.get("log2_hashmap_size", 19)))
        if self.cfg["n_levels"] > 16:
            raise NotImplementedError("at most 16 levels")
        total, self.offsets, self.resolutions, self.scales = grid_levels(**self.cfg)
        self.n_output_dims = self.cfg["n_levels"] * 2
        g = torch.Generator().manual_seed(seed)
        # tcnn: U(-1e-4, 1e-4).  (A GridParameter IS an nn.Parameter; its `.grad` completes a deferred scatter first.)
        self.params = GridParameter((torch.rand(total * 2, generator=g) * 2 - 1) * 1e-4)

    def forward(self, x):
        if not x.is_cuda:
            raise L.Mi3dError("tinycudann.Encoding runs on the GPU only")
        return _hashgrid.apply(x, self.params, self.cfg)

    def extra_repr(self):
        return f"n_input_dims=3, n_output_dims={self.n_output_dims}, seed={self.seed}, {self.encoding_config}"


def _unsupported(name):
    def ctor(*a, **k):
        raise NotImplementedError(f"tinycudann.{name} is not used by Make-It-3D's hot path and is not provided; "
                                  "the fused field lives in mi3d.field")
    return ctor


Network = _unsupported("Network")
NetworkWithInputEncoding = _unsupported("NetworkWithInputEncoding")
