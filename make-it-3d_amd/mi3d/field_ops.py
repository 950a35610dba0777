"""The whole field evaluation - stencil-aware hash-grid encode + matrix-core MLP - as ONE autograd node.

forward : features = encode(all P stencil points of every sample) -> h = MLP(features)           (2 launches)
backward: MLP backward (feature gradient written as level-major planes, weight gradients reduced on chip)
          -> hash-grid gradient scatter: dense levels by merged atomics, hashed levels through the binned
             record path that avoids global atomics (csrc/hashgrid.hip)                           (2 + 2/slice launches)
The per-layer composition (grid_ops.encode_points + mlp_ops.fused_mlp) stays available and computes the same thing;
this node exists because the binned scatter wants the MLP's input gradient in a layout autograd tensors between two
separate nodes would not carry, and to hold the [m*P, 32] feature matrix once.
"""
import ctypes as C
import os

import numpy as np
import torch
from torch.autograd import Function

from . import _lib as L
from . import grid_ops

_WORKSPACE = {}  # device index -> uint8 tensor (record arena of the binned scatter), grown on demand


def scatter_workspace(device, needed, budget_fraction=0.5):
    """A cached device scratch buffer of min(needed, budget) bytes.  The budget is a fraction of the memory that is
    free right now (plus what the cached buffer already holds); MI3D_SCATTER_WORKSPACE_GB overrides it."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    cur = _WORKSPACE.get(idx)
    have = cur.numel() if cur is not None else 0
    if have >= needed:
        return cur
    env = os.environ.get("MI3D_SCATTER_WORKSPACE_GB")
    if env is not None:
        budget = int(float(env) * (1 << 30))
    else:
        free, _ = torch.cuda.mem_get_info(idx)
        budget = int((free + have) * budget_fraction)
    want = min(int(needed), budget)
    if want <= have:
        return cur
    _WORKSPACE[idx] = None
    del cur
    _WORKSPACE[idx] = torch.empty(want, dtype=torch.uint8, device=device) if want > 0 else None
    return _WORKSPACE[idx]


def scatter_binned(x, x2, offsets, P0, bound, dplanes, cfg, step, n_params, workspace_bytes=None, level_absmax=None):
    """grad_params [n_params] from level-major feature-gradient planes [L][n*P][2] (C ABI: mi3d_grid_scatter_binned).
    workspace_bytes: None = cached scratch sized by scatter_workspace; 0 = force the all-atomic path.
    level_absmax: float32[16] device tensor of per-level max |dplanes| (from the MLP backward) - selects the 8-byte
    binary16 records for the fine levels; None keeps fp32 records everywhere."""
    offs, offs_p = grid_ops._offs_arg(offsets)
    P, n = offs.shape[0], x.shape[0]
    dplanes = L.dev_f32(dplanes, "dplanes")
    if dplanes.numel() != cfg["n_levels"] * n * P * 2:
        raise L.Mi3dError(f"dplanes has {dplanes.numel()} elements, expected [L={cfg['n_levels']}][{n * P}][2]")
    grad = torch.zeros(n_params, dtype=torch.float32, device=x.device)
    lib = L.lib()
    ws, ws_bytes = None, 0
    if workspace_bytes != 0 and n > 0:
        needed = lib.mi3d_grid_scatter_binned_workspace(n, P, float(bound), float(step), cfg["n_levels"],
                                                        cfg["base_resolution"], cfg["per_level_scale"],
                                                        cfg["log2_hashmap_size"], 1 if level_absmax is not None else 0)
        if workspace_bytes is not None:
            needed = min(needed, int(workspace_bytes))
            ws = torch.empty(needed, dtype=torch.uint8, device=x.device) if needed else None
        elif needed:
            ws = scatter_workspace(x.device, needed)
        ws_bytes = ws.numel() if ws is not None else 0
    grid_ops._timed("scatter", lambda: L.call(
        "mi3d_grid_scatter_binned", L.ptr(x), L.ptr(x2), n, offs_p, int(P0), P, float(bound), L.ptr(dplanes),
        cfg["n_levels"], cfg["base_resolution"], cfg["per_level_scale"], cfg["log2_hashmap_size"], float(step),
        L.ptr(level_absmax), L.ptr(ws), C.c_size_t(ws_bytes), L.ptr(grad), L.stream()), n * P)
    return grad


class _FieldStencil(Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, params, W1, b1, W2, b2, W3, b3, x, x2, offsets, P0, bound, cfg, step, half_mode):
        x = L.dev_f32(x.contiguous().view(-1, 3), "x", 3)
        if x2 is not None:
            x2 = L.dev_f32(x2.contiguous().view(-1, 3), "x2", 3)
        params = L.dev_f32(params, "params")
        ws = [L.dev_f32(t.contiguous(), "weight") for t in (W1, b1, W2, b2, W3, b3)]
        offs, offs_p = grid_ops._offs_arg(offsets)
        P, n = offs.shape[0], x.shape[0]
        # features as level-major planes [L][n*P][2]: written by the per-XCD gather, read as such by both MLP kernels
        feats = torch.empty(cfg["n_levels"], n * P, 2, dtype=torch.float32, device=x.device)
        grid_ops._timed("encode", lambda: L.call(
            "mi3d_grid_encode_points_planes", L.ptr(x), L.ptr(x2), n, offs_p, int(P0), P, float(bound), L.ptr(params),
            cfg["n_levels"], cfg["base_resolution"], cfg["per_level_scale"], cfg["log2_hashmap_size"], L.ptr(feats),
            L.stream()), n * P)
        dims = (W1.shape[1], W1.shape[0], W3.shape[0])
        h = torch.empty(n * P, dims[2], dtype=torch.float32, device=x.device)
        grid_ops._timed("mlp_fwd", lambda: L.call(
            "mi3d_mlp_forward", L.ptr(feats), 1, n * P, *[L.ptr(t) for t in ws], *dims, int(half_mode), L.ptr(h),
            L.stream()), n * P)
        ctx.save_for_backward(x, x2 if x2 is not None else x, feats, *ws)
        ctx.meta = (offs, int(P0), float(bound), cfg, float(step), x2 is not None, params.numel(), dims,
                    int(half_mode))
        return h

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dh):
        x, x2, feats, *ws = ctx.saved_tensors
        offs, P0, bound, cfg, step, has_x2, n_params, dims, half_mode = ctx.meta
        rows = feats.shape[1]
        dh = L.dev_f32(dh.float().contiguous(), "dh", dims[2])
        dplanes = torch.empty(cfg["n_levels"], rows, 2, dtype=torch.float32, device=x.device)
        grads = [torch.zeros_like(t) for t in ws]
        # under autocast the feature gradients are binary16-precise: the scatter may then use its 8-byte records,
        # scaled per level by the maxima the MLP backward collects
        absmax = torch.zeros(cfg["n_levels"], dtype=torch.float32, device=x.device) if half_mode else None
        grid_ops._timed("mlp_bwd", lambda: L.call(
            "mi3d_mlp_backward", L.ptr(feats), 1, L.ptr(dh), rows, *[L.ptr(t) for t in ws], *dims, half_mode,
            L.ptr(dplanes), 1, L.ptr(absmax), *[L.ptr(g) for g in grads], L.stream()), rows)
        gp = scatter_binned(x, x2 if has_x2 else None, offs, P0, bound, dplanes, cfg, step, n_params,
                            level_absmax=absmax)
        return (gp, *grads, None, None, None, None, None, None, None, None)


def field_stencil(params, layers, x, offsets, cfg, bound=1.0, x2=None, P0=None, step=0.0, half_mode=None):
    """h [n*P, 4] (row = sample*P + point): the MLP output at clamp(base + offsets[p]) for every sample, differentiable
    w.r.t. the hash table and the MLP weights.  `layers`: the three nn.Linear modules of sigma_net."""
    P = np.asarray(offsets).reshape(-1, 3).shape[0]
    if P0 is None:
        P0 = P
    if half_mode is None:
        half_mode = torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.float16
    l1, l2, l3 = layers
    return _FieldStencil.apply(params, l1.weight, l1.bias, l2.weight, l2.bias, l3.weight, l3.bias, x, x2, offsets, P0,
                               bound, cfg, step, bool(half_mode))


class _FieldHead(Function):
    """sigma, albedo, normal(x), normal(x2) from h [n*P, 4] in one elementwise kernel per direction (C ABI Part 5)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, h, x, x2, offsets, bound, blob_density, blob_radius, epsilon):
        offs, offs_p = grid_ops._offs_arg(offsets)
        P = offs.shape[0]
        x = L.dev_f32(x.contiguous().view(-1, 3), "x", 3)
        n = x.shape[0]
        h = L.dev_f32(h.contiguous().view(n * P, 4), "h", 4)
        if x2 is not None:
            x2 = L.dev_f32(x2.contiguous().view(-1, 3), "x2", 3)
        dev = x.device
        sigma = torch.empty(n, dtype=torch.float32, device=dev)
        albedo = torch.empty(n, 3, dtype=torch.float32, device=dev)
        normal = torch.empty(n, 3, dtype=torch.float32, device=dev)
        normal2 = torch.empty(n, 3, dtype=torch.float32, device=dev) if P == 13 else None
        grid_ops._timed("head_fwd", lambda: L.call(
            "mi3d_field_head_forward", L.ptr(h), L.ptr(x), L.ptr(x2), n, offs_p, P, float(bound), float(blob_density),
            float(blob_radius), float(epsilon), L.ptr(sigma), L.ptr(albedo), L.ptr(normal), L.ptr(normal2),
            L.stream()), n)
        ctx.save_for_backward(h, x, x2 if x2 is not None else x)
        ctx.meta = (offs, P, float(bound), float(blob_density), float(blob_radius), float(epsilon), x2 is not None)
        if normal2 is None:
            return sigma, albedo, normal
        return sigma, albedo, normal, normal2

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, *grads):
        h, x, x2 = ctx.saved_tensors
        offs, P, bound, blob_density, blob_radius, epsilon, has_x2 = ctx.meta
        _, offs_p = grid_ops._offs_arg(offs)
        g = [None if t is None else L.dev_f32(t.float().contiguous(), "grad") for t in grads] + [None]
        dh = torch.empty_like(h)
        n = x.shape[0]
        grid_ops._timed("head_bwd", lambda: L.call(
            "mi3d_field_head_backward", L.ptr(h), L.ptr(x), L.ptr(x2 if has_x2 else None), n, offs_p, P, bound,
            blob_density, blob_radius, epsilon, L.ptr(g[0]), L.ptr(g[1]), L.ptr(g[2]), L.ptr(g[3]), L.ptr(dh),
            L.stream()), n)
        return dh, None, None, None, None, None, None, None


def field_head(h, x, offsets, bound, blob_density, blob_radius, x2=None, epsilon=grid_ops.EPS):
    """(sigma [n], albedo [n,3], normal [n,3], normal_jitter [n,3] or None) from h [n*P, 4]; P must be 7 or 13."""
    out = _FieldHead.apply(h, x, x2, offsets, bound, blob_density, blob_radius, epsilon)
    return out if len(out) == 4 else (*out, None)
