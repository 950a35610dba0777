"""The field's MLP as ONE matrix-core kernel per direction (csrc/field.hip, Part 4 of include/mi3d.h) behind
torch.autograd.  Under torch.autocast(float16) the kernel runs its binary16-MFMA mode, which rounds exactly where
autocast rounds an nn.Linear stack; otherwise it is exact fp32."""
import torch
from torch.autograd import Function

from . import _lib as L
from . import grid_ops


def supported(dim_in, dim_hidden, dim_out, num_layers):
    return bool(L.lib().mi3d_mlp_supported(int(dim_in), int(dim_hidden), int(dim_out), int(num_layers)))


def layer_args(layers):
    """(W1, b1, W2, b2, W_last, b_last) of a 2- or 3-layer stack; the middle pair is None for two layers (the C ABI
    reads a null W2 / b2 as "two layers")."""
    layers = list(layers)
    if len(layers) == 3:
        l1, l2, l3 = layers
        return (l1.weight, l1.bias, l2.weight, l2.bias, l3.weight, l3.bias)
    if len(layers) == 2:
        l1, l3 = layers
        return (l1.weight, l1.bias, None, None, l3.weight, l3.bias)
    raise L.Mi3dError(f"the matrix-core MLP has 2 or 3 layers, not {len(layers)}")


def _weights(ts):
    return [None if t is None else L.dev_f32(t.contiguous(), "weight") for t in ts]


class _FusedMLP(Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x, W1, b1, W2, b2, W3, b3, half_mode):
        x = L.dev_f32(x.contiguous(), "x", W1.shape[1])
        ws = _weights((W1, b1, W2, b2, W3, b3))
        n = x.shape[0]
        out = torch.empty(n, W3.shape[0], dtype=torch.float32, device=x.device)
        dims = (W1.shape[1], W1.shape[0], W3.shape[0])
        with L.on(x):
            grid_ops._timed("mlp_fwd", lambda: L.call(
                "mi3d_mlp_forward", L.ptr(x), 0, 0, n, *[L.ptr(t) for t in ws], *dims, int(half_mode), L.ptr(out),
                L.stream(x)), n)
        ctx.save_for_backward(x, *ws)
        ctx.meta = (dims, int(half_mode))
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dout):
        x, *ws = ctx.saved_tensors
        dims, half_mode = ctx.meta
        dout = L.dev_f32(dout.float().contiguous(), "dout", dims[2])
        n = x.shape[0]
        dx = torch.empty_like(x)
        grads = [None if t is None else torch.zeros_like(t) for t in ws]
        with L.on(x):
            grid_ops._timed("mlp_bwd", lambda: L.call(
                "mi3d_mlp_backward", L.ptr(x), 0, 0, L.ptr(dout), n, *[L.ptr(t) for t in ws], *dims, half_mode, L.ptr(dx),
                0, *[L.ptr(g) for g in grads], L.stream(x)), n)
        return (dx, *grads, None)


def fused_mlp(x, layers, half_mode=None):
    """layers: the two or three nn.Linear modules.  half_mode None = follow torch.autocast."""
    if half_mode is None:
        half_mode = torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.float16
    return _FusedMLP.apply(x, *layer_args(layers), bool(half_mode))
