"""The field's MLP as ONE matrix-core kernel per direction (csrc/field.hip, Part 4 of include/mi3d.h) behind
torch.autograd.  Under torch.autocast(float16) the kernel runs its binary16-MFMA mode, which rounds exactly where
autocast rounds an nn.Linear stack; otherwise it is exact fp32."""
import torch
from torch.autograd import Function

from . import _lib as L
from . import grid_ops


def supported(dim_in, dim_hidden, dim_out, num_layers):
    return bool(L.lib().mi3d_mlp_supported(int(dim_in), int(dim_hidden), int(dim_out), int(num_layers)))


class _FusedMLP(Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x, W1, b1, W2, b2, W3, b3, half_mode):
        x = L.dev_f32(x.contiguous(), "x", W1.shape[1])
        ws = [L.dev_f32(t.contiguous(), "weight") for t in (W1, b1, W2, b2, W3, b3)]
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
        grads = [torch.zeros_like(t) for t in ws]
        with L.on(x):
            grid_ops._timed("mlp_bwd", lambda: L.call(
                "mi3d_mlp_backward", L.ptr(x), 0, 0, L.ptr(dout), n, *[L.ptr(t) for t in ws], *dims, half_mode, L.ptr(dx),
                0, *[L.ptr(g) for g in grads], L.stream(x)), n)
        return (dx, *grads, None)


def fused_mlp(x, layers, half_mode=None):
    """layers: the three nn.Linear modules.  half_mode None = follow torch.autocast."""
    if half_mode is None:
        half_mode = torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.float16
    l1, l2, l3 = layers
    return _FusedMLP.apply(x, l1.weight, l1.bias, l2.weight, l2.bias, l3.weight, l3.bias, bool(half_mode))
