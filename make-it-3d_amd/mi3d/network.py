"""NeRFNetwork - the reference's hash-grid field (`nerf.network_tcnn.NeRFNetwork`,
/root/reference/nerf/network_tcnn.py:37-206) on the MI355X kernels.

Same constructor (`NeRFNetwork(opt, bg_color=None, num_layers=3, hidden_dim=64, ...)`), same parameters and
state_dict keys (`encoder.params`, `sigma_net.net.{l}.{weight,bias}`), same methods
(`common_forward / density / normal / finite_difference_normal / forward / gaussian / get_params`).
The difference is `field_stencil`: the value at x, the six finite-difference neighbours of x and the six
neighbours of the jittered x2 (13 encoder+MLP passes in the reference) are evaluated - and back-propagated -
as ONE stencil-aware encode (mi3d.grid_ops) followed by ONE MLP over [13*m, 2L] rows (point-major: row = p*m + s).
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

import tinycudann as tcnn

from . import field_ops, grid_ops, mlp_ops
from .renderer import NeRFRenderer, safe_normalize


class _TruncExp(torch.autograd.Function):
    """exp forward, gradient computed at min(x, 15) (activation.py:5-18); fp32 under autocast."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g * torch.exp(x.clamp(max=15))


trunc_exp = _TruncExp.apply


class MLP(nn.Module):
    """Linear/ReLU stack with the reference's module layout (`net.{l}`), network_tcnn.py:13-32."""

    def __init__(self, dim_in, dim_out, dim_hidden, num_layers, bias=True):
        super().__init__()
        self.dim_in, self.dim_out, self.dim_hidden, self.num_layers = dim_in, dim_out, dim_hidden, num_layers
        self.net = nn.ModuleList([
            nn.Linear(dim_in if l == 0 else dim_hidden, dim_out if l == num_layers - 1 else dim_hidden, bias=bias)
            for l in range(num_layers)])

    def fused_ok(self, x):
        """The matrix-core kernels cover every shape network_tcnn.py:37-45,67 can build around this field: input width
        2 x levels <= 32, hidden 32 or 64, 2 or 3 layers, 4 outputs, biased (csrc/field.hip: k_mlp_fwd_g / k_mlp_bwd_g)."""
        return (x.is_cuda and all(l.bias is not None for l in self.net)
                and mlp_ops.supported(self.dim_in, self.dim_hidden, self.dim_out, self.num_layers))

    def forward(self, x):
        if self.fused_ok(x):
            return mlp_ops.fused_mlp(x, self.net)
        if x.is_cuda:  # no eager path on the GPU: a shape the kernels do not cover is an error, not a silent library GEMM
            raise mlp_ops.L.Mi3dError(
                f"MLP {self.dim_in} -> {self.num_layers - 1} x {self.dim_hidden} -> {self.dim_out} (bias "
                f"{all(l.bias is not None for l in self.net)}) is not covered by the matrix-core kernels "
                f"(dim_in even and <= 32, hidden 32 or 64, 2 or 3 layers, 4 outputs, biased)")
        for l, layer in enumerate(self.net):  # CPU tensors only: the host-side tests of the module surface
            x = layer(x)
            if l != self.num_layers - 1:
                x = F.relu(x, inplace=True)
        return x


class NeRFNetwork(NeRFRenderer):
    def __init__(self, opt, bg_color=None, num_layers=3, hidden_dim=64, num_layers_bg=2, hidden_dim_bg=64,
                 n_levels=16, log2_hashmap_size=19, base_resolution=16, per_level_scale=None):
        super().__init__(opt)
        self.num_layers, self.hidden_dim = num_layers, hidden_dim
        if per_level_scale is None:  # network_tcnn.py:52
            per_level_scale = np.exp2(np.log2(2048 * self.bound / 16) / (16 - 1))
        self.encoder = tcnn.Encoding(n_input_dims=3, encoding_config={
            "otype": "HashGrid", "n_levels": n_levels, "n_features_per_level": 2,
            "log2_hashmap_size": log2_hashmap_size, "base_resolution": base_resolution,
            "per_level_scale": per_level_scale}, dtype=torch.float32)
        self.sigma_net = MLP(self.encoder.n_output_dims, 4, hidden_dim, num_layers, bias=True)
        if self.bg_radius > 0:
            raise NotImplementedError("bg_radius > 0 (background network) is not on Make-It-3D's path (main.py:54)")
        self.bg_net = None
        if bg_color is not None:
            bg64 = F.interpolate(bg_color, size=64, mode="bicubic", align_corners=True)
            self.register_buffer("bg_color", bg64.permute(0, 2, 3, 1).reshape(-1, 3).clone(), persistent=False)
        else:
            self.register_buffer("bg_color", torch.rand((4096, 3)), persistent=False)

    # ------------------------------------------------------------------ field pieces
    def gaussian(self, x):
        d = (x ** 2).sum(-1)
        return self.opt.blob_density * torch.exp(-d / (2 * self.opt.blob_radius ** 2))

    @staticmethod
    def _sigmoid(h):
        """network_tcnn.py:110 under torch.autocast(float16): the MLP output is binary16 there, so torch.sigmoid returns
        binary16.  The fused MLP hands back an fp32 buffer of binary16-representable values: cast it back first."""
        if h.is_cuda and field_ops._half_mode(None):
            h = h.to(torch.float16)
        return torch.sigmoid(h)

    def _head(self, h, x):
        sigma = trunc_exp(h[..., 0] + self.gaussian(x))
        albedo = self._sigmoid(h[..., 1:])
        return sigma, albedo

    def _encode(self, x, offsets, x2=None, P0=None, step=0.0):
        return grid_ops.encode_points(self.encoder.params, x, offsets, self.encoder.cfg, float(self.bound), x2, P0,
                                      step)

    _CENTER = np.zeros((1, 3), np.float32)

    def common_forward(self, x):
        """sigma [n], albedo [n,3] at x in [-bound, bound]^3 (network_tcnn.py:102-112)."""
        x = x.reshape(-1, 3).float()
        return self._head(self.sigma_net(self._encode(x, self._CENTER)), x)

    def _stencil_sigma(self, x, offsets, x2=None, P0=None, step=0.0):
        """sigma at clamp(base + offsets[p]) for every sample: [n, P]; plus the MLP output h [n, P, 4] (a view of the
        point-major [P, n, 4] the kernels produce)."""
        n, P = x.shape[0], offsets.shape[0]
        if self.sigma_net.fused_ok(x) and self.encoder.cfg["n_levels"] * 2 == self.sigma_net.dim_in:
            h = field_ops.field_stencil(self.encoder.params, self.sigma_net.net, x, offsets, self.encoder.cfg,
                                        float(self.bound), x2, P0, step)
        else:
            h = self.sigma_net(self._encode(x, offsets, x2, P0, step))
        h = h.view(P, n, 4).transpose(0, 1)
        offs = torch.from_numpy(offsets).to(x.device)
        base = x.unsqueeze(1).expand(n, P, 3)
        if x2 is not None:
            base = torch.cat([base[:, :P0], x2.unsqueeze(1).expand(n, P - P0, 3)], 1)
        pts = (base + offs).clamp(-self.bound, self.bound)
        sigma = trunc_exp(h[..., 0] + self.gaussian(pts))
        return sigma, h

    @staticmethod
    def _normal_from(s6, epsilon=grid_ops.EPS):
        """-(central differences) -> safe_normalize -> nan_to_num (network_tcnn.py:124-138). s6: [n,6]."""
        g = torch.stack([0.5 * (s6[:, 0] - s6[:, 1]) / epsilon, 0.5 * (s6[:, 2] - s6[:, 3]) / epsilon,
                         0.5 * (s6[:, 4] - s6[:, 5]) / epsilon], -1)
        return torch.nan_to_num(safe_normalize(-g))

    def finite_difference_normal(self, x, epsilon=grid_ops.EPS):
        offs, _ = grid_ops.stencil_offsets(center=False)
        if epsilon != grid_ops.EPS:
            offs = offs / np.float32(grid_ops.EPS) * np.float32(epsilon)
        s6, _ = self._stencil_sigma(x.reshape(-1, 3).float(), offs)
        return -torch.stack([0.5 * (s6[:, 0] - s6[:, 1]) / epsilon, 0.5 * (s6[:, 2] - s6[:, 3]) / epsilon,
                             0.5 * (s6[:, 4] - s6[:, 5]) / epsilon], -1)

    def normal(self, x):
        return torch.nan_to_num(safe_normalize(self.finite_difference_normal(x)))

    def field_stencil(self, x, x2=None, step=0.0):
        x = x.reshape(-1, 3).float()
        offs, P0 = grid_ops.stencil_offsets(center=True, second=x2 is not None)
        if x.is_cuda and self.sigma_net.fused_ok(x) and self.encoder.cfg["n_levels"] * 2 == self.sigma_net.dim_in:
            # ONE node for encode + MLP + head (sigma, albedo, both finite-difference normals); its backward runs over
            # the stencil points the upstream gradient actually reaches (field_ops._Field)
            return field_ops.field(self.encoder.params, self.sigma_net.net, x, offs, self.encoder.cfg, float(self.bound),
                                   self.opt.blob_density, self.opt.blob_radius, x2, P0 if x2 is not None else None, step)
        sig, h = self._stencil_sigma(x, offs, x2, P0 if x2 is not None else None, step)
        albedo = self._sigmoid(h[:, 0, 1:])
        normals = self._normal_from(sig[:, 1:7])
        normals_jitter = self._normal_from(sig[:, 7:13]) if x2 is not None else None
        return sig[:, 0], albedo, normals, normals_jitter

    def shade(self, albedo, normal, light_d, ratio, shading, batch_rows=None):
        """network_tcnn.py:146-168, including the silent skip of shading for batches of >= 1e6 samples.  `batch_rows`:
        the size of the batch the REFERENCE would have handed over where the buffers here are larger than it (the
        compact inference rounds evaluate a buffer of `budget` rows; the reference's eval batches hold <= N + 128)."""
        if shading == "albedo" or (normal.shape[0] if batch_rows is None else batch_rows) >= 1e6:
            return albedo
        lambertian = ratio + (1 - ratio) * (normal @ light_d).clamp(min=0.1)
        if shading == "textureless":
            return lambertian.unsqueeze(-1).repeat(1, 3)
        if shading == "normal":
            return (normal + 1) / 2
        return albedo * lambertian.unsqueeze(-1)

    def forward(self, x, d, l=None, ratio=1, shading="albedo"):
        """sigma [n], color [n,3], normal [n,3] (network_tcnn.py:140-170); 7 evaluations in one stencil pass."""
        rows = getattr(self, "_infer_rows", None)
        if (rows is not None and not torch.is_grad_enabled() and x.is_cuda and self.sigma_net.fused_ok(x)
                and self.encoder.cfg["n_levels"] * 2 == self.sigma_net.dim_in):
            # the inference loop (renderer.run_cuda, eval): the round's row count lives on the device
            offs, _ = grid_ops.stencil_offsets(center=True, second=False)
            sigma, albedo, normal = field_ops.field_rows(self.encoder.params, self.sigma_net.net, x.reshape(-1, 3), offs,
                                                         self.encoder.cfg, float(self.bound), self.opt.blob_density,
                                                         self.opt.blob_radius, rows)
            return sigma, self.shade(albedo, normal, l, ratio, shading, getattr(self, "_infer_shade_rows", None)), normal
        sigma, albedo, normal, _ = self.field_stencil(x)
        return sigma, self.shade(albedo, normal, l, ratio, shading), normal

    def density(self, x):
        sigma, albedo = self.common_forward(x)
        return {"sigma": sigma, "albedo": albedo}

    def background(self, d):
        raise NotImplementedError("no background network (bg_radius <= 0 on this path)")

    def get_params(self, lr):
        return [{"params": self.encoder.parameters(), "lr": lr * 10}, {"params": self.sigma_net.parameters(), "lr": lr}]
