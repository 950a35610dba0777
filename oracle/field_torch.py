"""ORACLE (test infrastructure, NOT product code).

Pure-PyTorch (CPU) restatement of the field, differentiable, used as the gradient oracle:

* `HashGridTorch`  - the tiny-cuda-nn HashGrid the reference constructs at
  /root/reference/nerf/network_tcnn.py:54-65.  PARITY UNPINNED (tcnn is an un-vendored,
  un-pinned dependency, see oracle/hashgrid_ref.c); bit-for-bit the same index/weight
  arithmetic as hashgrid_ref.c (checked in tests/test_oracle_cpu.py), gradients by autograd
  (index_add == tcnn's atomicAdd scatter up to summation order).
* `FieldTorch`     - network_tcnn.py:13-32 (MLP), :94-170 (blob, common_forward,
  finite_difference_normal, normal, forward) and activation.py:5-18 (trunc_exp incl. its
  clamped backward), restated.  Validated against the reference's own NeRFNetwork class
  imported from /root/reference by tests/golden/make_golden_py.py (runs only where the
  reference tree exists) and through the committed fixtures elsewhere.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

PRIME_Y, PRIME_Z = 2654435761, 805459861


class HashGridTorch(nn.Module):
    def __init__(self, cfg, params=None):
        super().__init__()
        self.cfg = cfg
        self.n_output_dims = cfg.n_output_dims
        if params is None:
            g = torch.Generator().manual_seed(1337)
            params = (torch.rand(cfg.n_params, generator=g) * 2 - 1) * 1e-4
        self.params = nn.Parameter(torch.as_tensor(params, dtype=torch.float32).clone())

    def _level(self, x, l):
        cfg = self.cfg
        scale = float(cfg.scales[l])
        res = int(cfg.resolutions[l])
        off, hs = int(cfg.offsets[l]), int(cfg.offsets[l + 1] - cfg.offsets[l])
        # pos_fract: fmaf(scale, x, 0.5) - emulate the single rounding through float64
        pos = (x.double() * scale + 0.5).float()
        fl = torch.floor(pos)
        g = fl.to(torch.int64)
        w = pos - fl
        hashed = None
        # grid_index: which dims the stride loop covers
        stride, n_dims = 1, 0
        for _ in range(3):
            if stride > hs:
                break
            stride *= res
            n_dims += 1
        hashed = hs < stride
        feats = 0
        table = self.params.view(-1, 2)
        for k in range(8):
            wk = 1
            q = []
            for d in range(3):
                if (k >> d) & 1:
                    wk = wk * w[:, d]
                    q.append(g[:, d] + 1)
                else:
                    wk = wk * (1 - w[:, d])
                    q.append(g[:, d])
            q = [qi & 0xFFFFFFFF for qi in q]
            if hashed:
                idx = (q[0] ^ ((q[1] * PRIME_Y) & 0xFFFFFFFF) ^ ((q[2] * PRIME_Z) & 0xFFFFFFFF))
            else:
                idx, s = 0, 1
                for d in range(n_dims):
                    idx = idx + q[d] * s
                    s *= res
                idx = idx & 0xFFFFFFFF
            idx = idx % hs + off
            feats = feats + wk[:, None] * table[idx]
        return feats

    def forward(self, x):
        x = x.float().reshape(-1, 3)
        return torch.cat([self._level(x, l) for l in range(self.cfg.n_levels)], -1)


class _TruncExp(torch.autograd.Function):
    """activation.py:5-18"""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g * torch.exp(x.clamp(max=15))


def safe_normalize(x, eps=1e-20):
    return x / torch.sqrt(torch.clamp(torch.sum(x * x, -1, keepdim=True), min=eps, max=1e32))


class FieldTorch(nn.Module):
    def __init__(self, fp):
        """fp: oracle.FieldParams (numpy) - copied into torch parameters."""
        super().__init__()
        self.fp = fp
        self.bound = fp.bound
        self.encoder = HashGridTorch(fp.cfg, fp.params)
        self.W = nn.ParameterList([nn.Parameter(torch.from_numpy(w.copy())) for w in fp.W])
        self.B = nn.ParameterList([nn.Parameter(torch.from_numpy(b.copy())) for b in fp.B])

    def mlp(self, h):
        for l in range(len(self.W)):
            h = F.linear(h, self.W[l], self.B[l])
            if l != len(self.W) - 1:
                h = F.relu(h)
        return h

    def blob(self, x):
        d = (x ** 2).sum(-1)
        return self.fp.blob_density * torch.exp(-d / (2 * self.fp.blob_radius ** 2))

    def common_forward(self, x):
        h = (x + self.bound) / (2 * self.bound)
        h = self.mlp(self.encoder(h))
        sigma = _TruncExp.apply(h[..., 0] + self.blob(x))
        albedo = torch.sigmoid(h[..., 1:])
        return sigma, albedo

    def stencil_sigmas(self, x, epsilon=1e-2):
        offs = torch.tensor([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]],
                            dtype=torch.float32) * epsilon
        return torch.stack([self.common_forward((x + o).clamp(-self.bound, self.bound))[0] for o in offs], -1)

    @staticmethod
    def normal_from_stencil(s6, epsilon=1e-2):
        g = torch.stack([0.5 * (s6[:, 0] - s6[:, 1]) / epsilon, 0.5 * (s6[:, 2] - s6[:, 3]) / epsilon,
                         0.5 * (s6[:, 4] - s6[:, 5]) / epsilon], -1)
        return torch.nan_to_num(safe_normalize(-g))

    def normal(self, x):
        return self.normal_from_stencil(self.stencil_sigmas(x))

    def forward(self, x, d, l=None, ratio=1, shading="albedo"):
        sigma, albedo = self.common_forward(x)
        normal = self.normal(x)
        if shading == "albedo" or normal.shape[0] >= 1e6:
            color = albedo
        else:
            lam = ratio + (1 - ratio) * (normal @ l).clamp(min=0.1)
            if shading == "textureless":
                color = lam.unsqueeze(-1).repeat(1, 3)
            elif shading == "normal":
                color = (normal + 1) / 2
            else:
                color = albedo * lam.unsqueeze(-1)
        return sigma, color, normal

    def density(self, x):
        sigma, albedo = self.common_forward(x)
        return {"sigma": sigma, "albedo": albedo}
