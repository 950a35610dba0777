"""GPU parity of the fused field head (csrc/field.hip Part 5: sigma, albedo and both finite-difference normals from the
MLP output of the 7 / 13 stencil points) against the plain PyTorch composition of the same ops, forward and backward,
including the degenerate rows where safe_normalize's clamp and nan_to_num decide the gradient."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _torch_head(h, x, x2, offs, bound, density, radius, eps=1e-2):
    n, P = x.shape[0], offs.shape[0]
    h = h.view(P, n, 4).transpose(0, 1)   # rows are point-major: p*n + s
    o = torch.from_numpy(offs).to(x.device)
    base = x.unsqueeze(1).expand(n, P, 3)
    if x2 is not None:
        base = torch.cat([base[:, :7], x2.unsqueeze(1).expand(n, P - 7, 3)], 1)
    pts = (base + o).clamp(-bound, bound)
    gauss = density * torch.exp(-(pts ** 2).sum(-1) / (2 * radius ** 2))
    from mi3d.network import trunc_exp
    from mi3d.renderer import safe_normalize
    sig = trunc_exp(h[..., 0] + gauss)

    def normal(s6):
        g = torch.stack([0.5 * (s6[:, 0] - s6[:, 1]) / eps, 0.5 * (s6[:, 2] - s6[:, 3]) / eps,
                         0.5 * (s6[:, 4] - s6[:, 5]) / eps], -1)
        return torch.nan_to_num(safe_normalize(-g))
    out = [sig[:, 0], torch.sigmoid(h[:, 0, 1:]), normal(sig[:, 1:7])]
    if P == 13:
        out.append(normal(sig[:, 7:13]))
    return out


@pytest.mark.parametrize("second", [False, True])
def test_head_matches_torch_composition(cuda, second):
    from mi3d import field_ops, grid_ops
    torch.manual_seed(3)
    n = 5000
    offs, P0 = grid_ops.stencil_offsets(center=True, second=second)
    P = offs.shape[0]
    x = (torch.rand(n, 3, device=cuda) * 2 - 1) * 0.9
    x[:50] *= 0.05           # inside the blob: large sigma
    x[50:60] = 1.0           # stencil clamps at the box
    x2 = (x + torch.randn_like(x) * 0.01) if second else None
    h = (torch.randn(n * P, 4, device=cuda) * 0.5)
    hv = h.view(P, n, 4).transpose(0, 1)
    hv[100:200, 1:, 0] = hv[100:200, 1:2, 0]   # identical neighbours: zero finite difference -> clamp regime
    hv[200:210, :, 0] = 30.0                  # beyond trunc_exp's derivative clamp
    hv[:, 1:7, 0] = hv[:, 1:7, 0] * 0.01 + hv[:, :1, 0]   # neighbours close to the centre value
    h = h.detach().requires_grad_(True)
    got = field_ops.field_head(h, x, offs, 1.0, 5.0, 0.1, x2)
    got = [g for g in got if g is not None]
    href = h.detach().clone().requires_grad_(True)
    want = _torch_head(href, x, x2, offs, 1.0, 5.0, 0.1)
    assert len(got) == len(want)
    for a, b, name in zip(got, want, ["sigma", "albedo", "normal", "normal2"]):
        # normals are ratios of differences of nearly equal exponentials: 1 ulp of expf moves them by ~1e-5
        np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=2e-5,
                                   atol=5e-5 if name.startswith("normal") else 2e-6, err_msg=name)
    gs = [torch.randn_like(t) for t in want]
    torch.autograd.backward(got, gs)
    torch.autograd.backward(want, gs)
    a, b = h.grad.view(P, n, 4).transpose(0, 1), href.grad.view(P, n, 4).transpose(0, 1)
    scale = b.abs().amax(dim=(1, 2), keepdim=True) + 1e-20   # per sample: the normal gradients span many decades
    assert float(((a - b).abs() / scale).max()) < 2e-3
    assert torch.isfinite(a).all() == torch.isfinite(b).all()
