"""GPU parity of the matrix-core MLP (csrc/field.hip, include/mi3d.h Part 4) against a plain PyTorch reference of
the same op: nn.Linear(32,64)-ReLU-nn.Linear(64,64)-ReLU-nn.Linear(64,4) (network_tcnn.py:13-32,67).
fp32 mode: 1e-4 relative (BASELINE.json); fp16 mode: against torch.autocast(float16) of the same stack, to binary16
resolution."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _mlp(dev, seed=0, scale=1.0):
    from mi3d.network import MLP
    torch.manual_seed(seed)
    m = MLP(32, 4, 64, 3).to(dev)
    with torch.no_grad():
        for l in m.net:
            l.weight.mul_(scale)
            l.bias.uniform_(-0.3, 0.3)
    return m


def _torch_forward(m, x):
    for i, l in enumerate(m.net):
        x = F.linear(x, l.weight, l.bias)
        if i != len(m.net) - 1:
            x = F.relu(x)
    return x


@pytest.mark.parametrize("n", [1, 31, 32, 33, 1000, 200003])
def test_forward_fp32(cuda, n):
    from mi3d import mlp_ops
    m = _mlp(cuda)
    x = torch.randn(n, 32, device=cuda)
    assert m.fused_ok(x)
    with torch.no_grad():
        got = mlp_ops.fused_mlp(x, m.net, half_mode=False)
        want = _torch_forward(m.double(), x.double()).float()
    assert got.shape == (n, 4)
    np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=1e-4, atol=1e-5)


def test_forward_is_asymmetric_weight_exact(cuda):
    """A = I style check with asymmetric weights: W1 picks features, W2 permutes, W3 reads four distinct channels -
    any row/column or k-slot mix-up in the register chaining shows up as a wrong (not merely inexact) value."""
    from mi3d import mlp_ops
    m = _mlp(cuda)
    with torch.no_grad():
        for l in m.net:
            l.weight.zero_(); l.bias.zero_()
        for i in range(64):
            m.net[0].weight[i, (7 * i + 3) % 32] = 1.0 + i / 64       # hidden1[i] = c_i * x[(7i+3)%32]
            m.net[1].weight[i, (5 * i + 11) % 64] = 0.5 + i / 128     # hidden2[i] = d_i * hidden1[(5i+11)%64]
        for o in range(4):
            m.net[2].weight[o, 13 * o + 2] = 1.0 + o
        m.net[2].bias.copy_(torch.tensor([0.1, 0.2, 0.3, 0.4]))
    x = torch.rand(777, 32, device=cuda) + 0.1  # positive: ReLU is the identity
    with torch.no_grad():
        got = mlp_ops.fused_mlp(x, m.net, half_mode=False)
    want = _torch_forward(m, x)
    np.testing.assert_allclose(got.cpu().numpy(), want.detach().cpu().numpy(), rtol=1e-6)


@pytest.mark.parametrize("n", [33, 4096, 200003])
def test_backward_fp32(cuda, n):
    from mi3d import mlp_ops
    m = _mlp(cuda, seed=1)
    x = torch.randn(n, 32, device=cuda, requires_grad=True)
    g = torch.randn(n, 4, device=cuda)
    g[::7] = 0  # rows with no gradient
    y = mlp_ops.fused_mlp(x, m.net, half_mode=False)
    y.backward(g)
    got = [x.grad.clone()] + [p.grad.clone() for p in m.parameters()]
    x.grad = None
    m.zero_grad()
    md = _mlp(cuda, seed=1).double()
    xd = x.detach().double().requires_grad_()
    _torch_forward(md, xd).backward(g.double())
    want = [xd.grad] + [p.grad for p in md.parameters()]
    names = ["dx"] + [n_ for n_, _ in m.named_parameters()]
    for nm, a, b in zip(names, got, want):
        b = b.float()
        scale = float(b.abs().max()) + 1e-12
        err = float((a - b).abs().max())
        assert err <= 1e-4 * scale, (nm, err, scale)


@pytest.mark.parametrize("n", [1000, 131072])
def test_fp16_mode_matches_autocast(cuda, n):
    from mi3d import mlp_ops
    m = _mlp(cuda, seed=2)
    x = (torch.randn(n, 32, device=cuda) * 0.5).requires_grad_()
    g = torch.randn(n, 4, device=cuda) * 64
    with torch.autocast("cuda", dtype=torch.float16):
        y = m(x)  # the module picks the fused kernel and follows autocast
    assert y.dtype == torch.float32
    y.backward(g)
    got = [y.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in m.parameters()]
    x.grad = None
    m.zero_grad()
    with torch.autocast("cuda", dtype=torch.float16):
        yr = _torch_forward(m, x)
    yr.backward(g.half())
    want = [yr.detach().float(), x.grad.float()] + [p.grad.float() for p in m.parameters()]
    names = ["y", "dx"] + [n_ for n_, _ in m.named_parameters()]
    for nm, a, b in zip(names, got, want):
        scale = float(b.abs().max()) + 1e-12
        err = float((a - b).abs().max())
        # binary16 has 11 significant bits: one rounding flip per layer ~ 1e-3 of the value; the weight gradients
        # are sums over n rows of such values (torch additionally rounds ITS sums to binary16)
        assert err <= 8e-3 * scale, (nm, err, scale)
    # forward against an explicit emulation of the roundings (what oracle/field_ref.c half_mode does)
    with torch.no_grad():
        h = x.detach().half().float()
        for i, l in enumerate(m.net):
            h = (h @ l.weight.half().float().t() + l.bias.half().float()).half().float()
            if i != 2:
                h = F.relu(h)
    np.testing.assert_allclose(got[0].cpu().numpy(), h.cpu().numpy(), rtol=4e-3, atol=4e-3)


def test_unsupported_shape_uses_library_gemms(cuda):
    from mi3d.network import MLP
    m = MLP(8, 4, 32, 2).to(cuda)  # BASELINE config 1's head
    x = torch.randn(100, 8, device=cuda)
    assert not m.fused_ok(x)
    assert m(x).shape == (100, 4)
