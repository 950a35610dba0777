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


SHAPES = [(8, 32, 2), (8, 32, 3), (32, 32, 3), (32, 64, 2), (16, 64, 3), (2, 32, 2), (30, 64, 3)]  # (dim_in, hidden, layers)


def _mlp_g(dev, din, hid, layers, seed=0):
    from mi3d.network import MLP
    torch.manual_seed(seed)
    m = MLP(din, 4, hid, layers).to(dev)
    with torch.no_grad():
        for l in m.net:
            l.bias.uniform_(-0.3, 0.3)
    return m


@pytest.mark.parametrize("din,hid,layers", SHAPES)
@pytest.mark.parametrize("n", [33, 5000])
def test_generic_shapes_fp32(cuda, din, hid, layers, n):
    """Every shape the reference's MLP class can build around this field (network_tcnn.py:13-32,37-45; BASELINE config 1
    is 8 -> 32 -> 4) on the matrix cores, forward + every gradient, against fp64 torch."""
    from mi3d import mlp_ops
    m = _mlp_g(cuda, din, hid, layers, seed=3)
    x = torch.randn(n, din, device=cuda, requires_grad=True)
    assert m.fused_ok(x)
    g = torch.randn(n, 4, device=cuda)
    g[::5] = 0
    y = mlp_ops.fused_mlp(x, m.net, half_mode=False)
    y.backward(g)
    got = [y.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in m.parameters()]
    md = _mlp_g(cuda, din, hid, layers, seed=3).double()
    xd = x.detach().double().requires_grad_()
    yd = _torch_forward(md, xd)
    yd.backward(g.double())
    want = [yd.detach(), xd.grad] + [p.grad for p in md.parameters()]
    names = ["y", "dx"] + [n_ for n_, _ in m.named_parameters()]
    for nm, a, b in zip(names, got, want):
        b = b.float()
        scale = float(b.abs().max()) + 1e-12
        assert float((a - b).abs().max()) <= 1e-4 * scale, (nm, din, hid, layers)


@pytest.mark.parametrize("din,hid,layers", SHAPES)
def test_generic_shapes_fp16_mode_matches_autocast(cuda, din, hid, layers):
    m = _mlp_g(cuda, din, hid, layers, seed=4)
    n = 20000
    x = (torch.randn(n, din, device=cuda) * 0.5).requires_grad_()
    g = torch.randn(n, 4, device=cuda) * 64
    with torch.autocast("cuda", dtype=torch.float16):
        y = m(x)
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
        assert float((a - b).abs().max()) <= 8e-3 * scale, (nm, din, hid, layers)


@pytest.mark.parametrize("din,hid,layers", [(32, 64, 3), (8, 32, 2), (16, 64, 2)])
@pytest.mark.parametrize("half", [False, True])
def test_plane_layouts_equal_row_layout(cuda, din, hid, layers, half):
    """x / dx as level-major planes [din/2][rows][2] (fp32 and binary16 pairs) through the C ABI = the [n, din] rows."""
    from mi3d import _lib as L, mlp_ops
    m = _mlp_g(cuda, din, hid, layers, seed=5)
    n, rows = 3000, 3333   # plane_rows > n: a prefix of wider planes
    x = torch.randn(n, din, device=cuda) * 0.5
    if half:
        x = x.half().float()
    g = torch.randn(n, 4, device=cuda)
    ws = [None if t is None else t.detach().contiguous() for t in mlp_ops.layer_args(m.net)]
    dims = (din, hid, 4)
    mode = 1 if half else 0

    def run(xt, x_rows, dx, dx_rows, ph):
        out = torch.empty(n, 4, device=cuda)
        grads = [None if t is None else torch.zeros_like(t) for t in ws]
        L.call("mi3d_mlp_forward", L.ptr(xt), x_rows, ph, n, *[L.ptr(t) for t in ws], *dims, mode, L.ptr(out), L.stream())
        L.call("mi3d_mlp_backward", L.ptr(xt), x_rows, ph, L.ptr(g), n, *[L.ptr(t) for t in ws], *dims, mode, L.ptr(dx),
               dx_rows, *[L.ptr(t) for t in grads], L.stream())
        return out, grads
    dx_rows_t = torch.empty(n, din, device=cuda)
    o0, g0 = run(x.contiguous(), 0, dx_rows_t, 0, 0)
    planes = torch.zeros(din // 2, rows, 2, device=cuda)
    planes[:, :n] = x.view(n, din // 2, 2).permute(1, 0, 2)
    if half:
        planes = planes.half()
    dxp = torch.zeros(din // 2, rows, 2, device=cuda, dtype=planes.dtype)
    o1, g1 = run(planes.contiguous(), rows, dxp, rows, int(half))
    assert torch.equal(o0, o1)
    dx1 = dxp[:, :n].permute(1, 0, 2).reshape(n, din).float()
    want = dx_rows_t.half().float() if half else dx_rows_t
    np.testing.assert_allclose(dx1.cpu().numpy(), want.cpu().numpy(), rtol=0, atol=0)
    assert float(dxp[:, n:].abs().max()) == 0.0   # rows past n are not touched
    for a, b in zip(g0, g1):
        if a is not None:   # float-atomic sums over the workgroups: equal up to summation order
            np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=0, atol=2e-5 * float(a.abs().max()) + 1e-7)


def test_uncovered_shape_raises_on_the_gpu(cuda):
    """No eager / library-GEMM path on the GPU: a shape outside the kernels' coverage is an error."""
    from mi3d import _lib as L
    from mi3d.network import MLP
    for m in (MLP(8, 4, 48, 2), MLP(64, 4, 64, 3), MLP(8, 3, 32, 2), MLP(8, 4, 32, 4), MLP(8, 4, 32, 2, bias=False)):
        m = m.to(cuda)
        x = torch.randn(10, m.dim_in, device=cuda)
        assert not m.fused_ok(x)
        with pytest.raises(L.Mi3dError):
            m(x)
