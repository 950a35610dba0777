"""GPU parity: csrc/hashgrid.hip (through the `tinycudann.Encoding` drop-in -> C ABI) against the CPU oracle
(oracle/hashgrid_ref.c; tcnn restatement, PARITY UNPINNED for tcnn itself)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


CONFIGS = {
    "default16": dict(n_levels=16, log2_hashmap_size=19, base_resolution=16, per_level_scale=None),
    "c1_L4": dict(n_levels=4, log2_hashmap_size=19, base_resolution=16, per_level_scale=128 ** (1 / 3)),
    "small_hash": dict(n_levels=8, log2_hashmap_size=12, base_resolution=4, per_level_scale=1.7),
}


def _points(rng, n):
    x = rng.random((n, 3)).astype(np.float32)
    x[:16] = 1.0   # upper boundary: the +1 corner wraps in dense levels
    x[16:32] = 0.0
    x[32:40, 0] = 1.0
    x[40:48] = np.float32(0.5)
    return x


@pytest.mark.parametrize("name", list(CONFIGS))
def test_level_table_matches_oracle(cuda, oracle, name):
    import tinycudann as tcnn
    kw = CONFIGS[name]
    cfg = oracle.GridConfig(n_features_per_level=2, **kw)
    total, offs, res, scl = tcnn.grid_levels(cfg.n_levels, cfg.base_resolution, cfg.per_level_scale,
                                             cfg.log2_hashmap_size)
    assert total == cfg.n_entries
    assert np.array_equal(offs, cfg.offsets) and np.array_equal(res, cfg.resolutions)
    assert np.array_equal(scl, cfg.scales)  # bit-exact scales => bit-exact lattice cells


@pytest.mark.parametrize("name", list(CONFIGS))
@pytest.mark.parametrize("n", [1, 63, 64, 1000, 4099])
def test_forward(cuda, oracle, name, n):
    import tinycudann as tcnn
    kw = CONFIGS[name]
    cfg = oracle.GridConfig(n_features_per_level=2, **kw)
    enc = tcnn.Encoding(3, {"otype": "HashGrid", "n_levels": cfg.n_levels, "n_features_per_level": 2,
                            "log2_hashmap_size": cfg.log2_hashmap_size, "base_resolution": cfg.base_resolution,
                            "per_level_scale": cfg.per_level_scale}, dtype=torch.float32).to(cuda)
    assert enc.params.numel() == cfg.n_params and list(dict(enc.named_parameters())) == ["params"]
    rng = np.random.default_rng(7)
    params = rng.uniform(-1, 1, cfg.n_params).astype(np.float32)
    with torch.no_grad():
        enc.params.copy_(T(params, cuda))
    x = _points(rng, max(n, 48))[:n] if n >= 48 else rng.random((n, 3)).astype(np.float32)
    ref = oracle.hashgrid_forward(x, params, cfg)
    out = enc(T(x, cuda)).detach().cpu().numpy()
    assert out.shape == (n, cfg.n_levels * 2)
    # same indices and weights, 8-term fp32 dot product in a different association: a few ulp
    np.testing.assert_allclose(out, ref, rtol=1e-5, atol=1e-6)


def test_forward_default_init_and_autocast(cuda, oracle):
    import tinycudann as tcnn
    enc = tcnn.Encoding(3, {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 19,
                            "base_resolution": 16, "per_level_scale": 1.3819128},
                        dtype=torch.float32).to(cuda)
    p = enc.params.detach()
    assert p.numel() == 12196240 and float(p.abs().max()) <= 1e-4 and float(p.abs().max()) > 9e-5
    x = torch.rand(257, 3, device=cuda)
    with torch.autocast("cuda", dtype=torch.float16):
        y = enc(x.half())  # custom_fwd casts back to fp32
    assert y.dtype == torch.float32


@pytest.mark.parametrize("name", ["default16", "small_hash"])
def test_backward(cuda, oracle, name):
    import tinycudann as tcnn
    kw = CONFIGS[name]
    cfg = oracle.GridConfig(n_features_per_level=2, **kw)
    enc = tcnn.Encoding(3, {"otype": "HashGrid", "n_levels": cfg.n_levels, "n_features_per_level": 2,
                            "log2_hashmap_size": cfg.log2_hashmap_size, "base_resolution": cfg.base_resolution,
                            "per_level_scale": cfg.per_level_scale}).to(cuda)
    rng = np.random.default_rng(8)
    n = 3000
    x = _points(rng, n)
    dout = rng.normal(size=(n, cfg.n_levels * 2)).astype(np.float32)
    dout[100:164] = 0  # fully masked rows are skipped by the kernel
    ref = oracle.hashgrid_backward(x, dout, cfg)
    y = enc(T(x, cuda))
    y.backward(T(dout, cuda))
    g = enc.params.grad.cpu().numpy()
    # float atomics: order differs from the oracle's sequential sum
    scale = np.abs(ref).max()
    assert np.abs(g - ref).max() <= 1e-5 * scale + 1e-6
    assert np.array_equal(g != 0, ref != 0)
    # linearity / accumulation: a second backward doubles the gradient (grad_params is accumulated by autograd)
    y2 = enc(T(x, cuda))
    y2.backward(T(dout, cuda))
    g2 = enc.params.grad.cpu().numpy()
    assert np.abs(g2 - 2 * ref).max() <= 2e-5 * scale + 2e-6


def test_gradient_is_adjoint_of_forward(cuda, oracle):
    """Size-independent property at full scale: <E(p) , d> == <p , E^T d> since the encoding is linear in params."""
    import tinycudann as tcnn
    enc = tcnn.Encoding(3, {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 19,
                            "base_resolution": 16, "per_level_scale": 1.3819128}).to(cuda)
    g = torch.Generator(device="cuda").manual_seed(3)
    with torch.no_grad():
        enc.params.copy_(torch.randn(enc.params.shape, device=cuda, generator=g))
    n = 1 << 18
    x = torch.rand(n, 3, device=cuda, generator=g)
    d = torch.randn(n, 32, device=cuda, generator=g)
    y = enc(x)
    lhs = (y.double() * d.double()).sum()
    y.backward(d)
    rhs = (enc.params.detach().double() * enc.params.grad.double()).sum()
    assert abs(float(lhs - rhs)) <= 1e-4 * abs(float(lhs)) + 1e-2
