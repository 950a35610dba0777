"""GPU parity: the stencil-aware encode / scatter (mi3d.grid_ops -> mi3d_grid_encode_points / _scatter_points)
against the oracle evaluated point by point, the way the reference does it (13 separate encoder passes)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _ray_like_points(rng, n, bound, dt=0.0034):
    """Samples laid out like a marcher emits them: runs of consecutive points along straight segments."""
    out = np.zeros((n, 3), np.float32)
    i = 0
    while i < n:
        k = min(int(rng.integers(5, 200)), n - i)
        o = rng.uniform(-0.8, 0.8, 3) * bound
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        t = np.arange(k)[:, None] * dt
        out[i:i + k] = np.clip(o + t * d, -bound, bound)
        i += k
    return out.astype(np.float32)


def _points(x, x2, offs, P0, bound):
    pts = []
    for p in range(offs.shape[0]):
        base = x if p < P0 else x2
        pts.append(np.clip(base + offs[p], -bound, bound).astype(np.float32))
    return pts


@pytest.mark.parametrize("bound", [1.0, 2.0])
@pytest.mark.parametrize("second", [False, True])
def test_encode_points_matches_pointwise_oracle(cuda, oracle, bound, second):
    from mi3d import grid_ops
    rng = np.random.default_rng(3)
    cfg = oracle.GridConfig(bound=bound)
    kcfg = dict(n_levels=cfg.n_levels, base_resolution=cfg.base_resolution, per_level_scale=cfg.per_level_scale,
                log2_hashmap_size=cfg.log2_hashmap_size)
    n = 1000
    x = _ray_like_points(rng, n, bound)
    x[:8] = bound          # stencil clamps at the box
    x[8:16, 1] = -bound
    x2 = (x + rng.normal(size=x.shape).astype(np.float32) * np.float32(0.01)).astype(np.float32)
    offs, P0 = grid_ops.stencil_offsets(center=True, second=second)
    params = rng.uniform(-1, 1, cfg.n_params).astype(np.float32)
    out = grid_ops.encode_points(T(params, cuda), T(x, cuda), offs, kcfg, bound, T(x2, cuda) if second else None,
                                 P0).cpu().numpy().reshape(n, offs.shape[0], -1)
    for p, pts in enumerate(_points(x, x2, offs, P0, bound)):
        h01 = ((pts + np.float32(bound)) / np.float32(2 * bound)).astype(np.float32)
        ref = oracle.hashgrid_forward(h01, params, cfg)
        np.testing.assert_allclose(out[:, p], ref, rtol=1e-5, atol=1e-6, err_msg=f"point {p}")


@pytest.mark.parametrize("force_scatter", [None, 0, 16])
@pytest.mark.parametrize("second", [False, True])
def test_scatter_points_matches_pointwise_oracle(cuda, oracle, second, force_scatter, monkeypatch):
    """Backward of the stencil encode == sum of the oracle's per-point backward passes, for every merge regime."""
    import subprocess, sys, os, json
    if force_scatter is not None:
        # the A/B switch is read once per process: run this case in a child process
        env = dict(os.environ, MI3D_SCATTER=str(force_scatter), MI3D_CHILD="1")
        code = subprocess.run([sys.executable, "-m", "pytest", __file__, "-q", "-x", "-m", "gpu", "-k",
                               f"test_scatter_points_matches_pointwise_oracle and None and {second}"], env=env,
                              capture_output=True, text=True)
        assert code.returncode == 0, code.stdout[-2000:]
        return
    from mi3d import grid_ops
    rng = np.random.default_rng(4)
    bound = 1.0
    cfg = oracle.GridConfig(bound=bound)
    kcfg = dict(n_levels=cfg.n_levels, base_resolution=cfg.base_resolution, per_level_scale=cfg.per_level_scale,
                log2_hashmap_size=cfg.log2_hashmap_size)
    n = 1500
    x = _ray_like_points(rng, n, bound)
    x[:8] = bound
    x2 = (x + rng.normal(size=x.shape).astype(np.float32) * np.float32(0.01)).astype(np.float32)
    offs, P0 = grid_ops.stencil_offsets(center=True, second=second)
    P = offs.shape[0]
    dout = rng.normal(size=(n, P, 32)).astype(np.float32)
    dout[100:140] = 0
    dout[:, 2, 4:6] = 0
    params = torch.zeros(cfg.n_params, device=cuda, requires_grad=True)
    feats = grid_ops.encode_points(params, T(x, cuda), offs, kcfg, bound, T(x2, cuda) if second else None, P0,
                                   step=0.0034)
    feats.backward(T(dout.reshape(n * P, 32), cuda))
    g = params.grad.cpu().numpy()
    ref = np.zeros(cfg.n_params, np.float64)
    for p, pts in enumerate(_points(x, x2, offs, P0, bound)):
        h01 = ((pts + np.float32(bound)) / np.float32(2 * bound)).astype(np.float32)
        ref += oracle.hashgrid_backward(h01, dout[:, p], cfg)
    scale = np.abs(ref).max()
    assert np.abs(g - ref).max() <= 2e-5 * scale + 1e-6
    assert np.mean((g != 0) != (ref != 0)) < 1e-6


def test_count_caps_rows_without_host_sync(cuda, oracle):
    from mi3d import grid_ops
    cfg = oracle.GridConfig()
    kcfg = dict(n_levels=16, base_resolution=16, per_level_scale=cfg.per_level_scale, log2_hashmap_size=19)
    rng = np.random.default_rng(6)
    n, keep = 500, 321
    x = _ray_like_points(rng, n, 1.0)
    offs, P0 = grid_ops.stencil_offsets(center=True)
    params = torch.rand(cfg.n_params, device=cuda, requires_grad=True)
    count = torch.tensor([keep], dtype=torch.int32, device=cuda)
    full = grid_ops.encode_points(params.detach(), T(x, cuda), offs, kcfg)
    part = grid_ops.encode_points(params, T(x, cuda), offs, kcfg, count=count)
    P = offs.shape[0]
    assert torch.equal(part[:keep * P], full[:keep * P]) and float(part[keep * P:].abs().max()) == 0
    part.sum().backward()
    g_part = params.grad.clone()
    params.grad = None
    p2 = grid_ops.encode_points(params, T(x[:keep], cuda), offs, kcfg)
    p2.sum().backward()
    assert torch.allclose(g_part, params.grad, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("workspace", ["auto", "tiny", "none"])
@pytest.mark.parametrize("second", [False, True])
def test_binned_scatter_matches_pointwise_oracle(cuda, oracle, second, workspace):
    """mi3d_grid_scatter_binned (records + LDS accumulation on the hashed levels, merged atomics on the dense ones) ==
    sum of the oracle's per-point backward passes; 'tiny' forces several slices, 'none' the all-atomic fallback."""
    from mi3d import field_ops, grid_ops
    rng = np.random.default_rng(14)
    bound = 1.0
    cfg = oracle.GridConfig(bound=bound)
    kcfg = dict(n_levels=cfg.n_levels, base_resolution=cfg.base_resolution, per_level_scale=cfg.per_level_scale,
                log2_hashmap_size=cfg.log2_hashmap_size)
    n = 3000
    x = _ray_like_points(rng, n, bound)
    x[:8] = bound
    x2 = (x + rng.normal(size=x.shape).astype(np.float32) * np.float32(0.01)).astype(np.float32)
    offs, P0 = grid_ops.stencil_offsets(center=True, second=second)
    P = offs.shape[0]
    dout = rng.normal(size=(n, P, 16, 2)).astype(np.float32)
    dout[100:140] = 0
    dout[:, 2, 2] = 0
    planes = np.ascontiguousarray(dout.reshape(n * P, 16, 2).transpose(1, 0, 2))  # [L][rows][2]
    full = None
    ws = {"auto": None, "none": 0}.get(workspace, "tiny")
    if ws == "tiny":  # room for about a third of the samples per slice
        import ctypes
        from mi3d import _lib
        need = _lib.lib().mi3d_grid_scatter_binned_workspace(n // 3, P, 1.0, 0.0034, 16, 16, cfg.per_level_scale, 19, 0)
        ws = int(need)
    g = field_ops.scatter_binned(T(x, cuda), T(x2, cuda) if second else None, offs, P0, bound, T(planes, cuda), kcfg,
                                 0.0034, cfg.n_params, workspace_bytes=ws).cpu().numpy()
    ref = np.zeros(cfg.n_params, np.float64)
    for p, pts in enumerate(_points(x, x2, offs, P0, bound)):
        h01 = ((pts + np.float32(bound)) / np.float32(2 * bound)).astype(np.float32)
        ref += oracle.hashgrid_backward(h01, dout[:, p].reshape(n, 32), cfg)
    scale = np.abs(ref).max()
    assert np.abs(g - ref).max() <= 2e-5 * scale + 1e-6
    assert np.mean((g != 0) != (ref != 0)) < 1e-6


def test_field_stencil_node_equals_layer_composition(cuda, oracle):
    """The fused autograd node (encode + MLP, binned scatter) against the per-layer composition on the same inputs."""
    from mi3d import field_ops, grid_ops, mlp_ops
    from mi3d.network import MLP
    rng = np.random.default_rng(15)
    cfg = oracle.GridConfig()
    kcfg = dict(n_levels=16, base_resolution=16, per_level_scale=cfg.per_level_scale, log2_hashmap_size=19)
    n = 2000
    x = T(_ray_like_points(rng, n, 1.0), cuda)
    x2 = x + torch.randn_like(x) * 0.01
    offs, P0 = grid_ops.stencil_offsets(center=True, second=True)
    torch.manual_seed(0)
    mlp = MLP(32, 4, 64, 3).to(cuda)
    table = torch.empty(cfg.n_params, device=cuda).uniform_(-0.5, 0.5)
    g = torch.randn(n * offs.shape[0], 4, device=cuda)
    outs = []
    for fused in (True, False):
        params = table.clone().requires_grad_(True)
        mlp.zero_grad()
        if fused:
            h = field_ops.field_stencil(params, mlp.net, x, offs, kcfg, 1.0, x2, P0, step=0.0034, half_mode=False)
        else:
            h = mlp_ops.fused_mlp(grid_ops.encode_points(params, x, offs, kcfg, 1.0, x2, P0, step=0.0034), mlp.net,
                                  half_mode=False)
        h.backward(g)
        outs.append([h.detach().clone(), params.grad.clone()] + [p.grad.clone() for p in mlp.parameters()])
    for a, b in zip(*outs):
        scale = float(b.abs().max()) + 1e-12
        assert float((a - b).abs().max()) <= 2e-5 * scale


@pytest.mark.parametrize("workspace", ["auto", "tiny"])
def test_binned_scatter_half_records_match_oracle(cuda, oracle, workspace):
    """The 8-byte-record path (fine levels as {entry, binary16 pair} scaled per level, line-staged emit) - what the
    fused field node uses under torch.autocast.  Tolerance: binary16 resolution of each contribution."""
    from mi3d import field_ops, grid_ops
    rng = np.random.default_rng(24)
    cfg = oracle.GridConfig()
    kcfg = dict(n_levels=16, base_resolution=16, per_level_scale=cfg.per_level_scale, log2_hashmap_size=19)
    n = 6000
    x = _ray_like_points(rng, n, 1.0)
    x2 = (x + rng.normal(size=x.shape).astype(np.float32) * np.float32(0.01)).astype(np.float32)
    offs, P0 = grid_ops.stencil_offsets(center=True, second=True)
    P = offs.shape[0]
    # per-level magnitudes spread over 12 decades, plus rows of zeros
    mag = (10.0 ** rng.uniform(-8, 4, 16)).astype(np.float32)
    dout = (rng.normal(size=(n, P, 16, 2)).astype(np.float32) * mag[None, None, :, None]).astype(np.float32)
    dout[100:140] = 0
    planes = np.ascontiguousarray(dout.reshape(n * P, 16, 2).transpose(1, 0, 2))
    absmax = T(np.abs(planes).reshape(16, -1).max(1).astype(np.float32), cuda)
    ws = None
    if workspace == "tiny":
        from mi3d import _lib
        ws = int(_lib.lib().mi3d_grid_scatter_binned_workspace(n // 3, P, 1.0, 0.0034, 16, 16, cfg.per_level_scale, 19, 1))
    g = field_ops.scatter_binned(T(x, cuda), T(x2, cuda), offs, P0, 1.0, T(planes, cuda), kcfg, 0.0034, cfg.n_params,
                                 workspace_bytes=ws, level_absmax=absmax).cpu().numpy()
    ref = np.zeros(cfg.n_params, np.float64)
    for p, pts in enumerate(_points(x, x2, offs, P0, 1.0)):
        h01 = ((pts + np.float32(1.0)) / np.float32(2.0)).astype(np.float32)
        ref += oracle.hashgrid_backward(h01, dout[:, p].reshape(n, 32), cfg)
    for l in range(16):
        a, b = int(cfg.offsets[l]) * 2, int(cfg.offsets[l + 1]) * 2
        scale = np.abs(ref[a:b]).max()
        tol = 2e-5 if l < 8 else 2e-3   # levels with cells shorter than 3 steps (>= 8 here) carry binary16 records
        assert np.abs(g[a:b] - ref[a:b]).max() <= tol * scale + 1e-30, (l, np.abs(g[a:b] - ref[a:b]).max(), scale)


def test_binned_scatter_half_records_axis_parallel_rays(cuda, oracle):
    """Samples marching along +x keep (y, z) lattice coordinates - and with them the hash bin - for whole waves: all 64
    lanes of an append hit ONE bin, far more than a staging ring holds.  (This input hung the first staged emit.)"""
    from mi3d import field_ops, grid_ops
    rng = np.random.default_rng(31)
    cfg = oracle.GridConfig()
    kcfg = dict(n_levels=16, base_resolution=16, per_level_scale=cfg.per_level_scale, log2_hashmap_size=19)
    n = 4096
    t = (np.arange(n) % 512)[:, None] * np.float32(0.0034)
    x = np.concatenate([-0.85 + t, np.full((n, 1), 0.3137, np.float32) + (np.arange(n) // 512)[:, None] * 0.05,
                        np.full((n, 1), -0.2241, np.float32)], 1).astype(np.float32)
    offs, P0 = grid_ops.stencil_offsets(center=True, second=False)
    P = offs.shape[0]
    dout = rng.normal(size=(n, P, 16, 2)).astype(np.float32)
    planes = np.ascontiguousarray(dout.reshape(n * P, 16, 2).transpose(1, 0, 2))
    absmax = T(np.abs(planes).reshape(16, -1).max(1).astype(np.float32), cuda)
    g = field_ops.scatter_binned(T(x, cuda), None, offs, P0, 1.0, T(planes, cuda), kcfg, 0.0034, cfg.n_params,
                                 level_absmax=absmax).cpu().numpy()
    ref = np.zeros(cfg.n_params, np.float64)
    for p, pts in enumerate(_points(x, x, offs, P0, 1.0)):
        ref += oracle.hashgrid_backward(((pts + np.float32(1.0)) / np.float32(2.0)).astype(np.float32),
                                        dout[:, p].reshape(n, 32), cfg)
    for l in range(16):
        a, b = int(cfg.offsets[l]) * 2, int(cfg.offsets[l + 1]) * 2
        scale = np.abs(ref[a:b]).max()
        assert np.abs(g[a:b] - ref[a:b]).max() <= (2e-5 if l < 8 else 2e-3) * scale + 1e-30, l
