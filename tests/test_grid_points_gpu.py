"""GPU parity: the stencil-aware encode / scatter (mi3d.grid_ops -> mi3d_grid_encode_points / _scatter_points)
against the oracle evaluated point by point, the way the reference does it (13 separate encoder passes)."""
import numpy as np
import pytest
import torch

from conftest import record_scatter_workspaces

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
                                 P0).cpu().numpy().reshape(offs.shape[0], n, -1)   # rows are point-major: p*n + s
    for p, pts in enumerate(_points(x, x2, offs, P0, bound)):
        h01 = ((pts + np.float32(bound)) / np.float32(2 * bound)).astype(np.float32)
        ref = oracle.hashgrid_forward(h01, params, cfg)
        np.testing.assert_allclose(out[p], ref, rtol=1e-5, atol=1e-6, err_msg=f"point {p}")


@pytest.mark.parametrize("step", [0.0034, 100.0, 1e-7])
@pytest.mark.parametrize("second", [False, True])
def test_scatter_points_matches_pointwise_oracle(cuda, oracle, second, step):
    """Backward of the stencil encode == sum of the oracle's per-point backward passes, for every merge regime: `step`
    only steers which levels sum equal-cell runs before their atomics (0.0034: levels 0-7, 100: none, 1e-7: all 16),
    never the result."""
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
                                   step=step)
    feats.backward(T(dout.transpose(1, 0, 2).reshape(P * n, 32), cuda))   # point-major rows
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
    part_v, full_v = part.view(P, n, -1), full.view(P, n, -1)   # the row stride between points stays n
    assert torch.equal(part_v[:, :keep], full_v[:, :keep]) and float(part_v[:, keep:].abs().max()) == 0
    part.sum().backward()
    g_part = params.grad.clone()
    params.grad = None
    p2 = grid_ops.encode_points(params, T(x[:keep], cuda), offs, kcfg)
    assert torch.equal(p2.view(P, keep, -1), full_v[:, :keep])
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
    planes = np.ascontiguousarray(dout.transpose(2, 1, 0, 3).reshape(16, P * n, 2))  # [L][p*n + s][2]
    full = None
    ws = {"auto": None, "none": 0}.get(workspace, "tiny")
    import ctypes
    from mi3d import _lib
    if ws == "tiny":  # room for about a third of the samples per slice
        need = _lib.lib().mi3d_grid_scatter_binned_workspace(n // 3, P, 1.0, 0.0034, 16, 16, cfg.per_level_scale, 19)
        ws = int(need)
    with record_scatter_workspaces() as arenas:
        g = field_ops.scatter_binned(T(x, cuda), T(x2, cuda) if second else None, offs, P0, bound, T(planes, cuda), kcfg,
                                     0.0034, cfg.n_params, workspace_bytes=ws).cpu().numpy()
    if workspace == "none":
        assert arenas == []                                   # the all-atomic path was asked for
    else:  # the record path really ran; 'tiny' really cut the samples into several slices
        assert len(arenas) == 1 and arenas[0] > 0, arenas
        plan = (ctypes.c_ulonglong * (6 + 7 * 16))()
        _lib.call("mi3d_grid_scatter_plan", n, P, 1.0, 0.0034, 16, 16, float(cfg.per_level_scale), 19,
                  ctypes.c_size_t(arenas[0]), plan)
        assert (plan[0] < n) == (workspace == "tiny"), (plan[0], n)
    ref = np.zeros(cfg.n_params, np.float64)
    for p, pts in enumerate(_points(x, x2, offs, P0, bound)):
        h01 = ((pts + np.float32(bound)) / np.float32(2 * bound)).astype(np.float32)
        ref += oracle.hashgrid_backward(h01, dout[:, p].reshape(n, 32), cfg)
    scale = np.abs(ref).max()
    assert np.abs(g - ref).max() <= 2e-5 * scale + 1e-6
    assert np.mean((g != 0) != (ref != 0)) < 1e-6


@pytest.mark.parametrize("workspace", ["auto", "tiny", "none"])
@pytest.mark.parametrize("half", [False, True])
def test_binned_scatter_with_a_second_pair_for_point_0(cuda, oracle, half, workspace):
    """mi3d_grid_scatter_binned_plus: the 13-point scatter that also takes the point-0 planes of an earlier backward pass
    (`extra0`) == the oracle's per-point backward passes + one more pass over point 0 with the extra pair.  fp32 planes
    (16-byte pair records, the two point-0 pairs summed in fp32) and binary16 planes (12-byte records: the pair as raw
    bits, weight and x fraction as 23-bit fixed point, the extra pair as a record of its own); one slice, several slices,
    and the all-atomic fallback (which scatters the extra pair in a pass of its own)."""
    import ctypes
    from mi3d import _lib, field_ops, grid_ops
    rng = np.random.default_rng(19)
    cfg = oracle.GridConfig()
    kcfg = dict(n_levels=cfg.n_levels, base_resolution=cfg.base_resolution, per_level_scale=cfg.per_level_scale,
                log2_hashmap_size=cfg.log2_hashmap_size)
    n = 3000
    x = _ray_like_points(rng, n, 1.0)
    x[:8] = 1.0
    x2 = (x + rng.normal(size=x.shape).astype(np.float32) * np.float32(0.01)).astype(np.float32)
    offs, P0 = grid_ops.stencil_offsets(center=True, second=True)
    P = offs.shape[0]
    dt = np.float16 if half else np.float32
    dout = rng.normal(size=(n, P, 16, 2)).astype(dt)
    dout[100:140] = 0
    dout[:, 2, 2] = 0
    extra = rng.normal(size=(n, 16, 2)).astype(dt)
    extra[50:120] = 0
    extra[200:260, 12] = -dout[200:260, 0, 12]          # the two point-0 pairs cancel exactly on one fine level
    planes = np.ascontiguousarray(dout.transpose(2, 1, 0, 3).reshape(16, P * n, 2))   # [L][p*n + s][2]
    eplanes = np.ascontiguousarray(extra.transpose(1, 0, 2))                          # [L][n][2]
    ws = {"auto": None, "none": 0}.get(workspace, "tiny")
    if ws == "tiny":
        ws = int(_lib.lib().mi3d_grid_scatter_binned_workspace(n // 3, P, 1.0, 0.0034, 16, 16, cfg.per_level_scale, 19))
    with record_scatter_workspaces() as arenas:
        g = field_ops.scatter_binned(T(x, cuda), T(x2, cuda), offs, P0, 1.0, T(planes, cuda), kcfg, 0.0034, cfg.n_params,
                                     workspace_bytes=ws, extra0=T(eplanes, cuda)).cpu().numpy()
    assert (arenas == []) if workspace == "none" else (len(arenas) == 1 and arenas[0] > 0)
    ref = np.zeros(cfg.n_params, np.float64)
    pts = _points(x, x2, offs, P0, 1.0)
    for p, q in enumerate(pts):
        h01 = ((q + np.float32(1.0)) / np.float32(2.0)).astype(np.float32)
        ref += oracle.hashgrid_backward(h01, dout[:, p].astype(np.float32).reshape(n, 32), cfg)
    ref += oracle.hashgrid_backward(((pts[0] + np.float32(1.0)) / np.float32(2.0)).astype(np.float32),
                                    extra.astype(np.float32).reshape(n, 32), cfg)
    scale = np.abs(ref).max()
    assert np.abs(g - ref).max() <= 2e-5 * scale + 1e-6
    with pytest.raises(_lib.Mi3dError):     # the extra planes must match the main ones in type
        field_ops.scatter_binned(T(x, cuda), T(x2, cuda), offs, P0, 1.0, T(planes, cuda), kcfg, 0.0034, cfg.n_params,
                                 extra0=T(eplanes.astype(np.float16 if not half else np.float32), cuda))


@pytest.mark.parametrize("half", [False, True])
def test_binned_scatter_when_the_regions_overflow(cuda, oracle, half):
    """A skewed problem - 30 000 samples inside one cell of the finest level, all 13 stencil offsets zero - sends every
    record of a level to the same four bins: the (wave, bin) regions, sized for the uniform share, overflow by an order of
    magnitude.  A full region is not an error: what does not fit goes to the table with float atomics (the sorted flush's
    per-chunk vote, the direct appends and the coarse role's records all have that path).  Same gradient as the oracle."""
    from mi3d import field_ops
    rng = np.random.default_rng(23)
    cfg = oracle.GridConfig()
    kcfg = dict(n_levels=cfg.n_levels, base_resolution=cfg.base_resolution, per_level_scale=cfg.per_level_scale,
                log2_hashmap_size=cfg.log2_hashmap_size)
    n, P = 30000, 13
    x = (np.array([0.31, -0.22, 0.47], np.float32) + rng.uniform(-1e-4, 1e-4, (n, 3)).astype(np.float32)).astype(np.float32)
    offs = np.zeros((P, 3), np.float32)
    dt = np.float16 if half else np.float32
    dout = rng.normal(size=(n, P, 16, 2)).astype(dt)
    planes = np.ascontiguousarray(dout.transpose(2, 1, 0, 3).reshape(16, P * n, 2))
    with record_scatter_workspaces() as arenas:
        g = field_ops.scatter_binned(T(x, cuda), None, offs, P, 1.0, T(planes, cuda), kcfg, 0.0034, cfg.n_params).cpu().numpy()
    assert len(arenas) == 1 and arenas[0] > 0
    h01 = ((x + np.float32(1.0)) / np.float32(2.0)).astype(np.float32)
    ref = oracle.hashgrid_backward(h01, dout.astype(np.float32).sum(1).reshape(n, 32), cfg)   # 13 identical points
    scale = np.abs(ref).max()
    # (float atomics in arrival order: 390 000 contributions of O(1) per entry - the reference's own summation noise)
    assert np.abs(g - ref).max() <= 2e-4 * scale
    assert np.array_equal(g != 0, ref != 0)


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
    for i, (a, b) in enumerate(zip(*outs)):
        scale = float(b.abs().max()) + 1e-12
        # MLP parameter gradients (i >= 5) are float-atomic sums over 39 000 rows of O(1) terms: run-to-run noise ~1e-4
        assert float((a - b).abs().max()) <= (2e-4 if i >= 5 else 2e-5) * scale, i


@pytest.mark.parametrize("bad", ["inf", "nan"])
def test_binned_scatter_propagates_non_finite_gradients(cuda, oracle, bad):
    """A non-finite feature gradient must reach encoder.params.grad (GradScaler's overflow check reads it there) on
    EXACTLY the entries the reference's float atomics would put it on - the corners of that (sample, point, level) -
    and nowhere else: the record path hands such contributions to float atomics instead of its fixed-point sums."""
    from mi3d import field_ops, grid_ops
    rng = np.random.default_rng(41)
    cfg = oracle.GridConfig()
    kcfg = dict(n_levels=16, base_resolution=16, per_level_scale=cfg.per_level_scale, log2_hashmap_size=19)
    n = 2000
    x = _ray_like_points(rng, n, 1.0)
    offs, P0 = grid_ops.stencil_offsets(center=True, second=False)
    P = offs.shape[0]
    dout = rng.normal(size=(n, P, 16, 2)).astype(np.float32)
    for lvl in (3, 12):   # one run-merged coarse level, one fine level
        dout[777, 2, lvl, 1] = np.inf if bad == "inf" else np.nan
    planes = T(np.ascontiguousarray(dout.transpose(2, 1, 0, 3).reshape(16, P * n, 2)), cuda)
    with record_scatter_workspaces() as arenas:
        g = field_ops.scatter_binned(T(x, cuda), None, offs, P0, 1.0, planes, kcfg, 0.0034, cfg.n_params).cpu().numpy()
    assert len(arenas) == 1 and arenas[0] > 0, arenas   # through k_bin_emit's non-finite handling, not the atomic fallback
    ref = np.zeros(cfg.n_params, np.float64)
    with np.errstate(invalid="ignore"):
        for p, pts in enumerate(_points(x, x, offs, P0, 1.0)):
            ref += oracle.hashgrid_backward(((pts + np.float32(1.0)) / np.float32(2.0)).astype(np.float32),
                                            dout[:, p].reshape(n, 32), cfg)
    bad_ref = ~np.isfinite(ref)
    assert 2 <= bad_ref.sum() <= 32          # the 8 corners of two (sample, point, level) evaluations, one feature each
    assert np.array_equal(~np.isfinite(g), bad_ref)
    ok = ~bad_ref
    for l in range(16):
        a, b = int(cfg.offsets[l]) * 2, int(cfg.offsets[l + 1]) * 2
        m = ok[a:b]
        assert np.abs(g[a:b][m] - ref[a:b][m]).max() <= 2e-5 * np.abs(ref[a:b][m]).max(), l


@pytest.mark.parametrize("reach", [1, 7, 13])
def test_fused_field_node_runs_only_the_reached_stencil_prefix(cuda, oracle, reach):
    """field_ops.field (encode + MLP + head, one node) against field_stencil + field_head (two nodes, dense dh) when
    the upstream gradient reaches only sigma / albedo (point 0), the normal as well (points 0-6) or everything: same
    outputs, same parameter gradients - the fused node runs MLP backward and scatter over the reached prefix only."""
    from mi3d import field_ops, grid_ops
    from mi3d.network import MLP
    rng = np.random.default_rng(51)
    cfg = oracle.GridConfig()
    kcfg = dict(n_levels=16, base_resolution=16, per_level_scale=cfg.per_level_scale, log2_hashmap_size=19)
    n = 3000
    x = T(_ray_like_points(rng, n, 1.0), cuda)
    x2 = x + torch.randn_like(x) * 0.01
    offs, P0 = grid_ops.stencil_offsets(center=True, second=True)
    torch.manual_seed(1)
    mlp = MLP(32, 4, 64, 3).to(cuda)
    table = torch.empty(cfg.n_params, device=cuda).uniform_(-0.5, 0.5)
    gs = [torch.randn(n, device=cuda), torch.randn(n, 3, device=cuda), torch.randn(n, 3, device=cuda),
          torch.randn(n, 3, device=cuda)]
    outs = []
    for fused in (True, False):
        params = table.clone().requires_grad_(True)
        mlp.zero_grad()
        if fused:
            o = field_ops.field(params, mlp.net, x, offs, kcfg, 1.0, 5.0, 0.1, x2, P0, step=0.0034, half_mode=False)
        else:
            h = field_ops.field_stencil(params, mlp.net, x, offs, kcfg, 1.0, x2, P0, step=0.0034, half_mode=False)
            o = field_ops.field_head(h, x, offs, 1.0, 5.0, 0.1, x2)
        k = {1: 2, 7: 3, 13: 4}[reach]
        torch.autograd.backward(list(o[:k]), gs[:k])
        outs.append([t.detach().clone() for t in o] + [params.grad.clone()] + [p.grad.clone() for p in mlp.parameters()])
    for i, (a, b) in enumerate(zip(*outs)):
        scale = float(b.abs().max()) + 1e-12
        # MLP parameter gradients (i >= 5) are float-atomic sums over 39 000 rows of O(1) terms: run-to-run noise ~1e-4
        assert float((a - b).abs().max()) <= (2e-4 if i >= 5 else 2e-5) * scale, i


@pytest.mark.parametrize("bound", [1.0, 2.0, 1.5])
@pytest.mark.parametrize("levels,base,log2", [(16, 16, 19), (4, 16, 19), (6, 8, 12)])
def test_plane_kernels_are_bit_identical_to_the_row_kernel(cuda, oracle, bound, levels, base, log2):
    """The plane gather takes short index routes (dense levels: strided index, one 16-byte load per x-pair, served from LDS
    where the tables fit; power-of-two hashed levels: mask, one predicate for the four pairs; power-of-two 2 * bound: a
    multiplication for the division) - the row kernel takes grid_entry()'s general one.  Same cells, same entries, same
    order of operations: equal to the bit, also on the box faces and in its far corner, where the +1 corners of a dense
    level wrap (bound 1.5: the division path)."""
    from mi3d import _lib as L, grid_ops
    rng = np.random.default_rng(17)
    cfg = oracle.GridConfig(bound=bound, n_levels=levels, base_resolution=base, log2_hashmap_size=log2)
    kcfg = dict(n_levels=levels, base_resolution=base, per_level_scale=cfg.per_level_scale, log2_hashmap_size=log2)
    n = 4096 + 37
    x = _ray_like_points(rng, n, bound)
    x[:64] = bound                                                     # the far corner: every +1 corner wraps
    x[64:128] = np.where(rng.random((64, 3)) < 0.5, bound, x[64:128])   # faces and edges
    x[128:160] = -bound
    x[160:224] = bound * (1 - rng.random((64, 3)).astype(np.float32) * np.float32(0.02))   # the last cells
    x2 = (x + rng.normal(size=x.shape).astype(np.float32) * np.float32(0.01)).astype(np.float32)
    offs, P0 = grid_ops.stencil_offsets(center=True, second=True)
    P = offs.shape[0]
    params = T(rng.uniform(-1, 1, cfg.n_params).astype(np.float32), cuda)
    xd, x2d = T(x, cuda), T(x2, cuda)
    rows = grid_ops.encode_points(params, xd, offs, kcfg, bound, x2d, P0)          # [P n, 2 L], general index route
    _, offs_p = grid_ops._offs_arg(offs)
    for half in (0, 1):
        planes = torch.full((levels, P * n, 2), 7.0, device=cuda, dtype=torch.float16 if half else torch.float32)
        with L.on(xd):
            L.call("mi3d_grid_encode_points_planes", L.ptr(xd), L.ptr(x2d), n, offs_p, int(P0), P, float(bound),
                   L.ptr(params), levels, base, cfg.per_level_scale, log2, 2 * 3 ** 0.5 / 1024 * bound, L.ptr(planes),
                   half, L.stream(xd))
        got = planes.permute(1, 0, 2).reshape(P * n, 2 * levels)
        want = rows.half() if half else rows
        assert torch.equal(got, want), (half, float((got.float() - want.float()).abs().max()))


@pytest.mark.parametrize("log2", [10, 12])
@pytest.mark.parametrize("half", [False, True])
def test_binned_scatter_with_hash_tables_smaller_than_a_bin(cuda, oracle, half, log2):
    """log2_hashmap_size below 13: a hashed level is smaller than one reduce bin (8192 entries), so the x + 1 partner of a
    pair record - e1 = e0 ^ (2^t - 1), t = 1 + trailing ones of cx - must be masked with the LEVEL's size, not the bin's:
    with res 2048 and cx ending in ten or more ones the unmasked flip lands past the table (the 12-byte
    branch of k_bin_reduce masked with the bin only).  Both record types against the oracle."""
    from mi3d import field_ops, grid_ops
    rng = np.random.default_rng(29)
    cfg = oracle.GridConfig(log2_hashmap_size=log2)
    kcfg = dict(n_levels=cfg.n_levels, base_resolution=cfg.base_resolution, per_level_scale=cfg.per_level_scale,
                log2_hashmap_size=log2)
    n = 4000
    x = _ray_like_points(rng, n, 1.0)
    # cells whose x index ends in many ones on the finest levels (res 2048: cx = 1023, 511, 2047 - the last clamps)
    for i, cx in enumerate((1023, 511, 1535, 255, 767)):
        x[200 * i:200 * i + 100, 0] = np.float32((cx + 0.3) / 2048.0 * 2.0 - 1.0)
    offs, P0 = grid_ops.stencil_offsets(center=True, second=False)
    P = offs.shape[0]
    dt = np.float16 if half else np.float32
    dout = rng.normal(size=(n, P, 16, 2)).astype(dt)
    planes = np.ascontiguousarray(dout.transpose(2, 1, 0, 3).reshape(16, P * n, 2))
    with record_scatter_workspaces() as arenas:
        g = field_ops.scatter_binned(T(x, cuda), None, offs, P0, 1.0, T(planes, cuda), kcfg, 0.0034, cfg.n_params).cpu().numpy()
    assert len(arenas) == 1 and arenas[0] > 0
    ref = np.zeros(cfg.n_params, np.float64)
    for p, pts in enumerate(_points(x, x, offs, P0, 1.0)):
        h01 = ((pts + np.float32(1.0)) / np.float32(2.0)).astype(np.float32)
        ref += oracle.hashgrid_backward(h01, dout[:, p].astype(np.float32).reshape(n, 32), cfg)
    for l in range(16):   # per level: a misplaced partner shows on its own level, whatever the others' magnitudes are
        a, b = int(cfg.offsets[l]) * 2, int(cfg.offsets[l + 1]) * 2
        assert np.abs(g[a:b] - ref[a:b]).max() <= 5e-5 * np.abs(ref[a:b]).max() + 1e-6, l


def test_large_scatters_share_one_persistent_placed_arena(cuda, oracle, monkeypatch):
    """mi3d.field_ops keeps ONE record arena per device for requests of PLACED_MIN_BYTES and more and chooses it among
    candidate blocks by timing the caller's own scatter on each (the emit's time depends on the arena's
    physical placement).  With the threshold lowered to this test's size: the calibration runs once (candidates timed,
    one kept, logged), later calls are served from the same block - also smaller ones, as a prefix - the gradient is what
    a per-call allocation gives, a larger request replaces the arena, release_scatter_arena() lets go of it."""
    from mi3d import field_ops, grid_ops
    rng = np.random.default_rng(23)
    cfg = oracle.GridConfig()
    kcfg = dict(n_levels=cfg.n_levels, base_resolution=cfg.base_resolution, per_level_scale=cfg.per_level_scale,
                log2_hashmap_size=cfg.log2_hashmap_size)
    n = 4000
    x = _ray_like_points(rng, n, 1.0)
    x2 = (x + rng.normal(size=x.shape).astype(np.float32) * np.float32(0.01)).astype(np.float32)
    offs, P0 = grid_ops.stencil_offsets(center=True, second=True)
    P = offs.shape[0]
    planes = rng.normal(size=(16, P * n, 2)).astype(np.float16)
    args = (T(x, cuda), T(x2, cuda), offs, P0, 1.0, T(planes, cuda), kcfg, 0.0034, cfg.n_params)
    field_ops.release_scatter_arena()
    want = field_ops.scatter_binned(*args).cpu().numpy()               # per-call allocation (below the threshold)
    assert not field_ops._ARENAS
    monkeypatch.setattr(field_ops, "PLACED_MIN_BYTES", 1 << 20)
    monkeypatch.setattr(field_ops, "PLACEMENT_TRIALS", 3)
    log0 = len(field_ops.PLACEMENT_LOG)
    try:
        got = field_ops.scatter_binned(*args).cpu().numpy()
        assert len(field_ops.PLACEMENT_LOG) == log0 + 1
        rec = field_ops.PLACEMENT_LOG[-1]
        assert len(rec["candidates_ms"]) == 3 and 0 <= rec["kept"] < 3 and all(t > 0 for t in rec["candidates_ms"])
        arena = field_ops._ARENAS[cuda.index]
        scale = np.abs(want).max()
        assert np.abs(got - want).max() <= 2e-6 * scale               # (the order of the table's float atomics)
        again = field_ops.scatter_binned(*args).cpu().numpy()
        assert field_ops._ARENAS[cuda.index] is arena and len(field_ops.PLACEMENT_LOG) == log0 + 1
        assert np.abs(again - want).max() <= 2e-6 * scale
        # a smaller request is a prefix of the same block; a larger one replaces it
        m = 2500
        small = (T(x[:m], cuda), T(x2[:m], cuda), offs, P0, 1.0,
                 T(np.ascontiguousarray(planes.reshape(16, P, n, 2)[:, :, :m].reshape(16, P * m, 2)), cuda), kcfg, 0.0034, cfg.n_params)
        field_ops.scatter_binned(*small)
        assert field_ops._ARENAS[cuda.index] is arena
        ws = field_ops.scatter_workspace(cuda, arena.numel() + (1 << 20))
        assert ws.numel() == arena.numel() + (1 << 20) and field_ops._ARENAS[cuda.index] is not arena
    finally:
        field_ops.release_scatter_arena()
    assert not field_ops._ARENAS
