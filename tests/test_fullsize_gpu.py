"""GPU, BASELINE config 2 at FULL size (128x128 rays, max_steps 1024, all-ones occupancy: ~10.9 M samples, 141 M field
evaluations) through size-independent properties: the march's slabs tile the sample buffer and per-ray counts agree
with the oracle on a ray sample; the gather and the binned scatter are adjoint, <encode(p), g> = <p, scatter(g)>, over
all 141 M evaluations - a checksum of the whole forward/backward pair that no small case exercises (two sample slices,
full region occupancy, every bin)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_c2_full_size_march_invariants_and_gather_scatter_adjoint(cuda, oracle):
    import raymarching
    from mi3d import _lib as L
    from mi3d import field_ops, grid_ops, rays as R
    H = W = 128
    ro, rd, _ = R.view_rays(H, W, device=cuda)
    ro, rd = ro.view(-1, 3).contiguous(), rd.view(-1, 3).contiguous()
    N = ro.shape[0]
    aabb = torch.tensor([-1, -1, -1, 1, 1, 1.0], device=cuda)
    nears, fars = raymarching.near_far_from_aabb(ro, rd, aabb)
    bits = torch.full((128 ** 3 // 8,), 255, dtype=torch.uint8, device=cuda)
    counter = torch.zeros(2, dtype=torch.int32, device=cuda)
    xyzs, dirs, deltas, rays = raymarching.march_rays_train(ro, rd, 1.0, bits, 1, 128, nears, fars, counter, -1, False, 128,
                                                            True, 0, 1024)
    m = int(counter[0])
    assert int(counter[1]) == N and 10_000_000 < m < 12_000_000 and xyzs.shape[0] == m + 128 - m % 128
    r = rays.cpu().numpy()
    order = np.argsort(r[:, 1], kind="stable")
    offs_, cnts = r[order, 1].astype(np.int64), r[order, 2].astype(np.int64)
    assert offs_[0] == 0 and np.array_equal(offs_[1:], np.cumsum(cnts)[:-1]) and offs_[-1] + cnts[-1] == m
    idx = np.random.default_rng(0).choice(N, 96, replace=False)
    o_h, d_h = ro.cpu().numpy()[idx], rd.cpu().numpy()[idx]
    n_h, f_h = oracle.near_far_from_aabb(o_h, d_h, aabb.cpu().numpy())
    _, _, _, r_o = oracle.march_rays_train(o_h, d_h, 1.0, bits.cpu().numpy(), 1, 128, n_h, f_h, align=-1, max_steps=1024)
    assert np.array_equal(r_o[:, 2], r[idx, 2])          # per-ray sample counts, bit-exact
    assert not xyzs[m:].any() and not deltas[m:].any()   # the padding rows the wrapper exposes are zero

    # ---- adjointness of gather and scatter over every evaluation (positive data: no cancellation in the checksums)
    cfg = dict(n_levels=16, base_resolution=16, per_level_scale=1.3819128274917603, log2_hashmap_size=19)
    n_params = 12196240
    x = xyzs[:m].contiguous()
    x2 = (x + torch.randn_like(x) * 0.01).contiguous()
    offs, P0 = grid_ops.stencil_offsets(center=True, second=True)
    P = offs.shape[0]
    _, offs_p = grid_ops._offs_arg(offs)
    params = torch.empty(n_params, device=cuda).uniform_(0.5, 1.0)
    feats = torch.empty(16, m * P, 2, device=cuda)
    L.call("mi3d_grid_encode_points_planes", L.ptr(x), L.ptr(x2), m, offs_p, int(P0), P, 1.0, L.ptr(params), 16, 16,
           cfg["per_level_scale"], 19, 2 * 3 ** 0.5 / 1024, L.ptr(feats), 0, L.stream())
    assert float(feats.min()) >= 0.5 - 1e-4 and float(feats.max()) <= 1.0 + 1e-4  # convex combinations of the table
    g = torch.empty(16, m * P, 2, device=cuda).uniform_(0.5, 1.5)
    lhs = sum(float(torch.dot(feats[l].reshape(-1).double(), g[l].reshape(-1).double())) for l in range(16))
    del feats
    grad = field_ops.scatter_binned(x, x2, offs, P0, 1.0, g, cfg, 2 * 3 ** 0.5 / 1024, n_params)
    rhs = float(torch.dot(params.double(), grad.double()))
    assert abs(lhs - rhs) <= 1e-5 * abs(lhs), (lhs, rhs)
    # all contributions are positive; the view frustum crosses about a fifth of the coarsest level's entries
    assert float(grad.min()) >= 0.0 and float((grad[: 4096 * 2] > 0).float().mean()) > 0.1
