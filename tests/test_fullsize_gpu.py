"""GPU, BASELINE config 2 at FULL size (128x128 rays, max_steps 1024, all-ones occupancy: ~10.9 M samples, 141 M field
evaluations) through size-independent properties: the march's slabs tile the sample buffer and per-ray counts agree
with the oracle on a ray sample; the gather and the binned scatter are adjoint, <encode(p), g> = <p, scatter(g)>, over
all 141 M evaluations - a checksum of the whole forward/backward pair that no small case exercises (two sample slices,
full region occupancy, every bin)."""
import ctypes

import numpy as np
import pytest
import torch

C_ull = ctypes.c_ulonglong

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
    del g, grad

    # ---- the same identity on the layout the headline runs: BINARY16 feature / gradient planes (torch.autocast).  The
    # gather rounds every feature pair to binary16 (unbiased, 2^-11 relative): over 4.5 G positive terms the checksum
    # moves by far less than the tolerance; the gradient planes are binary16-exact by construction.
    feats = torch.empty(16, m * P, 2, device=cuda, dtype=torch.float16)
    L.call("mi3d_grid_encode_points_planes", L.ptr(x), L.ptr(x2), m, offs_p, int(P0), P, 1.0, L.ptr(params), 16, 16,
           cfg["per_level_scale"], 19, 2 * 3 ** 0.5 / 1024, L.ptr(feats), 1, L.stream())
    g = torch.empty(16, m * P, 2, device=cuda).uniform_(0.5, 1.5).to(torch.float16)
    lhs = sum(float(torch.dot(feats[l].reshape(-1).double(), g[l].reshape(-1).double())) for l in range(16))
    del feats
    grad = field_ops.scatter_binned(x, x2, offs, P0, 1.0, g, cfg, 2 * 3 ** 0.5 / 1024, n_params)
    rhs = float(torch.dot(params.double(), grad.double()))
    assert abs(lhs - rhs) <= 1e-5 * abs(lhs), (lhs, rhs)


def test_c4_size_march_invariants_and_multi_slice_adjoint(cuda, oracle):
    """BASELINE config 4's shape - 256 x 256 rays, max_steps 2048, occupancy = sphere of radius 0.5 (SURVEY 8(d):
    ~36 M samples per view), 7 field evaluations per sample: the march's slabs tile the buffer, per-ray counts equal
    the oracle's on a ray sample (bit-exact), and gather / scatter stay adjoint over all ~250 M evaluations with the
    scatter cut into several sample slices (binary16 planes)."""
    import raymarching
    from mi3d import _lib as L
    from mi3d import field_ops, grid_ops, network, rays as R, sds_step
    H = W = 256
    ro, rd, _ = R.view_rays(H, W, device=cuda)
    ro, rd = ro.view(-1, 3).contiguous(), rd.view(-1, 3).contiguous()
    N = ro.shape[0]
    model = network.NeRFNetwork(sds_step.make_opt(max_steps=2048)).to(cuda)
    sds_step.set_bitfield(model, 0.5)
    bits = model.density_bitfield
    nears, fars = raymarching.near_far_from_aabb(ro, rd, model.aabb_train)
    counter = torch.zeros(2, dtype=torch.int32, device=cuda)
    xyzs, dirs, deltas, rays = raymarching.march_rays_train(ro, rd, 1.0, bits, 1, 128, nears, fars, counter, -1, True, 128,
                                                            True, 0, 2048)
    m = int(counter[0])
    assert int(counter[1]) == N and 25_000_000 < m < 45_000_000 and xyzs.shape[0] == m + 128 - m % 128
    r = rays.cpu().numpy()
    order = np.argsort(r[:, 1], kind="stable")
    offs_, cnts = r[order, 1].astype(np.int64), r[order, 2].astype(np.int64)
    assert offs_[0] == 0 and np.array_equal(offs_[1:], np.cumsum(cnts)[:-1]) and offs_[-1] + cnts[-1] == m
    # every emitted sample lies in an occupied cell's neighbourhood of the sphere (pruning really prunes)
    assert float(xyzs[:m].norm(dim=-1).max()) < 0.5 + 3 * 2 / 128
    # perturb=True draws its own noise on the device: the count check uses an unperturbed march of a ray sample
    idx = np.random.default_rng(1).choice(N, 64, replace=False)
    sel = torch.from_numpy(idx).to(cuda)
    c2 = torch.zeros(2, dtype=torch.int32, device=cuda)
    _, _, _, r_g = raymarching.march_rays_train(ro[sel].contiguous(), rd[sel].contiguous(), 1.0, bits, 1, 128,
                                                nears[sel].contiguous(), fars[sel].contiguous(), c2, -1, False, 128, True,
                                                0, 2048)
    o_h, d_h = ro.cpu().numpy()[idx], rd.cpu().numpy()[idx]
    n_h, f_h = oracle.near_far_from_aabb(o_h, d_h, model.aabb_train.cpu().numpy())
    _, _, _, r_o = oracle.march_rays_train(o_h, d_h, 1.0, bits.cpu().numpy(), 1, 128, n_h, f_h, align=-1, max_steps=2048)
    r_g = r_g.cpu().numpy()
    got = np.zeros(64, np.int64)
    got[r_g[:, 0]] = r_g[:, 2]
    assert np.array_equal(r_o[:, 2], got)

    cfg = dict(n_levels=16, base_resolution=16, per_level_scale=1.3819128274917603, log2_hashmap_size=19)
    n_params = 12196240
    x = xyzs[:m].contiguous()
    offs, P0 = grid_ops.stencil_offsets(center=True, second=False)
    P = offs.shape[0]
    _, offs_p = grid_ops._offs_arg(offs)
    step = 2 * 3 ** 0.5 / 2048
    params = torch.empty(n_params, device=cuda).uniform_(0.5, 1.0)
    feats = torch.empty(16, m * P, 2, device=cuda, dtype=torch.float16)
    L.call("mi3d_grid_encode_points_planes", L.ptr(x), L.ptr(None), m, offs_p, int(P0), P, 1.0, L.ptr(params), 16, 16,
           cfg["per_level_scale"], 19, step, L.ptr(feats), 1, L.stream())
    g = torch.empty(16, m * P, 2, device=cuda).uniform_(0.5, 1.5).to(torch.float16)
    lhs = sum(float(torch.dot(feats[l].reshape(-1).double(), g[l].reshape(-1).double())) for l in range(16))
    del feats
    need = L.lib().mi3d_grid_scatter_binned_workspace(m, P, 1.0, step, 16, 16, cfg["per_level_scale"], 19)
    assert need > field_ops.WORKSPACE_CAP_BYTES        # i.e. the call below really runs in several slices
    grad = field_ops.scatter_binned(x, None, offs, P0, 1.0, g, cfg, step, n_params)
    rhs = float(torch.dot(params.double(), grad.double()))
    assert abs(lhs - rhs) <= 1e-5 * abs(lhs), (lhs, rhs)
