"""GPU: the product kernels AND the C oracle against the REFERENCE'S OWN raymarching kernels
(/root/reference/raymarching/src/raymarching.cu built for gfx950 by oracle/build_ref.py into
oracle/_ref/_raymarching_ref.so - git-ignored, shipped with the snapshot).  This is what pins the oracle's restatement
of raymarching.cu: integer work bit-exact, float work 1e-4 (the reference's rows are in atomic-arrival order, so
everything is compared per ray id)."""
import numpy as np
import pytest
import torch

from conftest import make_rays, random_bitfield, sphere_bitfield

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ref():
    from oracle import build_ref
    m = build_ref.load()
    if m is None:
        pytest.skip("oracle/_ref/_raymarching_ref.so not built (python oracle/build_ref.py where /root/reference exists)")
    return m


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def test_integer_kernels_bit_exact(cuda, oracle, ref):
    import raymarching
    ax = torch.arange(128, dtype=torch.int32, device=cuda)
    co = torch.stack(torch.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(-1, 3).contiguous()
    N = co.shape[0]
    idx_ref = torch.empty(N, dtype=torch.int32, device=cuda)
    ref.morton3D(co, N, idx_ref)
    torch.cuda.synchronize()
    assert torch.equal(raymarching.morton3D(co), idx_ref)
    assert np.array_equal(oracle.morton3D(co.cpu().numpy()), idx_ref.cpu().numpy())
    back = torch.empty(N, 3, dtype=torch.int32, device=cuda)
    ref.morton3D_invert(idx_ref, N, back)
    torch.cuda.synchronize()
    assert torch.equal(back, co) and torch.equal(raymarching.morton3D_invert(idx_ref), co)
    rng = np.random.default_rng(0)
    grid = rng.normal(size=(2, 128 ** 3)).astype(np.float32)
    grid[0, :64] = 0.25
    g = T(grid, cuda)
    bits_ref = torch.empty(2 * 128 ** 3 // 8, dtype=torch.uint8, device=cuda)
    ref.packbits(g, bits_ref.numel(), 0.25, bits_ref)
    torch.cuda.synchronize()
    assert torch.equal(raymarching.packbits(g, 0.25), bits_ref)
    assert np.array_equal(oracle.packbits(grid, 0.25), bits_ref.cpu().numpy())


def test_near_far_bit_exact(cuda, oracle, ref):
    import raymarching
    rng = np.random.default_rng(1)
    o, d = make_rays(rng, 5000)
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    n_ref, f_ref = torch.empty(5000, device=cuda), torch.empty(5000, device=cuda)
    ref.near_far_from_aabb(T(o, cuda), T(d, cuda), T(aabb, cuda), 5000, 0.2, n_ref, f_ref)
    torch.cuda.synchronize()
    n_p, f_p = raymarching.near_far_from_aabb(T(o, cuda), T(d, cuda), T(aabb, cuda), 0.2)
    n_o, f_o = oracle.near_far_from_aabb(o, d, aabb, 0.2)
    for a, b in ((n_p, n_ref), (f_p, f_ref)):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    assert np.array_equal(n_o.view(np.uint32), n_ref.cpu().numpy().view(np.uint32))
    assert np.array_equal(f_o.view(np.uint32), f_ref.cpu().numpy().view(np.uint32))


@pytest.mark.parametrize("C_,bound,kind,dt_gamma,max_steps", [(1, 1.0, "sphere", 0.0, 256), (1, 1.0, "random", 1 / 128, 128),
                                                                (2, 2.0, "sphere", 0.0, 256), (1, 1.0, "ones", 0.0, 512)])
def test_march_rays_train_against_reference_kernel(cuda, oracle, ref, C_, bound, kind, dt_gamma, max_steps):
    import raymarching
    rng = np.random.default_rng(5)
    N, H = 3000, 128
    o, d = make_rays(rng, N, bound)
    aabb = np.array([-bound] * 3 + [bound] * 3, np.float32)
    nears, fars = oracle.near_far_from_aabb(o, d, aabb)
    bits = {"ones": lambda: np.full(C_ * H ** 3 // 8, 255, np.uint8), "sphere": lambda: sphere_bitfield(oracle, C_, H, 0.45 * bound),
            "random": lambda: random_bitfield(rng, C_, H)}[kind]()
    noises = rng.uniform(0, 1, N).astype(np.float32)
    M = N * max_steps
    xr, dr = torch.zeros(M, 3, device=cuda), torch.zeros(M, 3, device=cuda)
    lr = torch.zeros(M, 2, device=cuda)
    rr = torch.zeros(N, 3, dtype=torch.int32, device=cuda)
    cr = torch.zeros(2, dtype=torch.int32, device=cuda)
    ref.march_rays_train(T(o, cuda), T(d, cuda), T(bits, cuda), bound, dt_gamma, max_steps, N, C_, H, M, T(nears, cuda),
                         T(fars, cuda), xr, dr, lr, rr, cr, T(noises, cuda))
    torch.cuda.synchronize()
    rr_h, m_ref = rr.cpu().numpy(), int(cr[0])
    assert int(cr[1]) == N
    by_ray = np.empty((N, 2), np.int64)
    by_ray[rr_h[:, 0]] = rr_h[:, 1:]
    # C oracle and product (through the C ABI): per-ray sample counts and, ray by ray, the samples themselves.
    # hipcc is free to contract a*b+c differently from the explicit fmaf pattern the product and the oracle share, and one
    # ulp at a voxel face flips an occupancy decision, so the bar is SURVEY 7's: counts equal on >= 99.9 % of the rays,
    # samples of equal-count rays within 2e-6; bit-exactness is reported.
    xo, do, lo, ro_, co_ = oracle.march_rays_train(o, d, bound, bits, C_, H, nears, fars, noises, align=-1,
                                                   dt_gamma=dt_gamma, max_steps=max_steps, return_counter=True)
    from mi3d import _lib as L
    xp, dp = torch.empty(M, 3, device=cuda), torch.empty(M, 3, device=cuda)
    lp = torch.empty(M, 2, device=cuda)
    rp = torch.empty(N, 3, dtype=torch.int32, device=cuda)
    cp = torch.zeros(2, dtype=torch.int32, device=cuda)
    keep = [T(a, cuda) for a in (o, d, bits, nears, fars, noises)]  # raw pointers: the tensors must outlive the launch
    L.call("mi3d_march_rays_train", L.ptr(keep[0]), L.ptr(keep[1]), L.ptr(keep[2]), float(bound), float(dt_gamma),
           max_steps, N, C_, H, M, L.ptr(keep[3]), L.ptr(keep[4]), L.ptr(xp), L.ptr(dp), L.ptr(lp), L.ptr(rp),
           L.ptr(cp), L.ptr(keep[5]), L.stream())
    torch.cuda.synchronize()
    rp_h = rp.cpu().numpy()
    same_o = ro_[:, 2] == by_ray[:, 1]
    same_p = rp_h[:, 2] == by_ray[:, 1]
    assert same_o.mean() >= 0.999 and same_p.mean() >= 0.999, (same_o.mean(), same_p.mean())
    xr_h, lr_h, xp_h, lp_h = xr.cpu().numpy(), lr.cpu().numpy(), xp.cpu().numpy(), lp.cpu().numpy()
    checked = exact = 0
    for n in rng.choice(N, 400, replace=False):
        cnt = int(by_ray[n, 1])
        if cnt == 0 or not (same_o[n] and same_p[n]):
            continue
        a, b, c = int(by_ray[n, 0]), int(rp_h[n, 1]), int(ro_[n, 1])
        np.testing.assert_allclose(xp_h[b:b + cnt], xr_h[a:a + cnt], rtol=0, atol=2e-6)   # product vs reference
        np.testing.assert_allclose(lp_h[b:b + cnt], lr_h[a:a + cnt], rtol=0, atol=2e-6)
        np.testing.assert_allclose(xo[c:c + cnt], xr_h[a:a + cnt], rtol=0, atol=2e-6)     # oracle vs reference
        np.testing.assert_allclose(lo[c:c + cnt], lr_h[a:a + cnt], rtol=0, atol=2e-6)
        exact += int(np.array_equal(xp_h[b:b + cnt].view(np.uint32), xr_h[a:a + cnt].view(np.uint32)))
        checked += 1
    assert checked > 100
    print(f"march vs reference kernel: counts equal on {same_p.mean():.4%} of rays (oracle {same_o.mean():.4%}), "
          f"{exact}/{checked} sampled rays bit-exact, total samples {int(cp[0])} vs {m_ref}")
    # what DESIGN.md section 4 claims, asserted: the product agrees with the reference kernel on EVERY ray's sample count
    # and bit for bit on every sampled ray (both spell the same fused multiply-adds); the C oracle is held to 99.9 %
    assert same_p.all() and exact == checked and int(cp[0]) == m_ref, (same_p.mean(), exact, checked)
    import json, os
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "march_parity.jsonl"), "a") as f:
        f.write(json.dumps({"config": [C_, bound, kind, dt_gamma, max_steps], "rays_equal_counts": float(same_p.mean()),
                            "oracle_rays_equal_counts": float(same_o.mean()), "sampled_rays_bit_exact": [exact, checked],
                            "total_samples": [int(cp[0]), m_ref]}) + "\n")


@pytest.mark.parametrize("T_thresh", [1e-4, 0.0])
def test_composite_train_against_reference_kernel(cuda, oracle, ref, T_thresh):
    import raymarching
    rng = np.random.default_rng(8)
    N, S = 700, 150
    cnt = rng.integers(0, S + 1, N).astype(np.int32)
    cnt[:3] = [0, 1, S]
    offs = np.concatenate([[0], np.cumsum(cnt)[:-1]]).astype(np.int32)
    M = int(cnt.sum())
    rays = np.stack([rng.permutation(N).astype(np.int32), offs, cnt], 1)
    sig = rng.uniform(0, 80, M).astype(np.float32)
    rgb = rng.uniform(0, 1, (M, 3)).astype(np.float32)
    deltas = np.stack([rng.uniform(0.002, 0.01, M), rng.uniform(0.002, 0.02, M)], 1).astype(np.float32)
    ws_r, dep_r, img_r = torch.empty(N, device=cuda), torch.empty(N, device=cuda), torch.empty(N, 3, device=cuda)
    ref.composite_rays_train_forward(T(sig, cuda), T(rgb, cuda), T(deltas, cuda), T(rays, cuda), M, N, T_thresh, ws_r, dep_r,
                                     img_r)
    torch.cuda.synchronize()
    s_t, c_t = T(sig, cuda).requires_grad_(), T(rgb, cuda).requires_grad_()
    ws_p, dep_p, img_p = raymarching.composite_rays_train(s_t, c_t, T(deltas, cuda), T(rays, cuda), T_thresh)
    ws_o, dep_o, img_o = oracle.composite_rays_train(sig, rgb, deltas, rays, T_thresh)
    for got, name in ((ws_p, "ws"), (dep_p, "depth"), (img_p, "image")):
        want = {"ws": ws_r, "depth": dep_r, "image": img_r}[name]
        np.testing.assert_allclose(got.detach().cpu().numpy(), want.cpu().numpy(), rtol=1e-4, atol=2e-6, err_msg=name)
    np.testing.assert_allclose(ws_o, ws_r.cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(img_o, img_r.cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(dep_o, dep_r.cpu().numpy(), rtol=1e-5, atol=1e-6)
    gws, gim = rng.normal(size=N).astype(np.float32), rng.normal(size=(N, 3)).astype(np.float32)
    gs_r, gc_r = torch.zeros(M, device=cuda), torch.zeros(M, 3, device=cuda)
    ref.composite_rays_train_backward(T(gws, cuda), T(gim, cuda), T(sig, cuda), T(rgb, cuda), T(deltas, cuda), T(rays, cuda), ws_r,
                                      img_r, M, N, T_thresh, gs_r, gc_r)
    torch.cuda.synchronize()
    torch.autograd.backward([ws_p, img_p], [T(gws, cuda), T(gim, cuda)])
    scale = float(gs_r.abs().max())
    assert float((s_t.grad - gs_r).abs().max()) <= 2e-4 * scale
    np.testing.assert_allclose(c_t.grad.cpu().numpy(), gc_r.cpu().numpy(), rtol=1e-4, atol=1e-6)
    gs_o, gc_o = oracle.composite_rays_train_backward(gws, gim, sig, rgb, deltas, rays, ws_o, img_o, T_thresh)
    assert np.abs(gs_o - gs_r.cpu().numpy()).max() <= 2e-5 * scale
    np.testing.assert_allclose(gc_o, gc_r.cpu().numpy(), rtol=1e-5, atol=1e-6)


def test_inference_march_composite_against_reference_kernel(cuda, oracle, ref):
    from mi3d import _lib as L
    rng = np.random.default_rng(12)
    N, H, n_step = 900, 128, 4
    o, d = make_rays(rng, N)
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    nears, fars = oracle.near_far_from_aabb(o, d, aabb)
    bits = sphere_bitfield(oracle, 1, H, 0.5)
    alive = np.flatnonzero(nears < fars).astype(np.int32)
    n_alive = alive.shape[0]
    noises = rng.uniform(0, 1, n_alive).astype(np.float32)
    M = n_alive * n_step
    outs = []
    for which in ("ref", "product"):
        xyzs, dirs = torch.zeros(M, 3, device=cuda), torch.zeros(M, 3, device=cuda)
        deltas = torch.zeros(M, 2, device=cuda)
        args = (T(alive, cuda), T(nears, cuda), T(o, cuda), T(d, cuda), T(bits, cuda), T(nears, cuda), T(fars, cuda),
                T(noises, cuda))  # kept alive across the raw-pointer launch
        if which == "ref":
            ref.march_rays(n_alive, n_step, args[0], args[1], args[2], args[3], 1.0, 0.0, 256, 1, H, args[4],
                           args[5], args[6], xyzs, dirs, deltas, args[7])
        else:
            L.call("mi3d_march_rays", n_alive, n_step, L.ptr(args[0]), L.ptr(args[1]), L.ptr(args[2]), L.ptr(args[3]), 1.0, 0.0,
                   256, 1, H, L.ptr(args[4]), L.ptr(args[5]), L.ptr(args[6]), L.ptr(xyzs), L.ptr(dirs),
                   L.ptr(deltas), L.ptr(args[7]), L.stream())
        torch.cuda.synchronize()
        outs.append((xyzs.cpu().numpy(), deltas.cpu().numpy()))
    # rows of a ray that ends before n_step stay zero in both; a one-ulp occupancy flip shifts a whole ray, so compare
    # the fraction of identical rays
    ray_same = (np.abs(outs[0][0] - outs[1][0]).reshape(n_alive, -1).max(1) <= 2e-6)
    assert ray_same.mean() >= 0.999, ray_same.mean()
    xo, do, lo = oracle.march_rays(n_alive, n_step, alive, nears, o, d, 1.0, bits, 1, H, nears, fars, noises=noises, max_steps=256)
    ray_same_o = (np.abs(xo[:M] - outs[0][0]).reshape(n_alive, -1).max(1) <= 2e-6)
    assert ray_same_o.mean() >= 0.999, ray_same_o.mean()
    # one composite_rays round on the marched samples
    sig = rng.uniform(0, 40, M).astype(np.float32)
    rgb = rng.uniform(0, 1, (M, 3)).astype(np.float32)
    nrm = rng.uniform(0, 1, (M, 3)).astype(np.float32)
    res = []
    for which in ("ref", "product"):
        ra, rt = T(alive.copy(), cuda), T(nears.copy(), cuda)
        ws, dep = torch.zeros(N, device=cuda), torch.zeros(N, device=cuda)
        img, nor = torch.zeros(N, 3, device=cuda), torch.zeros(N, 3, device=cuda)
        dl, sg, cl, nm = T(outs[0][1], cuda), T(sig, cuda), T(rgb, cuda), T(nrm, cuda)
        if which == "ref":
            ref.composite_rays(n_alive, n_step, 1e-2, ra, rt, sg, cl, nm, dl, ws, dep, img, nor)
        else:
            L.call("mi3d_composite_rays", n_alive, n_step, 1e-2, L.ptr(ra), L.ptr(rt), L.ptr(sg), L.ptr(cl),
                   L.ptr(nm), L.ptr(dl), L.ptr(ws), L.ptr(dep), L.ptr(img), L.ptr(nor), L.stream())
        torch.cuda.synchronize()
        res.append([t.cpu().numpy() for t in (ra, rt, ws, dep, img, nor)])
    assert np.array_equal(res[0][0], res[1][0])
    for a, b in zip(res[0][1:], res[1][1:]):
        np.testing.assert_allclose(b, a, rtol=1e-5, atol=1e-6)


def test_device_driven_inference_loop_against_reference_kernels_round_by_round(cuda, oracle, ref):
    """The whole eval loop (nerf/renderer.py:526-551): the REFERENCE kernels driven by the reference's host logic
    (n_step from the host-side alive count, boolean-mask compaction) against the product's device-driven loop
    (C ABI Part 1b: mi3d_infer_begin / march_rays_ctl / composite_rays_ctl / compact_alive_ctl) - the alive list
    bit-exact after every round, the control block equal to the host's state, the accumulated outputs at the end.
    The field is an analytic density so both sides see identical sigmas for identical samples."""
    import raymarching
    rng = np.random.default_rng(21)
    N, H, max_steps, T_thresh, align = 2000, 128, 128, 1e-2, 128
    o, d = make_rays(rng, N)
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    nears, fars = oracle.near_far_from_aabb(o, d, aabb)
    bits = T(sphere_bitfield(oracle, 1, H, 0.6), cuda)
    O, D, NE, FA = T(o, cuda), T(d, cuda), T(nears, cuda), T(fars, cuda)

    def field(xyzs):
        r2 = (xyzs ** 2).sum(-1)
        sig = 8.0 * torch.exp(-r2 / 0.08)            # soft blob: rays die at different rounds
        rgb = torch.sigmoid(xyzs * 3.0)
        nrm = torch.tanh(xyzs) * 0.5 + 0.5
        return sig.contiguous(), rgb.contiguous(), nrm.contiguous()

    # ---- reference kernels under the reference's host loop
    ws_r, dep_r = torch.zeros(N, device=cuda), torch.zeros(N, device=cuda)
    img_r, nor_r = torch.zeros(N, 3, device=cuda), torch.zeros(N, 3, device=cuda)
    alive_r = torch.arange(N, dtype=torch.int32, device=cuda)
    rt_r = NE.clone()
    # ---- product loop, device-driven
    ws_p, dep_p = torch.zeros(N, device=cuda), torch.zeros(N, device=cuda)
    img_p, nor_p = torch.zeros(N, 3, device=cuda), torch.zeros(N, 3, device=cuda)
    rt_p = NE.clone()
    rows_cap = N + 2 * align
    xb, db = torch.zeros(rows_cap, 3, device=cuda), torch.zeros(rows_cap, 3, device=cuda)
    lb = torch.zeros(rows_cap, 2, device=cuda)
    ctl, alive_p = raymarching.infer_begin(N, cuda, align)
    spare = torch.empty_like(alive_p)

    step = rounds = 0
    while step < max_steps:
        n_alive = alive_r.shape[0]
        state = ctl.tolist()
        assert state[0] == n_alive and state[3] == step and state[4] == rounds
        if n_alive <= 0:
            break
        n_step = max(min(N // n_alive, 8), 1)
        M = n_alive * n_step
        M += align - (M % align)
        assert state[1] == n_step and state[2] == M
        assert torch.equal(alive_p[:n_alive], alive_r)
        xyzs, dirs = torch.zeros(M, 3, device=cuda), torch.zeros(M, 3, device=cuda)
        deltas = torch.zeros(M, 2, device=cuda)
        noises = torch.zeros(n_alive, device=cuda)
        ref.march_rays(n_alive, n_step, alive_r, rt_r, O, D, 1.0, 0.0, max_steps, 1, H, bits, NE, FA, xyzs, dirs, deltas,
                       noises)
        sg, cl, nm = field(xyzs)
        ref.composite_rays(n_alive, n_step, T_thresh, alive_r, rt_r, sg, cl, nm, deltas, ws_r, dep_r, img_r, nor_r)
        alive_r = alive_r[alive_r >= 0]
        # the product round is launched for the initial upper bound (N rays), as a host that never looks would
        raymarching.march_rays_ctl(ctl, N, alive_p, rt_p, O, D, 1.0, bits, 1, H, FA, xb, db, lb, None, 0.0, max_steps)
        assert torch.allclose(xb[:M], xyzs, rtol=0, atol=2e-6) and torch.allclose(lb[:M], deltas, rtol=0, atol=2e-6)
        assert torch.equal(db[:M], dirs)
        sg2, cl2, nm2 = field(xb)
        raymarching.composite_rays_ctl(ctl, N, alive_p, rt_p, sg2, cl2, nm2, lb, ws_p, dep_p, img_p, nor_p, T_thresh)
        raymarching.compact_alive_ctl(ctl, alive_p, spare, N, align, max_steps)
        alive_p, spare = spare, alive_p
        step += n_step
        rounds += 1
    assert rounds > 5
    assert ctl.tolist()[0] == 0 or step >= max_steps
    for a, b in ((ws_r, ws_p), (dep_r, dep_p), (img_r, img_p), (nor_r, nor_p), (rt_r, rt_p)):
        np.testing.assert_allclose(b.cpu().numpy(), a.cpu().numpy(), rtol=1e-5, atol=1e-6)
