"""GPU parity: the HIP raymarching operators (through the `raymarching` drop-in package -> C ABI) against the
CPU oracle on identical seeded inputs.  Integer outputs and everything derived from the DDA walk are bit-exact;
composited floats are compared at the tolerance BASELINE.json states (1e-4 relative)."""
import numpy as np
import pytest
import torch

from conftest import make_rays, random_bitfield, sphere_bitfield

pytestmark = pytest.mark.gpu

RTOL = 1e-4  # north_star: "within 1e-4 relative fp32"


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def test_loads_native_library(cuda):
    from mi3d import _lib
    assert _lib.lib().mi3d_abi_version() == 5


def test_near_far_bit_exact(cuda, oracle):
    import raymarching
    rng = np.random.default_rng(1)
    o, d = make_rays(rng, 5000)
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    for min_near in (0.2, 0.05):
        n_ref, f_ref = oracle.near_far_from_aabb(o, d, aabb, min_near)
        n, f = raymarching.near_far_from_aabb(T(o, cuda), T(d, cuda), T(aabb, cuda), min_near)
        assert np.array_equal(n.cpu().numpy(), n_ref) and np.array_equal(f.cpu().numpy(), f_ref)
    assert (n_ref == np.finfo(np.float32).max).any()  # the miss path was exercised


def test_sph_from_ray(cuda, oracle):
    import raymarching
    rng = np.random.default_rng(2)
    o = rng.uniform(-0.5, 0.5, (1000, 3)).astype(np.float32)
    d = rng.normal(size=(1000, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    ref = oracle.sph_from_ray(o, d, 2.0)
    out = raymarching.sph_from_ray(T(o, cuda), T(d, cuda), 2.0).cpu().numpy()
    np.testing.assert_allclose(out, ref, rtol=1e-4, atol=2e-5)


def test_morton_all_coords_bit_exact(cuda, oracle):
    import raymarching
    co = np.stack(np.meshgrid(*[np.arange(128)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(np.int32)
    ref = oracle.morton3D(co)
    idx = raymarching.morton3D(T(co, cuda))
    assert np.array_equal(idx.cpu().numpy(), ref)
    back = raymarching.morton3D_invert(idx)
    assert np.array_equal(back.cpu().numpy(), co)
    # 10-bit range
    big = np.array([[1023, 0, 0], [0, 1023, 0], [0, 0, 1023], [1023, 1023, 1023]], np.int32)
    assert np.array_equal(raymarching.morton3D(T(big, cuda)).cpu().numpy(), oracle.morton3D(big))
    assert raymarching.morton3D(torch.zeros(0, 3, dtype=torch.int32, device=cuda)).numel() == 0


def test_packbits_bit_exact(cuda, oracle):
    import raymarching
    rng = np.random.default_rng(3)
    grid = rng.exponential(1.0, (2, 128 ** 3)).astype(np.float32)
    grid[0, :64] = -1.0
    for thresh in (0.0, 0.7, 10.0):
        grid[1, 100:108] = thresh  # strict '>' at equality
        ref = oracle.packbits(grid, thresh)
        out = raymarching.packbits(T(grid, cuda), thresh)
        assert np.array_equal(out.cpu().numpy(), ref)


MARCH_CASES = [
    # C, bound, bitfield, dt_gamma, max_steps, perturb
    (1, 1.0, "ones", 0.0, 1024, True),
    (1, 1.0, "ones", 0.0, 64, False),
    (1, 1.0, "sphere", 0.0, 512, True),
    (1, 1.0, "random", 1 / 128, 256, True),
    (3, 4.0, "random", 1 / 128, 1024, True),
    (2, 2.0, "sphere", 0.0, 1024, False),
    (1, 1.0, "empty", 0.0, 128, False),
]


def _bits(kind, rng, oracle, C, H, bound):
    if kind == "ones":
        return np.full(C * H ** 3 // 8, 255, np.uint8)
    if kind == "empty":
        return np.zeros(C * H ** 3 // 8, np.uint8)
    if kind == "sphere":
        return sphere_bitfield(oracle, C, H, 0.3 * bound)
    return random_bitfield(rng, C, H)


def _per_ray(rays, *arrays):
    """Re-key slab-ordered outputs by ray id so two legal slab orders compare equal."""
    out = {}
    for rid, off, cnt in rays:
        out[int(rid)] = tuple(a[off:off + cnt] for a in arrays)
    return out


@pytest.mark.parametrize("C,bound,kind,dt_gamma,max_steps,perturb", MARCH_CASES)
def test_march_rays_train_bit_exact(cuda, oracle, C, bound, kind, dt_gamma, max_steps, perturb):
    import raymarching
    from mi3d import _lib as L
    H, N = 128, 3001  # ragged: not a multiple of the wave size
    rng = np.random.default_rng(10 + C)
    o, d = make_rays(rng, N, bound)
    aabb = np.array([-bound] * 3 + [bound] * 3, np.float32)
    nears, fars = oracle.near_far_from_aabb(o, d, aabb)
    bits = _bits(kind, rng, oracle, C, H, bound)
    noises = rng.random(N).astype(np.float32) if perturb else np.zeros(N, np.float32)
    xr, dr, lr, rr, cr = oracle.march_rays_train(o, d, bound, bits, C, H, nears, fars, noises=noises, align=128,
                                                 dt_gamma=dt_gamma, max_steps=max_steps, return_counter=True)
    # call the C ABI directly so the test controls `noises` (the wrapper draws them with torch.rand)
    M = N * max_steps
    dev = cuda
    xyzs = torch.full((M, 3), 7.0, device=dev)
    dirs = torch.full((M, 3), 7.0, device=dev)
    deltas = torch.full((M, 2), 7.0, device=dev)
    rays = torch.empty(N, 3, dtype=torch.int32, device=dev)
    counter = torch.zeros(2, dtype=torch.int32, device=dev)
    args = [T(o, dev), T(d, dev), T(bits, dev), T(nears, dev), T(fars, dev), T(noises, dev)]
    L.call("mi3d_march_rays_train", L.ptr(args[0]), L.ptr(args[1]), L.ptr(args[2]), float(bound), float(dt_gamma),
           max_steps, N, C, H, M, L.ptr(args[3]), L.ptr(args[4]), L.ptr(xyzs), L.ptr(dirs), L.ptr(deltas),
           L.ptr(rays), L.ptr(counter), L.ptr(args[5]), L.stream())
    L.call("mi3d_march_zero_tail", L.ptr(counter), 128, M, L.ptr(xyzs), L.ptr(dirs), L.ptr(deltas), L.stream())
    cnt = counter.cpu().numpy()
    assert np.array_equal(cnt, cr), (cnt, cr)
    m = int(cnt[0])
    rays_h = rays.cpu().numpy()
    # integer side: every ray's step count is bit-exact; slabs tile [0, m) without overlap
    assert np.array_equal(rays_h[:, 0], np.arange(N))
    assert np.array_equal(rays_h[:, 2], rr[:, 2])
    hit = rays_h[rays_h[:, 2] > 0]  # empty rays own no rows (their offset may coincide with a neighbour's)
    if len(hit):
        order = np.argsort(hit[:, 1], kind="stable")
        offs, cnts = hit[order, 1], hit[order, 2]
        assert offs[0] == 0 and np.array_equal(offs[1:], np.cumsum(cnts)[:-1]) and offs[-1] + cnts[-1] == m
    # float side: identical fused-multiply-add sites => identical bits, compared per ray id
    got = _per_ray(rays_h, xyzs[:m].cpu().numpy(), dirs[:m].cpu().numpy(), deltas[:m].cpu().numpy())
    ref = _per_ray(rr, xr, dr, lr)
    for rid in range(N):
        for a, b in zip(got[rid], ref[rid]):
            assert np.array_equal(a, b), rid
    # padding rows are zero, as the reference's zero-filled buffers leave them
    pad = 128 - m % 128
    assert float(xyzs[m:m + pad].abs().max()) == 0 and float(deltas[m:m + pad].abs().max()) == 0
    if kind == "empty":
        assert m == 0


def test_march_wrapper_contract(cuda, oracle):
    """The drop-in op: padding rule, dtypes, step_counter side effects, perturb RNG, overflow drop."""
    import raymarching
    rng = np.random.default_rng(5)
    N, H = 777, 128
    o, d = make_rays(rng, N)
    aabb = T(np.array([-1, -1, -1, 1, 1, 1], np.float32), cuda)
    bits = T(sphere_bitfield(oracle, 1, H, 0.4), cuda)
    nears, fars = raymarching.near_far_from_aabb(T(o, cuda), T(d, cuda), aabb)
    counter = torch.zeros(2, dtype=torch.int32, device=cuda)
    xyzs, dirs, deltas, rays = raymarching.march_rays_train(T(o, cuda), T(d, cuda), 1.0, bits, 1, H, nears, fars,
                                                            counter, -1, False, 128, True, 0, 512)
    m = int(counter[0])
    assert int(counter[1]) == N
    assert xyzs.shape[0] == m + (128 - m % 128) and xyzs.shape[0] % 128 == 0
    assert dirs.shape == xyzs.shape and deltas.shape == (xyzs.shape[0], 2) and rays.shape == (N, 3)
    ref = oracle.march_rays_train(o, d, 1.0, bits.cpu().numpy(), 1, H, nears.cpu().numpy(), fars.cpu().numpy(),
                                  align=128, max_steps=512)
    # slabs are reserved in wave-arrival order: compare per ray id
    got = _per_ray(rays.cpu().numpy(), xyzs.cpu().numpy(), deltas.cpu().numpy())
    want = _per_ray(ref[3], ref[0], ref[2])
    for rid in range(N):
        assert np.array_equal(got[rid][0], want[rid][0]) and np.array_equal(got[rid][1], want[rid][1])
    assert float(xyzs[m:].abs().max()) == 0 and float(deltas[m:].abs().max()) == 0
    # mean_count mode: M is capped, overflowing rays are dropped silently and composite zeroes them
    mc = max(128, m // 2)
    x2, d2, l2, r2 = raymarching.march_rays_train(T(o, cuda), T(d, cuda), 1.0, bits, 1, H, nears, fars, None, mc,
                                                  False, 128, False, 0, 512)
    assert x2.shape[0] == mc + (128 - mc % 128)
    sig = torch.ones(x2.shape[0], device=cuda)
    rgb = torch.ones(x2.shape[0], 3, device=cuda)
    ws, dep, img = raymarching.composite_rays_train(sig, rgb, l2, r2)
    r2h = r2.cpu().numpy()
    dropped = (r2h[:, 1] + r2h[:, 2] > x2.shape[0]) & (r2h[:, 2] > 0)
    assert dropped.any() and float(ws[torch.from_numpy(r2h[dropped, 0]).long()].abs().max()) == 0


def _composite_inputs(rng, oracle, N=1500, max_steps=256, kind="sphere"):
    o, d = make_rays(rng, N)
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    nears, fars = oracle.near_far_from_aabb(o, d, aabb)
    bits = sphere_bitfield(oracle, 1, 128, 0.45) if kind == "sphere" else np.full(128 ** 3 // 8, 255, np.uint8)
    xyzs, dirs, deltas, rays = oracle.march_rays_train(o, d, 1.0, bits, 1, 128, nears, fars,
                                                       noises=rng.random(N).astype(np.float32), align=128,
                                                       max_steps=max_steps)
    m = xyzs.shape[0]
    sig = rng.exponential(1.0, m).astype(np.float32)
    return sig, rng.random((m, 3)).astype(np.float32), deltas, rays


@pytest.mark.parametrize("density,T_thresh", [(3.0, 1e-4), (60.0, 1e-4), (400.0, 1e-2), (0.0, 1e-4)])
@pytest.mark.parametrize("sdf", [False, True])
def test_composite_train_forward_backward(cuda, oracle, density, T_thresh, sdf):
    import raymarching
    rng = np.random.default_rng(20)
    sig, rgb, deltas, rays = _composite_inputs(rng, oracle, kind="ones" if density < 100 else "sphere")
    sig = sig * density
    if sdf:
        sig = np.clip(sig * 0.01, 0, 0.95).astype(np.float32)  # alpha = sigma must stay in [0,1)
    rays = rays.copy()
    rays[5, 1] = sig.shape[0] - 1  # a ray whose slab overflows M: must come out as zeros
    rays[5, 2] = 64
    ws_r, dep_r, img_r = oracle.composite_rays_train(sig, rgb, deltas, rays, T_thresh, sdf=sdf)
    fn = raymarching.composite_sdf_rays_train if sdf else raymarching.composite_rays_train
    s_t = T(sig, cuda).requires_grad_(True)
    c_t = T(rgb, cuda).requires_grad_(True)
    ws, dep, img = fn(s_t, c_t, T(deltas, cuda), T(rays, cuda), T_thresh)
    np.testing.assert_allclose(ws.detach().cpu().numpy(), ws_r, rtol=RTOL, atol=1e-6)
    np.testing.assert_allclose(img.detach().cpu().numpy(), img_r, rtol=RTOL, atol=1e-6)
    np.testing.assert_allclose(dep.detach().cpu().numpy(), dep_r, rtol=RTOL, atol=1e-5)
    assert ws_r[rays[5, 0]] == 0 and float(ws.detach()[rays[5, 0]]) == 0

    g_ws = rng.normal(size=ws_r.shape).astype(np.float32)
    g_img = rng.normal(size=img_r.shape).astype(np.float32)
    g_dep = rng.normal(size=ws_r.shape).astype(np.float32)  # must be ignored
    torch.autograd.backward([ws, dep, img], [T(g_ws, cuda), T(g_dep, cuda), T(g_img, cuda)])
    gs_r, gc_r = oracle.composite_rays_train_backward(g_ws, g_img, sig, rgb, deltas, rays, ws_r, img_r, T_thresh,
                                                      sdf=sdf)
    gs, gc = s_t.grad.cpu().numpy(), c_t.grad.cpu().numpy()
    np.testing.assert_allclose(gc, gc_r, rtol=RTOL, atol=1e-6)
    # d sigma sums terms of mixed sign: tolerance relative to the per-ray gradient scale
    scale = np.abs(gs_r).max() + 1e-12
    assert np.abs(gs - gs_r).max() <= 2e-4 * scale
    # samples behind an early-terminated ray keep exactly zero gradient
    # (the sample where T crosses T_thresh can differ by one when the two transmittances differ by an ulp)
    assert np.mean((gs_r == 0) != (gs == 0)) < 2e-3


def test_composite_empty_and_single(cuda, oracle):
    import raymarching
    rays = np.array([[0, 0, 0], [1, 0, 1], [2, 1, 3]], np.int32)
    sig = np.array([2.0, 0.5, 100.0, 3.0], np.float32)
    rgb = np.arange(12, dtype=np.float32).reshape(4, 3) / 12
    deltas = np.full((4, 2), 0.1, np.float32)
    ref = oracle.composite_rays_train(sig, rgb, deltas, rays)
    out = raymarching.composite_rays_train(T(sig, cuda), T(rgb, cuda), T(deltas, cuda), T(rays, cuda))
    for a, b in zip(out, ref):
        np.testing.assert_allclose(a.cpu().numpy(), b, rtol=RTOL, atol=1e-7)
    e = raymarching.composite_rays_train(torch.zeros(0, device=cuda), torch.zeros(0, 3, device=cuda),
                                         torch.zeros(0, 2, device=cuda), torch.zeros(0, 3, dtype=torch.int32,
                                                                                    device=cuda))
    assert e[0].numel() == 0


def test_inference_march_composite_loop(cuda, oracle):
    """The reference's test-time loop (nerf/renderer.py:526-551) driven through both implementations."""
    import raymarching
    rng = np.random.default_rng(30)
    N, H = 600, 128
    o, d = make_rays(rng, N)
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    nears, fars = oracle.near_far_from_aabb(o, d, aabb)
    bits = sphere_bitfield(oracle, 1, H, 0.5)

    def field(x):  # deterministic toy field evaluated on the host for both sides
        s = (40 * np.exp(-np.sum(x * x, 1) / 0.05)).astype(np.float32)
        c = (0.5 + 0.5 * np.sin(x * 7)).astype(np.float32)
        nrm = (x / (np.linalg.norm(x, axis=1, keepdims=True) + 1e-6)).astype(np.float32)
        return s, c, nrm

    # oracle
    ws_r, dep_r = np.zeros(N, np.float32), np.zeros(N, np.float32)
    img_r, nrm_r = np.zeros((N, 3), np.float32), np.zeros((N, 3), np.float32)
    alive_r, t_r = np.arange(N, dtype=np.int32), nears.copy()
    # device
    ws, dep = torch.zeros(N, device=cuda), torch.zeros(N, device=cuda)
    img, nrm = torch.zeros(N, 3, device=cuda), torch.zeros(N, 3, device=cuda)
    alive, t = torch.arange(N, dtype=torch.int32, device=cuda), T(nears, cuda).clone()
    step = 0
    while step < 256:
        n_alive = alive_r.shape[0]
        assert alive.shape[0] == n_alive
        if n_alive <= 0:
            break
        n_step = max(min(N // n_alive, 8), 1)
        xr, dr, lr = oracle.march_rays(n_alive, n_step, alive_r, t_r, o, d, 1.0, bits, 1, H, nears, fars, align=128,
                                       max_steps=256)
        x, dd, l = raymarching.march_rays(n_alive, n_step, alive, t, T(o, cuda), T(d, cuda), 1.0, T(bits, cuda), 1,
                                          H, T(nears, cuda), T(fars, cuda), 128, False, 0, 256)
        assert np.array_equal(x.cpu().numpy(), xr) and np.array_equal(l.cpu().numpy(), lr)
        s, c, nm = field(xr)
        oracle.composite_rays(n_alive, n_step, alive_r, t_r, s, c, nm, lr, ws_r, dep_r, img_r, nrm_r, 1e-2)
        raymarching.composite_rays(n_alive, n_step, alive, t, T(s, cuda), T(c, cuda), T(nm, cuda), l, ws, dep, img,
                                   nrm, 1e-2)
        assert np.array_equal(alive.cpu().numpy() >= 0, alive_r >= 0)
        alive_r = alive_r[alive_r >= 0]
        alive = alive[alive >= 0]
        step += n_step
    np.testing.assert_allclose(ws.cpu().numpy(), ws_r, rtol=RTOL, atol=1e-6)
    np.testing.assert_allclose(img.cpu().numpy(), img_r, rtol=RTOL, atol=1e-6)
    np.testing.assert_allclose(nrm.cpu().numpy(), nrm_r, rtol=RTOL, atol=1e-6)
    np.testing.assert_allclose(dep.cpu().numpy(), dep_r, rtol=RTOL, atol=1e-5)


def test_compact_budget_rounds_kernels_against_oracle(cuda, oracle):
    """The four entry points of the budget loop (C ABI Part 1b, second half: mi3d_infer_begin2, mi3d_march_rays_compact_ctl,
    mi3d_composite_rays_compact_ctl, mi3d_compact_alive_ctl2) driven directly, round by round, against the CPU oracle's
    march_rays / composite_rays (raymarching.cu:906-1115 restated): the plan in the control block, the slab packing (one
    slab per ray, slabs tile [0, rows) exactly), every slab row bit-exact, the accumulators, the death rule, the
    order-preserving compaction, the restart point (t_next = the march's own t, which the reference's running sum of
    deltas[:, 1] reproduces to float rounding) and the step cap (the only coverage used to be the
    end-to-end render comparison)."""
    import raymarching
    rng = np.random.default_rng(31)
    N, H, max_steps, budget = 600, 128, 96, 2500
    o, d = make_rays(rng, N)
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    nears, fars = oracle.near_far_from_aabb(o, d, aabb)
    bits = sphere_bitfield(oracle, 1, H, 0.6)

    def field(x):
        s = (6 * np.exp(-np.sum(x * x, 1) / 0.08)).astype(np.float32)
        c = (0.5 + 0.5 * np.sin(x * 7)).astype(np.float32)
        nrm = (x / (np.linalg.norm(x, axis=1, keepdims=True) + 1e-6)).astype(np.float32)
        return s, c, nrm

    ws_r, dep_r = np.zeros(N, np.float32), np.zeros(N, np.float32)
    img_r, nrm_r = np.zeros((N, 3), np.float32), np.zeros((N, 3), np.float32)
    alive_r, t_r = np.arange(N, dtype=np.int32), nears.copy()
    ws, dep = torch.zeros(N, device=cuda), torch.zeros(N, device=cuda)
    img, nrm = torch.zeros(N, 3, device=cuda), torch.zeros(N, 3, device=cuda)
    t = T(nears, cuda).clone()
    rows_cap = max(budget, N)
    xyzs, dirs = torch.zeros(rows_cap, 3, device=cuda), torch.zeros(rows_cap, 3, device=cuda)
    deltas = torch.zeros(rows_cap, 2, device=cuda)
    slab, t_next = torch.zeros(N, 2, dtype=torch.int32, device=cuda), torch.zeros(N, device=cuda)
    ctl, alive = raymarching.infer_begin2(N, cuda, budget, 1, max_steps)
    spare = torch.empty_like(alive)
    assert ctl.numel() == 16
    od, dd, bd, fd = T(o, cuda), T(d, cuda), T(bits, cuda), T(fars, cuda)
    done, rounds = 0, 0
    while True:
        st = ctl.tolist()
        n_alive, n_step = st[0], st[1]
        assert n_alive == alive_r.shape[0] and st[3] == done and st[4] == rounds and st[8] == 0
        if n_alive == 0:
            break
        assert n_step == max(1, min(budget // n_alive, max_steps, max_steps - done)), (st, done)   # the plan, capped
        raymarching.march_rays_compact_ctl(ctl, N, alive, t, od, dd, 1.0, bd, 1, H, fd, xyzs, dirs, deltas, slab, t_next,
                                           None, 0, max_steps)
        rows = int(ctl[2])
        xr, dr, lr = oracle.march_rays(n_alive, n_step, alive_r, t_r, o, d, 1.0, bits, 1, H, nears, fars, align=-1,
                                       max_steps=max_steps)
        cnt_r = (lr[:, 0].reshape(n_alive, n_step) != 0).sum(1)
        sl = slab[:n_alive].cpu().numpy()
        assert np.array_equal(sl[:, 1], cnt_r) and rows == int(cnt_r.sum())
        order = np.argsort(sl[:, 0], kind="stable")
        nz = order[sl[order, 1] > 0]
        assert np.array_equal(sl[nz, 0], np.concatenate([[0], np.cumsum(sl[nz, 1])[:-1]]))      # slabs tile [0, rows)
        xg, dg, lg = xyzs[:rows].cpu().numpy(), dirs[:rows].cpu().numpy(), deltas[:rows].cpu().numpy()
        gather = np.concatenate([np.arange(i * n_step, i * n_step + c) for i, c in enumerate(cnt_r)]).astype(np.int64) \
            if rows else np.zeros(0, np.int64)
        place = np.concatenate([np.arange(sl[i, 0], sl[i, 0] + c) for i, c in enumerate(cnt_r)]).astype(np.int64) \
            if rows else np.zeros(0, np.int64)
        assert np.array_equal(xg[place], xr[gather]) and np.array_equal(dg[place], dr[gather])   # every slab row, bit-exact
        assert np.array_equal(lg[place], lr[gather])
        s_r, c_r, n_r = field(xr)
        s_g, c_g, n_g = field(xg)
        oracle.composite_rays(n_alive, n_step, alive_r, t_r, s_r, c_r, n_r, lr, ws_r, dep_r, img_r, nrm_r, 1e-2)
        pad = lambda a, w: T(np.concatenate([a, np.zeros((rows_cap - rows,) + a.shape[1:], np.float32)]), cuda)  # noqa: E731
        raymarching.composite_rays_compact_ctl(ctl, N, alive, t, slab, t_next, pad(s_g, 1), pad(c_g, 3), pad(n_g, 3), deltas,
                                               ws, dep, img, nrm, 1e-2)
        a_g = alive[:n_alive].cpu().numpy()
        assert np.array_equal(a_g >= 0, alive_r >= 0)                                            # the death rule
        np.testing.assert_allclose(ws.cpu().numpy(), ws_r, rtol=RTOL, atol=1e-6)     # (__expf on the device, expf in the oracle)
        np.testing.assert_allclose(img.cpu().numpy(), img_r, rtol=RTOL, atol=1e-6)
        np.testing.assert_allclose(nrm.cpu().numpy(), nrm_r, rtol=RTOL, atol=1e-6)
        np.testing.assert_allclose(dep.cpu().numpy(), dep_r, rtol=RTOL, atol=1e-5)
        keep = alive_r >= 0
        # the restart: the march's own t for a survivor; the reference's running sum agrees to float rounding
        t_g = t.cpu().numpy()
        np.testing.assert_allclose(t_g[alive_r[keep]], t_r[alive_r[keep]], rtol=0, atol=2e-5)
        assert np.array_equal(t_g[alive_r[keep]], t_next[:n_alive].cpu().numpy()[keep])
        t_r[alive_r[keep]] = t_g[alive_r[keep]]     # (both sides start the next round from the same t)
        raymarching.compact_alive_ctl2(ctl, alive, spare, N, max_steps)
        alive, spare = spare, alive
        alive_r = alive_r[keep]
        done += n_step
        rounds += 1
        if done >= max_steps:
            alive_r = alive_r[:0]
        assert np.array_equal(alive[:alive_r.shape[0]].cpu().numpy(), alive_r)                   # compaction keeps the order
    assert rounds >= 3 and done <= max_steps + 0 or alive_r.shape[0] == 0
    assert float(ws_r.max()) > 0.5

    # a sample buffer smaller than the round needs: the rows are COUNTED in ctl[8], not dropped silently
    ctl, alive = raymarching.infer_begin2(N, cuda, budget, 1, max_steps)
    small = 64
    raymarching.march_rays_compact_ctl(ctl, N, alive, T(nears, cuda).clone(), od, dd, 1.0, bd, 1, H, fd, xyzs[:small],
                                       dirs[:small], deltas[:small], slab, t_next, None, 0, max_steps)
    st = ctl.tolist()
    assert st[8] > 0 and st[8] + int(slab[:, 1].sum()) == st[2]


def test_rejects_bad_inputs(cuda):
    import raymarching
    from mi3d._lib import Mi3dError
    with pytest.raises(Mi3dError):
        raymarching.composite_rays_train(torch.ones(4, device=cuda), torch.ones(5, 3, device=cuda),
                                         torch.ones(4, 2, device=cuda), torch.zeros(1, 3, dtype=torch.int32,
                                                                                   device=cuda))
    with pytest.raises(Mi3dError):
        raymarching.march_rays_train(torch.zeros(4, 3, device=cuda), torch.ones(4, 3, device=cuda), 1.0,
                                     torch.zeros(16, dtype=torch.uint8, device=cuda), 1, 128,
                                     torch.zeros(4, device=cuda), torch.ones(4, device=cuda))
