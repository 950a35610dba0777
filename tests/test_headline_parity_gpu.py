"""GPU: the mode bench.py's headline runs, at the size it runs it, against the reference's own Python.

VERDICT round 3, "weak" 1-3: the headline is torch.autocast + binary16 feature / gradient planes + k_bin_emit<true> + the
sliced binned scatter over 10.9 M samples x 13 stencil points - and that combination was compared with the reference
route only at 32 x 32 rays.  Here, on BASELINE config 2's own shape (128 x 128 rays, max_steps 1024):

  * C2 DENSE under autocast, the reference's two-backward schedule (a gradient injected at the image first - what
    latents.backward does, nerf/sd.py:171: stencil point 0 only - then the loss-scaled regularisers, nerf/utils.py:983:
    all 13 points), product fast route vs the reference's NeRFNetwork + NeRFRenderer.run_cuda (nerf/renderer.py:481-583,
    nerf/network_tcnn.py:94-170; staged verbatim under oracle/_ref/py) running UNCHANGED on the drop-in packages:
    outputs at binary16 resolution, hash-table gradient by max-norm, cosine and per-level norms;
  * the same with the pruned occupancy of SURVEY 8(d);
  * C2 dense in fp32 (no autocast): outputs 1e-4 (BASELINE.json), gradients at the tolerance the last test MEASURES;
  * the 2e-3 x max gradient tolerance of tests/test_reference_glue_gpu.py measured instead of asserted: the reference
    route in fp32 and the product in fp32 against an fp64 evaluation of the same function on the same samples.

The smoothness jitter (`torch.randn_like(xyzs)`, renderer.py:522) is drawn per ROW and the marching waves' slabs arrive
in a different order in every run, so at these sizes the same sample would get a different jitter in each run.  The
tests replace `torch.randn_like` - for [m, 3] tensors, inside the render - by a fixed pseudo-random function of the
sample's POSITION (bit-identical in both runs: same march kernel, same rays, same per-ray noise): every sample keeps its
jitter whatever row it lands in, and loss_smooth - the term that makes the backward a 13-point pass - compares exactly.

Every measured figure is also written to gpurun_out/headline_parity.json."""
import contextlib
import json
import math
import os

import numpy as np
import pytest
import torch

from conftest import position_jitter, record_scatter_workspaces

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPORT = {}
_LAST_PAIR = {}
# autocast output bounds.  Round 4 asserted rtol 2e-2 / atol 2e-3 without recording what was measured; round 5 recorded it
# (image max |diff| 4.0e-5 dense / 8.4e-5 pruned, profiles/headline_parity_r05.json) and asserted 2e-4.  Round 6 found what
# that difference WAS: under autocast the reference's torch.sigmoid(h[..., 1:]) returns binary16 (network_tcnn.py:110), the
# product kept its albedo in fp32.  With the albedo rounded as the reference rounds it (mi3d/field_ops.py:_head_forward)
# the two routes differ by 6.8e-6 (dense) / 1.4e-5 (pruned) on the image, 2.4e-6 on the depth, 2.4e-7 on the weights
# (profiles/headline_parity_r06.json) - the order of the fp32 accumulations.  Bounds: ~3 x the largest measured use.
AUTOCAST_RTOL, AUTOCAST_ATOL = 3e-5, 3e-5
# ... and against the CPU oracle chain in half_mode (oracle/field_ref.c: autocast restated for the nn.Linear stack and the
# sigmoid), 256 rays, no libmi3d.so on that side.  Measured (same file, `cpu_oracle_subset_half_mode`): image 4.2e-6 (dense)
# / 8.6e-6 (pruned), depth 1.4e-5 of 1.92, weights 1.5e-6; the bound is ~3 x the largest measured use, and the figures are
# written to the report before anything is asserted.
ORACLE_HALF_RTOL, ORACLE_HALF_ATOL = 2e-5, 2e-5


@pytest.fixture(scope="module")
def ref():
    from oracle import ref_import
    if not ref_import.available():
        pytest.skip("reference sources not staged (oracle/build_ref.py runs in the build container)")
    ref_import.install()
    yield ref_import
    if REPORT:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "headline_parity.json"), "w") as f:
            json.dump(REPORT, f, indent=1)


def _pair(ref, cuda, **opt_kw):
    from mi3d.network import NeRFNetwork
    kw = dict(cuda_ray=True, lambda_smooth=1.0, max_steps=1024)
    kw.update(opt_kw)
    opt = ref.default_opt(**kw)
    torch.manual_seed(0)
    ours = NeRFNetwork(opt).to(cuda)
    with torch.no_grad():
        ours.encoder.params.uniform_(-0.3, 0.3)
    theirs = ref.reference_network(opt, "dropin").to(cuda)
    theirs.load_state_dict(ours.state_dict())
    return theirs, ours, opt


def _headline_step(model, rays, seed, max_steps, autocast, scale, inject):
    """One NeRF-side training step of the reference schedule without the diffusion model: render under autocast;
    backward #1 = a fixed gradient on the image (the SDS injection; reaches sigma / albedo of stencil point 0 only);
    backward #2 = scale x regularisers (orientation + smoothness + opacity + entropy: all 13 points)."""
    from mi3d import sds_step
    ro, rd, ds = rays
    model.train()
    model.zero_grad()
    torch.manual_seed(seed)
    opt = sds_step.make_opt(max_steps=max_steps)
    with position_jitter():
        with torch.autocast("cuda", dtype=torch.float16, enabled=autocast):
            out = model.render(ro, rd, depth_scale=ds, bg_color=torch.full((3,), 0.7, device=ro.device), perturb=True,
                               ambient_ratio=1.0, shading="albedo", force_all_rays=True, dt_gamma=0,
                               max_steps=max_steps)
            loss = sds_step.regularisers(opt, out, out["weights_sum"].reshape(1, 1, -1, 1))
        out["image"].backward(inject.view_as(out["image"]), retain_graph=True)
        (scale * loss).backward()
    out["loss"] = loss.detach()
    return out


def _grad_report(name, theirs, ours):
    """max-norm error relative to the largest gradient, cosine, per-level norm ratios of the hash-table gradient, and
    the same two scalars for every MLP tensor."""
    g_ref, g = theirs.encoder.params.grad.double(), ours.encoder.params.grad.double()
    assert torch.isfinite(g_ref).all() and torch.isfinite(g).all()
    rep = {"table_max": float(g_ref.abs().max()),
           "table_max_err_rel": float((g_ref - g).abs().max() / g_ref.abs().max()),
           "table_cosine": float((g_ref * g).sum() / (g_ref.norm() * g.norm())),
           "table_norm_ratio": float(g.norm() / g_ref.norm())}
    offs = [int(o) * 2 for o in ours.encoder.offsets]
    ratios, cosines = [], []
    for l in range(len(offs) - 1):
        a, b = g_ref[offs[l]:offs[l + 1]], g[offs[l]:offs[l + 1]]
        ratios.append(float(b.norm() / a.norm()))
        cosines.append(float((a * b).sum() / (a.norm() * b.norm())))
    rep["level_norm_ratio"], rep["level_cosine"] = ratios, cosines
    mlp = {}
    for (n, p), q in zip(list(theirs.named_parameters())[1:], list(ours.parameters())[1:]):
        a, b = p.grad.double(), q.grad.double()
        mlp[n] = {"max_err_rel": float((a - b).abs().max() / a.abs().max()),
                  "cosine": float((a * b).sum() / (a.norm() * b.norm()))}
    rep["mlp"] = mlp
    REPORT[name] = rep
    return rep


def _outputs_close(a, b, rtol, atol, name=None):
    """image / depth / weights_sum of the product (b) against the reference route (a); with `name` the measured errors go
    into the report: max |diff|, max |diff| / |reference| over the elements above 1e-3 of the largest one, and the
    largest value of |diff| / (atol + rtol |reference|) - the share of the asserted bound that is used."""
    for k in ("image", "depth", "weights_sum"):
        x, y = a[k].detach().double().cpu().numpy(), b[k].detach().double().cpu().numpy()
        if name is not None:
            d = np.abs(y - x)
            big = np.abs(x) > 1e-3 * np.abs(x).max()
            REPORT.setdefault(name, {}).setdefault("outputs", {})[k] = {
                "max_abs_err": float(d.max()), "max_rel_err": float((d[big] / np.abs(x[big])).max()),
                "max_abs_value": float(np.abs(x).max()), "bound_used": float((d / (atol + rtol * np.abs(x))).max()),
                "asserted": {"rtol": rtol, "atol": atol}}
        np.testing.assert_allclose(y, x, rtol=rtol, atol=atol, err_msg=k)
    assert torch.equal(a["mask"], b["mask"])


def _oracle_subset(name, ours, rays, seed, out, n_rays=256, max_steps=1024, bg=0.7, tol=1e-4, atol=None, half_mode=False):
    """The ONE check at this size that does not run libmi3d.so on both sides: `n_rays` evenly
    spaced rays of the same view through the CPU oracle chain - oracle.march_rays_train -> field_forward (hash grid, MLP,
    head in C / numpy: oracle/raymarching_ref.c, hashgrid_ref.c, field_ref.c) -> composite_rays_train, with the march
    jitter the product drew (same seed, same draw order: light direction, then torch.rand(N)) - against the product's
    image / depth / weights_sum of those rays (renderer.py:481-583, raymarching.cu:311-577).  Rays are independent, so a
    subset costs seconds where the whole view would cost minutes.

    `half_mode` (round 6): the same chain with oracle/field_ref.c's restatement of what
    torch.autocast(float16) does to the reference's nn.Linear stack (network_tcnn.py:13-32 under nerf/utils.py:977: input,
    weights and bias rounded to binary16, fp32 accumulation, every layer's output rounded to binary16; the head in fp32) -
    the headline MODE at the headline SIZE against a side that never loads libmi3d.so."""
    from oracle import oracle as O
    ro, rd, ds = rays
    dev = ro.device
    N = ro.view(-1, 3).shape[0]
    torch.manual_seed(seed)
    torch.randn(3, device=dev)                      # run_cuda's light direction (drawn first, unused under 'albedo')
    noises = torch.rand(N, device=dev)              # march_rays_train's jitter
    idx = torch.linspace(0, N - 1, n_rays).round().long().unique()
    cfg = O.GridConfig()
    fp = O.FieldParams(cfg)
    fp.params = ours.encoder.params.detach().cpu().numpy()
    fp.W = [l.weight.detach().cpu().numpy() for l in ours.sigma_net.net]
    fp.B = [l.bias.detach().cpu().numpy() for l in ours.sigma_net.net]
    o = ro.view(-1, 3)[idx.to(dev)].cpu().numpy()
    d = rd.view(-1, 3)[idx.to(dev)].cpu().numpy()
    nears, fars = O.near_far_from_aabb(o, d, ours.aabb_train.cpu().numpy())
    xyzs, dirs, deltas, rr = O.march_rays_train(o, d, float(ours.bound), ours.density_bitfield.cpu().numpy(), ours.cascade,
                                                ours.grid_size, nears, fars, noises=noises[idx.to(dev)].cpu().numpy(),
                                                align=128, max_steps=max_steps)
    sig, col, _ = O.field_forward(xyzs, dirs, fp, half_mode=half_mode)
    ws, dep, img = O.composite_rays_train(sig, col, deltas, rr)
    img = img + (1 - ws)[:, None] * np.float32(bg)
    dep = (dep + (1 - ws) * np.float32(ours.opt.max_depth)) * ds.view(-1)[idx.to(dev)].cpu().numpy()
    got = {"image": out["image"].detach().view(-1, 3)[idx.to(dev)].cpu().numpy(),
           "depth": out["depth"].detach().view(-1)[idx.to(dev)].cpu().numpy(),
           "weights_sum": out["weights_sum"].detach().view(-1)[idx.to(dev)].cpu().numpy()}
    want = {"image": img, "depth": dep, "weights_sum": ws}
    atol = tol * 0.1 if atol is None else atol
    rep = {"rays": int(idx.numel()), "samples": int(rr[:, 2].sum()), "tolerance": tol, "atol": atol,
           "half_mode": bool(half_mode)}
    for k in got:      # (measured first, asserted after: a failing run still leaves its figures in the report)
        x, y = want[k].astype(np.float64), got[k].astype(np.float64)
        dlt = np.abs(y - x)
        big = np.abs(x) > 1e-3 * np.abs(x).max()
        rep[k] = {"max_abs_err": float(dlt.max()), "max_rel_err": float((dlt[big] / np.abs(x[big])).max()),
                  "max_abs_value": float(np.abs(x).max()), "bound_used": float((dlt / (atol + tol * np.abs(x))).max())}
    REPORT.setdefault(name, {})["cpu_oracle_subset" + ("_half_mode" if half_mode else "")] = rep
    for k in got:
        np.testing.assert_allclose(got[k], want[k], rtol=tol, atol=atol, err_msg="oracle subset: " + k)
    return rep


def _run_pair(ref, cuda, name, bitfield, autocast, scale=4.0, seed=31, defer=True):
    """`defer`: the product parks the first pass's point-0 planes and scatters them with the second pass (the default,
    grid_ops.DEFER_POINT0: ONE scatter call) or scatters every pass on its own (round 3's behaviour: TWO calls)."""
    from mi3d import grid_ops, rays as R, sds_step
    theirs, ours, opt = _pair(ref, cuda)
    for m in (theirs, ours):
        sds_step.set_bitfield(m, bitfield)
    rays = R.view_rays(128, 128, device=cuda)
    g = torch.Generator(device="cpu").manual_seed(9)
    inject = (torch.randn(128 * 128, 3, generator=g) * 2e-3).to(cuda)   # the size of an SDS gradient on a 128 x 128 image
    a = _headline_step(theirs, rays, seed, 1024, autocast, scale, inject)
    torch.cuda.empty_cache()
    was, grid_ops.DEFER_POINT0 = grid_ops.DEFER_POINT0, defer
    try:
        with record_scatter_workspaces() as arenas:
            b = _headline_step(ours, rays, seed, 1024, autocast, scale, inject)
    finally:
        grid_ops.DEFER_POINT0 = was
    n = int(ours.step_counter[0, 0])
    assert n == int(theirs.step_counter[0, 0])
    # both backward passes went through the records: as one call (point 0 of the first pass riding along) or as two
    assert len(arenas) == (1 if defer else 2) and all(x > 0 for x in arenas), arenas
    assert not ours.encoder.params.__dict__.get("_mi3d_pending")
    rep = _grad_report(name, theirs, ours)
    rep["samples"], rep["arena_bytes"] = n, arenas
    for k in ("loss_orient", "loss_smooth", "loss"):
        rep[k] = [float(a[k]), float(b[k])]
    _LAST_PAIR.clear()
    _LAST_PAIR.update(ours=ours, rays=rays, seed=seed)     # (for the CPU-oracle subset of the fp32 test)
    return a, b, rep, n


def test_c2_dense_autocast_headline_mode(ref, cuda):
    """(a) the headline: C2 dense, autocast, two-backward schedule, binary16 planes, sliced binned scatter."""
    a, b, rep, n = _run_pair(ref, cuda, "c2_dense_autocast", "dense", True)
    assert 10_000_000 < n < 12_000_000
    # (the bound was asserted at binary16 resolution, 2e-2, and never recorded - see AUTOCAST_RTOL)
    _outputs_close(a, b, rtol=AUTOCAST_RTOL, atol=AUTOCAST_ATOL, name="c2_dense_autocast")
    # 256 rays of this view through the CPU oracle chain in its autocast restatement: no libmi3d.so on that side
    _oracle_subset("c2_dense_autocast", _LAST_PAIR["ours"], _LAST_PAIR["rays"], _LAST_PAIR["seed"], b,
                   tol=ORACLE_HALF_RTOL, atol=ORACLE_HALF_ATOL, half_mode=True)
    _LAST_PAIR.clear()
    assert abs(rep["loss_orient"][1] - rep["loss_orient"][0]) <= 2e-2 * abs(rep["loss_orient"][0])
    assert abs(rep["loss_smooth"][1] - rep["loss_smooth"][0]) <= 2e-2 * abs(rep["loss_smooth"][0])
    # measured (profiles/headline_parity_r04.json): 9.3e-4 x max, cosine 1.0000, per-level cosines >= 0.99975, per-level
    # norms within 2e-4; the bounds leave a factor of ~5
    assert rep["table_max_err_rel"] <= 5e-3, rep
    assert rep["table_cosine"] >= 0.9999, rep
    assert min(rep["level_cosine"]) >= 0.998, rep["level_cosine"]
    assert all(abs(r - 1.0) <= 2e-3 for r in rep["level_norm_ratio"]), rep["level_norm_ratio"]
    for k, v in rep["mlp"].items():
        assert v["max_err_rel"] <= 5e-3 and v["cosine"] >= 0.9999, (k, v)


@pytest.mark.parametrize("defer", [True, False])
def test_c2_pruned_autocast_headline_mode(ref, cuda, defer):
    """(b) the same with the pruned occupancy (sphere 0.3: ~2.3 M samples; one scatter slice), with the first pass's
    point-0 planes deferred into the second pass's scatter and with every pass scattered on its own."""
    a, b, rep, n = _run_pair(ref, cuda, "c2_pruned_autocast" + ("" if defer else "_two_scatters"), 0.3, True, defer=defer)
    assert 1_500_000 < n < 3_500_000
    _outputs_close(a, b, rtol=AUTOCAST_RTOL, atol=AUTOCAST_ATOL,
                   name="c2_pruned_autocast" + ("" if defer else "_two_scatters"))
    if defer:   # (the forward does not depend on `defer`: once)
        _oracle_subset("c2_pruned_autocast", _LAST_PAIR["ours"], _LAST_PAIR["rays"], _LAST_PAIR["seed"], b,
                       tol=ORACLE_HALF_RTOL, atol=ORACLE_HALF_ATOL, half_mode=True)
    _LAST_PAIR.clear()
    assert rep["table_max_err_rel"] <= 5e-3, rep
    assert rep["table_cosine"] >= 0.9999, rep
    assert min(rep["level_cosine"]) >= 0.995, rep["level_cosine"]
    assert all(abs(r - 1.0) <= 2e-3 for r in rep["level_norm_ratio"]), rep["level_norm_ratio"]


def test_c2_dense_fp32_outputs_1e4(ref, cuda):
    """(c) C2 dense without autocast (fp32 planes, k_bin_emit<false>, exact fp32 MFMA): rendered outputs within
    BASELINE.json's 1e-4, both normal regularisers 2e-4, gradients by the same three measures."""
    a, b, rep, n = _run_pair(ref, cuda, "c2_dense_fp32", "dense", False, scale=1.0)
    assert 10_000_000 < n < 12_000_000
    _outputs_close(a, b, rtol=1e-4, atol=1e-6, name="c2_dense_fp32")
    # 256 rays through the CPU oracle chain: no libmi3d.so on that side
    _oracle_subset("c2_dense_fp32", _LAST_PAIR["ours"], _LAST_PAIR["rays"], _LAST_PAIR["seed"], b)
    _LAST_PAIR.clear()
    assert abs(rep["loss_orient"][1] - rep["loss_orient"][0]) <= 2e-4 * abs(rep["loss_orient"][0])
    assert abs(rep["loss_smooth"][1] - rep["loss_smooth"][0]) <= 2e-4 * abs(rep["loss_smooth"][0])
    assert rep["table_max_err_rel"] <= 1e-3, rep          # measured 2.8e-4 (both routes sit 8e-4 from fp64, test below)
    assert rep["table_cosine"] >= 0.999999, rep
    assert all(abs(r - 1.0) <= 1e-4 for r in rep["level_norm_ratio"]), rep["level_norm_ratio"]


# ---------------------------------------------------------------------------------------------------------------------
# (e) the gradient tolerance, measured: an fp64 evaluation of the same function on the same samples

class _Grid64(torch.nn.Module):
    """The hash grid of oracle/field_torch.py with fp64 parameters and sums: cell indices and trilinear weights are the
    fp32 numbers the kernels compute (they are part of the function, not of its rounding), everything after them is
    binary64."""

    def __init__(self, cfg, params):
        super().__init__()
        self.params = torch.nn.Parameter(params.detach().double().clone())
        self.cfg = cfg

    def forward(self, x01):
        from oracle import field_torch as FT
        cfg, table, feats = self.cfg, self.params.view(-1, 2), []
        x = x01.float()
        for l in range(cfg.n_levels):
            scale, res = float(cfg.scales[l]), int(cfg.resolutions[l])
            off, hs = int(cfg.offsets[l]), int(cfg.offsets[l + 1] - cfg.offsets[l])
            pos = (x.double() * scale + 0.5).float()       # fmaf(scale, x, 0.5): one rounding
            fl = torch.floor(pos)
            g, w = fl.to(torch.int64), (pos - fl)
            stride, n_dims = 1, 0
            for _ in range(3):
                if stride > hs:
                    break
                stride *= res
                n_dims += 1
            hashed = hs < stride
            f = 0
            for k in range(8):
                wk, q = 1, []
                for d in range(3):
                    if (k >> d) & 1:
                        wk = wk * w[:, d]
                        q.append((g[:, d] + 1) & 0xFFFFFFFF)
                    else:
                        wk = wk * (1 - w[:, d])
                        q.append(g[:, d] & 0xFFFFFFFF)
                if hashed:
                    idx = q[0] ^ ((q[1] * FT.PRIME_Y) & 0xFFFFFFFF) ^ ((q[2] * FT.PRIME_Z) & 0xFFFFFFFF)
                else:
                    idx, s = 0, 1
                    for d in range(n_dims):
                        idx = idx + q[d] * s
                        s *= res
                    idx = idx & 0xFFFFFFFF
                f = f + wk.double()[:, None] * table[idx % hs + off]   # fp32 weight (as computed), fp64 product and sum
            feats.append(f)
        return torch.cat(feats, -1)


def _truth_fp64(model, xyzs, dirs, deltas, rays, bg, eps=1e-2):
    """The training branch of run_cuda + common_forward / normal (renderer.py:503-524, network_tcnn.py:102-138) and
    composite_rays_train's forward (raymarching.cu:500-600) in binary64 autograd on the given samples.  Returns the loss
    of _render_loss and leaves fp64 gradients on the returned parameter list (table, W1, b1, ...)."""
    from oracle import oracle as O
    from oracle.field_torch import safe_normalize
    enc = model.encoder
    cfg = O.GridConfig(n_levels=enc.cfg["n_levels"], per_level_scale=enc.cfg["per_level_scale"],
                       base_resolution=enc.cfg["base_resolution"], log2_hashmap_size=enc.cfg["log2_hashmap_size"])
    grid = _Grid64(cfg, enc.params).to(xyzs.device)
    Ws = [torch.nn.Parameter(l.weight.detach().double().clone()) for l in model.sigma_net.net]
    Bs = [torch.nn.Parameter(l.bias.detach().double().clone()) for l in model.sigma_net.net]
    bound, bd, br = float(model.bound), float(model.opt.blob_density), float(model.opt.blob_radius)

    def common(x):                       # x fp32 positions (the points the kernels see), arithmetic in fp64
        h = grid(((x + bound) / (2 * bound)))
        for i, (W, B) in enumerate(zip(Ws, Bs)):
            h = torch.nn.functional.linear(h, W, B)
            if i != len(Ws) - 1:
                h = torch.relu(h)
        xd = x.double()
        blob = bd * torch.exp(-(xd ** 2).sum(-1) / (2 * br ** 2))
        sigma = torch.exp(h[:, 0] + blob)            # (h + blob stays far below trunc_exp's clamp at 15 here)
        return sigma, torch.sigmoid(h[:, 1:])

    def normal(x):
        e = torch.eye(3, device=x.device, dtype=torch.float32) * np.float32(eps)
        s = [common((x + sgn * e[d]).clamp(-bound, bound))[0] for d in range(3) for sgn in (1.0, -1.0)]
        g = torch.stack([0.5 * (s[0] - s[1]) / eps, 0.5 * (s[2] - s[3]) / eps, 0.5 * (s[4] - s[5]) / eps], -1)
        return torch.nan_to_num(safe_normalize(-g))

    sigma, albedo = common(xyzs)
    normals = normal(xyzs)
    with position_jitter():
        x2 = xyzs + torch.randn_like(xyzs) * 1e-2
    normals2 = normal(x2)
    # composite (per ray: T before sample i; a ray uses its samples while T stayed >= 1e-4).  Rays without samples keep
    # the background; the <= 128 zero rows the march pads behind the last slab (raymarching.py:176-178 returns xyzs[:m +
    # pad]) are no ray's samples but DO enter the two normal losses' means - in the reference as well.
    N = rays.shape[0]
    r = rays[rays[:, 2] > 0].long()
    r = r[torch.argsort(r[:, 1])]
    ridx, off, cnt = r[:, 0], r[:, 1], r[:, 2]
    assert int(off[0]) == 0 and torch.equal(off[1:], torch.cumsum(cnt, 0)[:-1])   # the slabs tile [0, m)
    m = int(cnt.sum())
    ray_of = torch.repeat_interleave(ridx, cnt)
    first = torch.repeat_interleave(off, cnt)
    tau = sigma[:m] * deltas[:m, 0].double()
    cs = torch.cumsum(tau, 0)
    excl = cs - tau
    T = torch.exp(-(excl - excl[first]))
    alpha = 1 - torch.exp(-tau)
    used = (T >= 1e-4) | (torch.arange(m, device=xyzs.device) == first)
    wgt = alpha * T * used
    ws = torch.zeros(N, dtype=torch.float64, device=xyzs.device).index_add(0, ray_of, wgt)
    img = torch.zeros(N, 3, dtype=torch.float64, device=xyzs.device).index_add(0, ray_of, wgt[:, None] * albedo[:m])
    img = img + (1 - ws)[:, None] * bg.double()
    w = 1 - torch.exp(-sigma)
    loss_orient = (w.detach() * (normals * dirs.double()).sum(-1).clamp(min=0) ** 2).mean()
    loss_smooth = (normals - normals2).abs().mean()
    loss = (img ** 2).mean() + (ws ** 2).mean() + 0.1 * loss_orient + loss_smooth
    loss.backward()
    return loss, [grid.params] + [t for pair in zip(Ws, Bs) for t in pair]


def _render_loss(model, rays, seed, max_steps):
    ro, rd, ds = rays
    model.train()
    model.zero_grad()
    torch.manual_seed(seed)
    with position_jitter():
        out = model.render(ro, rd, depth_scale=ds, bg_color=torch.full((3,), 0.7, device=ro.device), perturb=True,
                           ambient_ratio=1.0, shading="albedo", force_all_rays=True, dt_gamma=0, max_steps=max_steps)
        loss = ((out["image"] ** 2).mean() + (out["weights_sum"] ** 2).mean() + 0.1 * out["loss_orient"]
                + out["loss_smooth"])
    loss.backward()
    return float(loss)


def test_gradient_tolerance_is_measured_against_fp64(ref, cuda):
    """(e) 32 x 32 rays x 256 steps, fp32, all four loss terms.  The reference route (13 encoder passes, nn.Linear,
    float atomics) and the product (one stencil gather, MFMA MLP in exact fp32, 64-bit fixed-point sums) are both
    compared with a binary64 evaluation of the same function on the same samples.  What the two fp32 routes differ by
    from EACH OTHER (the 2e-3 x max bound of tests/test_reference_glue_gpu.py) is then read against how far each is
    from the truth: the product must be no further from fp64 than 2 x the reference route is (+ 1e-5 x max)."""
    import raymarching
    from mi3d import rays as R, sds_step
    theirs, ours, opt = _pair(ref, cuda, max_steps=256)
    for m in (theirs, ours):
        sds_step.set_bitfield(m, 0.6)
    rays = R.view_rays(32, 32, device=cuda)
    la = _render_loss(theirs, rays, 77, 256)
    lb = _render_loss(ours, rays, 77, 256)
    # the same samples, marched once more with the same seed (light_d is drawn first, as run_cuda does)
    ro, rd = rays[0].view(-1, 3).contiguous(), rays[1].view(-1, 3).contiguous()
    torch.manual_seed(77)
    nears, fars = raymarching.near_far_from_aabb(ro, rd, ours.aabb_train)
    torch.randn(3, device=cuda)
    counter = torch.zeros(2, dtype=torch.int32, device=cuda)
    xyzs, dirs, deltas, rr = raymarching.march_rays_train(ro, rd, ours.bound, ours.density_bitfield, ours.cascade,
                                                          ours.grid_size, nears, fars, counter, -1, True, 128, True, 0,
                                                          256)
    assert int(counter[0]) == int(ours.step_counter[0, 0])
    loss64, g64 = _truth_fp64(ours, xyzs, dirs, deltas, rr, torch.full((3,), 0.7, device=cuda))
    assert abs(la - float(loss64)) <= 1e-4 * abs(float(loss64)) and abs(lb - float(loss64)) <= 1e-4 * abs(float(loss64))
    rep = {"loss": [la, lb, float(loss64)], "tensors": {}}
    worst_ref = worst_ours = worst_pair = 0.0
    for (name, p), q, t in zip(theirs.named_parameters(), ours.parameters(), g64):
        truth = t.grad.view_as(p.grad)
        scale = float(truth.abs().max())
        e_ref = float((p.grad.double() - truth).abs().max()) / scale
        e_ours = float((q.grad.double() - truth).abs().max()) / scale
        e_pair = float((p.grad.double() - q.grad.double()).abs().max()) / scale
        rep["tensors"][name] = {"max": scale, "reference_route_vs_fp64": e_ref, "product_vs_fp64": e_ours,
                                "product_vs_reference_route": e_pair}
        worst_ref, worst_ours, worst_pair = max(worst_ref, e_ref), max(worst_ours, e_ours), max(worst_pair, e_pair)
        assert e_ours <= 2 * e_ref + 1e-5, (name, e_ours, e_ref)
    rep["worst"] = {"reference_route_vs_fp64": worst_ref, "product_vs_fp64": worst_ours,
                    "product_vs_reference_route": worst_pair}
    REPORT["gradient_tolerance_fp64"] = rep
    assert worst_pair <= 2e-3    # the bound the glue tests assert, now with its two halves on record


# ---------------------------------------------------------------------------------------------------------------------
# (d) the fused Adan kernel against the reference's own optimizer

def test_adan_kernel_against_reference_golden_trajectory(cuda):
    """csrc/optim.hip on the GPU against tests/golden/adan.npz - six steps of the reference's own optimizer.py
    (make_golden_adan.py: two groups, lr 5e-2 / 5e-3, one step large enough to clip)."""
    from mi3d.optim import Adan
    g = np.load(os.path.join(ROOT, "tests", "golden", "adan.npz"))
    p1 = torch.nn.Parameter(torch.from_numpy(g["table0"].copy()).to(cuda))
    p2 = torch.nn.Parameter(torch.from_numpy(g["w0"].copy()).to(cuda))
    opt = Adan([{"params": [p1], "lr": 5e-2}, {"params": [p2], "lr": 5e-3}], eps=1e-8, weight_decay=2e-5,
               max_grad_norm=5.0)
    for i in range(6):
        p1.grad, p2.grad = torch.from_numpy(g[f"g1_{i}"].copy()).to(cuda), torch.from_numpy(g[f"g2_{i}"].copy()).to(cuda)
        assert opt._fused_ok()                                   # the HIP kernel, not the torch-op path
        opt.step()
        np.testing.assert_allclose(p1.detach().cpu().numpy(), g[f"table_{i}"], rtol=2e-5, atol=2e-6, err_msg=f"step {i}")
        np.testing.assert_allclose(p2.detach().cpu().numpy(), g[f"w_{i}"], rtol=2e-5, atol=2e-6, err_msg=f"step {i}")


def test_adan_kernel_against_the_reference_optimizer_on_the_gpu(ref, cuda):
    """The reference's optimizer.py (staged verbatim, oracle/_ref/py/optimizer.py:23-249, foreach=False as main.py:132
    leaves it) stepping CUDA tensors next to the fused kernel: table-sized odd tensor + MLP-shaped ones, the reference's
    hyper-parameters, eight steps with clipping active and inactive."""
    import importlib.util
    from mi3d.optim import Adan
    path = os.path.join(ref.REFERENCE, "optimizer.py")
    spec = importlib.util.spec_from_file_location("ref_optimizer_gpu", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    torch.manual_seed(4)
    shapes = [(1_000_003,), (64, 32), (64,), (4, 64)]
    a_p = [torch.nn.Parameter(torch.randn(s, device=cuda) * 0.1) for s in shapes]
    b_p = [torch.nn.Parameter(p.detach().clone()) for p in a_p]
    groups = lambda ps: [{"params": ps[:1], "lr": 5e-2}, {"params": ps[1:], "lr": 5e-3}]
    a = mod.Adan(groups(a_p), eps=1e-8, weight_decay=2e-5, max_grad_norm=5.0, foreach=False)
    b = Adan(groups(b_p), eps=1e-8, weight_decay=2e-5, max_grad_norm=5.0)
    for it, s in enumerate((1e-4, 3e-3, 0.5, 1e-3, 20.0, 1e-2, 1e-5, 2.0)):
        for p, q in zip(a_p, b_p):
            gr = torch.randn_like(p) * s
            p.grad, q.grad = gr.clone(), gr.clone()
        assert b._fused_ok()
        a.step()
        b.step()
        for p, q in zip(a_p, b_p):
            assert torch.allclose(p, q, rtol=2e-5, atol=2e-6), (it, float((p - q).abs().max()))
    for p, q in zip(a_p, b_p):
        for k in ("exp_avg", "exp_avg_sq", "exp_avg_diff"):
            assert torch.allclose(a.state[p][k], b.state[q][k], rtol=5e-5, atol=1e-9), k


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE config 4's shape and the eval loop at BASELINE config 2's size, on the reference route

def test_c4_view_forward_render_matches_the_reference_route(ref, cuda):
    """One view of BASELINE config 4 - 256 x 256 rays, max_steps 2048, pruned occupancy (sphere 0.5: ~36 M samples x 7
    field evaluations), training-mode forward render, fp32 - through the reference's run_cuda on the drop-in packages and
    through the product: image / depth / weights 1e-4, the orientation loss 2e-4."""
    from mi3d import rays as R, sds_step
    theirs, ours, opt = _pair(ref, cuda, lambda_smooth=0.0, max_steps=2048)
    for m in (theirs, ours):
        sds_step.set_bitfield(m, 0.5)
    ro, rd, ds = R.view_rays(256, 256, device=cuda)
    outs = []
    for model in (theirs, ours):
        model.train()
        torch.manual_seed(13)
        with torch.no_grad():
            outs.append(model.render(ro, rd, depth_scale=ds, bg_color=torch.full((3,), 0.7, device=cuda), perturb=True,
                                     ambient_ratio=1.0, shading="albedo", force_all_rays=True, dt_gamma=0, max_steps=2048))
        torch.cuda.empty_cache()
    n = int(ours.step_counter[0, 0])
    assert n == int(theirs.step_counter[0, 0]) and 25_000_000 < n < 45_000_000
    _outputs_close(outs[0], outs[1], rtol=1e-4, atol=1e-6, name="c4_view_forward_fp32")
    lo = [float(o["loss_orient"]) for o in outs]
    assert abs(lo[1] - lo[0]) <= 2e-4 * abs(lo[0])
    REPORT.setdefault("c4_view_forward_fp32", {}).update(
        {"samples": n, "loss_orient": lo, "image_max_abs_diff": float((outs[0]["image"] - outs[1]["image"]).abs().max())})


def test_eval_loop_at_c2_size_matches_the_reference_route(ref, cuda):
    """The inference branch (renderer.py:526-551) at BASELINE config 2's ray count and step budget on a refreshed
    occupancy grid: the reference's host loop on the drop-in march_rays / composite_rays against the product's
    graph-replayed, device-driven loop - image, depth, weights 1e-4, normal map 2e-3 absolute."""
    from mi3d import rays as R
    theirs, ours, opt = _pair(ref, cuda, lambda_smooth=0.0, max_steps=1024)
    torch.manual_seed(3)
    ours.update_extra_state()
    theirs.density_grid.copy_(ours.density_grid)
    theirs.density_bitfield.copy_(ours.density_bitfield)
    theirs.mean_density = ours.mean_density
    ro, rd, ds = R.view_rays(128, 128, device=cuda)
    outs = []
    for m in (theirs, ours):
        m.eval()
        with torch.no_grad():
            torch.manual_seed(11)
            outs.append(m.render(ro, rd, depth_scale=ds, bg_color=torch.ones(3, device=cuda), perturb=False,
                                 ambient_ratio=1.0, shading="albedo", dt_gamma=0, max_steps=1024))
    a, b = outs
    assert float(a["weights_sum"].max()) > 0.5 and ours.infer_stats["graph_replays"] >= 1
    _outputs_close(a, b, rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(b["normal"].cpu().numpy(), a["normal"].cpu().numpy(), rtol=0, atol=2e-3)
    REPORT["eval_loop_c2_size"] = {"rounds_launched": ours.infer_stats["rounds_launched"],
                                   "host_reads": ours.infer_stats["host_reads"],
                                   "image_max_abs_diff": float((a["image"] - b["image"]).abs().max())}
