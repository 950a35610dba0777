"""GPU parity of the refine-stage point renderer (csrc/raster.hip, C ABI Part 7; mi3d.refine.render_point) against the
numpy oracle (oracle/raster_ref.py: refine_utils.py:306-333 with pytorch3d's rasterize_points / alpha_composite
restated - PARITY UNPINNED for those two, pytorch3d is absent).  Indices bit-exact, squared distances and composited
features to fp32 rounding, feature gradients against the oracle's closed form; at BASELINE config 5's size (512 x 512,
half a million points) through properties checked against a brute-force torch evaluation of sampled pixels."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _cloud(rng, P, spread=0.35):
    """points on a noisy sphere shell around the origin + a few degenerate ones"""
    d = rng.normal(size=(P, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    pts = (d * spread * (1 + 0.05 * rng.normal(size=(P, 1)))).astype(np.float32)
    return pts


def _camera(dev, radius=1.25):
    from mi3d import rays as R
    c2w = R.orbit_pose(radius, 80.0, 30.0, device=dev)[0]
    return torch.linalg.inv(c2w)


@pytest.mark.parametrize("H,W,K", [(64, 64, 8), (48, 80, 8), (40, 40, 1), (33, 70, 3)])
def test_rasterize_and_composite_match_the_oracle(cuda, H, W, K):
    from mi3d import refine
    from oracle import raster_ref as O
    rng = np.random.default_rng(H * 100 + K)
    P, Cn = 6000, 19
    pts = _cloud(rng, P)
    pts[:40] = pts[40:80]                     # coincident points: equal depth, the index decides
    pts[100:120, 2] += 5.0                    # far behind the camera after projection? (depends on the pose) - keep finite
    w2c = _camera(cuda)
    focal = 1.0 / (2 * np.tan(np.radians(20) / 2))
    Kmat = refine.intrinsics(focal, H, W, cuda)
    x = T(pts, cuda)
    proj = torch.matmul(x, w2c[:3, :3].T) + w2c[:3, 3]
    proj = torch.matmul(proj, Kmat.T)
    proj[:, 0:2] = proj[:, 0:2] / proj[:, 2:]
    proj[:, 0] = (proj[:, 0] / W * 2 - 1.0) * -1
    proj[:, 1] = (proj[:, 1] / H * 2 - 1.0) * -1
    proj[200:230, 2] = -0.5                   # behind the camera: must be skipped
    radius = 2.0 / H * 2.0
    idx, zbuf, dists = refine.rasterize_points(proj, (H, W), radius, K)
    ndc = proj.cpu().numpy()
    idx_o, zbuf_o, dists_o = O.rasterize_points(ndc, H, W, radius, K)
    assert (idx_o >= 0).mean() > 0.05
    assert np.array_equal(idx.cpu().numpy(), idx_o)
    np.testing.assert_allclose(dists.cpu().numpy(), dists_o, rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(zbuf.cpu().numpy(), zbuf_o, rtol=0, atol=0)
    # composite forward / backward on these hits
    feats = rng.uniform(0, 1, (P, Cn)).astype(np.float32)
    f = T(feats, cuda).requires_grad_(True)
    out = refine._PointComposite.apply(f, idx, dists, float(radius))
    img_o, w_o = O.alpha_composite(idx_o, O.point_alphas(dists_o, radius), feats)
    np.testing.assert_allclose(out.detach().cpu().numpy(), img_o, rtol=1e-5, atol=1e-6)
    dout = rng.normal(size=(Cn, H, W)).astype(np.float32)
    out.backward(T(dout, cuda))
    g_o = O.alpha_composite_backward(idx_o, w_o, dout, P)
    scale = np.abs(g_o).max()
    assert np.abs(f.grad.cpu().numpy() - g_o).max() <= 2e-5 * scale


def test_render_point_end_to_end_against_the_oracle(cuda):
    """mi3d.refine.render_point (torch projection + HIP rasteriser + compositor) vs the oracle's own projection and
    brute-force rasteriser: a pixel may differ only where a point sits within rounding of the radius."""
    from mi3d import refine
    from oracle import raster_ref as O
    rng = np.random.default_rng(5)
    P, Cn, H, W = 8000, 19, 64, 64
    pts = _cloud(rng, P)
    feats = rng.uniform(0, 1, (P, Cn)).astype(np.float32)
    w2c = _camera(cuda)
    focal = 1.0 / (2 * np.tan(np.radians(20) / 2))
    Kmat = refine.intrinsics(focal, H, W, cuda)
    radius = 2.0 / H * 2.0
    out = refine.render_point(T(pts, cuda), T(feats, cuda), H, W, Kmat, w2c, (H, W), radius, 8)[0].cpu().numpy()
    img_o, _, _, _ = O.render_point(pts, feats, H, W, Kmat.cpu().numpy(), w2c.cpu().numpy(), radius, 8)
    bad = np.abs(out - img_o).max(0) > 1e-4
    assert bad.mean() < 2e-3, bad.mean()


def test_config5_size_properties(cuda):
    """512 x 512, 500 000 points, 19 channels, radius 2 px, 8 points per pixel: every pixel's slots are sorted by
    (depth, index), hold only covering points in front of the camera, and agree with a brute-force top-K over ALL
    points on a sample of pixels; composited coverage stays in [0, 1]; gradient = adjoint of the forward."""
    from mi3d import refine
    rng = np.random.default_rng(9)
    P, Cn, H, W, K = 500_000, 19, 512, 512, 8
    pts = T(_cloud(rng, P), cuda)
    w2c = _camera(cuda)
    focal = 1.0 / (2 * np.tan(np.radians(20) / 2))
    Kmat = refine.intrinsics(focal, H, W, cuda)
    proj = torch.matmul(torch.matmul(pts, w2c[:3, :3].T) + w2c[:3, 3], Kmat.T)
    proj[:, 0:2] = proj[:, 0:2] / proj[:, 2:]
    proj[:, 0] = (proj[:, 0] / W * 2 - 1.0) * -1
    proj[:, 1] = (proj[:, 1] / H * 2 - 1.0) * -1
    radius = 2.0 / H * 2.0
    idx, zbuf, dists = refine.rasterize_points(proj, (H, W), radius, K)
    used = idx >= 0
    assert float(used[..., 0].float().mean()) > 0.05
    assert bool((used[..., 1:] <= used[..., :-1]).all())                       # used slots form a prefix
    z = torch.where(used, zbuf, torch.full_like(zbuf, float("inf")))
    later = (z[..., 1:] > z[..., :-1]) | ((z[..., 1:] == z[..., :-1]) & ((idx[..., 1:] > idx[..., :-1]) | ~used[..., 1:]))
    assert bool(later.all())
    assert bool((dists[used] < radius * radius).all() and (dists[used] >= 0).all())
    # brute force over all points for sampled pixels
    ys = torch.randint(0, H, (64,), device=cuda)
    xs = torch.randint(0, W, (64,), device=cuda)
    xf = 1 - (2 * xs.float() + 1) / W
    yf = 1 - (2 * ys.float() + 1) / H
    d2 = (xf[:, None] - proj[None, :, 0]) ** 2 + (yf[:, None] - proj[None, :, 1]) ** 2
    cover = (d2 < radius * radius) & (proj[None, :, 2] >= 0)
    for i in range(64):
        cand = torch.nonzero(cover[i]).flatten()
        zc = proj[cand, 2]
        order = np.lexsort((cand.cpu().numpy(), zc.cpu().numpy()))[:K]
        want = cand.cpu().numpy()[order]
        got = idx[ys[i], xs[i]].cpu().numpy()
        assert np.array_equal(got[:want.size], want) and (got[want.size:] == -1).all(), i
    feats = torch.rand(P, Cn, device=cuda, requires_grad=True)
    out = refine._PointComposite.apply(feats, idx, dists, float(radius))
    ones = refine._PointComposite.apply(torch.ones(P, 1, device=cuda), idx, dists, float(radius))
    assert float(ones.min()) >= 0 and float(ones.max()) <= 1 + 1e-5
    g = torch.rand_like(out)
    out.backward(g)
    lhs = float((out.detach().double() * g.double()).sum())
    rhs = float((feats.detach().double() * feats.grad.double()).sum())
    assert abs(lhs - rhs) <= 1e-5 * abs(lhs)


def test_refine_train_step_draws_its_own_timestep(cuda):
    """mi3d.refine.refine_train_step with t=None (nerf/utils.py:839-894): without a CLIP model every draw must land on
    the SDS branch (the denoise + CLIP branch of sd.py:153 cannot run without one); with the CLIP stand-in, ref_rgb and
    ref_text the t <= 400 draws take that branch and its loss value joins the step's loss."""
    from mi3d import rays as R, refine, sd_standin as S
    torch.manual_seed(0)
    P, H = 4000, 64
    d = torch.randn(P, 3, device=cuda)
    points = (d / d.norm(dim=-1, keepdim=True) * 0.35).contiguous()
    colour = torch.nn.Parameter(torch.rand(P, 3, device=cuda))
    feat = torch.nn.Parameter(torch.randn(P, 16, device=cuda))
    origin = colour.detach().clone()
    unet = refine.UNet(num_input_channels=19).to(cuda).train()
    opt = torch.optim.Adam([colour, feat] + list(unet.parameters()), lr=1e-3)
    g = S.StableDiffusionStandIn(cuda, dtype=torch.float32, with_decoder=True,
                                 unet_kw=dict(ch=(64, 64, 64, 64), ctx_dim=32, layers=1),
                                 vae_kw=dict(ch=(32, 32, 32, 32), layers=1), decoder_kw=dict(ch=(32, 32, 32, 32), layers=1))
    text_z = torch.randn(2, 77, 32, device=cuda)
    w2c = torch.linalg.inv(R.orbit_pose(1.25, 80.0, 30.0, device=cuda)[0])
    focal = 1.0 / (2 * np.tan(np.radians(20) / 2))
    args = (unet, {"colour": colour, "feat": feat}, opt, g, text_z, points, w2c, focal, H, H, 2.0 / H * 2.0, 8, origin)
    for _ in range(12):   # (0.2, 0.6) step range: about half of unrestricted draws would be <= 400
        loss = refine.refine_train_step(*args)
        assert torch.isfinite(loss)
    clip = S.CLIPStandIn(width=64, layers=2, heads=2, embed=32, text_width=32, text_layers=2, text_heads=2).to(cuda)
    for p in clip.parameters():
        p.requires_grad_(False)
    ref_rgb = torch.rand(1, 3, 512, 512, device=cuda)
    base = float(refine.refine_train_step(*args, t=500, clip_model=clip, ref_rgb=ref_rgb, ref_text="a toy"))
    with_clip = float(refine.refine_train_step(*args, t=300, clip_model=clip, ref_rgb=ref_rgb, ref_text="a toy"))
    assert np.isfinite(base) and np.isfinite(with_clip)
    for _ in range(6):
        assert torch.isfinite(refine.refine_train_step(*args, clip_model=clip, ref_rgb=ref_rgb, ref_text="a toy"))


def test_refine_train_step_with_the_trainer_level_terms(cuda):
    """nerf/utils.py:872-884 as the trainer issues them: the front view's masked L1 against the reference image, the novel
    view's guidance step + 10 x CLIP image-image + contextual (VGG19 relu5_4) terms - both change the point colours, the
    learned features and the U-Net; the novel view's extra terms really back-propagate (the step differs from one
    without them)."""
    from mi3d import rays as R, refine, sd_standin as S
    torch.manual_seed(0)
    P, H = 4000, 64
    d = torch.randn(P, 3, device=cuda)
    points = (d / d.norm(dim=-1, keepdim=True) * 0.35).contiguous()
    g = S.StableDiffusionStandIn(cuda, dtype=torch.float32, unet_kw=dict(ch=(64, 64, 64, 64), ctx_dim=32, layers=1),
                                 vae_kw=dict(ch=(32, 32, 32, 32), layers=1))
    text_z = torch.randn(2, 77, 32, device=cuda)
    w2c = torch.linalg.inv(R.orbit_pose(1.25, 80.0, 30.0, device=cuda)[0])
    focal = 1.0 / (2 * np.tan(np.radians(20) / 2))
    clip = S.CLIPStandIn(width=64, layers=2, heads=2, embed=32, text_width=32, text_layers=2, text_heads=2).to(cuda)
    cx = refine.ContextualLoss().to(cuda)
    for p in list(clip.parameters()) + list(cx.parameters()):
        p.requires_grad_(False)
    ref_rgb = torch.rand(1, 3, H, H, device=cuda)
    gt_mask = (torch.rand(1, 1, H, H, device=cuda) > 0.3).float()

    def fresh():
        torch.manual_seed(1)
        colour = torch.nn.Parameter(torch.rand(P, 3, device=cuda))
        feat = torch.nn.Parameter(torch.randn(P, 16, device=cuda))
        unet = refine.UNet(num_input_channels=19).to(cuda).train()
        opt = torch.optim.Adam([colour, feat] + list(unet.parameters()), lr=1e-3)
        return colour, feat, unet, (unet, {"colour": colour, "feat": feat}, opt, g, text_z, points, w2c, focal, H, H,
                                   2.0 / H * 2.0, 8, colour.detach().clone())
    # front view
    colour, feat, unet, args = fresh()
    c0, f0, w0 = colour.detach().clone(), feat.detach().clone(), unet.start.block["conv_f"].weight.detach().clone()
    loss = refine.refine_train_step(*args, is_front=True, ref_rgb=ref_rgb, gt_mask=gt_mask)
    assert torch.isfinite(loss) and float(loss) > 0
    assert float((colour - c0).abs().max()) > 0 and float((feat - f0).abs().max()) > 0
    assert float((unet.start.block["conv_f"].weight - w0).abs().max()) > 0
    with pytest.raises(ValueError):
        refine.refine_train_step(*args, is_front=True)
    # novel view, with and without the trainer-level terms (same seeds, same timestep: the SDS branch)
    outs = []
    for extra in (False, True):
        colour, feat, unet, args = fresh()
        torch.manual_seed(7)
        kw = dict(clip_model=clip, ref_rgb=ref_rgb, ref_text="a toy", cx_model=cx) if extra else {}
        loss = refine.refine_train_step(*args, t=500, **kw)
        assert torch.isfinite(loss)
        outs.append((float(loss), feat.detach().clone()))
    assert outs[1][0] != outs[0][0] and float((outs[1][1] - outs[0][1]).abs().max()) > 0
