"""CPU: the refine stage's U-Net (mi3d.refine.UNet) against the reference's own nerf/unet.py - same state_dict keys,
bit-identical output for shared weights - and the numpy oracle of the point renderer against hand-checkable cases."""
import numpy as np
import pytest
import torch


def test_unet_equals_the_reference_unet():
    from oracle import ref_import
    if not ref_import.available():
        pytest.skip("reference sources not available")
    ref_import.install()
    from nerf.unet import UNet as RefUNet
    from mi3d.refine import UNet
    torch.manual_seed(0)
    ref = RefUNet(num_input_channels=19).eval()
    ours = UNet(num_input_channels=19).eval()
    assert set(ref.state_dict().keys()) == set(ours.state_dict().keys())
    ours.load_state_dict(ref.state_dict())
    x = [torch.rand(1, 19, 64 // s, 64 // s) for s in (1, 2, 4)]
    with torch.no_grad():
        a, b = ref(x), ours(x)
    assert a.shape == (1, 3, 64, 64) and torch.equal(a, b)
    ref.train(); ours.train()   # BatchNorm in training mode (what the refine loop runs)
    a, b = ref(x), ours(x)
    assert torch.allclose(a, b, atol=1e-6)


def test_raster_oracle_on_hand_checkable_cases():
    from oracle import raster_ref as R
    H = W = 8
    # pixel centres: NDC x of output column xi is 1 - (2 xi + 1) / W  (+X left)
    assert np.allclose(R.pix_to_ndc(W - 1 - np.arange(W), W, H), 1 - (2 * np.arange(W) + 1) / W)
    # one point exactly on the centre of pixel (row 2, col 5), one behind the camera, one farther on the same pixel
    cx, cy = 1 - (2 * 5 + 1) / W, 1 - (2 * 2 + 1) / H
    pts = np.array([[cx, cy, 2.0], [cx, cy, -1.0], [cx + 0.01, cy, 3.0]], np.float32)
    idx, zbuf, dists = R.rasterize_points(pts, H, W, radius=0.1, K=4)
    assert list(idx[2, 5]) == [0, 2, -1, -1] and dists[2, 5, 0] == 0 and abs(dists[2, 5, 1] - 1e-4) < 1e-6
    assert (idx[np.arange(H) != 2] == -1).all()
    a = R.point_alphas(dists, 0.1)
    feats = np.array([[1.0, 0.0], [9.0, 9.0], [0.0, 1.0]], np.float32)
    img, w = R.alpha_composite(idx, a, feats)
    a0, a1 = 1 - np.sqrt(1e-3), 1 - np.sqrt(max(0.1 * 1e-4 / 0.01, 1e-3))
    assert np.allclose(img[:, 2, 5], [a0, (1 - a0) * a1], atol=1e-6)
    g = R.alpha_composite_backward(idx, w, np.ones((2, H, W), np.float32), 3)
    assert np.allclose(g[0], [a0, a0], atol=1e-6) and np.allclose(g[1], 0) and np.allclose(g[2], (1 - a0) * a1, atol=1e-6)


def test_contextual_loss_properties():
    """mi3d.refine.contextual_loss (the `contextual_loss` package's cosine form restated; nerf/utils.py:810,881): zero for
    identical feature maps, positive otherwise, unchanged when the positions of either map are permuted (it matches
    positions by feature similarity, not by place), and differentiable w.r.t. its first argument."""
    from mi3d import refine
    torch.manual_seed(0)
    y = torch.randn(1, 16, 6, 6)
    assert float(refine.contextual_loss(y.clone(), y)) < 1e-3
    x = torch.randn(1, 16, 6, 6, requires_grad=True)
    l = refine.contextual_loss(x, y)
    assert float(l) > 0.5
    perm = torch.randperm(36)
    xp = x.reshape(1, 16, 36)[:, :, perm].reshape(1, 16, 6, 6)
    yp = y.reshape(1, 16, 36)[:, :, torch.randperm(36)].reshape(1, 16, 6, 6)
    assert abs(float(refine.contextual_loss(xp, y)) - float(l)) < 1e-5
    assert abs(float(refine.contextual_loss(x, yp)) - float(l)) < 1e-5
    l.backward()
    assert torch.isfinite(x.grad).all() and float(x.grad.abs().max()) > 0
    # moving x towards y lowers the loss
    with torch.no_grad():
        closer = 0.2 * x + 0.8 * y
    assert float(refine.contextual_loss(closer, y)) < float(l)


def test_vgg19_feature_stack_and_trainer_level_clip_term():
    """VGG19 `features` to relu5_4 (16 convolutions, 4 pools: 1/16 resolution, 512 channels, 20.0 M parameters) and the
    trainer's CLIP image-image term (nerf/utils.py:434-441) - which, unlike the guidance's denoise branch, back-propagates
    into the render."""
    from mi3d import refine, sd_standin as S
    vgg = refine.VGG19Features()
    assert sum(p.numel() for p in vgg.parameters()) == 20_024_384          # torchvision vgg19.features[:36]
    assert sum(isinstance(m, torch.nn.Conv2d) for m in vgg.features) == 16
    with torch.no_grad():
        f = vgg(torch.rand(1, 3, 64, 64))
    assert f.shape == (1, 512, 4, 4) and float(f.min()) >= 0
    cx = refine.ContextualLoss()
    clip = S.CLIPStandIn(width=32, layers=1, heads=2, embed=16, text_width=16, text_layers=1, text_heads=2)
    for p in clip.parameters():
        p.requires_grad_(False)
    rgb = torch.rand(1, 3, 48, 48, requires_grad=True)
    ref = torch.rand(1, 3, 64, 64)
    loss = 10 * refine.img_clip_loss(clip, rgb, ref) + cx(rgb, ref)
    loss.backward()
    assert torch.isfinite(loss) and torch.isfinite(rgb.grad).all() and float(rgb.grad.abs().max()) > 0
    assert -10.0 - 1e-4 <= float(10 * refine.img_clip_loss(clip, ref, ref)) <= -10.0 + 1e-3   # identical images: cosine 1
