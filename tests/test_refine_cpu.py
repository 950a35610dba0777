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
