"""Both branches of the guidance's train_step (nerf/sd.py:117-174) on tiny CPU networks: the branch rule, what each
branch returns and back-propagates, and the DDIM step's algebra."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "make-it-3d_amd"))


def test_ddim_step_is_the_eta0_update():
    from mi3d.sd_standin import ddim_step
    torch.manual_seed(0)
    betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float64) ** 2
    ac = torch.cumprod(1 - betas, 0)
    x0, noise = torch.randn(1, 4, 8, 8, dtype=torch.float64), torch.randn(1, 4, 8, 8, dtype=torch.float64)
    for tt in (0, 1, 250, 400, 999):
        t = torch.tensor([tt])
        x_t = ac[t].sqrt() * x0 + (1 - ac[t]).sqrt() * noise
        a_prev = ac[tt - 1] if tt > 0 else ac[0]  # set_alpha_to_one = False: the final alpha is alphas_cumprod[0]
        # given the true noise, the step lands on the same x0 re-noised to level t-1 with the same noise
        want = a_prev.sqrt() * x0 + (1 - a_prev).sqrt() * noise
        assert torch.allclose(ddim_step(ac, noise, t, x_t), want, atol=1e-12)
        # and for an arbitrary prediction it is the published closed form
        eps = torch.randn_like(noise)
        closed = (a_prev / ac[tt]).sqrt() * (x_t - (1 - ac[tt]).sqrt() * eps) + (1 - a_prev).sqrt() * eps
        assert torch.allclose(ddim_step(ac, eps, t, x_t), closed, atol=1e-12)


@pytest.fixture(scope="module")
def tiny():
    from mi3d import sd_standin as S
    dev = torch.device("cpu")
    g = S.StableDiffusionStandIn(dev, dtype=torch.float32, with_decoder=True,
                                 unet_kw=dict(ch=(64, 64, 64, 64), ctx_dim=32, layers=1),
                                 vae_kw=dict(ch=(32, 32, 32, 32), layers=1),
                                 decoder_kw=dict(ch=(32, 32, 32, 32), layers=1))
    clip = S.CLIPStandIn(width=64, layers=2, heads=2, embed=32, text_width=32, text_layers=2, text_heads=2)
    for p in clip.parameters():
        p.requires_grad_(False)
    text_z = torch.randn(2, 77, 32)
    return g, clip, text_z


def test_branch_rule_and_what_each_branch_does(tiny):
    g, clip, text_z = tiny
    ref_rgb = torch.rand(1, 3, 512, 512)

    def step(t, islarge=False):
        rgb = torch.rand(1, 3, 16, 16, requires_grad=True)
        out = g.train_step(text_z, rgb * 1.0, ref_rgb=ref_rgb, islarge=islarge, ref_text="a toy", clip_model=clip,
                           guidance_scale=10, t=t)
        return out, rgb.grad

    # t/1000 <= 0.4 and not is_large: denoise + CLIP branch - a loss VALUE and the decoded image, nothing back-propagated
    (loss, imgs), grad = step(400)
    assert torch.is_tensor(loss) and loss.ndim == 0 and torch.isfinite(loss) and not loss.requires_grad
    assert -20.0 <= float(loss) <= 20.0                      # 10 * (-cos) + 10 * (-cos)
    assert imgs.shape == (1, 3, 512, 512) and float(imgs.min()) >= 0 and float(imgs.max()) <= 1
    assert grad is None
    # t/1000 > 0.4: SDS branch - (0, None) and the gradient is injected into the render
    (loss, imgs), grad = step(401)
    assert loss == 0 and imgs is None and grad is not None and torch.isfinite(grad).all() and float(grad.abs().max()) > 0
    # is_large views take the SDS branch at any t
    (loss, imgs), grad = step(300, islarge=True)
    assert loss == 0 and imgs is None and grad is not None
    # a tensor t is accepted as well
    (loss, imgs), _ = step(torch.tensor([250]))
    assert imgs is not None
    with pytest.raises(ValueError):
        g.train_step(text_z, torch.rand(1, 3, 16, 16), t=100)    # the CLIP branch without a CLIP model


def test_clip_standin_shapes():
    from mi3d.sd_standin import CLIPStandIn
    clip = CLIPStandIn(width=64, layers=1, heads=2, embed=48, text_width=32, text_layers=1, text_heads=2)
    z = clip.encode_image(torch.rand(2, 3, 224, 224))
    tok = clip.tokenize(["a toy", "a much longer prompt " * 20])
    assert z.shape == (2, 48) and tok.shape == (2, 77) and int(tok[1].max()) == 49407
    assert clip.encode_text(tok).shape == (2, 48)
