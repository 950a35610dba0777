"""GPU: the SDS training step (mi3d.sds_step) - both backward schedules give the same parameter gradients, a step
changes the parameters, fp16 autocast stays finite."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(dev, seed=0, fp16=True, init_scale=None):
    from mi3d import rays as R, sd_standin, sds_step
    opt = sds_step.make_opt(max_steps=64, fp16=fp16)
    model, optimizer, scaler = sds_step.build_training_state(opt, dev, seed=seed, bitfield=0.5, init_scale=init_scale)
    with torch.no_grad():
        model.encoder.params.uniform_(-0.1, 0.1)
    ro, rd, ds = R.view_rays(32, 32, device=dev)
    return opt, model, optimizer, scaler, (ro, rd, ds)


class _TinyGuidance(torch.nn.Module):
    """A small conv stand-in so the test is fast; same sds_gradient/train_step contract as the SD stand-in."""

    def __init__(self, dev, deterministic=False):
        super().__init__()
        from mi3d.sd_standin import StableDiffusionStandIn
        self.impl = StableDiffusionStandIn.__new__(StableDiffusionStandIn)
        torch.nn.Module.__init__(self.impl)
        from mi3d import sd_standin as S
        torch.manual_seed(0)
        self.impl.device = dev
        if deterministic:  # backward-schedule equivalence must not depend on which conv solver MIOpen picks
            class _U(torch.nn.Module):
                def __init__(s):
                    super().__init__()
                    s.conv_in = torch.nn.Conv2d(4, 4, 1)

                def forward(s, x, t, encoder_hidden_states):
                    return torch.tanh(0.7 * x + 0.1 * encoder_hidden_states.float().mean())

            class _V(torch.nn.Module):
                def forward(s, x):
                    p = torch.nn.functional.avg_pool2d(x, 8)
                    return torch.cat([p, p[:, :1], 0.1 * p, 0.1 * p[:, :1]], 1)
            self.impl.unet, self.impl.vae_encoder = _U().to(dev), _V().to(dev)
        else:
            self.impl.unet = S.UNetSD2(ch=(64, 128, 128, 128), ctx_dim=64).to(dev).half()
            self.impl.vae_encoder = S.VAEEncoderSD(ch=(32, 32, 64, 64)).to(dev)
        for p in self.impl.parameters():
            p.requires_grad_(False)
        self.impl.num_train_timesteps, self.impl.min_step, self.impl.max_step = 1000, 200, 600
        betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000) ** 2
        self.impl.register_buffer("alphas", torch.cumprod(1 - betas, 0).to(dev), persistent=False)

    def __getattr__(self, k):
        if k == "impl":
            return super().__getattr__(k)
        return getattr(self.impl, k)


def _captured_grads(cuda, mode, fp16=False, **opt_over):
    from mi3d import sds_step
    opt, model, optimizer, scaler, (ro, rd, ds) = _setup(cuda, fp16=fp16, init_scale=8.0 if fp16 else None)
    for k, v in opt_over.items():
        setattr(opt, k, v)
        setattr(model.opt, k, v)
    guidance = _TinyGuidance(cuda, deterministic=True)
    text_z = torch.randn(2, 77, 64, generator=torch.Generator().manual_seed(1)).to(cuda)
    captured = {}
    orig = torch.nn.utils.clip_grad_norm_
    torch.nn.utils.clip_grad_norm_ = lambda params, max_norm: captured.update(
        {n: p.grad.detach().clone() for n, p in model.named_parameters()})
    try:
        torch.manual_seed(5)
        sds_step.sds_train_step(model, guidance, text_z, optimizer, scaler, ro, rd, ds, 32, 32, opt,
                                sds_backward=mode, t=500)  # > 400: the SDS branch (sd.py:153)
    finally:
        torch.nn.utils.clip_grad_norm_ = orig
    return captured


def test_single_backward_equals_reference_two_backward(cuda):
    """SDS injection + loss backward as ONE graph walk gives the parameter gradients of the reference's two walks
    (nerf/sd.py:171 then nerf/utils.py:983).  The normal regularisers are switched off here: sample rows are laid out
    in slab-arrival order (one atomic per wave, as the reference's per-ray atomics), the smoothness jitter
    `randn_like(xyzs)` is drawn per ROW, so which sample gets which jitter - and with it loss_smooth and its
    gradient - changes run to run even for ONE schedule (see the next test)."""
    off = dict(lambda_smooth=0.0, lambda_orient=0.0)
    single, ref = _captured_grads(cuda, "single", **off), _captured_grads(cuda, "reference", **off)
    for n in single:
        a, b = single[n], ref[n]
        assert torch.isfinite(a).all()
        scale = float(b.abs().max()) + 1e-12
        assert float((a - b).abs().max()) <= 1e-3 * scale, n


def test_overlapped_schedule_equals_reference_two_backward(cuda):
    """The U-Net on a second HIP stream under the regulariser pass, the SDS pass after it: the same two gradient terms
    as the reference's order, summed the other way round (commutative) - equal up to the float-atomic noise of the MLP
    weight gradients, with the regularisers that do not depend on row order switched on."""
    off = dict(lambda_smooth=0.0, lambda_orient=0.0)
    ovl, ref = _captured_grads(cuda, "overlapped", **off), _captured_grads(cuda, "reference", **off)
    ref2 = _captured_grads(cuda, "reference", **off)
    for n in ovl:
        a, b = ovl[n], ref[n]
        assert torch.isfinite(a).all()
        scale = float(b.abs().max()) + 1e-12
        noise = float((ref2[n] - b).abs().max())
        assert float((a - b).abs().max()) <= 4 * noise + 1e-5 * scale, (n, noise, scale)
    # under autocast too (binary16 planes, loss scale 8): the same gradients within the run-to-run noise of the schedule
    ovl, ref, ref2 = (_captured_grads(cuda, m, fp16=True, **off) for m in ("overlapped", "reference", "reference"))
    for n in ovl:
        a, b = ovl[n], ref[n]
        assert torch.isfinite(a).all() and torch.isfinite(b).all(), n
        scale = float(b.abs().max()) + 1e-12
        noise = float((ref2[n] - b).abs().max())
        assert float((a - b).abs().max()) <= 4 * noise + 1e-4 * scale, (n, noise, scale)


def test_single_backward_with_regularisers_within_run_to_run_noise(cuda):
    """All regularisers on: the two schedules differ by no more than the same schedule differs from itself when
    repeated (row order -> jitter assignment, and float-atomic summation order; the reference has both properties)."""
    r1, r2 = _captured_grads(cuda, "reference"), _captured_grads(cuda, "reference")
    s1 = _captured_grads(cuda, "single")
    for n in s1:
        scale = float(r1[n].abs().max()) + 1e-12
        noise = float((r1[n] - r2[n]).abs().max())
        diff = float((s1[n] - r1[n]).abs().max())
        assert torch.isfinite(s1[n]).all()
        # one pair of runs is a noisy estimate of the noise itself (the bias gradients are float-atomic sums over every row)
        assert diff <= 6 * noise + 2e-3 * scale, (n, diff, noise, scale)


def test_fp16_step_updates_parameters_and_stays_finite(cuda):
    from mi3d import sds_step
    opt, model, optimizer, scaler, (ro, rd, ds) = _setup(cuda, fp16=True)
    guidance = _TinyGuidance(cuda)
    text_z = torch.randn(2, 77, 64, device=cuda)
    before = model.encoder.params.detach().clone()
    w0 = model.sigma_net.net[0].weight.detach().clone()
    for _ in range(3):
        loss = sds_step.sds_train_step(model, guidance, text_z, optimizer, scaler, ro, rd, ds, 32, 32, opt)
        assert torch.isfinite(loss)
    assert torch.isfinite(model.encoder.params).all()
    assert float((model.encoder.params - before).abs().max()) > 0
    assert float((model.sigma_net.net[0].weight - w0).abs().max()) > 0


def test_fused_adan_kernel_matches_the_torch_op_sequence(cuda):
    """csrc/optim.hip (one kernel per tensor, clip factor on device) against the same update through torch ops - the
    path tests/test_optim_cpu.py pins on a trajectory of the reference's own optimizer.py - over 5 steps, odd sizes,
    with the global-norm clip active and inactive."""
    from mi3d import optim
    torch.manual_seed(0)
    shapes = [(1000003,), (64, 32), (7,), (4, 64)]
    for scale in (1e-3, 50.0):   # small gradients: clip factor 1; large: clipping
        ref_p = [torch.randn(s, device=cuda) for s in shapes]
        fus_p = [p.clone() for p in ref_p]
        for q in ref_p + fus_p:
            q.requires_grad_(True)
        groups = lambda ps: [{"params": ps[:1], "lr": 5e-2}, {"params": ps[1:], "lr": 5e-3}]
        a = optim.Adan(groups(ref_p), eps=1e-8, weight_decay=2e-5, max_grad_norm=5.0)
        b = optim.Adan(groups(fus_p), eps=1e-8, weight_decay=2e-5, max_grad_norm=5.0)
        a._fused_ok = lambda: False   # force the torch-op path
        for it in range(5):
            gs = [torch.randn(s, device=cuda) * scale for s in shapes]
            for p, q, g in zip(ref_p, fus_p, gs):
                p.grad, q.grad = g.clone(), g.clone()
            a.step()
            b.step()
            for p, q in zip(ref_p, fus_p):
                # a few ulp: torch evaluates value * m / denom, the kernel value * (m / denom)
                assert torch.allclose(p, q, rtol=1e-5, atol=5e-6), (scale, it, float((p - q).abs().max()))
                assert torch.allclose(p.grad, q.grad, rtol=1e-6, atol=0)   # the clipped gradient, left in place
        for p, q in zip(ref_p, fus_p):
            for k in ("exp_avg", "exp_avg_sq", "exp_avg_diff", "neg_pre_grad"):
                assert torch.allclose(a.state[p][k], b.state[q][k], rtol=2e-5, atol=1e-8), k


@pytest.mark.gpu
def test_denoise_clip_branch_full_size_networks(cuda):
    """The t <= 400 branch of the guidance (nerf/sd.py:153-159) with the full-size VAE decoder and ViT-B/16 stand-ins
    (the U-Net shrunk: it is exercised at full size by bench.py): forward work only, nothing reaches the render."""
    from mi3d import sd_standin as S
    g = S.StableDiffusionStandIn(cuda, with_decoder=True, unet_kw=dict(ch=(64, 128, 128, 128), ctx_dim=1024))
    clip = S.CLIPStandIn().to(cuda).half()
    for prm in clip.parameters():
        prm.requires_grad_(False)  # frozen, as the guidance networks are
    assert sum(p.numel() for p in clip.parameters()) > 140e6          # ViT-B/16 + text tower: ~150 M parameters
    assert sum(p.numel() for p in g.vae_decoder.parameters()) > 45e6  # the SD VAE decoder: ~49 M
    rgb = torch.rand(1, 3, 128, 128, device=cuda, requires_grad=True)
    ref = torch.rand(1, 3, 512, 512, device=cuda)
    with torch.autocast("cuda", dtype=torch.float16):
        loss, imgs = g.train_step(g.get_text_embeds(), rgb * 1.0, ref_rgb=ref, ref_text="a toy", clip_model=clip, t=300)
    assert imgs.shape == (1, 3, 512, 512) and torch.isfinite(imgs).all()
    assert torch.isfinite(loss) and not loss.requires_grad and rgb.grad is None
    loss, imgs = g.train_step(g.get_text_embeds(), rgb * 1.0, ref_rgb=ref, ref_text="a toy", clip_model=clip, t=450)
    assert loss == 0 and imgs is None and rgb.grad is not None
