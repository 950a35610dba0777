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


# ---------------------------------------------------------------------------------------------------------------------
# the point-0 pass of the two-backward schedule, deferred into the pass that follows it (grid_ops.DEFER_POINT0)

def _two_pass_state(cuda, fp16):
    from mi3d import rays as R, sds_step
    opt = sds_step.make_opt(max_steps=128, fp16=fp16)
    model, _, _ = sds_step.build_training_state(opt, cuda, seed=3, bitfield=0.5)
    with torch.no_grad():
        model.encoder.params.uniform_(-0.3, 0.3)
    rays = R.view_rays(32, 32, device=cuda)
    inject = (torch.randn(1024, 3, generator=torch.Generator().manual_seed(2)) * 2e-3).to(cuda)
    return opt, model, rays, inject


def _two_passes(model, opt, rays, inject, fp16, between=None, second=True, scale=4.0):
    """render -> image.backward(inject, retain_graph=True) [point 0 only] -> `between()` -> (scale * loss).backward()."""
    from conftest import position_jitter
    from mi3d import sds_step
    ro, rd, ds = rays
    model.zero_grad(set_to_none=True)
    torch.manual_seed(11)
    with position_jitter():
        with torch.autocast("cuda", dtype=torch.float16, enabled=fp16):
            out = model.render(ro, rd, depth_scale=ds, bg_color=torch.full((3,), 0.7, device=ro.device), perturb=True,
                               ambient_ratio=1.0, shading="albedo", force_all_rays=True, dt_gamma=0, max_steps=128)
            loss = sds_step.regularisers(opt, out, out["weights_sum"].reshape(1, 1, 32, 32))
        out["image"].backward(inject.view_as(out["image"]), retain_graph=True)
        mid = between(model) if between is not None else None
        if second:
            (scale * loss).backward()
    return {n: p.grad.detach().clone() for n, p in model.named_parameters()}, mid


def _parked(model):
    return len(model.encoder.params.__dict__.get("_mi3d_pending") or [])


def _raw_grad(model):
    return torch.Tensor.grad.__get__(model.encoder.params)   # what autograd accumulated, without completing anything


@pytest.mark.parametrize("fp16", [False, True])
def test_deferred_point0_scatter_equals_two_scatters(cuda, fp16):
    """The reference schedule with the first pass's point-0 planes riding along in the second pass's scatter (ONE
    mi3d_grid_scatter_binned_plus call) against every pass scattered on its own (two calls): the same table gradient up
    to fp32 rounding of w (a + b) against w a + w b, every other gradient bit-identical; while the planes are parked the
    raw accumulator holds nothing of the first pass, and nothing stays parked afterwards."""
    from conftest import record_scatter_workspaces
    from mi3d import grid_ops
    opt, model, rays, inject = _two_pass_state(cuda, fp16)
    grid_ops.DEFER_POINT0 = False
    try:
        with record_scatter_workspaces() as calls_two:
            two, _ = _two_passes(model, opt, rays, inject, fp16)
    finally:
        grid_ops.DEFER_POINT0 = True
    seen = {}

    def between(m):
        seen["parked"], seen["raw"] = _parked(m), _raw_grad(m)
    with record_scatter_workspaces() as calls_one:
        one, _ = _two_passes(model, opt, rays, inject, fp16, between=between)
    assert len(calls_two) == 2 and len(calls_one) == 1 and all(c > 0 for c in calls_two + calls_one)
    assert seen["parked"] == 1 and (seen["raw"] is None or float(seen["raw"].abs().max()) == 0.0)
    assert _parked(model) == 0
    g2, g1 = two["encoder.params"], one["encoder.params"]
    scale = float(g2.abs().max())
    assert scale > 0 and float((g1 - g2).abs().max()) <= 2e-6 * scale
    # (the same entries are touched; a handful of sums that cancel to within the 64-bit fixed point's last digit are an
    # exact zero in one order of addition and not in the other)
    assert abs(int((g1 != 0).sum()) - int((g2 != 0).sum())) <= 1e-4 * int((g2 != 0).sum())
    for n in two:
        if n != "encoder.params":   # MLP weight gradients: float-atomic sums, run-to-run noise
            assert float((one[n] - two[n]).abs().max()) <= 2e-4 * float(two[n].abs().max()), n


def test_reading_grad_completes_a_deferred_scatter(cuda):
    """Whoever reads encoder.params.grad sees the complete gradient: between the passes (the parked planes are scattered
    then, and the second pass has nothing left to take along), after a first pass that no second one follows, through a
    flat all-reduce bucket's view, and not at all after `.grad = None`."""
    from mi3d import dp, grid_ops
    opt, model, rays, inject = _two_pass_state(cuda, False)
    grid_ops.DEFER_POINT0 = False
    try:
        two, _ = _two_passes(model, opt, rays, inject, False)
        first_only, _ = _two_passes(model, opt, rays, inject, False, second=False)
    finally:
        grid_ops.DEFER_POINT0 = True
    tol = lambda ref: 2e-6 * float(ref.abs().max())
    # (1) a reader between the passes
    got, mid = _two_passes(model, opt, rays, inject, False,
                           between=lambda m: (_parked(m), m.encoder.params.grad.detach().clone(), _parked(m)))
    assert mid[0] == 1 and mid[2] == 0
    assert float((mid[1] - first_only["encoder.params"]).abs().max()) <= tol(first_only["encoder.params"])
    assert float((got["encoder.params"] - two["encoder.params"]).abs().max()) <= tol(two["encoder.params"])
    # (2) no second pass: the first reader gets the first pass's gradient
    got, _ = _two_passes(model, opt, rays, inject, False, second=False)
    assert _parked(model) == 0
    assert float((got["encoder.params"] - first_only["encoder.params"]).abs().max()) <= tol(first_only["encoder.params"])
    # (3) .grad = None between the passes drops the first pass, parked or not
    grid_ops.DEFER_POINT0 = False
    try:
        want, _ = _two_passes(model, opt, rays, inject, False, between=lambda m: setattr(m.encoder.params, "grad", None))
    finally:
        grid_ops.DEFER_POINT0 = True
    got, _ = _two_passes(model, opt, rays, inject, False, between=lambda m: setattr(m.encoder.params, "grad", None))
    assert float((got["encoder.params"] - want["encoder.params"]).abs().max()) <= tol(want["encoder.params"])
    # (4) gradients as views into one flat bucket (mi3d/dp.py): the completed scatter lands in the bucket, in place
    bucket = dp.FlatGradBucket(model.parameters())
    from conftest import position_jitter
    from mi3d import sds_step
    ro, rd, ds = rays
    bucket.zero()
    torch.manual_seed(11)
    with position_jitter():
        out = model.render(ro, rd, depth_scale=ds, bg_color=torch.full((3,), 0.7, device=cuda), perturb=True,
                           ambient_ratio=1.0, shading="albedo", force_all_rays=True, dt_gamma=0, max_steps=128)
        loss = sds_step.regularisers(opt, out, out["weights_sum"].reshape(1, 1, 32, 32))
        out["image"].backward(inject.view_as(out["image"]), retain_graph=True)
        assert _parked(model) == 1
        (4.0 * loss).backward()
    bucket.all_reduce_mean()        # (no process group: checks the views, reads every .grad)
    n_table = model.encoder.params.numel()
    assert model.encoder.params.grad.data_ptr() == bucket.flat.data_ptr()
    assert float((bucket.flat[:n_table] - two["encoder.params"]).abs().max()) <= tol(two["encoder.params"])
    for p in model.parameters():
        p.grad = None


def test_autograd_grad_is_never_deferred(cuda):
    """torch.autograd.grad hands gradients BACK instead of accumulating them: the node must return the table gradient
    itself, retain_graph or not - nothing is parked."""
    from conftest import position_jitter
    from mi3d import grid_ops
    opt, model, rays, inject = _two_pass_state(cuda, False)
    ro, rd, ds = rays
    outs = []
    for mode in ("grad", "backward"):
        model.zero_grad(set_to_none=True)
        torch.manual_seed(11)
        with position_jitter():
            out = model.render(ro, rd, depth_scale=ds, bg_color=torch.full((3,), 0.7, device=cuda), perturb=True,
                               ambient_ratio=1.0, shading="albedo", force_all_rays=True, dt_gamma=0, max_steps=128)
        if mode == "grad":
            (g,) = torch.autograd.grad(out["image"], [model.encoder.params], inject.view_as(out["image"]),
                                       retain_graph=True)
            assert _parked(model) == 0 and _raw_grad(model) is None
        else:
            out["image"].backward(inject.view_as(out["image"]), retain_graph=True)
            assert _parked(model) == 1
            g = model.encoder.params.grad.detach().clone()
        outs.append(g)
    assert float(outs[0].abs().max()) > 0
    assert float((outs[0] - outs[1]).abs().max()) <= 2e-6 * float(outs[0].abs().max())
    assert grid_ops.DEFER_POINT0


def test_reference_schedule_does_not_grow_memory_and_parks_nothing_across_steps(cuda):
    """Thirty steps of the reference's two-backward schedule (the deferred point-0 scatter in every one of them): nothing
    stays parked on encoder.params once a step is over, the allocator's live bytes are flat from step 5 on, and the
    parameters keep moving."""
    from mi3d import sds_step
    opt, model, optimizer, scaler, (ro, rd, ds) = _setup(cuda, fp16=True, init_scale=8.0)
    guidance = _TinyGuidance(cuda, deterministic=True)
    text_z = torch.randn(2, 77, 64, generator=torch.Generator().manual_seed(1)).to(cuda)
    live, before = [], model.encoder.params.detach().clone()
    for it in range(30):
        sds_step.sds_train_step(model, guidance, text_z, optimizer, scaler, ro, rd, ds, 32, 32, opt,
                                sds_backward="reference", t=500)
        assert not model.encoder.params.__dict__.get("_mi3d_pending")
        torch.cuda.synchronize()
        live.append(torch.cuda.memory_allocated(cuda))
    assert max(live[5:]) - min(live[5:]) <= 8 << 20, live          # (the sample count moves a little from step to step)
    assert torch.isfinite(model.encoder.params).all()
    assert float((model.encoder.params - before).abs().max()) > 0
