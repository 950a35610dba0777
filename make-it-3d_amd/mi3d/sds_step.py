"""One coarse-stage SDS training step on a novel view - the inner loop of the reference Trainer
(/root/reference/nerf/utils.py:461-563 `train_step` + :977-986 of `train_one_epoch`) restated for benchmarking and
tests, minus the CLIP terms (openai `clip` weights are unavailable offline and outside the hot path).

    render (march -> 13-point field -> composite, normal regularisers)
    -> SD guidance: VAE-encode(grad) + U-Net eps-prediction x2 (no grad) -> SDS gradient
    -> regularisers (opacity, entropy, orientation, smoothness)
    -> backward -> clip_grad_norm -> GradScaler.step(Adan)

`sds_backward`:
  "reference": two NeRF backward passes, exactly as the reference does it - `latents.backward(grad,
               retain_graph=True)` inside guidance.train_step (nerf/sd.py:171), then `scaler.scale(loss).backward()`.
  "single"   : one backward of  scale*loss + <latents, grad.detach()> .  The parameter gradients are the same sum
               (the SDS term bypasses the loss scale in both, quirk SURVEY 9.10); the NeRF graph is walked once.
               (tests/test_sds_step_gpu.py checks the two agree.)
  "overlapped": the reference's two backward passes with the U-Net off the critical path: the no-grad U-Net half of
               guidance.train_step is queued on a second HIP stream, the regulariser backward pass (which needs nothing
               from the diffusion model) runs beside it, then `latents.backward(grad)`.  Every parameter gradient is
               the sum of the same two terms as in "reference", added in the other order (a + b == b + a in floating
               point), every random draw happens in the same order on the same stream.
"""
import math
import types

import torch
import torch.nn.functional as F

DEFAULT_OPT = dict(bound=1.0, cuda_ray=True, min_near=0.1, density_thresh=10.0, bg_radius=-1, blob_density=5.0,
                   blob_radius=0.1, max_depth=10.0, dt_gamma=0.0, max_steps=1024, lambda_entropy=1.0,
                   lambda_opacity=1e-3, lambda_orient=1e-2, lambda_smooth=1.0, guidance_scale=10.0, lr=1e-3,
                   fp16=True)


def make_opt(**over):
    d = dict(DEFAULT_OPT)
    d.update(over)
    return types.SimpleNamespace(**d)


def render_kwargs(opt):
    """What `**vars(self.opt)` forwards that run_cuda actually reads (utils.py:496-498)."""
    return dict(dt_gamma=opt.dt_gamma, max_steps=opt.max_steps)


def regularisers(opt, outputs, pred_ws, is_large=False, past_diff_iters=True):
    """utils.py:519-548."""
    loss = 0
    if opt.lambda_opacity > 0:
        lo = (pred_ws ** 2).mean()
        loss = loss + opt.lambda_opacity * lo * (10 if is_large else 1)
    if opt.lambda_entropy > 0:
        a = pred_ws.clamp(1e-5, 1 - 1e-5)
        le = (-a * torch.log2(a) - (1 - a) * torch.log2(1 - a)).mean()
        loss = loss + opt.lambda_entropy * le * (10 if past_diff_iters else 1)
    if opt.lambda_orient > 0 and "loss_orient" in outputs:
        lo = outputs["loss_orient"]
        loss = loss + opt.lambda_orient * lo + opt.lambda_orient * lo * (10 if past_diff_iters else 1)
    if opt.lambda_smooth > 0 and "loss_smooth" in outputs:
        loss = loss + opt.lambda_smooth * outputs["loss_smooth"]
    return loss


def sds_train_step(model, guidance, text_z, optimizer, scaler, rays_o, rays_d, depth_scale, H, W, opt,
                   shading="albedo", ambient_ratio=1.0, sds_backward="single", t=None, grad_sync=None, clip_model=None,
                   ref_rgb=None, ref_text=None):
    """Returns the (unscaled) loss tensor.  `grad_sync` (callable or None) runs between backward and the optimizer step -
    the data-parallel all-reduce hook.  `clip_model`, `ref_rgb`, `ref_text` ("reference" schedule only): what the
    guidance's OTHER branch needs - for t/1000 <= 0.4 on a view that is not `is_large` guidance.train_step takes one
    DDIM step, decodes, and returns 10 x (CLIP image-image + image-text) similarity of the DENOISED image as a loss value
    (nerf/sd.py:153-159; decoded under no_grad: a constant w.r.t. the NeRF) which the trainer adds to the regularisers
    (nerf/utils.py:515-516); no SDS gradient is injected on such a step."""
    from .grid_ops import phase
    optimizer.zero_grad(set_to_none=False)
    B = rays_o.shape[0]
    with torch.autocast("cuda", dtype=torch.float16, enabled=opt.fp16):
        with phase("render_forward"):
            bg_color = torch.rand(3, device=rays_o.device)
            outputs = model.render(rays_o, rays_d, depth_scale=depth_scale, bg_color=bg_color, staged=False,
                                   perturb=True, ambient_ratio=ambient_ratio, shading=shading, force_all_rays=True,
                                   **render_kwargs(opt))
            pred_rgb = outputs["image"].reshape(B, H, W, 3).permute(0, 3, 1, 2).contiguous()
            pred_ws = outputs["weights_sum"].reshape(B, 1, H, W)
        with phase("guidance_train_step"):  # VAE encode + U-Net x2 (+ NeRF backward #1 in the reference schedule)
            guidance_loss = 0
            if sds_backward == "reference":
                # NeRF backward #1 inside (t/1000 > 0.4), or the denoise + CLIP branch's loss value (no backward)
                guidance_loss, _ = guidance.train_step(text_z, pred_rgb, ref_rgb=ref_rgb, ref_text=ref_text,
                                                       clip_model=clip_model, guidance_scale=opt.guidance_scale, t=t)
                sds_term = None
            elif sds_backward == "overlapped":
                latents, finish_guidance = guidance.sds_gradient_async(text_z, pred_rgb, opt.guidance_scale, t)
                sds_term = None
            else:
                latents, grad = guidance.sds_gradient(text_z, pred_rgb, opt.guidance_scale, t)
                sds_term = (latents.float() * grad.float()).sum()
        with phase("regularisers_backward"):
            loss = regularisers(opt, outputs, pred_ws)
            if torch.is_tensor(guidance_loss):
                loss = guidance_loss + loss       # utils.py:515-516: `loss` starts as the guidance's return value
    with phase("regularisers_backward"):
        total = scaler.scale(loss)
        if sds_term is not None:
            total = total + sds_term  # the SDS gradient is never loss-scaled (sd.py:171)
        total.backward(retain_graph=sds_backward == "overlapped")
    if sds_backward == "overlapped":
        with phase("guidance_train_step"):
            latents.backward(gradient=finish_guidance())   # sd.py:171, after the pass it no longer has to wait for
    with phase("sync_clip_optimizer"):
        if grad_sync is not None:
            grad_sync()
        torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=10)  # on still-scaled grads, as utils.py:984
        scaler.step(optimizer)
        scaler.update()
    return loss.detach()


def build_training_state(opt, device, seed=0, bitfield="dense", init_scale=None, **net_kw):
    """Model + Adan + GradScaler with the reference's hyper-parameters (main.py:132, utils.py:309) and an analytic
    occupancy bitfield: 'dense' = all ones, float r = sphere of radius r."""
    from . import network, optim
    torch.manual_seed(seed)
    model = network.NeRFNetwork(opt, **net_kw).to(device)
    model.train()
    set_bitfield(model, bitfield)
    optimizer = optim.Adan(model.get_params(5 * opt.lr), eps=1e-8, weight_decay=2e-5, max_grad_norm=5.0)
    # init_scale None = torch's default 65536 (what utils.py:309 constructs); a scaler that overflows skips the
    # optimizer step and halves, so a benchmark passes the value the scaler settles at (see bench.py)
    kw = {} if init_scale is None else {"init_scale": float(init_scale)}
    scaler = torch.amp.GradScaler("cuda", enabled=opt.fp16, **kw)
    return model, optimizer, scaler


def set_bitfield(model, kind):
    import raymarching
    H, C = model.grid_size, model.cascade
    dev = model.density_bitfield.device
    if kind == "dense":
        model.density_bitfield.fill_(255)
        return
    r = float(kind)
    ax = torch.arange(H, dtype=torch.int32, device=dev)
    gx, gy, gz = torch.meshgrid(ax, ax, ax, indexing="ij")
    coords = torch.stack([gx.reshape(-1), gy.reshape(-1), gz.reshape(-1)], -1)
    slots = raymarching.morton3D(coords).long()
    grid = torch.zeros(C, H ** 3, device=dev)
    for c in range(C):
        b = min(2 ** c, model.bound)
        centre = ((coords.float() + 0.5) / H * 2 - 1) * b
        grid[c, slots] = (centre.norm(dim=-1) < r).float()
    model.density_bitfield = raymarching.packbits(grid, 0.5, model.density_bitfield)
