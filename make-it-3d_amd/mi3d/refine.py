"""Refine stage (BASELINE config 5): the textured-point-cloud renderer and its deferred-rendering U-Net.

`render_point` mirrors /root/reference/nerf/refine_utils.py:306-333 - same name, arguments and result - with the two
pytorch3d calls (rasterize_points, compositing.alpha_composite) replaced by the HIP kernels of csrc/raster.hip
(C ABI Part 7); the projection in front of them is the reference's own handful of torch ops, kept verbatim in
meaning so the NDC coordinates are the ones pytorch3d would have been given.  Differentiable w.r.t. the point
features (colours + 16 learned channels), which is what the reference optimises (nerf/utils.py:826-831).

`UNet` restates nerf/unet.py:111-172 (gated convolutions, multi-scale input, bilinear upsampling) on stock torch
modules with the reference's module names, so its state_dict loads reference checkpoints.

`refine_train_step` is the inner loop of nerf/utils.py:839-894 for a novel view: three renders at H, H/2, H/4, the
U-Net, the full-resolution coverage-mask render + 5x5 max-pool, SDS guidance (scale 5), colour regulariser, background
loss, Adam - minus the CLIP / contextual (VGG) terms, whose weights do not exist offline.
"""
import ctypes as C

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function

from . import _lib as L


class _PointComposite(Function):
    @staticmethod
    def forward(ctx, feats, idx, dists, radius):
        feats = L.dev_f32(feats.contiguous(), "features")
        H, W, K = idx.shape
        P, Cn = feats.shape
        out = torch.empty(Cn, H, W, dtype=torch.float32, device=feats.device)
        L.launch("mi3d_points_composite_forward", feats, L.ptr(idx), L.ptr(dists), H, W, K, L.ptr(feats), Cn,
                 C.c_double(radius), L.ptr(out))
        ctx.save_for_backward(idx, dists)
        ctx.meta = (P, Cn, float(radius))
        return out

    @staticmethod
    def backward(ctx, dout):
        idx, dists = ctx.saved_tensors
        P, Cn, radius = ctx.meta
        H, W, K = idx.shape
        dout = L.dev_f32(dout.float().contiguous(), "grad")
        grad = torch.zeros(P, Cn, dtype=torch.float32, device=dout.device)
        L.launch("mi3d_points_composite_backward", dout, L.ptr(idx), L.ptr(dists), H, W, K, L.ptr(dout), Cn,
                 C.c_double(radius), L.ptr(grad))
        return grad, None, None, None


def rasterize_points(points_ndc, image_size, radius, points_per_pixel):
    """pytorch3d.renderer.points.rasterize_points for one cloud: (idx int32 [H,W,K], zbuf, dists) from NDC points
    [P,3]; unused slots hold -1."""
    H, W = (image_size, image_size) if isinstance(image_size, int) else image_size
    pts = L.dev_f32(points_ndc.detach().contiguous(), "points", 3)
    P, dev = pts.shape[0], pts.device
    K = int(points_per_pixel)
    need = L.lib().mi3d_points_rasterize_workspace(P, H, W, float(radius))
    ws = torch.empty(max(int(need), 64), dtype=torch.uint8, device=dev)
    idx = torch.empty(H, W, K, dtype=torch.int32, device=dev)
    zbuf = torch.empty(H, W, K, dtype=torch.float32, device=dev)
    dists = torch.empty(H, W, K, dtype=torch.float32, device=dev)
    L.launch("mi3d_points_rasterize", pts, L.ptr(pts), P, H, W, float(radius), K, L.ptr(ws), C.c_size_t(ws.numel()),
             L.ptr(idx), L.ptr(zbuf), L.ptr(dists))
    return idx, zbuf, dists


def render_point(points_xyz_org, points_color, H, W, K, world2cam, image_size, radius, ppp, bg_feat=None,
                 acc="alphacomposite"):
    """refine_utils.py:306-333.  points [P,3] world, features [P,C] -> [1,C,H,W].  (`bg_feat` and `acc` are accepted
    and unused, as in the reference.)"""
    if points_xyz_org.requires_grad:
        raise NotImplementedError("point positions carry no gradient on this path (the reference optimises features)")
    proj_xyz = torch.matmul(points_xyz_org, world2cam[:3, :3].T) + world2cam[:3, 3]
    proj_xyz = torch.matmul(proj_xyz, K.T)                       # perspective projection
    proj_xyz[:, 0:2] = proj_xyz[:, 0:2] / proj_xyz[:, 2:]
    proj_xyz[:, 0] = proj_xyz[:, 0] / W * 2 - 1.0
    proj_xyz[:, 1] = proj_xyz[:, 1] / H * 2 - 1.0
    proj_xyz[:, 0] = proj_xyz[:, 0] * -1
    proj_xyz[:, 1] = proj_xyz[:, 1] * -1
    idx, _, dists = rasterize_points(proj_xyz, image_size, radius, ppp)
    return _PointComposite.apply(points_color.float(), idx, dists, float(radius)).unsqueeze(0)


# ----------------------------------------------------------------------------------------------- nerf/unet.py:111-172
class _Identity(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()

    def forward(self, x):
        return x


class GatedBlock(nn.Module):
    """conv_f -> ELU, conv_m -> sigmoid, product, norm (unet.py:33-60); module names as in the reference."""

    def __init__(self, cin, cout, kernel_size=3, normalization=nn.BatchNorm2d):
        super().__init__()
        pad = (kernel_size - 1) // 2
        self.block = nn.ModuleDict({
            "conv_f": nn.Conv2d(cin, cout, kernel_size, padding=pad), "act_f": nn.ELU(),
            "conv_m": nn.Conv2d(cin, cout, kernel_size, padding=pad), "act_m": nn.Sigmoid(),
            "norm": normalization(cout)})

    def forward(self, x, *a, **k):
        b = self.block
        return b["norm"](b["act_f"](b["conv_f"](x)) * b["act_m"](b["conv_m"](x)))


class _Down(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = GatedBlock(cin, cout)
        self.down = nn.AvgPool2d(2, 2)

    def forward(self, x, mask=None):
        return self.conv(self.down(x))


class _Up(nn.Module):
    def __init__(self, cout):
        super().__init__()
        self.up = nn.Sequential(nn.Upsample(scale_factor=2, mode="bilinear"),
                                nn.Sequential(nn.Conv2d(cout * 2, cout, 3, padding=1)))
        self.conv = GatedBlock(cout * 2, cout, normalization=_Identity)

    def forward(self, low, skip):
        return self.conv(torch.cat([self.up(low), skip], 1))


class UNet(nn.Module):
    """Rendering network with multi-scale input (unet.py:111-172): feature_scale 4 -> 16/32/64 channels."""

    def __init__(self, num_input_channels=3, num_output_channels=3, feature_scale=4):
        super().__init__()
        f = [x // feature_scale for x in (64, 128, 256, 512, 1024)]
        c = num_input_channels
        self.start = GatedBlock(c, f[0])
        self.down1 = _Down(f[0], f[1] - c)
        self.down2 = _Down(f[1], f[2] - c)
        self.up2 = _Up(f[1])
        self.up1 = _Up(f[0])
        self.final = nn.Sequential(nn.Sequential(nn.Conv2d(f[0], num_output_channels, 1)), nn.Sigmoid())

    def forward(self, inputs):
        in64 = self.start(inputs[0])
        down1 = torch.cat([self.down1(in64), inputs[1]], 1)
        down2 = torch.cat([self.down2(down1), inputs[2]], 1)
        return self.final(self.up1(self.up2(down2, down1), in64))


def intrinsics(focal, h, w, device):
    """nerf/utils.py:857-858: K for an h x w render with normalised focal length."""
    return torch.tensor(np.array([[focal * w, 0, 0.5 * w], [0, focal * h, 0.5 * h], [0, 0, 1]]), device=device).float()


def refine_render(unet, points, feats, world2cam, focal, H, W, radius, ppp):
    """Three renders at H, H/2, H/4 -> U-Net -> rgb [1,3,H,W], plus the dilated coverage mask (utils.py:850-869)."""
    preds, scale = [], 1
    for _ in range(3):
        h, w = H // scale, W // scale
        preds.append(render_point(points, feats, h, w, intrinsics(focal, h, w, points.device), world2cam, (h, w), radius,
                                  ppp))
        scale *= 2
    rgb = unet(preds)
    mask = render_point(points, torch.ones_like(points), H, W, intrinsics(focal, H, W, points.device), world2cam, (H, W),
                        radius, ppp)
    return rgb, F.max_pool2d(mask, kernel_size=5, stride=1, padding=2)


def refine_train_step(unet, params, optimizer, guidance, text_z, points, world2cam, focal, H, W, radius, ppp,
                      colour_origin, guidance_scale=5.0, t=None, clip_model=None, ref_rgb=None, ref_text=None):
    """One novel-view iteration of nerf/utils.py:839-894.  params = dict(colour [P,3], feat [P,16]) (nn.Parameters).

    `clip_model`, `ref_rgb`, `ref_text` are what the reference hands the guidance (utils.py:878-879): with them, a draw
    of t <= 400 takes the guidance's denoise + CLIP branch and its loss VALUE joins `loss` as `clip_loss` does there (it
    carries no gradient: the reference decodes under no_grad).  Without a CLIP model that branch cannot
    run, so t is then drawn from the SDS part of the guidance's range only instead of raising mid-training.  The
    trainer-level CLIP / contextual terms of utils.py:880-883 need weights that do not exist offline (out of
    scope)."""
    feats = torch.cat((params["colour"], params["feat"]), -1).float()
    rgb, mask = refine_render(unet, points, feats, world2cam, focal, H, W, radius, ppp)
    have_clip = clip_model is not None and ref_rgb is not None and ref_text is not None
    if t is None and not have_clip:
        lo = max(guidance.min_step, int(0.4 * guidance.num_train_timesteps) + 1)   # sd.py:153: t/1000 <= 0.4 -> CLIP branch
        t = int(torch.randint(lo, max(lo, guidance.max_step) + 1, [1]).item())
    # sd.py:171 injects the SDS gradient with latents.backward(retain_graph=True) INSIDE train_step ...
    clip_loss, _ = guidance.train_step(text_z, rgb, ref_rgb=ref_rgb, islarge=False, ref_text=ref_text,
                                       clip_model=clip_model, guidance_scale=guidance_scale, t=t)
    bg_loss = 1e-3 * (1 - rgb * (1 - mask)).sum()
    reg_loss = F.mse_loss(params["colour"], colour_origin) * 1e3
    loss = clip_loss + reg_loss + bg_loss
    # ... and the reference then zeroes the gradients before its own backward (nerf/utils.py:888-890): on the SDS
    # branch the guidance gradient never reaches the optimiser.  Kept as the reference has it - the work is the same.
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()
    return loss.detach()
