"""Refine stage (BASELINE config 5): the textured-point-cloud renderer and its deferred-rendering U-Net.

`render_point` mirrors /root/reference/nerf/refine_utils.py:306-333 - same name, arguments and result - with the two
pytorch3d calls (rasterize_points, compositing.alpha_composite) replaced by the HIP kernels of csrc/raster.hip
(C ABI Part 7); the projection in front of them is the reference's own handful of torch ops, kept verbatim in
meaning so the NDC coordinates are the ones pytorch3d would have been given.  Differentiable w.r.t. the point
features (colours + 16 learned channels), which is what the reference optimises (nerf/utils.py:826-831).

`UNet` restates nerf/unet.py:111-172 (gated convolutions, multi-scale input, bilinear upsampling) on stock torch
modules with the reference's module names, so its state_dict loads reference checkpoints.

`refine_train_step` is the inner loop of nerf/utils.py:839-894: three renders at H, H/2, H/4, the U-Net, the
full-resolution coverage-mask render + 5x5 max-pool, then - as the trainer issues them - the FRONT view's masked L1
against the reference image (:872-874) or the NOVEL view's guidance step (scale 5) + 10 x CLIP image-image similarity +
contextual loss on VGG19 relu5_4 features (:875-883), the background and colour regularisers, Adam.  The CLIP and
VGG19 weights do not exist offline: `mi3d.sd_standin.CLIPStandIn` and `VGG19Features` below are shape-faithful
random-weight stand-ins (the work and the gradient paths are the reference's, the numbers mean nothing), and
`ContextualLoss` restates the un-vendored `contextual_loss` package (S-aiueo32/contextual_loss_pytorch, imported at
nerf/utils.py:36, version not pinned by the reference) from its published functional form - PARITY UNPINNED.
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


# ------------------------------------------------------------------------------ nerf/utils.py:810,880-883: contextual loss
class VGG19Features(nn.Module):
    """torchvision VGG19 `features` up to relu5_4 (16 3x3 convolutions, 4 max-pools), as contextual_loss's VGG19 wrapper
    slices it; random weights (torchvision's ImageNet weights are not available offline), frozen."""
    CFG = (64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512)

    def __init__(self):
        super().__init__()
        layers, c = [], 3
        for v in self.CFG:
            if v == "M":
                layers.append(nn.MaxPool2d(2, 2))
            else:
                layers += [nn.Conv2d(c, v, 3, padding=1), nn.ReLU(inplace=True)]
                c = v
        self.features = nn.Sequential(*layers)
        for prm in self.parameters():
            prm.requires_grad_(False)
        self.register_buffer("mean", torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1), persistent=False)
        self.register_buffer("std", torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1), persistent=False)

    def forward(self, x):
        return self.features((x - self.mean) / self.std)


def contextual_loss(x, y, band_width=0.5):
    """contextual_loss.functional.contextual_loss(loss_type='cosine') for feature maps x, y [N,C,H,W]: cosine distances
    between all positions after centring on y's channel means, distances relative to each row's minimum, softmax-like
    affinities with bandwidth h, CX = mean over y's positions of the best affinity, loss = -log(CX)."""
    N, C = x.shape[:2]
    y_mu = y.mean(dim=(0, 2, 3), keepdim=True)
    xn = F.normalize(x - y_mu, p=2, dim=1).reshape(N, C, -1)
    yn = F.normalize(y - y_mu, p=2, dim=1).reshape(N, C, -1)
    dist = 1 - torch.bmm(xn.transpose(1, 2), yn)                       # [N, HW, HW]
    dist = dist / (dist.min(dim=2, keepdim=True)[0] + 1e-5)
    w = torch.exp((1 - dist) / band_width)
    cx = w / w.sum(dim=2, keepdim=True)
    cx = cx.max(dim=1)[0].mean(dim=1)
    return torch.mean(-torch.log(cx + 1e-5))


class ContextualLoss(nn.Module):
    """`cl.ContextualLoss(use_vgg=True, vgg_layer='relu5_4')` (nerf/utils.py:810)."""

    def __init__(self, band_width=0.5):
        super().__init__()
        self.band_width = band_width
        self.vgg = VGG19Features()

    def forward(self, x, y):
        return contextual_loss(self.vgg(x), self.vgg(y), self.band_width)


def clip_aug(rgb):
    """nerf/utils.py:323-326: Resize((224, 224)) + CLIP normalisation."""
    x = F.interpolate(rgb, (224, 224), mode="bilinear", align_corners=False, antialias=True)
    mean = torch.tensor((0.48145466, 0.4578275, 0.40821073), device=rgb.device).view(1, 3, 1, 1)
    std = torch.tensor((0.26862954, 0.26130258, 0.27577711), device=rgb.device).view(1, 3, 1, 1)
    return (x - mean) / std


def img_clip_loss(clip_model, rgb1, rgb2):
    """nerf/utils.py:434-441 - the trainer's own CLIP image-image term: it DOES back-propagate into rgb1."""
    z1, z2 = clip_model.encode_image(clip_aug(rgb1)), clip_model.encode_image(clip_aug(rgb2))
    z1, z2 = z1 / z1.norm(dim=-1, keepdim=True), z2 / z2.norm(dim=-1, keepdim=True)
    return -(z1 * z2).sum(-1).mean()


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
                      colour_origin, guidance_scale=5.0, t=None, clip_model=None, ref_rgb=None, ref_text=None,
                      is_front=False, gt_mask=None, cx_model=None):
    """One iteration of nerf/utils.py:839-894.  params = dict(colour [P,3], feat [P,16]) (nn.Parameters).

    Front view (`is_front`, utils.py:872-874): 1000 x L1 between the masked render and the masked reference image
    (`ref_rgb` [1,3,H,W], `gt_mask` [1,1,H,W]).  Novel view (:875-883): the guidance's train_step - with `clip_model`,
    `ref_rgb`, `ref_text` a draw of t <= 400 takes its denoise + CLIP branch, whose loss VALUE joins the loss (it carries no
    gradient: the reference decodes under no_grad); without a CLIP model that branch cannot run, so t is then
    drawn from the SDS part of the range instead of raising mid-training - then, when `clip_model` and `ref_rgb` are given,
    the trainer's own 10 x CLIP image-image term and, with `cx_model`, the contextual loss; both back-propagate into the
    render.  Every view: background regulariser on the dilated coverage mask, colour regulariser, Adam."""
    feats = torch.cat((params["colour"], params["feat"]), -1).float()
    rgb, mask = refine_render(unet, points, feats, world2cam, focal, H, W, radius, ppp)
    if is_front:
        if ref_rgb is None or gt_mask is None:
            raise ValueError("the front view compares with the reference image: pass ref_rgb and gt_mask")
        clip_loss = 1000 * F.l1_loss(rgb * gt_mask, ref_rgb * gt_mask)
    else:
        have_clip = clip_model is not None and ref_rgb is not None and ref_text is not None
        if t is None and not have_clip:
            lo = max(guidance.min_step, int(0.4 * guidance.num_train_timesteps) + 1)   # sd.py:153: t/1000 <= 0.4 -> CLIP branch
            t = int(torch.randint(lo, max(lo, guidance.max_step) + 1, [1]).item())
        # sd.py:171 injects the SDS gradient with latents.backward(retain_graph=True) INSIDE train_step ...
        clip_loss, _ = guidance.train_step(text_z, rgb, ref_rgb=ref_rgb, islarge=False, ref_text=ref_text,
                                           clip_model=clip_model, guidance_scale=guidance_scale, t=t)
        if clip_model is not None and ref_rgb is not None:
            clip_loss = clip_loss + 10 * img_clip_loss(clip_model, rgb, ref_rgb)          # utils.py:880
        if cx_model is not None and ref_rgb is not None:
            clip_loss = clip_loss + cx_model(rgb, ref_rgb)                                 # utils.py:881-882
    bg_loss = 1e-3 * (1 - rgb * (1 - mask)).sum()
    reg_loss = F.mse_loss(params["colour"], colour_origin) * 1e3
    loss = clip_loss + reg_loss + bg_loss
    # ... and the reference then zeroes the gradients before its own backward (nerf/utils.py:888-890): on the SDS
    # branch the guidance gradient never reaches the optimiser.  Kept as the reference has it - the work is the same.
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()
    return loss.detach()
