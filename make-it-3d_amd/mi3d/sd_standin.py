"""Stable-Diffusion-2-base-shaped guidance stand-in (random weights) for benchmarking the SDS step.

The reference's guidance (`nerf/sd.py`) loads `stabilityai/stable-diffusion-2-base` through `diffusers`;
neither the package nor the weights exist in this image (no network), and the north-star leaves this half on
stock PyTorch-ROCm.  This module rebuilds the two networks the SDS branch touches - the U-Net
(`UNet2DConditionModel`: in/out 4 ch, blocks 320/640/1280/1280, 2 res-blocks per level, cross-attention dim 1024,
head dim 64, linear transformer projections, GroupNorm 32, time embedding 1280) and the VAE encoder
(`AutoencoderKL` encoder: 128/256/512/512, 2 res-blocks per level, mid attention, 8 moment channels) - layer for
layer from memory [PARITY UNPINNED: shapes/FLOPs faithful, numerics meaningless with random weights], and restates
`StableDiffusion.train_step`'s SDS branch (nerf/sd.py:117-151,163-172) and `encode_imgs` (:212-220).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


class ResBlock(nn.Module):
    def __init__(self, cin, cout, temb=None, eps=1e-5):
        super().__init__()
        self.norm1 = nn.GroupNorm(32, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.time = nn.Linear(temb, cout) if temb else None
        self.norm2 = nn.GroupNorm(32, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.skip = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x, temb=None):
        h = self.conv1(F.silu(self.norm1(x)))
        if self.time is not None:
            h = h + self.time(F.silu(temb))[:, :, None, None]
        h = self.conv2(F.silu(self.norm2(h)))
        return (x if self.skip is None else self.skip(x)) + h


class Attention(nn.Module):
    def __init__(self, dim, ctx_dim, heads):
        super().__init__()
        self.heads = heads
        self.q = nn.Linear(dim, dim, bias=False)
        self.k = nn.Linear(ctx_dim, dim, bias=False)
        self.v = nn.Linear(ctx_dim, dim, bias=False)
        self.o = nn.Linear(dim, dim)

    def forward(self, x, ctx=None):
        ctx = x if ctx is None else ctx
        B, N, C = x.shape
        q = self.q(x).view(B, N, self.heads, -1).transpose(1, 2)
        k = self.k(ctx).view(B, ctx.shape[1], self.heads, -1).transpose(1, 2)
        v = self.v(ctx).view(B, ctx.shape[1], self.heads, -1).transpose(1, 2)
        o = F.scaled_dot_product_attention(q, k, v)
        return self.o(o.transpose(1, 2).reshape(B, N, C))


class TransformerBlock(nn.Module):
    def __init__(self, dim, ctx_dim, heads):
        super().__init__()
        self.n1, self.a1 = nn.LayerNorm(dim), Attention(dim, dim, heads)
        self.n2, self.a2 = nn.LayerNorm(dim), Attention(dim, ctx_dim, heads)
        self.n3 = nn.LayerNorm(dim)
        self.ff_in = nn.Linear(dim, dim * 8)  # GEGLU: 2 x 4*dim
        self.ff_out = nn.Linear(dim * 4, dim)

    def forward(self, x, ctx):
        x = x + self.a1(self.n1(x))
        x = x + self.a2(self.n2(x), ctx)
        h, gate = self.ff_in(self.n3(x)).chunk(2, -1)
        return x + self.ff_out(h * F.gelu(gate))


class SpatialTransformer(nn.Module):
    def __init__(self, ch, ctx_dim, head_dim=64):
        super().__init__()
        self.norm = nn.GroupNorm(32, ch, eps=1e-6)
        self.proj_in = nn.Linear(ch, ch)
        self.block = TransformerBlock(ch, ctx_dim, ch // head_dim)
        self.proj_out = nn.Linear(ch, ch)

    def forward(self, x, ctx):
        B, C, H, W = x.shape
        h = self.norm(x).permute(0, 2, 3, 1).reshape(B, H * W, C)
        h = self.proj_out(self.block(self.proj_in(h), ctx))
        return x + h.reshape(B, H, W, C).permute(0, 3, 1, 2)


class UNetSD2(nn.Module):
    def __init__(self, ch=(320, 640, 1280, 1280), ctx_dim=1024, layers=2):
        super().__init__()
        temb = ch[0] * 4
        self.ch0 = ch[0]
        self.time1, self.time2 = nn.Linear(ch[0], temb), nn.Linear(temb, temb)
        self.conv_in = nn.Conv2d(4, ch[0], 3, padding=1)
        self.down = nn.ModuleList()
        skips, c = [ch[0]], ch[0]
        for lvl, co in enumerate(ch):
            for _ in range(layers):
                self.down.append(nn.ModuleList([ResBlock(c, co, temb),
                                                SpatialTransformer(co, ctx_dim) if lvl < 3 else nn.Identity()]))
                c = co
                skips.append(c)
            if lvl < len(ch) - 1:
                self.down.append(nn.ModuleList([nn.Conv2d(c, c, 3, stride=2, padding=1)]))
                skips.append(c)
        self.mid = nn.ModuleList([ResBlock(c, c, temb), SpatialTransformer(c, ctx_dim), ResBlock(c, c, temb)])
        self.up = nn.ModuleList()
        for lvl, co in reversed(list(enumerate(ch))):
            for i in range(layers + 1):
                blk = [ResBlock(c + skips.pop(), co, temb), SpatialTransformer(co, ctx_dim) if lvl < 3 else nn.Identity()]
                c = co
                if i == layers and lvl > 0:
                    blk.append(nn.Conv2d(c, c, 3, padding=1))  # after nearest x2 upsample
                self.up.append(nn.ModuleList(blk))
        self.norm_out, self.conv_out = nn.GroupNorm(32, c), nn.Conv2d(c, 4, 3, padding=1)

    def forward(self, x, t, encoder_hidden_states):
        half = self.ch0 // 2
        freqs = torch.exp(-math.log(10000) * torch.arange(half, device=x.device, dtype=torch.float32) / half)
        ang = t.float().view(-1, 1).expand(x.shape[0], 1) * freqs[None]
        temb = torch.cat([torch.cos(ang), torch.sin(ang)], -1).to(x.dtype)
        temb = self.time2(F.silu(self.time1(temb)))
        h = self.conv_in(x)
        hs = [h]
        for blk in self.down:
            if len(blk) == 1:
                h = blk[0](h)
            else:
                h = blk[0](h, temb)
                h = blk[1](h, encoder_hidden_states) if isinstance(blk[1], SpatialTransformer) else h
            hs.append(h)
        h = self.mid[2](self.mid[1](self.mid[0](h, temb), encoder_hidden_states), temb)
        for blk in self.up:
            h = blk[0](torch.cat([h, hs.pop()], 1), temb)
            h = blk[1](h, encoder_hidden_states) if isinstance(blk[1], SpatialTransformer) else h
            if len(blk) == 3:
                h = blk[2](F.interpolate(h, scale_factor=2.0, mode="nearest"))
        return self.conv_out(F.silu(self.norm_out(h)))


class VAEEncoderSD(nn.Module):
    def __init__(self, ch=(128, 256, 512, 512), layers=2):
        super().__init__()
        self.conv_in = nn.Conv2d(3, ch[0], 3, padding=1)
        blocks, c = [], ch[0]
        for lvl, co in enumerate(ch):
            for _ in range(layers):
                blocks.append(ResBlock(c, co, None, eps=1e-6))
                c = co
            if lvl < len(ch) - 1:
                blocks.append(nn.Conv2d(c, c, 3, stride=2, padding=0))
        self.blocks = nn.ModuleList(blocks)
        self.mid1, self.mid2 = ResBlock(c, c, None, eps=1e-6), ResBlock(c, c, None, eps=1e-6)
        self.mid_norm, self.mid_attn = nn.GroupNorm(32, c, eps=1e-6), Attention(c, c, 1)
        self.norm_out, self.conv_out = nn.GroupNorm(32, c, eps=1e-6), nn.Conv2d(c, 8, 3, padding=1)
        self.quant_conv = nn.Conv2d(8, 8, 1)

    def forward(self, x):
        h = self.conv_in(x)
        for b in self.blocks:
            if isinstance(b, nn.Conv2d):
                h = b(F.pad(h, (0, 1, 0, 1)))
            else:
                h = b(h)
        h = self.mid1(h)
        B, C, H, W = h.shape
        a = self.mid_attn(self.mid_norm(h).permute(0, 2, 3, 1).reshape(B, H * W, C))
        h = self.mid2(h + a.reshape(B, H, W, C).permute(0, 3, 1, 2))
        return self.quant_conv(self.conv_out(F.silu(self.norm_out(h))))  # [B, 8, H/8, W/8] = (mean, logvar)


class StableDiffusionStandIn(nn.Module):
    """`StableDiffusion` surface used by the coarse stage: `train_step(text_embeddings, pred_rgb, ...)`."""

    def __init__(self, device, step_range=(0.2, 0.6), dtype=torch.float16, seed=0):
        super().__init__()
        self.device = device
        g = torch.random.fork_rng(devices=[])
        with g:
            torch.manual_seed(seed)
            self.unet = UNetSD2()
            self.vae_encoder = VAEEncoderSD()
        self.to(device)
        self.unet.to(dtype)  # frozen, weights kept in half like an inference deployment; VAE stays fp32 (has grad)
        for p in self.parameters():
            p.requires_grad_(False)
        self.num_train_timesteps = 1000
        self.min_step = int(self.num_train_timesteps * step_range[0])
        self.max_step = int(self.num_train_timesteps * step_range[1])
        betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float32) ** 2  # scaled_linear
        self.register_buffer("alphas", torch.cumprod(1 - betas, 0).to(device), persistent=False)

    def get_text_embeds(self, prompt=None, negative_prompt=None):
        """[2,77,1024] (uncond first, nerf/sd.py:84); random - there is no text tower offline."""
        g = torch.Generator(device="cpu").manual_seed(1)
        return torch.randn(2, 77, 1024, generator=g).to(self.device)

    def encode_imgs(self, imgs):
        moments = self.vae_encoder(2 * imgs - 1)
        mean, logvar = moments.chunk(2, 1)
        std = torch.exp(0.5 * logvar.clamp(-30, 20))
        return (mean + std * torch.randn_like(mean)) * 0.18215

    def sds_gradient(self, text_embeddings, pred_rgb, guidance_scale=10, t=None):
        """Returns (latents [1,4,64,64] with graph, grad [1,4,64,64]) - sd.py:124-151,163-170."""
        pred_rgb_512 = F.interpolate(pred_rgb, (512, 512), mode="bilinear", align_corners=False)
        if t is None:
            t = torch.randint(self.min_step, self.max_step + 1, [1], dtype=torch.long, device=self.device)
        latents = self.encode_imgs(pred_rgb_512)
        with torch.no_grad():
            noise = torch.randn_like(latents)
            a = self.alphas[t].view(-1, 1, 1, 1)
            noisy = a.sqrt() * latents + (1 - a).sqrt() * noise
            eps = self.unet(torch.cat([noisy] * 2).to(self.unet.conv_in.weight.dtype), t,
                            encoder_hidden_states=text_embeddings.to(self.unet.conv_in.weight.dtype)).float()
            eps_uncond, eps_text = eps.chunk(2)
            eps = eps_text + guidance_scale * (eps_text - eps_uncond)  # sic: anchored on eps_text (sd.py:151)
            grad = torch.nan_to_num((1 - self.alphas[t]) * (eps - noise))
        return latents, grad

    def train_step(self, text_embeddings, pred_rgb, ref_rgb=None, noise=None, islarge=False, ref_text=None,
                   clip_model=None, guidance_scale=10, t=None):
        """SDS branch of sd.py:117-174: injects the gradient with latents.backward(retain_graph=True)."""
        latents, grad = self.sds_gradient(text_embeddings, pred_rgb, guidance_scale, t)
        latents.backward(gradient=grad, retain_graph=True)
        return 0, None
