"""Stable-Diffusion-2-base-shaped guidance stand-in (random weights) for benchmarking the SDS step.

The reference's guidance (`nerf/sd.py`) loads `stabilityai/stable-diffusion-2-base` through `diffusers`;
neither the package nor the weights exist in this image (no network), and the north-star leaves this half on
stock PyTorch-ROCm.  This module rebuilds the two networks the SDS branch touches - the U-Net
(`UNet2DConditionModel`: in/out 4 ch, blocks 320/640/1280/1280, 2 res-blocks per level, cross-attention dim 1024,
head dim 64, linear transformer projections, GroupNorm 32, time embedding 1280) and the VAE encoder
(`AutoencoderKL` encoder: 128/256/512/512, 2 res-blocks per level, mid attention, 8 moment channels) - layer for
layer from memory [PARITY UNPINNED: shapes/FLOPs faithful, numerics meaningless with random weights], and restates
`StableDiffusion.train_step` (nerf/sd.py:117-174) with BOTH of its branches and `encode_imgs` / `decode_latents`
(:201-220): the SDS branch (:160-172), and - for t/1000 <= 0.4 on a view that is not `is_large` - the denoise + CLIP
branch (:153-159): one DDIM step to t-1 (diffusers `DDIMScheduler.step`, eta 0, epsilon prediction, restated from its
published formula), VAE decode, CLIP image-image and image-text similarity.  That branch runs under no_grad in the
reference (decode_latents :203) - it yields a loss VALUE and the denoised image, no gradient - so it is forward work
only; its networks (`VAEDecoderSD`, `CLIPStandIn` = ViT-B/16 as nerf/utils.py:248 loads it) are shape-faithful
random-weight stand-ins like the rest.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


# Stock-PyTorch knobs of the diffusion half, each switchable for an A/B (tools/sd_knobs.py):
#   GN_SPLIT_STATS  F.group_norm's forward statistics kernel (at::native RowwiseMomentsCUDAKernel) runs ONE workgroup per
#                   (sample, group): 32 workgroups on a 256-CU chip for the batch-1 VAE encoder - 0.88 ms for one
#                   512 x 512 x 128 activation, 150 GB/s; 6.7 ms of a step (profiles/kernel_stats_r04_bench_steps.csv).
#                   The same normalisation through torch.var_mean (a multi-workgroup reduction) and ONE fused
#                   multiply-add pass, in fp32 as autocast runs group_norm; autograd differentiates it.
#   VAE_HALF_CACHE  under autocast every fp32 conv weight of the frozen VAE encoder is cast to binary16 on every call
#                   (autocast only caches casts of tensors that require grad): keep a binary16 copy and use it whenever
#                   autocast(float16) is on - the same rounded weights, no casts.
# Measured (tools/sd_knobs.py, profiles/sd_knobs_r04.json; one box, 10 calls each): GN_SPLIT_STATS 34.1 -> 32.2 ms per
# guidance call (VAE encode 11.4 -> 9.5, backward unchanged) - ON; VAE_HALF_CACHE 34.07 -> 34.25 / 32.22 -> 32.13, i.e.
# nothing (the casts are 60 small kernels on otherwise idle CUs) - OFF, the encoder keeps the reference's fp32 weights.
#   VAE_GRAPH       the VAE encoder's forward AND backward (it is differentiated down to the rendered image every step,
#                   nerf/sd.py:171) as two captured hipGraphs inside one autograd node (torch.cuda.make_graphed_callables):
#                   ~300 + ~600 launches per step replayed instead of issued.  Fixed shape ([1, 3, 512, 512], what
#                   sd.py:124 always feeds it), frozen weights, autocast(float16) only; anything else runs eagerly.
#                   Measured: 32.10 -> 32.05 ms per guidance call (profiles/sd_knobs_r04_vae_graph.json): nothing - OFF.
GN_SPLIT_STATS = True
VAE_HALF_CACHE = False
VAE_GRAPH = False


class _SplitStatsGroupNorm(torch.autograd.Function):
    """group_norm with the forward statistics through torch.var_mean (see GN_SPLIT_STATS) and ONE fused multiply-add pass;
    the backward is ATen's own native_group_norm_backward on the saved (mean, rstd) - those kernels were never the slow
    ones (differentiating the var_mean / addcmul composition through autograd costs more passes than it saves:
    profiles/sd_knobs_r04.json, 7.4 -> 11.6 ms for the VAE encoder's backward)."""

    @staticmethod
    def forward(ctx, x, weight, bias, groups, eps):
        B, C = x.shape[0], x.shape[1]
        xf = x.float().contiguous()
        var, mean = torch.var_mean(xf.view(B, groups, -1), dim=-1, unbiased=False)                 # [B, G]
        rstd = torch.rsqrt(var + eps)
        per = C // groups
        wf, bf = weight.float(), bias.float()
        a = rstd.repeat_interleave(per, 1) * wf                                                      # [B, C]
        b = bf - mean.repeat_interleave(per, 1) * a
        shape = (B, C) + (1,) * (x.dim() - 2)
        ctx.save_for_backward(xf, mean, rstd, wf)
        ctx.meta = (groups, x.dtype, weight.dtype)
        return torch.addcmul(b.view(shape), xf, a.view(shape))

    @staticmethod
    def backward(ctx, g):
        xf, mean, rstd, wf = ctx.saved_tensors
        groups, x_dtype, w_dtype = ctx.meta
        B, C = xf.shape[0], xf.shape[1]
        need = [ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]]
        gi, gw, gb = torch.ops.aten.native_group_norm_backward(g.float().contiguous(), xf, mean, rstd, wf, B, C,
                                                               xf[0, 0].numel(), groups, need)
        return (None if gi is None else gi.to(x_dtype), None if gw is None else gw.to(w_dtype),
                None if gb is None else gb.to(w_dtype), None, None)


class GroupNorm(nn.GroupNorm):
    def forward(self, x):
        B = x.shape[0]
        if not (GN_SPLIT_STATS and x.is_cuda and B * self.num_groups <= 256 and x[0].numel() // self.num_groups >= 65536):
            return super().forward(x)
        out_dtype = torch.float32 if torch.is_autocast_enabled("cuda") else x.dtype   # autocast: group_norm runs in fp32
        with torch.autocast("cuda", enabled=False):
            return _SplitStatsGroupNorm.apply(x, self.weight, self.bias, self.num_groups, self.eps).to(out_dtype)


class ResBlock(nn.Module):
    def __init__(self, cin, cout, temb=None, eps=1e-5):
        super().__init__()
        self.norm1 = GroupNorm(32, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.time = nn.Linear(temb, cout) if temb else None
        self.norm2 = GroupNorm(32, cout, eps=eps)
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
        self.norm = GroupNorm(32, ch, eps=1e-6)
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
        self.norm_out, self.conv_out = GroupNorm(32, c), nn.Conv2d(c, 4, 3, padding=1)

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
        self.mid_norm, self.mid_attn = GroupNorm(32, c, eps=1e-6), Attention(c, c, 1)
        self.norm_out, self.conv_out = GroupNorm(32, c, eps=1e-6), nn.Conv2d(c, 8, 3, padding=1)
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


class VAEDecoderSD(nn.Module):
    """`AutoencoderKL` decoder: post_quant 4->4, conv_in 4->512, mid (res, attention, res), up blocks 512/512/256/128
    with 3 res-blocks each and a nearest-x2 + conv upsampler between them, GroupNorm + SiLU + conv_out -> 3."""

    def __init__(self, ch=(512, 512, 256, 128), layers=3):
        super().__init__()
        self.post_quant_conv = nn.Conv2d(4, 4, 1)
        c = ch[0]
        self.conv_in = nn.Conv2d(4, c, 3, padding=1)
        self.mid1, self.mid2 = ResBlock(c, c, None, eps=1e-6), ResBlock(c, c, None, eps=1e-6)
        self.mid_norm, self.mid_attn = GroupNorm(32, c, eps=1e-6), Attention(c, c, 1)
        blocks = []
        for lvl, co in enumerate(ch):
            for _ in range(layers):
                blocks.append(ResBlock(c, co, None, eps=1e-6))
                c = co
            if lvl < len(ch) - 1:
                blocks.append(nn.Conv2d(c, c, 3, padding=1))  # after nearest x2
        self.blocks = nn.ModuleList(blocks)
        self.norm_out, self.conv_out = GroupNorm(32, c, eps=1e-6), nn.Conv2d(c, 3, 3, padding=1)

    def forward(self, z):
        h = self.mid1(self.conv_in(self.post_quant_conv(z)))
        B, C, H, W = h.shape
        a = self.mid_attn(self.mid_norm(h).permute(0, 2, 3, 1).reshape(B, H * W, C))
        h = self.mid2(h + a.reshape(B, H, W, C).permute(0, 3, 1, 2))
        for b in self.blocks:
            h = b(F.interpolate(h, scale_factor=2.0, mode="nearest")) if isinstance(b, nn.Conv2d) else b(h)
        return self.conv_out(F.silu(self.norm_out(h)))


class _ClipBlock(nn.Module):
    def __init__(self, width, heads):
        super().__init__()
        self.ln_1, self.attn = nn.LayerNorm(width), nn.MultiheadAttention(width, heads, batch_first=True)
        self.ln_2 = nn.LayerNorm(width)
        self.mlp = nn.Sequential(nn.Linear(width, width * 4), nn.GELU(), nn.Linear(width * 4, width))

    def forward(self, x, mask=None):
        h = self.ln_1(x)
        x = x + self.attn(h, h, h, need_weights=False, attn_mask=mask)[0]
        return x + self.mlp(self.ln_2(x))


class CLIPStandIn(nn.Module):
    """OpenAI CLIP ViT-B/16 (`clip.load("ViT-B/16")`, nerf/utils.py:248): vision tower 224 px / patch 16 / width 768 /
    12 layers / 12 heads, text tower context 77 / width 512 / 12 layers / 8 heads / vocabulary 49408, joint dimension
    512.  `encode_image`, `encode_text` as the reference calls them; `tokenize` stands in for clip.tokenize (hashes the
    prompt's bytes into token ids - there is no BPE vocabulary offline)."""

    def __init__(self, image_size=224, patch=16, width=768, layers=12, heads=12, embed=512, ctx=77, vocab=49408,
                 text_width=512, text_layers=12, text_heads=8):
        super().__init__()
        self.conv1 = nn.Conv2d(3, width, patch, stride=patch, bias=False)
        n_tok = (image_size // patch) ** 2 + 1
        self.class_embedding = nn.Parameter(torch.randn(width) * width ** -0.5)
        self.positional_embedding_v = nn.Parameter(torch.randn(n_tok, width) * width ** -0.5)
        self.ln_pre, self.ln_post = nn.LayerNorm(width), nn.LayerNorm(width)
        self.visual = nn.ModuleList([_ClipBlock(width, heads) for _ in range(layers)])
        self.proj = nn.Parameter(torch.randn(width, embed) * width ** -0.5)
        self.ctx = ctx
        self.token_embedding = nn.Embedding(vocab, text_width)
        self.positional_embedding_t = nn.Parameter(torch.randn(ctx, text_width) * 0.01)
        self.text = nn.ModuleList([_ClipBlock(text_width, text_heads) for _ in range(text_layers)])
        self.ln_final = nn.LayerNorm(text_width)
        self.text_projection = nn.Parameter(torch.randn(text_width, embed) * text_width ** -0.5)
        self.register_buffer("causal", torch.full((ctx, ctx), float("-inf")).triu_(1), persistent=False)

    def encode_image(self, image):
        x = self.conv1(image.to(self.conv1.weight.dtype)).flatten(2).transpose(1, 2)
        x = torch.cat([self.class_embedding.expand(x.shape[0], 1, -1), x], 1) + self.positional_embedding_v
        x = self.ln_pre(x)
        for blk in self.visual:
            x = blk(x)
        return self.ln_post(x[:, 0]) @ self.proj

    def tokenize(self, prompts):
        prompts = [prompts] if isinstance(prompts, str) else prompts
        out = torch.zeros(len(prompts), self.ctx, dtype=torch.long)
        for i, p in enumerate(prompts):
            ids = [49406] + [1 + (b * 193 + j * 7) % 49000 for j, b in enumerate(p.encode()[: self.ctx - 2])] + [49407]
            out[i, : len(ids)] = torch.tensor(ids)
        return out

    def encode_text(self, tokens):
        x = self.token_embedding(tokens) + self.positional_embedding_t
        for blk in self.text:
            x = blk(x, self.causal.to(x.dtype))
        x = self.ln_final(x)
        return x[torch.arange(x.shape[0]), tokens.argmax(-1)] @ self.text_projection  # the end-of-text token's state


def ddim_step(alphas_cumprod, noise_pred, t, sample):
    """diffusers DDIMScheduler.step after set_timesteps(num_train_timesteps) (sd.py:154-155): eta 0, epsilon
    prediction, no sample clipping, set_alpha_to_one False (the Stable-Diffusion scheduler config) - one step from t to
    t - 1:  x0 = (x_t - sqrt(1 - a_t) eps) / sqrt(a_t);  x_{t-1} = sqrt(a_{t-1}) x0 + sqrt(1 - a_{t-1}) eps."""
    a_t = alphas_cumprod[t].view(-1, 1, 1, 1)
    prev = t - 1
    a_prev = torch.where(prev >= 0, alphas_cumprod[prev.clamp(min=0)], alphas_cumprod[0]).view(-1, 1, 1, 1)
    x0 = (sample - (1 - a_t).sqrt() * noise_pred) / a_t.sqrt()
    return a_prev.sqrt() * x0 + (1 - a_prev).sqrt() * noise_pred


class StableDiffusionStandIn(nn.Module):
    """`StableDiffusion` surface used by the coarse stage: `train_step(text_embeddings, pred_rgb, ...)`."""

    def __init__(self, device, step_range=(0.2, 0.6), dtype=torch.float16, seed=0, with_decoder=False,
                 unet_kw=None, vae_kw=None, decoder_kw=None, graph_unet=True):
        super().__init__()
        self.device = device
        # the frozen U-Net's forward (no_grad, fixed shapes, ~1000 kernel launches) is captured once as a hipGraph and
        # replayed: same kernels, same arithmetic, no per-launch host work (tools/step_profile.py: 13.5 -> 11.7 ms)
        self.graph_unet = bool(graph_unet)
        self._graph = None
        g = torch.random.fork_rng(devices=[])
        with g:
            torch.manual_seed(seed)
            self.unet = UNetSD2(**(unet_kw or {}))
            self.vae_encoder = VAEEncoderSD(**(vae_kw or {}))
            # only the denoise + CLIP branch (t <= 400, sd.py:153) decodes; the SDS benchmark never builds it
            self.vae_decoder = VAEDecoderSD(**(decoder_kw or {})) if with_decoder else None
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

    def _vae_encoder_for(self, x):
        """The encoder to run: its binary16 copy under autocast(float16) on the GPU (VAE_HALF_CACHE), else the fp32 one."""
        if not (VAE_HALF_CACHE and x.is_cuda and torch.is_autocast_enabled("cuda")
                and torch.get_autocast_dtype("cuda") == torch.float16):
            return self.vae_encoder
        half = self.__dict__.get("_vae_encoder_half")
        if half is None:
            import copy
            half = copy.deepcopy(self.vae_encoder).half()
            for p in half.parameters():
                p.requires_grad_(False)
            self.__dict__["_vae_encoder_half"] = half      # (not a registered sub-module: state_dict stays the fp32 model's)
        return half

    def _vae_moments(self, x):
        """vae_encoder(x): through the graphed callable (VAE_GRAPH) when x is the [1, 3, 512, 512] image under
        autocast(float16) on the GPU, eagerly otherwise."""
        enc = self._vae_encoder_for(x)
        if not (VAE_GRAPH and x.is_cuda and tuple(x.shape) == (1, 3, 512, 512) and x.dtype == torch.float32
                and torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.float16
                and not torch.cuda.is_current_stream_capturing()):
            return enc(x)
        key = (id(enc), GN_SPLIT_STATS, x.requires_grad)
        cached = self.__dict__.get("_vae_graphed")
        if cached is None or cached[0] != key:
            if self.__dict__.get("_vae_graph_failed"):
                return enc(x)
            try:
                sample = torch.rand(1, 3, 512, 512, device=x.device, requires_grad=x.requires_grad)
                with torch.autocast("cuda", dtype=torch.float16, cache_enabled=False):   # (capture needs the cast cache off)
                    graphed = torch.cuda.make_graphed_callables(enc, (sample,), allow_unused_input=True)
                cached = (key, graphed)
                self.__dict__["_vae_graphed"] = cached
            except RuntimeError as e:
                if isinstance(e, torch.cuda.OutOfMemoryError):
                    raise
                import warnings
                warnings.warn(f"VAE encoder hipGraph capture unavailable, running eagerly: {str(e).splitlines()[0]}")
                self.__dict__["_vae_graph_failed"] = True
                torch.cuda.synchronize(x.device)
                return enc(x)
        return cached[1](x)

    def encode_imgs(self, imgs):
        moments = self._vae_moments(2 * imgs - 1)
        mean, logvar = moments.chunk(2, 1)
        std = torch.exp(0.5 * logvar.clamp(-30, 20))
        return (mean + std * torch.randn_like(mean)) * 0.18215

    def decode_latents(self, latents):
        """sd.py:201-209 (no_grad, as there)."""
        if self.vae_decoder is None:
            raise RuntimeError("construct StableDiffusionStandIn(with_decoder=True) for the denoise + CLIP branch")
        with torch.no_grad():
            imgs = self.vae_decoder((1 / 0.18215 * latents).to(self.vae_decoder.conv_in.weight.dtype)).float()
        return (imgs / 2 + 0.5).clamp(0, 1)

    def aug(self, rgb):
        """sd.py:49-52: Resize((224, 224)) + CLIP normalisation."""
        x = F.interpolate(rgb, (224, 224), mode="bilinear", align_corners=False, antialias=True)
        mean = torch.tensor((0.48145466, 0.4578275, 0.40821073), device=rgb.device).view(1, 3, 1, 1)
        std = torch.tensor((0.26862954, 0.26130258, 0.27577711), device=rgb.device).view(1, 3, 1, 1)
        return (x - mean) / std

    def img_clip_loss(self, clip_model, rgb1, rgb2):
        """sd.py:97-104."""
        z1, z2 = clip_model.encode_image(self.aug(rgb1)), clip_model.encode_image(self.aug(rgb2))
        z1, z2 = z1 / z1.norm(dim=-1, keepdim=True), z2 / z2.norm(dim=-1, keepdim=True)
        return -(z1 * z2).sum(-1).mean()

    def img_text_clip_loss(self, clip_model, rgb, prompt):
        """sd.py:106-114."""
        z1 = clip_model.encode_image(self.aug(rgb))
        z1 = z1 / z1.norm(dim=-1, keepdim=True)
        zt = clip_model.encode_text(clip_model.tokenize(prompt).to(self.device))
        zt = zt / zt.norm(dim=-1, keepdim=True)
        return -(z1 * zt).sum(-1).mean()

    def _encode_view(self, pred_rgb, t):
        """sd.py:124-136: 512 x 512 resize, timestep, VAE encode (with graph), the noise draw."""
        pred_rgb_512 = F.interpolate(pred_rgb, (512, 512), mode="bilinear", align_corners=False)
        # the branch rule (sd.py:153) needs t on the host; the reference draws it on the device and pays a sync in the
        # `if` - drawn on the host here (or passed in as an int) there is none
        if t is None:
            t = int(torch.randint(self.min_step, self.max_step + 1, [1]).item())
        t_host = int(t) if not torch.is_tensor(t) else int(t.item())
        t = torch.tensor([t_host], dtype=torch.long, device=self.device)
        latents = self.encode_imgs(pred_rgb_512)
        with torch.no_grad():
            noise = torch.randn_like(latents)
        return latents, noise, t, t_host

    def _guided_eps(self, text_embeddings, latents, noise, t, guidance_scale):
        """sd.py:138-151 (no grad): add noise, U-Net on the CFG pair, guidance."""
        with torch.no_grad():
            a = self.alphas[t].view(-1, 1, 1, 1)
            noisy = a.sqrt() * latents + (1 - a).sqrt() * noise
            eps = self._unet_forward(torch.cat([noisy] * 2).to(self.unet.conv_in.weight.dtype), t,
                                     text_embeddings.to(self.unet.conv_in.weight.dtype)).float()
            eps_uncond, eps_text = eps.chunk(2)
            eps = eps_text + guidance_scale * (eps_text - eps_uncond)  # sic: anchored on eps_text (sd.py:151)
        return noisy, eps

    def _predict_noise(self, text_embeddings, pred_rgb, guidance_scale, t):
        """sd.py:124-151: 512 x 512 resize, VAE encode (with graph), noise, U-Net on the CFG pair, guidance."""
        latents, noise, t, t_host = self._encode_view(pred_rgb, t)
        noisy, eps = self._guided_eps(text_embeddings, latents, noise, t, guidance_scale)
        return latents, noise, noisy, eps, t, t_host

    def sds_gradient_async(self, text_embeddings, pred_rgb, guidance_scale=10, t=None):
        """The SDS branch with the U-Net on a second HIP stream: returns (latents with graph, finish).  The VAE encode
        and every random draw stay on the calling stream in the reference's order (sd.py:124-136); the no-grad half
        (sd.py:138-151,163-170) is queued on the side stream behind an event, so whatever the caller launches next on
        its own stream - the regulariser backward pass in mi3d/sds_step.py - runs beside it.  `finish()` makes the
        calling stream wait for the side stream and returns the SDS gradient [1,4,64,64]."""
        from . import grid_ops
        main = torch.cuda.current_stream(self.device)
        side = getattr(self, "_side_stream", None)
        if side is None:
            side = self._side_stream = torch.cuda.Stream(device=self.device)
        latents, noise, t, _ = self._encode_view(pred_rgb, t)
        lat = latents.detach()
        ready = main.record_event()
        for v in (lat, noise, t):
            v.record_stream(side)   # allocated on the calling stream, read on the side stream
        with torch.cuda.stream(side):
            side.wait_event(ready)
            box = []

            def unet_half():
                _, eps = self._guided_eps(text_embeddings, lat, noise, t, guidance_scale)
                with torch.no_grad():
                    box.append(torch.nan_to_num((1 - self.alphas[t]) * (eps - noise)))
            grid_ops._timed("sd_unet_side_stream", unet_half, 1)   # (events on the side stream)
            grad = box[0]
            done = side.record_event()

        def finish():
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(done)
            grad.record_stream(cur)     # allocated on the side stream, consumed (and freed) on the caller's
            return grad
        return latents, finish

    def _unet_forward(self, x, t, ctx):
        """U-Net noise prediction; on the GPU through a captured hipGraph (torch.cuda.CUDAGraph) keyed on the shapes
        and the autocast state, eagerly if capture is unavailable."""
        if not (getattr(self, "graph_unet", False) and x.is_cuda):  # (objects assembled without __init__: eager)
            return self.unet(x, t, encoder_hidden_states=ctx)
        key = (tuple(x.shape), x.dtype, tuple(ctx.shape), tuple(t.shape), torch.is_autocast_enabled("cuda"))
        if getattr(self, "_graph", None) is None or self._graph[0] != key:
            try:
                gx, gt, gc = x.clone(), t.clone(), ctx.clone()
                side = torch.cuda.Stream(device=x.device)
                side.wait_stream(torch.cuda.current_stream(x.device))
                with torch.cuda.stream(side):
                    for _ in range(2):
                        self.unet(gx, gt, encoder_hidden_states=gc)
                torch.cuda.current_stream(x.device).wait_stream(side)
                g = torch.cuda.CUDAGraph()
                # thread_local: a collective's watchdog thread (RCCL runs) may touch the device while this captures
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    gy = self.unet(gx, gt, encoder_hidden_states=gc)
                self._graph = (key, g, gx, gt, gc, gy)
            except RuntimeError as e:
                # capture is an optimisation, never a requirement - but only a CAPTURE problem may switch it off, and not
                # silently: anything else (a shape or dtype bug, an out-of-memory) would hide behind the eager path
                msg = str(e)
                if isinstance(e, torch.cuda.OutOfMemoryError) or not any(
                        k in msg.lower() for k in ("captur", "graph", "stream")):
                    raise
                import warnings
                warnings.warn(f"U-Net hipGraph capture unavailable, running eagerly from now on: {msg.splitlines()[0]}")
                self.graph_unet = False
                torch.cuda.synchronize(x.device)
                return self.unet(x, t, encoder_hidden_states=ctx)
        _, g, gx, gt, gc, gy = self._graph
        cur = torch.cuda.current_stream(x.device)
        # the graph's input / output buffers are shared by every caller: a replay queued on ANOTHER stream (the side
        # stream of sds_gradient_async) must have finished with them before this stream overwrites its inputs
        done = self.__dict__.get("_graph_done")
        if done is not None:
            cur.wait_event(done)
        gx.copy_(x)
        gt.copy_(t)
        gc.copy_(ctx)
        g.replay()
        out = gy.clone()   # the graph's output buffer is overwritten by the next replay
        self.__dict__["_graph_done"] = cur.record_event()
        return out

    def sds_gradient(self, text_embeddings, pred_rgb, guidance_scale=10, t=None):
        """Returns (latents [1,4,64,64] with graph, grad [1,4,64,64]) - sd.py:124-151,163-170."""
        latents, noise, _, eps, t, _ = self._predict_noise(text_embeddings, pred_rgb, guidance_scale, t)
        with torch.no_grad():
            grad = torch.nan_to_num((1 - self.alphas[t]) * (eps - noise))
        return latents, grad

    def train_step(self, text_embeddings, pred_rgb, ref_rgb=None, noise=None, islarge=False, ref_text=None,
                   clip_model=None, guidance_scale=10, t=None):
        """sd.py:117-174.  t/1000 <= 0.4 on a view that is not `islarge`: one DDIM step, decode, CLIP similarities ->
        (loss value, denoised image), nothing is back-propagated (the reference decodes under no_grad).  Otherwise the
        SDS branch: injects the gradient with latents.backward(retain_graph=True) and returns (0, None).
        `t` (an int, or a [1] long tensor) overrides the random draw - benchmarks pin the branch with it."""
        latents, noise, noisy, eps, t, t_host = self._predict_noise(text_embeddings, pred_rgb, guidance_scale, t)
        if not islarge and t_host / self.num_train_timesteps <= 0.4:
            if clip_model is None or ref_rgb is None or ref_text is None:
                raise ValueError("the denoise + CLIP branch (t <= 400) needs clip_model, ref_rgb and ref_text")
            with torch.no_grad():
                de_latents = ddim_step(self.alphas, eps, t, noisy)
            imgs = self.decode_latents(de_latents)
            loss = 10 * self.img_clip_loss(clip_model, imgs, ref_rgb) + \
                10 * self.img_text_clip_loss(clip_model, imgs, ref_text)
            return loss, imgs
        with torch.no_grad():
            grad = torch.nan_to_num((1 - self.alphas[t]) * (eps - noise))
        latents.backward(gradient=grad, retain_graph=True)
        return 0, None
