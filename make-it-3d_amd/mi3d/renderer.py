"""NeRFRenderer - the reference's `nerf.renderer.NeRFRenderer` surface (/root/reference/nerf/renderer.py:99-677)
on the MI355X kernels: same constructor, buffers (state_dict keys aabb_train / aabb_infer / density_grid /
density_bitfield / step_counter), attributes and `render / run / run_cuda / update_extra_state` semantics.

What differs is how the work is issued, not what is computed:
  * training `run_cuda` asks the field for the whole 13-point stencil of every sample at once
    (`self.field_stencil`) instead of 13 separate `self(...)` / `self.normal(...)` passes;
  * RNG draws happen in the reference's order (light direction, march noise, smoothness jitter) so a
    seeded run sees the same random numbers.
`export_mesh` (marching cubes + xatlas + nvdiffrast, renderer.py:142-330) is outside the hot path and not provided.
"""
import math
import weakref

import torch
import torch.nn as nn

import raymarching


def safe_normalize(x, eps=1e-20):
    # nerf/utils.py:47-48
    return x / torch.sqrt(torch.clamp(torch.sum(x * x, -1, keepdim=True), min=eps, max=1e32))


def sample_pdf(bins, weights, n_samples, det=False):
    """Inverse-CDF resampling of `bins` [B,T] by `weights` [B,T-1] (renderer.py:16-50)."""
    weights = weights + 1e-5
    pdf = weights / weights.sum(-1, keepdim=True)
    cdf = torch.cat([torch.zeros_like(pdf[..., :1]), torch.cumsum(pdf, -1)], -1)
    if det:
        u = torch.linspace(0.5 / n_samples, 1.0 - 0.5 / n_samples, steps=n_samples, device=weights.device)
        u = u.expand(*cdf.shape[:-1], n_samples)
    else:
        u = torch.rand(*cdf.shape[:-1], n_samples).to(weights.device)
    u = u.contiguous()
    hi = torch.searchsorted(cdf, u, right=True)
    lo = (hi - 1).clamp(min=0)
    hi = hi.clamp(max=cdf.shape[-1] - 1)
    c_lo, c_hi = torch.gather(cdf, -1, lo), torch.gather(cdf, -1, hi)
    b_lo, b_hi = torch.gather(bins, -1, lo), torch.gather(bins, -1, hi)
    denom = c_hi - c_lo
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    return b_lo + (u - c_lo) / denom * (b_hi - b_lo)


def near_far_from_bound(rays_o, rays_d, bound, type="cube", min_near=0.05):
    """renderer.py:52-76 (fp32 regardless of autocast)."""
    with torch.autocast("cuda", enabled=False):
        radius = rays_o.norm(dim=-1, keepdim=True)
        if type == "sphere":
            return radius - bound, radius + bound
        tmin = (-bound - rays_o) / (rays_d + 1e-15)
        tmax = (bound - rays_o) / (rays_d + 1e-15)
        near = torch.where(tmin < tmax, tmin, tmax).max(dim=-1, keepdim=True)[0]
        far = torch.where(tmin > tmax, tmin, tmax).min(dim=-1, keepdim=True)[0]
        miss = far < near
        near[miss] = 1e9
        far[miss] = 1e9
        return torch.clamp(near, min=min_near), far


# per-model buffers and captured graphs of the inference loop (NeRFRenderer._infer_state): keyed weakly by the module
_INFER_CACHES = weakref.WeakKeyDictionary()


class NeRFRenderer(nn.Module):
    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        self.bound = opt.bound
        self.cascade = 1 + math.ceil(math.log2(opt.bound))
        self.grid_size = 128
        self.cuda_ray = opt.cuda_ray
        self.min_near = opt.min_near
        self.density_thresh = opt.density_thresh
        self.bg_radius = opt.bg_radius

        box = torch.FloatTensor([-opt.bound, -opt.bound, -opt.bound, opt.bound, opt.bound, opt.bound])
        self.register_buffer("aabb_train", box)
        self.register_buffer("aabb_infer", box.clone())
        if self.cuda_ray:
            self.register_buffer("density_grid", torch.zeros([self.cascade, self.grid_size ** 3]))
            self.register_buffer("density_bitfield",
                                 torch.zeros(self.cascade * self.grid_size ** 3 // 8, dtype=torch.uint8))
            self.mean_density = 0
            self.iter_density = 0
            self.register_buffer("step_counter", torch.zeros(16, 2, dtype=torch.int32))
            self.mean_count = 0
            self.local_step = 0

    # --- the field interface subclasses provide -------------------------------------------------
    def forward(self, x, d, l=None, ratio=1, shading="albedo"):
        raise NotImplementedError()

    def density(self, x):
        raise NotImplementedError()

    def normal(self, x):
        raise NotImplementedError()

    def field_stencil(self, x, x2=None, step=0.0):
        """sigma [m], albedo [m,3], normal(x) [m,3], normal(x2) [m,3] or None - all stencil points in one pass."""
        raise NotImplementedError()

    def shade(self, albedo, normal, light_d, ratio, shading):
        raise NotImplementedError()

    def reset_extra_state(self):
        """renderer.py:144-155."""
        if not self.cuda_ray:
            return
        self.density_grid.zero_()
        self.mean_density = 0
        self.iter_density = 0
        self.step_counter.zero_()
        self.mean_count = 0
        self.local_step = 0

    def export_mesh(self, path, resolution=None, S=128):
        """renderer.py:157-330 (marching cubes + xatlas + nvdiffrast texture baking, reached only through main.py's
        `--save_mesh`): out of this path's scope (mesh export) - said loudly instead of by AttributeError."""
        raise NotImplementedError("export_mesh (--save_mesh) is outside the SDS training hot path this package covers; load "
                                  "the checkpoint into the reference's own class (nerf.network_tcnn.NeRFNetwork_reference "
                                  "under mi3d.autopatch: same state_dict keys) to export a mesh")

    # --- pure-PyTorch sampler path (BASELINE config 1) --------------------------------------------
    def run(self, rays_o, rays_d, ref_bg=None, num_steps=128, upsample_steps=128, light_d=None, ambient_ratio=1.0,
            shading="albedo", bg_color=None, perturb=False, **kwargs):
        """Uniform + importance sampling between the bounding-sphere hits, alpha compositing without early
        termination (renderer.py:332-479).  rays [B,N,3] with B == 1; bg_color [B*N,3] or None."""
        prefix = rays_o.shape[:-1]
        rays_o = rays_o.contiguous().view(-1, 3)
        rays_d = rays_d.contiguous().view(-1, 3)
        N, device = rays_o.shape[0], rays_o.device
        aabb = self.aabb_train if self.training else self.aabb_infer
        out = {}

        nears, fars = near_far_from_bound(rays_o, rays_d, self.bound, type="sphere", min_near=self.min_near)
        if light_d is None:
            light_d = safe_normalize(rays_o[0] + torch.randn(3, device=device, dtype=torch.float))

        z = nears + (fars - nears) * torch.linspace(0.0, 1.0, num_steps, device=device).unsqueeze(0)  # [N,T]
        spacing = (fars - nears) / num_steps
        if perturb:
            z = z + (torch.rand(z.shape, device=device) - 0.5) * spacing

        def positions(zv):
            p = rays_o.unsqueeze(-2) + rays_d.unsqueeze(-2) * zv.unsqueeze(-1)
            return torch.min(torch.max(p, aabb[:3]), aabb[3:])

        def weights_of(zv, sigma):
            delta = torch.cat([zv[..., 1:] - zv[..., :-1], spacing * torch.ones_like(zv[..., :1])], -1)
            alpha = 1 - torch.exp(-delta * sigma)
            trans = torch.cumprod(torch.cat([torch.ones_like(alpha[..., :1]), 1 - alpha + 1e-15], -1), -1)[..., :-1]
            return alpha * trans, delta

        xyzs = positions(z)
        dens = {k: v.view(N, num_steps, -1) for k, v in self.density(xyzs.reshape(-1, 3)).items()}

        if upsample_steps > 0:
            with torch.no_grad():
                w, delta = weights_of(z, dens["sigma"].squeeze(-1))
                mid = z[..., :-1] + 0.5 * delta[..., :-1]
                z_new = sample_pdf(mid, w[:, 1:-1], upsample_steps, det=not self.training).detach()
                xyzs_new = positions(z_new)
            dens_new = {k: v.view(N, upsample_steps, -1) for k, v in self.density(xyzs_new.reshape(-1, 3)).items()}
            z, order = torch.sort(torch.cat([z, z_new], 1), dim=1)
            xyzs = torch.cat([xyzs, xyzs_new], 1)
            xyzs = torch.gather(xyzs, 1, order.unsqueeze(-1).expand_as(xyzs))
            for k in dens:
                both = torch.cat([dens[k], dens_new[k]], 1)
                dens[k] = torch.gather(both, 1, order.unsqueeze(-1).expand_as(both))

        weights, _ = weights_of(z, dens["sigma"].squeeze(-1))
        dirs = rays_d.view(-1, 1, 3).expand_as(xyzs)
        sigmas, rgbs, normals = self(xyzs.reshape(-1, 3), dirs.reshape(-1, 3), light_d, ratio=ambient_ratio,
                                     shading=shading)
        rgbs = rgbs.view(N, -1, 3)
        normal_map = None
        if normals is not None:
            normals = normals.view(N, -1, 3)
            normal_map = torch.sum(normals * weights[:, :, None], dim=1)
            out["loss_orient"] = (weights.detach() * (normals * dirs).sum(-1).clamp(min=0) ** 2).sum(-1).mean()
            if self.opt.lambda_smooth > 0:
                jitter = self.normal(xyzs.reshape(-1, 3) + torch.randn_like(xyzs).reshape(-1, 3) * 1e-2)
                out["loss_smooth"] = (normals - jitter.view(N, -1, 3)).abs().mean()

        weights_sum = weights.sum(-1)
        depth = torch.sum(weights * z, -1)
        image = torch.sum(weights.unsqueeze(-1) * rgbs, -2)
        if bg_color is None:
            bg_color = 1
        image = image + (1 - weights_sum).unsqueeze(-1) * bg_color

        out["image"] = image.view(*prefix, 3)
        out["depth"] = depth.view(*prefix, 1)
        out["weights_sum"] = weights_sum
        out["mask"] = (nears < fars).reshape(*prefix)
        out["normal"] = normal_map
        return out

    # --- occupancy-grid path (the hot path) ------------------------------------------------------------
    def run_cuda(self, rays_o, rays_d, depth_scale=None, bg_color=None, dt_gamma=0, light_d=None, ambient_ratio=1.0,
                 shading="albedo", perturb=False, force_all_rays=False, max_steps=1024, T_thresh=1e-4, **kwargs):
        """renderer.py:481-583.  Training: march -> field (13-point stencil) -> composite + normal regularisers.
        Eval: the march/composite loop over alive rays, its round state kept on the device."""
        prefix = rays_o.shape[:-1]
        rays_o = rays_o.contiguous().view(-1, 3)
        rays_d = rays_d.contiguous().view(-1, 3)
        N, device = rays_o.shape[0], rays_o.device

        # min_near is NOT forwarded here by the reference either: near_far_from_aabb's own default 0.2 applies
        nears, fars = raymarching.near_far_from_aabb(rays_o, rays_d, self.aabb_train if self.training else self.aabb_infer)
        if light_d is None:
            light_d = safe_normalize(rays_o[0] + torch.randn(3, device=device, dtype=torch.float))
        out = {}

        if self.training:
            counter = self.step_counter[self.local_step % 16]
            counter.zero_()
            self.local_step += 1
            xyzs, dirs, deltas, rays = raymarching.march_rays_train(
                rays_o, rays_d, self.bound, self.density_bitfield, self.cascade, self.grid_size, nears, fars, counter,
                self.mean_count, perturb, 128, force_all_rays, dt_gamma, max_steps)
            smooth = self.opt.lambda_smooth > 0
            x2 = xyzs + torch.randn_like(xyzs) * 1e-2 if smooth else None
            step = 2 * math.sqrt(3) / max_steps  # dt_min: only steers the scatter's merge heuristic
            sigmas, albedo, normals, normals_jitter = self.field_stencil(xyzs, x2, step)
            rgbs = self.shade(albedo, normals, light_d, ambient_ratio, shading)
            weights_sum, depth, image = raymarching.composite_rays_train(sigmas, rgbs, deltas, rays, T_thresh)
            if normals is not None:
                w = 1 - torch.exp(-sigmas)  # "not very exact in cuda ray mode": per-sample opacity proxy
                out["loss_orient"] = (w.detach() * (normals * dirs).sum(-1).clamp(min=0) ** 2).mean()
                if smooth:
                    out["loss_smooth"] = (normals - normals_jitter).abs().mean()
        else:
            weights_sum, depth, image, normal = self._infer_loop(rays_o, rays_d, nears, fars, light_d, ambient_ratio,
                                                                 shading, perturb, dt_gamma, max_steps, T_thresh)

        if bg_color is None:
            bg_color = 1
        image = (image + (1 - weights_sum).unsqueeze(-1) * bg_color).view(*prefix, 3)
        if not self.training:
            normal = (normal + (1 - weights_sum).unsqueeze(-1) * bg_color).view(*prefix, 3)
        depth = depth + (1 - weights_sum) * self.opt.max_depth
        depth = depth.view(*prefix, 1)
        if depth_scale is not None:
            depth = depth * depth_scale.view(*prefix, 1)

        out["image"] = image
        out["depth"] = depth
        out["weights_sum"] = weights_sum.reshape(*prefix)
        out["mask"] = (nears < fars).reshape(*prefix)
        if not self.training:
            out["normal"] = normal
        return out

    # --- the inference loop (renderer.py:526-551), driven from the device ------------------------------------------
    # "budget" (default): compact rounds under a row budget - a round takes n_step = clamp(budget / n_alive, 1,
    # max_steps) steps of every alive ray, packed into one slab per ray (raymarching.*_compact_ctl): a 128 x 128 render is
    # a handful of rounds, one or two graph replays, no row of a finished ray is evaluated.  "reference": the reference's
    # own round structure (n_step = clamp(N // n_alive, 1, 8), rows at n * n_step) on the device control block, bit-
    # identical to the reference's host loop round by round - what the kernel-level parity tests run, and what renders
    # of >= 10^6 rays with a shading that looks at normals use (network_tcnn.py:159 skips the shading per BATCH of >= 10^6
    # rows, and the reference's batches are its rounds).
    infer_schedule = "budget"
    infer_budget_rows = None     # rows per round; None: min(max(16 N, 2^18), max(2^21, 2 N))
    infer_budget_rounds = 4      # rounds per captured graph (budget schedule); 0 = launched from the host, one read per round
    # rounds per captured graph of the reference schedule (0 = launch every round from the host and read the alive count
    # back every 8 rounds, the round-3 loop).  An even number: the alive / spare lists swap every round.
    infer_graph_rounds = 32
    infer_max_graphs = 6         # captured graphs kept per ray count (they share one memory pool)

    def _infer_round(self, st, n_ub, light_d, ambient_ratio, shading, dt_gamma, max_steps, T_thresh):
        """One round of the reference loop - march n_step steps for every alive ray, evaluate the field on those rows,
        composite, compact the alive list, plan the next round - launched for an UPPER BOUND n_ub of the alive count:
        every kernel (march, gather, MLP, head, composite, compaction) reads the true counts from the device control
        block and skips what lies beyond them."""
        N, align = st["N"], st["align"]
        rows_ub = min(N, 8 * n_ub)
        rows_ub += align - rows_ub % align       # >= the device's n_alive * n_step rounded past `align`
        alive, spare = (st["alive"], st["spare"]) if st["parity"] == 0 else (st["spare"], st["alive"])
        raymarching.march_rays_ctl(st["ctl"], n_ub, alive, st["rays_t"], st["rays_o"], st["rays_d"], self.bound,
                                   self.density_bitfield, self.cascade, self.grid_size, st["fars"], st["xyzs"],
                                   st["dirs"], st["deltas"], st["noises"], dt_gamma, max_steps)
        self._infer_rows = st["ctl"][2:3]        # the round's true row count, on the device (field_ops.field_rows)
        try:
            sigmas, rgbs, normals = self(st["xyzs"][:rows_ub], st["dirs"][:rows_ub], light_d, ratio=ambient_ratio,
                                         shading=shading)
        finally:
            self._infer_rows = None
        raymarching.composite_rays_ctl(st["ctl"], n_ub, alive, st["rays_t"], sigmas, rgbs, (normals + 1) / 2,
                                       st["deltas"][:rows_ub], st["weights_sum"], st["depth"], st["image"],
                                       st["normal"], T_thresh)
        raymarching.compact_alive_ctl(st["ctl"], alive, spare, N, align, max_steps)
        st["parity"] ^= 1

    def _infer_round_budget(self, st, light_d, ambient_ratio, shading, dt_gamma, max_steps, T_thresh):
        """One compact round: every alive ray marches ctl's n_step steps into its own slab (ctl[2] = the rows the round
        really has), the field runs on those rows, every ray composites its slab, the alive list is compacted and the
        next round planned - all launched for N rays and the buffers' capacity; the kernels read the counts."""
        N = st["N"]
        alive, spare = (st["alive"], st["spare"]) if st["parity"] == 0 else (st["spare"], st["alive"])
        raymarching.march_rays_compact_ctl(st["ctl"], N, alive, st["rays_t"], st["rays_o"], st["rays_d"], self.bound,
                                           self.density_bitfield, self.cascade, self.grid_size, st["fars"], st["xyzs"],
                                           st["dirs"], st["deltas"], st["ray_slab"], st["t_next"], st["noises"], dt_gamma,
                                           max_steps)
        self._infer_rows = st["ctl"][2:3]
        # network_tcnn.py:159 skips lambertian / textureless / normal shading for a BATCH of >= 10^6 rows; the reference's
        # eval batches hold at most N (+ padding) rows - the rule must see that, not this round's buffer capacity
        self._infer_shade_rows = N
        try:
            sigmas, rgbs, normals = self(st["xyzs"], st["dirs"], light_d, ratio=ambient_ratio, shading=shading)
        finally:
            self._infer_rows = None
            self._infer_shade_rows = None
        raymarching.composite_rays_compact_ctl(st["ctl"], N, alive, st["rays_t"], st["ray_slab"], st["t_next"], sigmas,
                                               rgbs, (normals + 1) / 2, st["deltas"], st["weights_sum"], st["depth"],
                                               st["image"], st["normal"], T_thresh)
        raymarching.compact_alive_ctl2(st["ctl"], alive, spare, N, max_steps)
        st["parity"] ^= 1

    def _infer_state(self, N, device, perturb, rows_cap=None):
        """Caller-owned buffers of the loop for N rays, kept across renders: a captured graph of rounds holds their
        addresses."""
        cache = _INFER_CACHES.setdefault(self, {})   # (not an attribute: copy.deepcopy(model) must not meet a CUDAGraph)
        align = 128
        rows_cap = N + 2 * align if rows_cap is None else int(rows_cap)
        key = (N, str(device), rows_cap)
        st = cache.get(key)
        if st is None:
            f32 = dict(dtype=torch.float32, device=device)
            i32 = dict(dtype=torch.int32, device=device)
            st = {"N": N, "align": align, "rows_cap": rows_cap, "graphs": {}, "pool": None,
                  "rays_o": torch.empty(N, 3, **f32), "rays_d": torch.empty(N, 3, **f32),
                  "fars": torch.empty(N, **f32), "rays_t": torch.empty(N, **f32), "light_d": torch.empty(3, **f32),
                  "xyzs": torch.zeros(rows_cap, 3, **f32), "dirs": torch.zeros(rows_cap, 3, **f32),
                  "deltas": torch.zeros(rows_cap, 2, **f32), "noise_buf": torch.zeros(N, **f32),
                  "ctl": torch.zeros(16, **i32), "alive": torch.empty(N, **i32), "spare": torch.empty(N, **i32),
                  "ray_slab": torch.zeros(N, 2, **i32), "t_next": torch.zeros(N, **f32),
                  "weights_sum": torch.zeros(N, **f32), "depth": torch.zeros(N, **f32),
                  "image": torch.zeros(N, 3, **f32), "normal": torch.zeros(N, 3, **f32)}
            cache.clear()                       # one ray count at a time: the buffers of an older one are released
            cache[key] = st
        st["noises"] = st["noise_buf"] if perturb else None
        st["parity"] = 0
        return st

    def _graph_key_tail(self, shading, ambient_ratio, dt_gamma, max_steps, T_thresh, perturb):
        """Everything a captured round bakes in besides the buffers: the scalars handed to the kernels (bound, cascade,
        grid size included - ADVICE round 4), the autocast state AND dtype, and the addresses of bitfield and weights."""
        ac = torch.is_autocast_enabled("cuda")
        return (shading, float(ambient_ratio), float(dt_gamma), int(max_steps), float(T_thresh), bool(perturb), ac,
                str(torch.get_autocast_dtype("cuda")) if ac else None, float(self.bound), int(self.cascade),
                int(self.grid_size), self.density_bitfield.data_ptr(), tuple(p.data_ptr() for p in self.parameters()))

    def _capture_rounds(self, st, key, rounds, one_round):
        """`rounds` rounds as one torch.cuda.CUDAGraph over the loop's buffers.  Two eager rounds first (real rounds: they
        advance the loop): every kernel of a round has then run in this process, so the capture meets no first-use
        initialisation; two, so the alive / spare lists keep their parity.  The graphs of one ray count share a memory
        pool (their temporaries are dead when a replay ends and replays never overlap) and at most `infer_max_graphs` of
        them are kept."""
        from . import grid_ops
        for _ in range(2):
            one_round()
        profile, grid_ops.PROFILE = grid_ops.PROFILE, None   # (no event records inside a capture)
        try:
            if st["pool"] is None:
                st["pool"] = torch.cuda.graph_pool_handle()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=st["pool"], capture_error_mode="thread_local"):
                for _ in range(rounds):
                    one_round()
        finally:
            grid_ops.PROFILE = profile
        graphs = st["graphs"]
        for k in [k for k in graphs if k[-1] != key[-1]]:     # stale: other weights / bitfield / settings
            del graphs[k]
        while len(graphs) >= max(1, int(self.infer_max_graphs)):
            del graphs[next(iter(graphs))]                     # the oldest capture goes
        graphs[key] = g
        return g

    def budget_rows(self, N):
        """Rows a compact round may hold: enough that the first round takes ~16 steps of every ray, bounded so that the
        field's planes for one round stay around 2 GB (~1 KB per row in fp32)."""
        if self.infer_budget_rows is not None:
            return max(int(self.infer_budget_rows), 1)
        return min(max(16 * N, 1 << 18), max(1 << 21, 2 * N))

    def _infer_loop(self, rays_o, rays_d, nears, fars, light_d, ambient_ratio, shading, perturb, dt_gamma, max_steps,
                    T_thresh):
        """The reference loop (renderer.py:526-551) with its per-round state - n_alive, n_step, the alive list, the row
        count - kept on the DEVICE (raymarching.*_ctl, C ABI Part 1b) and the HOST OUT OF THE LOOP: rounds are captured
        as hipGraphs (torch.cuda.CUDAGraph: march -> counted gather / MLP / head -> composite -> compaction) and
        replayed until a single end-of-replay read of the control block says no ray is alive.  Rounds that run after the
        last ray died see n_alive = 0 on the device and do nothing.  Schedules: see `infer_schedule` above."""
        N, device = rays_o.shape[0], rays_o.device
        # (perturb in eval mode - the reference's own eval / test steps never ask for it, nerf/utils.py:589,596 - shifts
        # every round after the first back by the jitter in the reference, because its composite's running t starts at
        # `near`, not at the jittered t0 (raymarching.cu:1070): a quirk of ITS round structure, kept by running it)
        budget_mode = (self.infer_schedule == "budget" and not perturb
                       and not (N + 128 >= 1_000_000 and shading != "albedo"))
        capturing = rays_o.is_cuda and torch.cuda.is_current_stream_capturing()
        stats = {"schedule": "budget" if budget_mode else "reference", "rounds_launched": 0, "host_reads": 0,
                 "graph_replays": 0, "graphs_captured": 0}
        if budget_mode:
            budget = self.budget_rows(N)
            st = self._infer_state(N, device, perturb, rows_cap=max(budget, N) + 128)
            stats["budget_rows"] = budget
        else:
            st = self._infer_state(N, device, perturb)
        st["rays_o"].copy_(rays_o)
        st["rays_d"].copy_(rays_d)
        st["fars"].copy_(fars)
        st["rays_t"].copy_(nears)
        st["light_d"].copy_(light_d.reshape(3).float())
        if perturb:
            st["noise_buf"].copy_(torch.rand(N, dtype=torch.float32, device=device))
        for k in ("weights_sum", "depth", "image", "normal"):
            st[k].zero_()
        args = (st["light_d"], float(ambient_ratio), shading, float(dt_gamma), int(max_steps), float(T_thresh))
        state = [N, 0, 0, 0, 0, 0, 0, 0]
        key_tail = self._graph_key_tail(shading, ambient_ratio, dt_gamma, max_steps, T_thresh, perturb)

        if budget_mode:
            raymarching.infer_begin2(N, device, budget, 1, int(max_steps), st["ctl"], st["alive"])
            R = int(self.infer_budget_rounds)
            one_round = lambda: self._infer_round_budget(st, *args)   # noqa: E731
            if R <= 0 or R % 2 or capturing or not rays_o.is_cuda:
                alive = N
                while alive > 0:
                    one_round()
                    stats["rounds_launched"] += 1
                    state = st["ctl"].tolist()
                    alive = state[0]
                    stats["host_reads"] += 1
            else:
                key = ("budget", budget, R, key_tail)
                alive = N
                while alive > 0:
                    g = st["graphs"].get(key)
                    if g is None:
                        g = self._capture_rounds(st, key, R, one_round)
                        stats["rounds_launched"] += 2
                        stats["graphs_captured"] += 1
                    g.replay()
                    stats["rounds_launched"] += R
                    stats["graph_replays"] += 1
                    state = st["ctl"].tolist()               # the only synchronisation: once per replay
                    alive = state[0]
                    stats["host_reads"] += 1
            if len(state) > 8 and state[8]:
                # (cannot happen with the buffers this module allocates - rows_cap >= max(budget, N); the march counts what
                #  it could not place instead of dropping it silently: C ABI version 5, ctl[8])
                raise raymarching.L.Mi3dError(f"inference march dropped {state[8]} rows: the sample buffers hold fewer rows "
                                              f"than max(budget, N)")
        else:
            raymarching.infer_begin(N, device, st["align"], st["ctl"], st["alive"])
            R = int(self.infer_graph_rounds)
            use_graph = R > 0 and R % 2 == 0 and rays_o.is_cuda and not capturing
            if not use_graph:
                sync_every = 8
                n_ub, done_lb, since_sync, next_sync = N, 0, 0, sync_every
                while n_ub > 0 and done_lb < max_steps:
                    self._infer_round(st, n_ub, *args)
                    stats["rounds_launched"] += 1
                    done_lb += max(min(N // n_ub, 8), 1)     # the device's n_step is at least this
                    since_sync += 1
                    if since_sync >= next_sync:
                        state = st["ctl"].tolist()           # the only synchronisation
                        stats["host_reads"] += 1
                        # while rays are dying quickly the host's bound goes stale quickly (and every round evaluates the
                        # field on its rows_ub rows): read the count back every round then, every `sync_every` otherwise
                        next_sync = 1 if 4 * state[0] < 3 * n_ub else sync_every
                        n_ub, done_lb, since_sync = state[0], state[3], 0
            else:
                alive = N
                while alive > 0:
                    n_ub = N
                    while n_ub // 2 >= max(alive, 1024):     # the smallest bucket N / 2^k (>= 1024) that covers the alive count
                        n_ub //= 2
                    key = ("reference", n_ub, R, key_tail)
                    g = st["graphs"].get(key)
                    if g is None:
                        g = self._capture_rounds(st, key, R, lambda: self._infer_round(st, n_ub, *args))
                        stats["rounds_launched"] += 2
                        stats["graphs_captured"] += 1
                    g.replay()
                    stats["rounds_launched"] += R
                    stats["graph_replays"] += 1
                    state = st["ctl"].tolist()               # the only synchronisation: once per R rounds
                    alive = state[0]
                    stats["host_reads"] += 1
        stats["rounds_done"] = state[4]
        self.infer_stats = stats
        return st["weights_sum"].clone(), st["depth"].clone(), st["image"].clone(), st["normal"].clone()

    @torch.no_grad()
    def update_extra_state(self, decay=0.95, S=128):
        """Density-grid refresh (renderer.py:586-639): jittered cell-centre densities per cascade, EMA-max, mean
        threshold, repack the bitfield; refresh mean_count."""
        if not self.cuda_ray:
            return
        dev = self.aabb_train.device
        fresh = -torch.ones_like(self.density_grid)
        axis = torch.arange(self.grid_size, dtype=torch.int32, device=dev)
        for xs in axis.split(S):
            for ys in axis.split(S):
                for zs in axis.split(S):
                    gx, gy, gz = torch.meshgrid(xs, ys, zs, indexing="ij")
                    coords = torch.stack([gx.reshape(-1), gy.reshape(-1), gz.reshape(-1)], -1)
                    slots = raymarching.morton3D(coords).long()
                    centres = 2 * coords.float() / (self.grid_size - 1) - 1
                    for cas in range(self.cascade):
                        span = min(2 ** cas, self.bound)
                        half = span / self.grid_size
                        pts = centres * (span - half)
                        pts += (torch.rand_like(pts) * 2 - 1) * half
                        fresh[cas, slots] = self.density(pts)["sigma"].reshape(-1).detach().float()
        live = self.density_grid >= 0
        self.density_grid[live] = torch.maximum(self.density_grid[live] * decay, fresh[live])
        self.mean_density = torch.mean(self.density_grid[live]).item()
        self.iter_density += 1
        thresh = min(self.mean_density, self.density_thresh)
        self.density_bitfield = raymarching.packbits(self.density_grid, thresh, self.density_bitfield)
        total = min(16, self.local_step)
        if total > 0:
            self.mean_count = int(self.step_counter[:total, 0].sum().item() / total)
        self.local_step = 0

    def render(self, rays_o, rays_d, depth_scale=None, staged=False, max_ray_batch=4096, **kwargs):
        """renderer.py:642-677: dispatch to run_cuda / run; `staged` chunks rays on the non-cuda_ray path only."""
        if self.cuda_ray:
            return self.run_cuda(rays_o, rays_d, depth_scale, **kwargs)
        if not staged:
            return self.run(rays_o, rays_d, depth_scale, **kwargs)
        B, N = rays_o.shape[:2]
        dev = rays_o.device
        depth = torch.empty((B, N, 1), device=dev)
        image = torch.empty((B, N, 3), device=dev)
        weights_sum = torch.empty((B, N), device=dev)
        for b in range(B):
            for head in range(0, N, max_ray_batch):
                tail = min(head + max_ray_batch, N)
                part = self.run(rays_o[b:b + 1, head:tail], rays_d[b:b + 1, head:tail], **kwargs)
                depth[b:b + 1, head:tail] = part["depth"]
                weights_sum[b:b + 1, head:tail] = part["weights_sum"]
                image[b:b + 1, head:tail] = part["image"]
        return {"depth": depth, "image": image, "weights_sum": weights_sum}
