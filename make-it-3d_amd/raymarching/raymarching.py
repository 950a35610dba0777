"""`raymarching` - drop-in for the reference's operator package of the same name
(/root/reference/raymarching/raymarching.py), backed by the hand-written gfx950 kernels in
csrc/raymarching.hip through the C ABI of include/mi3d.h.

Same callables, same argument order/defaults, same return shapes and the same quirks:
  * `march_rays_train` pads the returned sample count UP PAST the next multiple of `align`
    (a full extra `align` when already aligned) - reference raymarching.py:237-238;
  * `composite_rays_train.backward` ignores grad_depth - reference raymarching.py:287;
  * inputs are cast to fp32 under autocast (custom_fwd(cast_inputs=float32)).
Differences (none observable through the API):
  * no 537 MB zero fill of the [N*max_steps] sample buffers - only the <= `align` padding rows are
    zeroed, on device; no torch.cuda.empty_cache();
  * `rays` rows come back in ray order (rays[n,0] == n); slabs are reserved with one atomic per
    64 rays.  The reference's order is atomic-arrival order, i.e. unspecified.
There is no CPU path: tensors are moved to the GPU like the reference does, and a missing
libmi3d.so raises.
"""
import torch
from torch.autograd import Function

from mi3d import _lib as L

_f32 = torch.float32


def _custom_fwd(fn):
    return torch.amp.custom_fwd(fn, device_type="cuda", cast_inputs=_f32)


def _custom_bwd(fn):
    return torch.amp.custom_bwd(fn, device_type="cuda")


def _rays(rays_o, rays_d):
    if not rays_o.is_cuda:
        rays_o = rays_o.cuda()
    if not rays_d.is_cuda:
        rays_d = rays_d.cuda()
    rays_o = L.dev_f32(rays_o.contiguous().view(-1, 3), "rays_o")
    rays_d = L.dev_f32(rays_d.contiguous().view(-1, 3), "rays_d")
    if rays_o.shape != rays_d.shape:
        raise L.Mi3dError("rays_o / rays_d shape mismatch")
    return rays_o, rays_d


def _aligned16(t):
    return t if t.data_ptr() % 16 == 0 else t.clone()


# ---------------------------------------------------------------------------- utils


class _near_far_from_aabb(Function):
    @staticmethod
    @_custom_fwd
    def forward(ctx, rays_o, rays_d, aabb, min_near=0.2):
        """rays_o/rays_d [N,3], aabb [6] = (xmin,ymin,zmin,xmax,ymax,zmax) -> nears [N], fars [N]."""
        rays_o, rays_d = _rays(rays_o, rays_d)
        aabb = L.dev_f32(aabb.to(rays_o.device, _f32).contiguous(), "aabb")
        N = rays_o.shape[0]
        nears = torch.empty(N, dtype=_f32, device=rays_o.device)
        fars = torch.empty(N, dtype=_f32, device=rays_o.device)
        L.launch("mi3d_near_far_from_aabb", rays_o, L.ptr(rays_o), L.ptr(rays_d), L.ptr(aabb), N, float(min_near),
               L.ptr(nears), L.ptr(fars))
        return nears, fars


near_far_from_aabb = _near_far_from_aabb.apply


class _sph_from_ray(Function):
    @staticmethod
    @_custom_fwd
    def forward(ctx, rays_o, rays_d, radius):
        """(theta, phi) in [-1,1]^2 of the far intersection with the sphere of `radius`: coords [N,2]."""
        rays_o, rays_d = _rays(rays_o, rays_d)
        N = rays_o.shape[0]
        coords = torch.empty(N, 2, dtype=_f32, device=rays_o.device)
        L.launch("mi3d_sph_from_ray", rays_o, L.ptr(rays_o), L.ptr(rays_d), float(radius), N, L.ptr(coords))
        return coords


sph_from_ray = _sph_from_ray.apply


class _morton3D(Function):
    @staticmethod
    def forward(ctx, coords):
        """coords int32 [N,3] in [0,1024) -> z-order indices int32 [N]."""
        if not coords.is_cuda:
            coords = coords.cuda()
        coords = coords.int().contiguous()
        N = coords.shape[0]
        indices = torch.empty(N, dtype=torch.int32, device=coords.device)
        L.launch("mi3d_morton3D", coords, L.ptr(coords), N, L.ptr(indices))
        return indices


morton3D = _morton3D.apply


class _morton3D_invert(Function):
    @staticmethod
    def forward(ctx, indices):
        """indices int32 [N] -> coords int32 [N,3]."""
        if not indices.is_cuda:
            indices = indices.cuda()
        indices = indices.int().contiguous()
        N = indices.shape[0]
        coords = torch.empty(N, 3, dtype=torch.int32, device=indices.device)
        L.launch("mi3d_morton3D_invert", indices, L.ptr(indices), N, L.ptr(coords))
        return coords


morton3D_invert = _morton3D_invert.apply


class _packbits(Function):
    @staticmethod
    @_custom_fwd
    def forward(ctx, grid, thresh, bitfield=None):
        """grid float [C, H^3] -> bitfield uint8 [C*H^3/8]; bit i of byte n = grid.flat[8n+i] > thresh."""
        if not grid.is_cuda:
            grid = grid.cuda()
        grid = _aligned16(L.dev_f32(grid.contiguous(), "grid"))
        C_, H3 = grid.shape[0], grid.shape[1]
        N = C_ * H3 // 8
        if bitfield is None:
            bitfield = torch.empty(N, dtype=torch.uint8, device=grid.device)
        L.dev_typed(bitfield, "bitfield", torch.uint8)
        L.launch("mi3d_packbits", grid, L.ptr(grid), N, float(thresh), L.ptr(bitfield))
        return bitfield


packbits = _packbits.apply

# ---------------------------------------------------------------------------- training


class _march_rays_train(Function):
    @staticmethod
    @_custom_fwd
    def forward(ctx, rays_o, rays_d, bound, density_bitfield, C, H, nears, fars, step_counter=None, mean_count=-1,
                perturb=False, align=-1, force_all_rays=False, dt_gamma=0, max_steps=1024):
        """March rays through the cascaded occupancy bitfield.
        Returns xyzs [m,3], dirs [m,3], deltas [m,2] (dt, t-advance), rays int32 [N,3] = (ray, offset, count)."""
        rays_o, rays_d = _rays(rays_o, rays_d)
        if not density_bitfield.is_cuda:
            density_bitfield = density_bitfield.cuda()
        bits = L.dev_typed(density_bitfield.contiguous(), "density_bitfield", torch.uint8)
        dev = rays_o.device
        N = rays_o.shape[0]
        M = N * max_steps
        exact = force_all_rays or mean_count <= 0  # sample count is read back and the outputs are sliced
        if not exact:
            if align > 0:
                mean_count += align - mean_count % align
            M = mean_count
        if bits.numel() * 8 < C * H * H * H:
            raise L.Mi3dError("density_bitfield smaller than C*H^3 bits")

        # exact mode: every row below the final count is written by the kernel, the <= align rows above it are
        # zeroed on device -> no full zero fill.  mean_count mode keeps the reference's zeros (dropped rays).
        alloc = torch.empty if exact else torch.zeros
        xyzs = alloc(M, 3, dtype=_f32, device=dev)
        dirs = alloc(M, 3, dtype=_f32, device=dev)
        deltas = alloc(M, 2, dtype=_f32, device=dev)
        rays = torch.empty(N, 3, dtype=torch.int32, device=dev)
        if step_counter is None:
            step_counter = torch.zeros(2, dtype=torch.int32, device=dev)
        L.dev_typed(step_counter, "step_counter", torch.int32)
        nears, fars = L.dev_f32(nears.contiguous(), "nears"), L.dev_f32(fars.contiguous(), "fars")
        noises = torch.rand(N, dtype=_f32, device=dev) if perturb else torch.zeros(N, dtype=_f32, device=dev)

        L.launch("mi3d_march_rays_train", rays_o, L.ptr(rays_o), L.ptr(rays_d), L.ptr(bits), float(bound), float(dt_gamma),
               int(max_steps), N, int(C), int(H), M, L.ptr(nears), L.ptr(fars), L.ptr(xyzs), L.ptr(dirs),
               L.ptr(deltas), L.ptr(rays), L.ptr(step_counter), L.ptr(noises))
        if exact:
            a = align if align > 0 else 1
            L.launch("mi3d_march_zero_tail", xyzs, L.ptr(step_counter), a, M, L.ptr(xyzs), L.ptr(dirs), L.ptr(deltas))
            m = int(step_counter[0].item())  # the op's contract returns sliced tensors: one D2H sync
            if align > 0:
                m += align - m % align
            xyzs, dirs, deltas = xyzs[:m], dirs[:m], deltas[:m]
        return xyzs, dirs, deltas, rays


march_rays_train = _march_rays_train.apply


def _composite_train_cls(fwd_name, bwd_name):
    class _composite(Function):
        @staticmethod
        @_custom_fwd
        def forward(ctx, sigmas, rgbs, deltas, rays, T_thresh=1e-4):
            """sigmas [M], rgbs [M,3], deltas [M,2], rays int32 [N,3] -> weights_sum [N], depth [N], image [N,3]."""
            sigmas = L.dev_f32(sigmas.contiguous(), "sigmas")
            rgbs = L.dev_f32(rgbs.contiguous(), "rgbs", 3)
            deltas = L.dev_f32(deltas.contiguous(), "deltas", 2)
            rays = L.dev_typed(rays.contiguous(), "rays", torch.int32)
            M, N = sigmas.shape[0], rays.shape[0]
            if rgbs.shape[0] != M or deltas.shape[0] != M:
                raise L.Mi3dError("sigmas / rgbs / deltas row counts differ")
            dev = sigmas.device
            weights_sum = torch.empty(N, dtype=_f32, device=dev)
            depth = torch.empty(N, dtype=_f32, device=dev)
            image = torch.empty(N, 3, dtype=_f32, device=dev)
            L.launch(fwd_name, sigmas, L.ptr(sigmas), L.ptr(rgbs), L.ptr(deltas), L.ptr(rays), M, N, float(T_thresh),
                   L.ptr(weights_sum), L.ptr(depth), L.ptr(image))
            ctx.save_for_backward(sigmas, rgbs, deltas, rays, weights_sum, depth, image)
            ctx.dims = [M, N, T_thresh]
            return weights_sum, depth, image

        @staticmethod
        @_custom_bwd
        def backward(ctx, grad_weights_sum, grad_depth, grad_image):
            # grad_depth is dropped, as in the reference (raymarching.py:287)
            grad_weights_sum = grad_weights_sum.contiguous()
            grad_image = grad_image.contiguous()
            sigmas, rgbs, deltas, rays, weights_sum, depth, image = ctx.saved_tensors
            M, N, T_thresh = ctx.dims
            grad_sigmas = torch.zeros_like(sigmas)  # samples past a ray's termination keep zero gradient
            grad_rgbs = torch.zeros_like(rgbs)
            L.launch(bwd_name, sigmas, L.ptr(grad_weights_sum), L.ptr(grad_image), L.ptr(sigmas), L.ptr(rgbs), L.ptr(deltas),
                   L.ptr(rays), L.ptr(weights_sum), L.ptr(image), M, N, float(T_thresh), L.ptr(grad_sigmas),
                   L.ptr(grad_rgbs))
            return grad_sigmas, grad_rgbs, None, None, None

    return _composite


_composite_rays_train = _composite_train_cls("mi3d_composite_rays_train_forward", "mi3d_composite_rays_train_backward")
composite_rays_train = _composite_rays_train.apply
_composite_sdf_rays_train = _composite_train_cls("mi3d_composite_sdf_rays_train_forward",
                                                 "mi3d_composite_sdf_rays_train_backward")
composite_sdf_rays_train = _composite_sdf_rays_train.apply

# ---------------------------------------------------------------------------- inference


class _march_rays(Function):
    @staticmethod
    @_custom_fwd
    def forward(ctx, n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, density_bitfield, C, H, near, far,
                align=-1, perturb=False, dt_gamma=0, max_steps=1024):
        """March up to n_step occupied steps for each alive ray: xyzs/dirs [M,3], deltas [M,2], M = n_alive*n_step
        (padded past the next multiple of `align`)."""
        rays_o, rays_d = _rays(rays_o, rays_d)
        dev = rays_o.device
        M = n_alive * n_step
        if align > 0:
            M += align - (M % align)
        xyzs = torch.zeros(M, 3, dtype=_f32, device=dev)  # rows past a ray's end must read delta == 0
        dirs = torch.zeros(M, 3, dtype=_f32, device=dev)
        deltas = torch.zeros(M, 2, dtype=_f32, device=dev)
        noises = (torch.rand if perturb else torch.zeros)(n_alive, dtype=_f32, device=dev)
        bits = L.dev_typed(density_bitfield.contiguous(), "density_bitfield", torch.uint8)
        L.launch("mi3d_march_rays", rays_o, int(n_alive), int(n_step), L.ptr(L.dev_typed(rays_alive, "rays_alive", torch.int32)),
               L.ptr(L.dev_f32(rays_t, "rays_t")), L.ptr(rays_o), L.ptr(rays_d), float(bound), float(dt_gamma),
               int(max_steps), int(C), int(H), L.ptr(bits), L.ptr(L.dev_f32(near, "near")),
               L.ptr(L.dev_f32(far, "far")), L.ptr(xyzs), L.ptr(dirs), L.ptr(deltas), L.ptr(noises))
        return xyzs, dirs, deltas


march_rays = _march_rays.apply


class _composite_rays(Function):
    @staticmethod
    @_custom_fwd
    def forward(ctx, n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, normals, deltas, weights_sum, depth, image,
                normal, T_thresh=1e-2):
        """In-place accumulation into weights_sum/depth/image/normal; finished rays get rays_alive = -1."""
        sigmas = L.dev_f32(sigmas.float().contiguous(), "sigmas")
        rgbs = L.dev_f32(rgbs.float().contiguous(), "rgbs")
        normals = L.dev_f32(normals.float().contiguous(), "normals")
        L.launch("mi3d_composite_rays", sigmas, int(n_alive), int(n_step), float(T_thresh),
               L.ptr(L.dev_typed(rays_alive, "rays_alive", torch.int32)), L.ptr(L.dev_f32(rays_t, "rays_t")),
               L.ptr(sigmas), L.ptr(rgbs), L.ptr(normals), L.ptr(L.dev_f32(deltas, "deltas")),
               L.ptr(L.dev_f32(weights_sum, "weights_sum")), L.ptr(L.dev_f32(depth, "depth")),
               L.ptr(L.dev_f32(image, "image")), L.ptr(L.dev_f32(normal, "normal")))
        return tuple()


composite_rays = _composite_rays.apply


class _composite_sdf_rays(Function):
    @staticmethod
    @_custom_fwd
    def forward(ctx, n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image,
                T_thresh=1e-2):
        sigmas = L.dev_f32(sigmas.float().contiguous(), "sigmas")
        rgbs = L.dev_f32(rgbs.float().contiguous(), "rgbs")
        L.launch("mi3d_composite_sdf_rays", sigmas, int(n_alive), int(n_step), float(T_thresh),
               L.ptr(L.dev_typed(rays_alive, "rays_alive", torch.int32)), L.ptr(L.dev_f32(rays_t, "rays_t")),
               L.ptr(sigmas), L.ptr(rgbs), L.ptr(L.dev_f32(deltas, "deltas")),
               L.ptr(L.dev_f32(weights_sum, "weights_sum")), L.ptr(L.dev_f32(depth, "depth")),
               L.ptr(L.dev_f32(image, "image")))
        return tuple()


composite_sdf_rays = _composite_sdf_rays.apply


# ---------------------------------------------------------------------------- inference loop driven from the device
# Extensions beyond the reference's op surface (C ABI Part 1b): the state the reference's eval loop keeps on the host
# (n_alive, n_step; nerf/renderer.py:526-551) lives in a device control block, so a round needs no synchronisation.
# mi3d.renderer.NeRFRenderer.run_cuda uses them; the reference's own loop keeps working on march_rays / composite_rays.


def infer_begin(N, device, align=128, ctl=None, rays_alive=None):
    """(ctl int32[8], rays_alive int32[N] = 0..N-1) for a fresh loop over N rays; `ctl` / `rays_alive`: re-initialise
    these caller-owned buffers instead of allocating (a captured graph of rounds keeps their addresses)."""
    if ctl is None:
        ctl = torch.zeros(8, dtype=torch.int32, device=device)
    else:
        L.dev_typed(ctl, "ctl", torch.int32).zero_()
    if rays_alive is None:
        rays_alive = torch.empty(N, dtype=torch.int32, device=device)
    L.dev_typed(rays_alive, "rays_alive", torch.int32)
    L.launch("mi3d_infer_begin", ctl, L.ptr(ctl), L.ptr(rays_alive), int(N), int(max(align, 0)))
    return ctl, rays_alive


def march_rays_ctl(ctl, n_alive_max, rays_alive, rays_t, rays_o, rays_d, bound, density_bitfield, C, H, fars, xyzs, dirs,
                   deltas, noises=None, dt_gamma=0, max_steps=1024):
    """march_rays for the device-side n_alive / n_step of `ctl`, into caller-owned buffers of >= N + align rows."""
    rays_o, rays_d = _rays(rays_o, rays_d)
    bits = L.dev_typed(density_bitfield, "density_bitfield", torch.uint8)
    L.launch("mi3d_march_rays_ctl", rays_o, L.ptr(L.dev_typed(ctl, "ctl", torch.int32)), int(n_alive_max),
             L.ptr(L.dev_typed(rays_alive, "rays_alive", torch.int32)), L.ptr(L.dev_f32(rays_t, "rays_t")), L.ptr(rays_o),
             L.ptr(rays_d), float(bound), float(dt_gamma), int(max_steps), int(C), int(H), L.ptr(bits),
             L.ptr(L.dev_f32(fars, "fars")), L.ptr(L.dev_f32(xyzs, "xyzs", 3)), L.ptr(L.dev_f32(dirs, "dirs", 3)),
             L.ptr(L.dev_f32(deltas, "deltas", 2)), L.ptr(noises))


def composite_rays_ctl(ctl, n_alive_max, rays_alive, rays_t, sigmas, rgbs, normals, deltas, weights_sum, depth, image,
                       normal, T_thresh=1e-2):
    sigmas = L.dev_f32(sigmas.float().contiguous(), "sigmas")
    rgbs = L.dev_f32(rgbs.float().contiguous(), "rgbs")
    normals = L.dev_f32(normals.float().contiguous(), "normals")
    L.launch("mi3d_composite_rays_ctl", sigmas, L.ptr(ctl), int(n_alive_max), float(T_thresh),
             L.ptr(L.dev_typed(rays_alive, "rays_alive", torch.int32)), L.ptr(L.dev_f32(rays_t, "rays_t")), L.ptr(sigmas),
             L.ptr(rgbs), L.ptr(normals), L.ptr(L.dev_f32(deltas, "deltas")), L.ptr(L.dev_f32(weights_sum, "weights_sum")),
             L.ptr(L.dev_f32(depth, "depth")), L.ptr(L.dev_f32(image, "image")), L.ptr(L.dev_f32(normal, "normal")))


def compact_alive_ctl(ctl, rays_alive_in, rays_alive_out, N, align=128, max_steps=1024):
    """rays_alive_out[:n'] = the entries >= 0 of rays_alive_in[:n_alive] (order kept); ctl advanced to the next round."""
    L.launch("mi3d_compact_alive_ctl", ctl, L.ptr(ctl), L.ptr(L.dev_typed(rays_alive_in, "rays_alive_in", torch.int32)),
             L.ptr(L.dev_typed(rays_alive_out, "rays_alive_out", torch.int32)), int(N), int(max(align, 0)),
             int(max_steps))


# ---- the same loop in compact rounds under a row budget (C ABI Part 1b, second half; csrc/raymarching.hip) ---------------
# A round takes n_step = clamp(budget / n_alive, step_min, step_max) steps of every alive ray and the march packs what the
# rays really emitted into one slab per ray: a 128 x 128 render is a handful of rounds instead of ~280, and rays that
# missed, finished or terminated cost no rows.  ctl[2] is the round's row count (written by the march).


def infer_begin2(N, device, budget_rows, step_min=1, step_max=1024, ctl=None, rays_alive=None):
    if ctl is None:
        ctl = torch.zeros(16, dtype=torch.int32, device=device)
    else:
        if ctl.numel() < 16:
            raise L.Mi3dError("the budget loop's ctl block is int32[16] (ABI version 5: ctl[8] counts dropped rows)")
        L.dev_typed(ctl, "ctl", torch.int32).zero_()
    if rays_alive is None:
        rays_alive = torch.empty(N, dtype=torch.int32, device=device)
    L.dev_typed(rays_alive, "rays_alive", torch.int32)
    L.launch("mi3d_infer_begin2", ctl, L.ptr(ctl), L.ptr(rays_alive), int(N), int(budget_rows), int(step_min), int(step_max))
    return ctl, rays_alive


def march_rays_compact_ctl(ctl, n_alive_max, rays_alive, rays_t, rays_o, rays_d, bound, density_bitfield, C, H, fars, xyzs,
                           dirs, deltas, ray_slab, t_next, noises=None, dt_gamma=0, max_steps=1024):
    """Up to ctl's n_step occupied steps of every alive ray into ONE slab per ray of the caller-owned sample buffers:
    ray_slab int32[>= n_alive_max, 2] = (first row, rows), t_next f32[>= n_alive_max] = where the ray's march stands;
    ctl[2] = the rows of the round."""
    rays_o, rays_d = _rays(rays_o, rays_d)
    bits = L.dev_typed(density_bitfield, "density_bitfield", torch.uint8)
    rows_cap = min(xyzs.shape[0], dirs.shape[0], deltas.shape[0])
    if ray_slab.shape[0] < n_alive_max or t_next.shape[0] < n_alive_max:
        raise L.Mi3dError("ray_slab / t_next hold fewer slots than n_alive_max")
    L.launch("mi3d_march_rays_compact_ctl", rays_o, L.ptr(L.dev_typed(ctl, "ctl", torch.int32)), int(n_alive_max),
             L.ptr(L.dev_typed(rays_alive, "rays_alive", torch.int32)), L.ptr(L.dev_f32(rays_t, "rays_t")), L.ptr(rays_o),
             L.ptr(rays_d), float(bound), float(dt_gamma), int(max_steps), int(C), int(H), L.ptr(bits),
             L.ptr(L.dev_f32(fars, "fars")), int(rows_cap), L.ptr(L.dev_f32(xyzs, "xyzs", 3)),
             L.ptr(L.dev_f32(dirs, "dirs", 3)), L.ptr(L.dev_f32(deltas, "deltas", 2)),
             L.ptr(L.dev_typed(ray_slab, "ray_slab", torch.int32)), L.ptr(L.dev_f32(t_next, "t_next")), L.ptr(noises))


def composite_rays_compact_ctl(ctl, n_alive_max, rays_alive, rays_t, ray_slab, t_next, sigmas, rgbs, normals, deltas,
                               weights_sum, depth, image, normal, T_thresh=1e-2):
    sigmas = L.dev_f32(sigmas.float().contiguous(), "sigmas")
    rgbs = L.dev_f32(rgbs.float().contiguous(), "rgbs")
    normals = L.dev_f32(normals.float().contiguous(), "normals")
    L.launch("mi3d_composite_rays_compact_ctl", sigmas, L.ptr(ctl), int(n_alive_max), float(T_thresh),
             L.ptr(L.dev_typed(rays_alive, "rays_alive", torch.int32)), L.ptr(L.dev_f32(rays_t, "rays_t")),
             L.ptr(L.dev_typed(ray_slab, "ray_slab", torch.int32)), L.ptr(L.dev_f32(t_next, "t_next")), L.ptr(sigmas),
             L.ptr(rgbs), L.ptr(normals), L.ptr(L.dev_f32(deltas, "deltas")), L.ptr(L.dev_f32(weights_sum, "weights_sum")),
             L.ptr(L.dev_f32(depth, "depth")), L.ptr(L.dev_f32(image, "image")), L.ptr(L.dev_f32(normal, "normal")))


def compact_alive_ctl2(ctl, rays_alive_in, rays_alive_out, N, max_steps=1024):
    L.launch("mi3d_compact_alive_ctl2", ctl, L.ptr(ctl), L.ptr(L.dev_typed(rays_alive_in, "rays_alive_in", torch.int32)),
             L.ptr(L.dev_typed(rays_alive_out, "rays_alive_out", torch.int32)), int(N), int(max_steps))
