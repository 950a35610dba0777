"""The 13-point hash-grid gather alone on the C2-dense view (for rocprofv3 --kernel-trace / --pmc passes).
    python tools/encode_bench.py [--iters 3] [--bitfield dense]"""
import argparse, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "make-it-3d_amd")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--bitfield", default="dense")
    ap.add_argument("--planes", action="store_true")
    a = ap.parse_args()
    import raymarching
    from mi3d import rays as R, grid_ops, sds_step, network
    dev = torch.device("cuda:0")
    cfg = dict(n_levels=16, base_resolution=16, per_level_scale=1.3819128, log2_hashmap_size=19)
    model = network.NeRFNetwork(sds_step.make_opt()).to(dev)
    sds_step.set_bitfield(model, a.bitfield if a.bitfield == "dense" else float(a.bitfield))
    ro, rd, _ = R.view_rays(128, 128, device=dev)
    ro, rd = ro.view(-1, 3), rd.view(-1, 3)
    nears, fars = raymarching.near_far_from_aabb(ro, rd, model.aabb_train)
    cnt = torch.zeros(2, dtype=torch.int32, device=dev)
    xs, dirs, deltas, rays = raymarching.march_rays_train(ro, rd, 1.0, model.density_bitfield, 1, 128, nears, fars, cnt,
                                                          -1, True, 128, True, 0, 1024)
    xs2 = (xs + torch.randn_like(xs) * 0.01).contiguous()
    offs, P0 = grid_ops.stencil_offsets(center=True, second=True)
    params = model.encoder.params.detach()
    for i in range(a.iters):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if a.planes:
            from mi3d import _lib as L
            offs_np, offs_p = grid_ops._offs_arg(offs)
            f = torch.empty(16, xs.shape[0] * 13, 2, device=dev)
            L.call("mi3d_grid_encode_points_planes", L.ptr(xs), L.ptr(xs2), xs.shape[0], offs_p, int(P0), 13, 1.0,
                   L.ptr(params), 16, 16, 1.3819128, 19, L.ptr(f), L.stream())
            f = f.view(16 * xs.shape[0] * 13, 2)[:xs.shape[0] * 13]
        else:
            f = grid_ops.encode_points(params, xs, offs, cfg, 1.0, xs2, P0)
        torch.cuda.synchronize()
        print(f"iter {i}: {1e3 * (time.perf_counter() - t0):.2f} ms, samples {xs.shape[0]}, evaluations {f.shape[0]}")
        del f


if __name__ == "__main__":
    main()
