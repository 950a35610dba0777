"""Development aid: the binned hash-grid scatter alone on the C2 view (run under rocprofv3 --kernel-trace --stats).
    python tools/scatter_binned_bench.py [--samples N] [--bitfield dense|0.3] [--ws-gb G]"""
import argparse, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "make-it-3d_amd")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=0)
    ap.add_argument("--bitfield", default="dense")
    ap.add_argument("--ws-gb", type=float, default=-1)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--half", action="store_true", help="8-byte binary16 records on the fine levels")
    a = ap.parse_args()
    import raymarching
    from mi3d import rays as R, grid_ops, field_ops, sds_step, network
    dev = torch.device("cuda:0")
    cfg = dict(n_levels=16, base_resolution=16, per_level_scale=1.3819128, log2_hashmap_size=19)
    opt = sds_step.make_opt()
    model = network.NeRFNetwork(opt).to(dev)
    sds_step.set_bitfield(model, a.bitfield if a.bitfield == "dense" else float(a.bitfield))
    ro, rd, _ = R.view_rays(128, 128, device=dev)
    ro, rd = ro.view(-1, 3), rd.view(-1, 3)
    nears, fars = raymarching.near_far_from_aabb(ro, rd, model.aabb_train)
    cnt = torch.zeros(2, dtype=torch.int32, device=dev)
    xyzs, dirs, deltas, rays = raymarching.march_rays_train(ro, rd, 1.0, model.density_bitfield, 1, 128, nears, fars,
                                                            cnt, -1, True, 128, True, 0, 1024)
    m = xyzs.shape[0] if a.samples <= 0 else min(a.samples, xyzs.shape[0])
    xs = xyzs[:m].contiguous()
    xs2 = (xs + torch.randn_like(xs) * 0.01).contiguous()
    offs, P0 = grid_ops.stencil_offsets(center=True, second=True)
    P = offs.shape[0]
    planes = torch.randn(16, m * P, 2, device=dev)
    ws = None if a.ws_gb < 0 else int(a.ws_gb * (1 << 30))
    absmax = planes.abs().amax(dim=(1, 2)).contiguous() if a.half else None
    for i in range(a.iters):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        g = field_ops.scatter_binned(xs, xs2, offs, P0, 1.0, planes, cfg, 2 * 3 ** 0.5 / 1024, 12196240, workspace_bytes=ws, level_absmax=absmax)
        torch.cuda.synchronize()
        print(f"iter {i}: {1e3 * (time.perf_counter() - t0):.1f} ms  (m={m}, |g|={float(g.abs().sum()):.3e})", flush=True)


if __name__ == "__main__":
    main()
