"""Per-level cost of the hash-grid gradient scatter (development aid): times k_scatter restricted to level subsets
on a 2M-sample slice of the C2 view, dense and pruned (sphere r=0.3) occupancy.
    python tools/scatter_levels.py [--out gpurun_out/scatter_levels.json]"""
import argparse, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "make-it-3d_amd")]


def timeit(fn, iters=5, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/scatter_levels.json")
    a = ap.parse_args()
    import raymarching
    from mi3d import rays as R, grid_ops, sds_step, network
    dev = torch.device("cuda:0")
    res = {}
    cfg = dict(n_levels=16, base_resolution=16, per_level_scale=1.3819128, log2_hashmap_size=19)
    for name, bf in (("dense", "dense"), ("pruned0.3", 0.3)):
        opt = sds_step.make_opt()
        model = network.NeRFNetwork(opt).to(dev)
        sds_step.set_bitfield(model, bf)
        ro, rd, _ = R.view_rays(128, 128, device=dev)
        ro, rd = ro.view(-1, 3), rd.view(-1, 3)
        nears, fars = raymarching.near_far_from_aabb(ro, rd, model.aabb_train)
        cnt = torch.zeros(2, dtype=torch.int32, device=dev)
        xyzs, dirs, deltas, rays = raymarching.march_rays_train(ro, rd, 1.0, model.density_bitfield, 1, 128, nears,
                                                                fars, cnt, -1, True, 128, True, 0, 1024)
        m = xyzs.shape[0]
        ms = min(m, 1 << 21)
        xs = xyzs[:ms].contiguous()
        xs2 = (xs + torch.randn_like(xs) * 0.01).contiguous()
        offs, P0 = grid_ops.stencil_offsets(center=True, second=True)
        p13 = torch.zeros(12196240, device=dev).uniform_(-1e-4, 1e-4).requires_grad_(True)
        feats = grid_ops.encode_points(p13, xs, offs, cfg, 1.0, xs2, P0, step=2 * 3 ** 0.5 / 1024)
        d13 = torch.randn_like(feats)
        r = {"samples_total": m, "samples_timed": ms}
        for label, mask in [("all", 0xFFFF), ("L0-4", 0x001F), ("L5-9", 0x03E0), ("L10-15", 0xFC00)] + \
                           [(f"L{l}", 1 << l) for l in range(16)]:
            os.environ["MI3D_SCATTER_LMASK"] = hex(mask)
            r[label] = timeit(lambda: torch.autograd.grad(feats, p13, d13, retain_graph=True))
        os.environ.pop("MI3D_SCATTER_LMASK")
        res[name] = r
        print(name, json.dumps(r))
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
