"""Per-kernel timing of the operator-level path on one GPU (development aid; bench.py is the contract).
    python tools/microbench.py [--res 128] [--max-steps 1024] [--out gpurun_out/micro.json]"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "make-it-3d_amd")]


def timeit(fn, iters=10, warmup=3):
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
    ap.add_argument("--res", type=int, default=128)
    ap.add_argument("--max-steps", type=int, default=1024)
    ap.add_argument("--out", default="gpurun_out/micro.json")
    a = ap.parse_args()
    import raymarching
    import tinycudann as tcnn
    from mi3d import rays as R
    from mi3d import _lib as L
    dev = torch.device("cuda:0")
    res = {}
    ro, rd, _ = R.view_rays(a.res, a.res, device=dev)
    ro, rd = ro.view(-1, 3), rd.view(-1, 3)
    N = ro.shape[0]
    aabb = torch.tensor([-1, -1, -1, 1, 1, 1.0], device=dev)
    nears, fars = raymarching.near_far_from_aabb(ro, rd, aabb)
    bits = torch.full((128 ** 3 // 8,), 255, dtype=torch.uint8, device=dev)
    cnt = torch.zeros(2, dtype=torch.int32, device=dev)

    def march():
        cnt.zero_()
        return raymarching.march_rays_train(ro, rd, 1.0, bits, 1, 128, nears, fars, cnt, -1, True, 128, True, 0,
                                            a.max_steps)
    xyzs, dirs, deltas, rays = march()
    m = xyzs.shape[0]
    res["rays"], res["samples"] = N, m
    res["march_ms"] = timeit(march)
    res["march_GBps_alg"] = m * 32 / res["march_ms"] / 1e6

    enc = tcnn.Encoding(3, {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 19,
                            "base_resolution": 16, "per_level_scale": 1.3819128}).to(dev)
    with torch.no_grad():
        enc.params.uniform_(-1, 1)
    x01 = ((xyzs + 1) / 2).contiguous()
    out = torch.empty(m, 32, device=dev)

    def hg_fwd():
        L.call("mi3d_hashgrid_forward", L.ptr(x01), m, L.ptr(enc.params), 16, 16, 1.3819128, 19, L.ptr(out),
               L.stream())
    res["hashgrid_fwd_ms"] = timeit(hg_fwd)
    res["hashgrid_fwd_GBps_alg"] = m * 1024 / res["hashgrid_fwd_ms"] / 1e6
    dout = torch.randn(m, 32, device=dev)
    grad = torch.zeros_like(enc.params)

    def hg_bwd():
        L.call("mi3d_hashgrid_backward", L.ptr(x01), m, L.ptr(dout), 16, 16, 1.3819128, 19, L.ptr(grad), L.stream())
    res["hashgrid_bwd_ms"] = timeit(hg_bwd, iters=5, warmup=1)
    res["hashgrid_bwd_GBps_alg"] = m * 2048 / res["hashgrid_bwd_ms"] / 1e6
    # random (incoherent) positions for contrast
    xr = torch.rand(m, 3, device=dev)

    def hg_fwd_rand():
        L.call("mi3d_hashgrid_forward", L.ptr(xr), m, L.ptr(enc.params), 16, 16, 1.3819128, 19, L.ptr(out),
               L.stream())
    res["hashgrid_fwd_random_ms"] = timeit(hg_fwd_rand)

    # stencil-aware encode + scatter, P = 13 (the reference's 13 evaluations), on a 1M-sample slice
    from mi3d import grid_ops
    cfg = dict(n_levels=16, base_resolution=16, per_level_scale=1.3819128, log2_hashmap_size=19)
    ms = min(m, 1 << 20)
    xs = xyzs[:ms].contiguous()
    xs2 = (xs + torch.randn_like(xs) * 0.01).contiguous()
    offs, P0 = grid_ops.stencil_offsets(center=True, second=True)
    P = offs.shape[0]
    p13 = enc.params.detach().clone().requires_grad_(True)
    feats = grid_ops.encode_points(p13, xs, offs, cfg, 1.0, xs2, P0, step=2 * 3 ** 0.5 / a.max_steps)
    res["encode13_1M_ms"] = timeit(lambda: grid_ops.encode_points(p13.detach(), xs, offs, cfg, 1.0, xs2, P0))
    d13 = torch.randn_like(feats)

    def sc13():
        torch.autograd.grad(feats, p13, d13, retain_graph=True)
    res["scatter13_1M_ms"] = timeit(sc13, iters=5, warmup=1)
    res["scatter13_full_step_est_ms"] = res["scatter13_1M_ms"] * m / ms
    res["encode13_full_step_est_ms"] = res["encode13_1M_ms"] * m / ms
    del feats, d13

    sig = torch.rand(m, device=dev) * 5
    rgb = torch.rand(m, 3, device=dev)
    res["composite_fwd_ms"] = timeit(lambda: raymarching.composite_rays_train(sig, rgb, deltas, rays, 1e-4))
    sig.requires_grad_(True)
    rgb.requires_grad_(True)
    ws, dep, img = raymarching.composite_rays_train(sig, rgb, deltas, rays, 1e-4)
    g1, g2 = torch.randn_like(ws), torch.randn_like(img)

    def comp_bwd():
        torch.autograd.grad([ws, img], [sig, rgb], [g1, g2], retain_graph=True)
    res["composite_bwd_ms"] = timeit(comp_bwd)
    # torch MLP at the reference's shapes, fp16 autocast, for scale
    mlp = torch.nn.Sequential(torch.nn.Linear(32, 64), torch.nn.ReLU(), torch.nn.Linear(64, 64), torch.nn.ReLU(),
                              torch.nn.Linear(64, 4)).to(dev)

    def mlp_fwd():
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            return mlp(out)
    res["torch_mlp_fwd_ms"] = timeit(mlp_fwd)
    print(json.dumps(res, indent=1))
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
