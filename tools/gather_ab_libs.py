"""A/B of the plane gather (k_grid_encode_planes + the LDS levels' kernel) across several PRODUCT-GRADE builds of the library in
ONE process, like tools/scatter_ab_libs.py for the scatter:
    python tools/gather_ab_libs.py --libs make-it-3d_amd/csrc/libmi3d.so,tools/bin/libmi3d_nosteal.so --out gpurun_out/gather_ab_libs.json
The 13-point gather of the C2-dense view into binary16 planes (the autocast layout), the libraries interleaved (A B A B ...)
`--rounds` times; every library's planes must be bit-identical to the first one's."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "make-it-3d_amd"), os.path.join(ROOT, "tools")]

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", required=True)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--out", default="gpurun_out/gather_ab_libs.json")
    a = ap.parse_args()
    import raymarching
    from scatter_ab_libs import load
    from mi3d import _lib as L, grid_ops, network, rays as R, sds_step
    paths = a.libs.split(",")
    libs = [load(p) for p in paths]
    L._lib = libs[0]
    dev = torch.device("cuda:0")
    pls = 1.3819128274917603
    model = network.NeRFNetwork(sds_step.make_opt()).to(dev)
    sds_step.set_bitfield(model, "dense")
    ro, rd, _ = R.view_rays(128, 128, device=dev)
    ro, rd = ro.view(-1, 3), rd.view(-1, 3)
    nears, fars = raymarching.near_far_from_aabb(ro, rd, model.aabb_train)
    cnt = torch.zeros(2, dtype=torch.int32, device=dev)
    xs, _, _, _ = raymarching.march_rays_train(ro, rd, 1.0, model.density_bitfield, 1, 128, nears, fars, cnt, -1, True, 128,
                                               True, 0, 1024)
    xs = xs.contiguous()
    n, P = xs.shape[0], 13
    xs2 = (xs + torch.randn_like(xs) * 0.01).contiguous()
    offs, P0 = grid_ops.stencil_offsets(center=True, second=True)
    _, offs_p = grid_ops._offs_arg(offs)
    params = torch.empty(12196240, device=dev).uniform_(-1, 1)
    feats = torch.empty(16, P * n, 2, device=dev, dtype=torch.float16)

    def encode():
        L.call("mi3d_grid_encode_points_planes", L.ptr(xs), L.ptr(xs2), n, offs_p, int(P0), P, 1.0, L.ptr(params), 16, 16, pls,
               19, 2 * 3 ** 0.5 / 1024, L.ptr(feats), 1, L.stream())

    def timeit(fn):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.iters

    res = {"libs": paths, "samples": n, "ms": {os.path.basename(p): [] for p in paths}}
    ref = None
    for p, lib in zip(paths, libs):
        L._lib = lib
        feats.zero_()
        encode()
        torch.cuda.synchronize()
        if ref is None:
            ref = feats.clone()
        else:
            res[f"{os.path.basename(p)}:bit_identical_to_first"] = bool(torch.equal(feats.view(torch.int16), ref.view(torch.int16)))
    for _ in range(a.rounds):
        for p, lib in zip(paths, libs):
            L._lib = lib
            res["ms"][os.path.basename(p)].append(timeit(encode))
    print(json.dumps(res, indent=1))
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
