"""what limits the 13-point gather on the fine hashed levels (11-15)?

For each probed level, on the C2-dense sample set (10.9 M samples x 13 stencil points):
  real     the product's gather restricted to that level (tools build, tunable MI3D_T_ENCODE_ONLY_LEVEL: every XCD on the
           same level, 3 workgroups per CU - DESIGN.md 3.1's per-level figure);
  probe    tools/gather_probe.hip: the level's EXACT address stream - four aligned 16-byte slot loads per point and, for odd
           cx, the four 8-byte x + 1 loads behind them, in the product's order - read from a precomputed 16-byte record per
           (point, sample), with none of the arithmetic (no position, no cells, no hash multiplies, no weights, no fused
           multiply-adds), same launch geometry, one 4-byte non-temporal store per point;
  stream   the probe without the table loads (what reading the precomputed stream and storing costs by itself).
probe ~ real  -> the level is bound by the memory path (L2 -> L1 line fills): the gather is done where it is;
probe << real -> the arithmetic between the loads is the lever.

    python tools/build_dev.py; hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/bin/libgather_probe.so tools/gather_probe.hip
    python tools/gather_probe.py --levels 9,11,13,15 --out gpurun_out/gather_probe.json"""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "make-it-3d_amd")]
os.environ.setdefault("MI3D_LIB", os.path.join(ROOT, "tools", "bin", "libmi3d_dev.so"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

T_ENCODE_ONLY_LEVEL = 2


def timeit(fn, iters=5):
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
    ap.add_argument("--levels", default="9,11,13,15")
    ap.add_argument("--wgs-per-cu", default="3,6")
    ap.add_argument("--out", default="gpurun_out/gather_probe.json")
    a = ap.parse_args()
    import raymarching
    from mi3d import _lib as L, grid_ops, network, rays as R, sds_step
    from tinycudann import grid_levels
    lib = L.lib()
    lib.mi3d_dev_set.argtypes = [C.c_int, C.c_int]
    probe = C.CDLL(os.path.join(ROOT, "tools", "bin", "libgather_probe.so"))
    vp, u32, f32, i32 = C.c_void_p, C.c_uint32, C.c_float, C.c_int
    probe.probe_offsets.argtypes = [vp, vp, u32, vp, u32, u32, f32, f32, u32, vp, vp]
    probe.probe_gather.argtypes = [vp, vp, u32, u32, u32, i32, i32, vp, vp]
    dev = torch.device("cuda:0")
    cfg = dict(n_levels=16, base_resolution=16, per_level_scale=1.3819128274917603, log2_hashmap_size=19)
    total, offsets, resolutions, scales = grid_levels(**cfg)
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
    offs = np.ascontiguousarray(offs, np.float32)
    offs_p = offs.ctypes.data_as(C.c_void_p)
    params = model.encoder.params.detach()
    step = 2 * 3 ** 0.5 / 1024
    feats = torch.empty(16, P * n, 2, dtype=torch.float16, device=dev)
    stream = torch.empty(P * n, 4, dtype=torch.int32, device=dev)
    out = torch.empty(P * n, dtype=torch.float32, device=dev)
    st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)   # noqa: E731

    def real():
        err = lib.mi3d_grid_encode_points_planes(L.ptr(xs), L.ptr(xs2), n, offs_p, P0, P, 1.0, L.ptr(params), 16, 16,
                                                 cfg["per_level_scale"], 19, step, L.ptr(feats), 1, st())
        assert err == 0, err

    res = {"samples": n, "points": P, "evaluations": n * P, "levels": {}}
    lib.mi3d_dev_set(T_ENCODE_ONLY_LEVEL, -1)
    res["whole_gather_ms"] = timeit(real, 3)
    for l in [int(v) for v in a.levels.split(",")]:
        size = int(offsets[l + 1] - offsets[l])
        rec = {"resolution": int(resolutions[l]), "entries": size, "cells_per_marching_step": float(scales[l]) * step / 2.0}
        lib.mi3d_dev_set(T_ENCODE_ONLY_LEVEL, l)
        rec["real_ms"] = timeit(real)
        lib.mi3d_dev_set(T_ENCODE_ONLY_LEVEL, -1)
        err = probe.probe_offsets(L.ptr(xs), L.ptr(xs2), n, offs_p, P0, P, 1.0, float(scales[l]), size, L.ptr(stream), st())
        assert err == 0, err
        torch.cuda.synchronize()
        odd = float(((stream[:, 0] >> 24) & 1).float().mean())
        rec["share_of_points_with_odd_cx"] = odd
        level_table = params[int(offsets[l]) * 2:]
        for w in [int(v) for v in a.wgs_per_cu.split(",")]:
            for mode, name in ((0, "probe"), (1, "stream_only")):
                def run(mode=mode, w=w):
                    e = probe.probe_gather(L.ptr(level_table), L.ptr(stream), n, P, size, w, mode, L.ptr(out), st())
                    assert e == 0, e
                rec[f"{name}_ms_{w}_wgs_per_cu"] = timeit(run)
        # the loads' line traffic: 4 slot lines per point (+ the x + 1 corner's line where it differs), 128 B each
        rec["line_fills_GB_at_one_line_per_load"] = n * P * (4 + 4 * odd) * 128 / 1e9
        res["levels"][l] = rec
        print(l, json.dumps(rec), flush=True)
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
