#!/usr/bin/env python
"""Development probe: how full do the scatter's (wave, bin) record regions get, per level?  (Needs tools/build_dev.py.)
Runs one binned scatter of the C2-dense stencil with dense random gradients on a workspace it owns, then reads the region
counters back: records per level, mean / max fill relative to the region capacity, fraction of regions that overflowed
(their surplus went to the table by global atomics)."""
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "make-it-3d_amd"))
os.environ["MI3D_LIB"] = os.path.join(ROOT, "tools", "bin", "libmi3d_dev.so")


def main():
    import raymarching
    from mi3d import _lib as L, grid_ops, network, rays as R, sds_step
    lib = L.lib()
    dev = torch.device("cuda:0")
    model = network.NeRFNetwork(sds_step.make_opt()).to(dev)
    sds_step.set_bitfield(model, "dense")
    ro, rd, _ = R.view_rays(128, 128, device=dev)
    ro, rd = ro.view(-1, 3), rd.view(-1, 3)
    nears, fars = raymarching.near_far_from_aabb(ro, rd, model.aabb_train)
    cnt = torch.zeros(2, dtype=torch.int32, device=dev)
    xs, _, _, _ = raymarching.march_rays_train(ro, rd, 1.0, model.density_bitfield, 1, 128, nears, fars, cnt, -1, False,
                                               128, True, 0.0, 1024)
    xs = xs.contiguous()
    n = xs.shape[0]
    xs2 = (xs + torch.randn_like(xs) * 0.01).contiguous()
    offs, P0 = grid_ops.stencil_offsets(center=True, second=True)
    _, offs_p = grid_ops._offs_arg(offs)
    P, pls, step = 13, 1.3819128274917603, 2 * 3 ** 0.5 / 1024
    g = torch.randn(16, P * n, 2, device=dev).half()
    lib.mi3d_grid_scatter_binned_workspace.restype = C.c_size_t
    n_slice = n
    while True:
        need = lib.mi3d_grid_scatter_binned_workspace(n_slice, P, C.c_float(1.0), C.c_float(step), 16, 16, C.c_float(pls), 19)
        if need <= 100 << 30:
            break
        n_slice = (n_slice + 1) // 2
    ws = torch.zeros(need, dtype=torch.uint8, device=dev)
    grad = torch.zeros(12196240, device=dev)
    # one slice only: hand the call exactly n_slice samples
    L.call("mi3d_grid_scatter_binned", L.ptr(xs), L.ptr(xs2), n_slice, offs_p, int(P0), P, 1.0, L.ptr(g), 1, 16, 16, pls,
           19, step, L.ptr(ws), C.c_size_t(need), L.ptr(grad), L.stream())
    torch.cuda.synchronize()
    out = (C.c_ulonglong * (6 + 7 * 16))()
    L.call("mi3d_grid_scatter_plan", n_slice, P, 1.0, step, 16, 16, pls, 19, C.c_size_t(need), out)
    assert out[0] == n_slice
    total_bytes, total_counts, merge = out[4], out[5], out[2]
    counts = ws[total_bytes: total_bytes + 4 * total_counts].view(torch.int32)
    res = {"samples_in_slice": n_slice, "merge_levels": int(merge), "levels": []}
    for l in range(16):
        bins, cap, waves, row, _split, _wg0, c0 = (int(out[6 + 7 * l + k]) for k in range(7))
        c = counts[c0: c0 + waves * bins].view(waves, bins).float()
        per_bin = c.sum(0)
        res["levels"].append({"level": l, "bins": bins, "cap": cap, "waves": waves, "row_records": bool(row),
                              "records": int(c.sum().item()), "mean_fill": float(c.mean().item()) / cap,
                              "max_fill": float(c.max().item()) / cap, "regions_full": float((c >= cap).float().mean().item()),
                              "records_in_fullest_bin": int(per_bin.max().item()),
                              "records_in_mean_bin": float(per_bin.mean().item())})
    print(json.dumps(res, indent=1))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "scatter_fill.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
