"""Per-XCD, per-segment timeline of the plane gather (development build's timestamps) beside the plan that cut the (level, tile)
list: how well does make_encode_plan's cost model balance the eight XCDs, and what does a tile of each level really cost?
    python tools/build_dev.py && python tools/encode_xcd_timeline.py --out gpurun_out/encode_xcd_timeline.json"""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "make-it-3d_amd")]
os.environ.setdefault("MI3D_LIB", os.path.join(ROOT, "tools", "bin", "libmi3d_dev.so"))

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/encode_xcd_timeline.json")
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    import raymarching
    from mi3d import _lib as L, grid_ops, network, rays as R, sds_step
    lib = L.lib()
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
    step = 2 * 3 ** 0.5 / 1024

    def encode():
        L.call("mi3d_grid_encode_points_planes", L.ptr(xs), L.ptr(xs2), n, offs_p, int(P0), P, 1.0, L.ptr(params), 16, 16, pls,
               19, step, L.ptr(feats), 1, L.stream())
    nseg = (C.c_uint32 * 8)()
    segs = (C.c_uint32 * (8 * 16 * 3))()
    L.call("mi3d_grid_encode_plan", n, 1.0, step, 16, 16, pls, 19, nseg, segs)
    plan = [[(segs[(x * 16 + i) * 3], segs[(x * 16 + i) * 3 + 1], segs[(x * 16 + i) * 3 + 2]) for i in range(nseg[x])] for x in range(8)]
    buf = (C.c_ulonglong * (8 * 17))()
    for _ in range(3):
        encode()
    torch.cuda.synchronize()
    runs = []
    for _ in range(a.reps):
        lib.mi3d_dev_encode_times(buf, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        encode()
        e1.record()
        torch.cuda.synchronize()
        lib.mi3d_dev_encode_times(buf, 0)
        t0 = min(buf[x * 17] for x in range(8))
        run = {"ms": e0.elapsed_time(e1), "xcd": []}
        for x in range(8):
            ends = [(buf[x * 17 + i] - t0) / 1e5 for i in range(1, nseg[x] + 1)]
            start = (buf[x * 17] - t0) / 1e5
            run["xcd"].append({"start_ms": round(start, 3), "segment_end_ms": [round(t, 3) for t in ends]})
        runs.append(run)
    res = {"samples": n, "plan": [[{"level": l, "tile0": t0_, "tile1": t1} for (l, t0_, t1) in plan[x]] for x in range(8)], "runs": runs}
    # per level: ms per 1000 tiles on one XCD, from the segments that are not an XCD's first (their start is known exactly)
    per_level = {}
    for run in runs:
        for x in range(8):
            prev = run["xcd"][x]["start_ms"]
            for (l, a0, a1), end in zip(plan[x], run["xcd"][x]["segment_end_ms"]):
                if a1 > a0:
                    per_level.setdefault(l, []).append((end - prev) / (a1 - a0) * 1000.0)
                prev = end
    res["ms_per_1000_tiles_one_xcd"] = {str(l): round(sum(v) / len(v), 4) for l, v in sorted(per_level.items())}
    res["xcd_end_ms_mean"] = [round(sum(r["xcd"][x]["segment_end_ms"][-1] for r in runs) / len(runs), 3) for x in range(8)]
    print(json.dumps({k: res[k] for k in ("ms_per_1000_tiles_one_xcd", "xcd_end_ms_mean")}, indent=1))
    print([round(r["ms"], 3) for r in runs])
    for x in range(8):
        print(x, [(s["level"], s["tile1"] - s["tile0"]) for s in res["plan"][x]], runs[-1]["xcd"][x])
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
