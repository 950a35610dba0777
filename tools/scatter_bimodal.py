"""the dense-gradient scatter call timed 47.9 / 48.6 ms on some boxes and 55.2 / 55.4 on
others with the same kernels.  ONE process, the 13-point scatter + deferred point-0 pair of the C2-dense view, dense and
real-census gradients alternating `--rounds` times, every call timed on its own with HIP events, with the arena it got,
the slices the plan cut and the shader clock / board power sampled around it; then the same again with `--hog-gib` of
other allocations live (bench.py times its dense call at the END of a run, next to the model, the guidance stand-ins and
whatever the caching allocator still holds).  Run it under `rocprofv3 --kernel-trace` to get every k_bin_emit /
k_bin_reduce dispatch's duration (tools/trace_sum.py lists averages; --per-dispatch below prints them in order).

    python tools/scatter_bimodal.py --rounds 10 --out gpurun_out/scatter_bimodal.json
    python tools/scatter_bimodal.py --per-dispatch <kernel_trace.csv>      # after a traced run
"""
import argparse
import csv
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "make-it-3d_amd")]


def per_dispatch(path):
    files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if "k_bin_emit" in name or "k_bin_reduce" in name:
                rows.append((int(r["Start_Timestamp"]), "emit" if "emit" in name else "reduce",
                             (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6))
    rows.sort()
    out = {"emit_ms": [round(d, 3) for _, k, d in rows if k == "emit"],
           "reduce_ms": [round(d, 3) for _, k, d in rows if k == "reduce"]}
    print(json.dumps(out))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=10)
    ap.add_argument("--hog-gib", type=float, default=60.0)
    ap.add_argument("--sustain-s", type=float, default=40.0)
    ap.add_argument("--out", default="gpurun_out/scatter_bimodal.json")
    ap.add_argument("--per-dispatch", default=None)
    a = ap.parse_args()
    if a.per_dispatch:
        per_dispatch(a.per_dispatch)
        return
    import torch
    import raymarching
    import bench
    from mi3d import _lib as L, field_ops, grid_ops, network, rays as R, sds_step
    dev = torch.device("cuda:0")
    cfg = dict(n_levels=16, base_resolution=16, per_level_scale=1.3819128274917603, log2_hashmap_size=19)
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
    REAL_NZ = [0.72, 0.85, 0.77, 0.8, 0.83, 0.85, 0.76, 0.84, 0.82, 0.77, 0.63, 0.7, 0.55, 0.64, 0.72, 0.59]
    ex = torch.randn(16, n, 2, device=dev).half()
    step = 2 * 3 ** 0.5 / 1024
    torch.manual_seed(5)
    g_dense = torch.randn(16, P * n, 2, device=dev).half()
    g_real = g_dense.clone()
    for l in range(16):
        keep = torch.rand(P * n, device=dev) < REAL_NZ[l]
        keep[:n] = torch.rand(n, device=dev) < 0.94
        g_real[l] *= keep[:, None].half()
    arenas = []
    real_ws = field_ops.scatter_workspace

    def recording_ws(device, needed, cap=None):
        t = real_ws(device, needed, cap)
        arenas.append(0 if t is None else t.numel())
        return t
    field_ops.scatter_workspace = recording_ws

    sampler = bench.ClockSampler(bench.ClockSampler.pci_address_of(dev), period=0.05).start()
    res = {"samples": n, "phases": {}}
    hog = None
    for phase in ("fresh", "hogged"):
        if phase == "hogged":
            hog = torch.empty(int(a.hog_gib * 2 ** 30), dtype=torch.uint8, device=dev)
        calls = []
        for r in range(a.rounds):
            for kind, g in (("dense", g_dense), ("real", g_real)):
                free, _ = torch.cuda.mem_get_info(dev)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                w0 = time.time()
                e0.record()
                out = field_ops.scatter_binned(xs, xs2, offs, P0, 1.0, g, cfg, step, 12196240, extra0=ex)
                e1.record()
                torch.cuda.synchronize()
                w1 = time.time()
                del out
                calls.append({"round": r, "kind": kind, "ms": e0.elapsed_time(e1), "arena_GiB": arenas[-1] / 2 ** 30,
                              "free_GiB_before": free / 2 ** 30, "t0": w0, "t1": w1})
        res["phases"][phase] = calls
    # SUSTAINED: what bench.py's dense call sees - it runs after a minute of steps.  The dense call back to back for
    # `--sustain-s` seconds (no extra0: bench's own call), every call timed, clocks / power sampled throughout.
    if hog is not None:
        del hog
        torch.cuda.empty_cache()
    calls, t_end = [], time.time() + a.sustain_s
    while time.time() < t_end:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        w0 = time.time()
        e0.record()
        out = field_ops.scatter_binned(xs, xs2, offs, P0, 1.0, g_dense, cfg, step, 12196240)
        e1.record()
        torch.cuda.synchronize()
        del out
        calls.append({"round": len(calls), "kind": "dense", "ms": e0.elapsed_time(e1), "arena_GiB": arenas[-1] / 2 ** 30,
                      "t0": w0, "t1": time.time()})
    res["phases"]["sustained_dense_no_extra0"] = calls
    sampler.stop()
    for phase, calls in res["phases"].items():
        for c in calls:
            c["clocks"] = sampler.summary(c.pop("t0"), c.pop("t1"))
        for kind in ("dense", "real"):
            ts = [c["ms"] for c in calls if c["kind"] == kind]
            if len(ts) > 40:   # the sustained phase: one figure per ~4 s, with the clock the card held then
                k = max(1, len(ts) // 10)
                sel = [c for c in calls if c["kind"] == kind][::k]
                print(phase, kind, " ".join(f"{c['ms']:.1f}@{(c['clocks'].get('sclk_mhz') or {}).get('median', 0):.0f}MHz/"
                                            f"{(c['clocks'].get('power_w') or {}).get('median', 0):.0f}W" for c in sel))
            elif ts:
                print(phase, kind, " ".join(f"{t:.2f}" for t in ts))
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
