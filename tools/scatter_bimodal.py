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
    ap.add_argument("--sweep-n", type=int, default=0, help="time the dense call at this many sample counts n - 64 k stride")
    ap.add_argument("--sweep-stride", type=int, default=1)
    ap.add_argument("--arenas", action="store_true", help="explicitly held arenas before / after SD steps, re-allocated, fragmented")
    ap.add_argument("--placements", type=int, default=0, help="that many freshly allocated arenas, every --libs build on each")
    ap.add_argument("--libs", default="make-it-3d_amd/csrc/libmi3d.so")
    ap.add_argument("--shift", type=int, default=0, help="that many spacer sizes (k x --shift-gib) in front of a fresh arena")
    ap.add_argument("--shift-gib", type=float, default=6.0)
    ap.add_argument("--recipes", action="store_true", help="allocator conditionings before the arena is allocated")
    ap.add_argument("--matrix", action="store_true", help="with / without / zero deferred pair, before and after SD stand-in steps")
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

    if a.matrix:
        # Round 6, runs 2-3: in ONE process the call took 50.2 ms with a deferred point-0 pair riding along (14 record points)
        # and 55.2 ms WITHOUT it (13 points: less work) - and the 13-point call took 47.9 ms inside bench.py on the same box.
        # Which of the two differences is the code path and which the process?  A = with the pair, B = without, C = with an
        # all-zero pair (the 14-point path, no extra records); then the same after the process has run what bench.py runs
        # before its dense call (convolutions, GEMMs, attention: kernels with large scratch and LDS footprints).
        ex0 = torch.zeros_like(ex)
        kinds = {"A_with_pair": ex, "B_without": None, "C_zero_pair": ex0}

        def one(kind):
            call = lambda: field_ops.scatter_binned(xs, xs2, offs, P0, 1.0, g_dense, cfg, step, 12196240, extra0=kinds[kind])   # noqa: E731
            call()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            call(); call()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / 2
        res = {"samples": n, "matrix": {}}
        order = ["B_without", "A_with_pair", "C_zero_pair", "B_without", "A_with_pair"]
        res["matrix"]["fresh_process"] = [(k, one(k)) for k in order]
        from mi3d import sd_standin
        guidance = sd_standin.StableDiffusionStandIn(dev)
        tz = guidance.get_text_embeds()
        img = torch.rand(1, 3, 128, 128, device=dev, requires_grad=True)
        for _ in range(3):
            with torch.autocast("cuda", dtype=torch.float16):
                guidance.train_step(tz, img, guidance_scale=100, t=500) if hasattr(guidance, "train_step") else None
        torch.cuda.synchronize()
        res["matrix"]["after_sd_standin_steps"] = [(k, one(k)) for k in order]
        for ph, rows in res["matrix"].items():
            print(ph, " ".join(f"{k}:{v:.2f}" for k, v in rows))
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        json.dump(res, open(a.out, "w"), indent=1)
        return
    if a.arenas:
        # Is the slow mode a property of WHERE the record arena lies (the driver's physical placement / page fragments at
        # allocation time) or of the process's state?  The call with explicitly held arenas: W1 before and after SD stand-in
        # steps (same memory), a second arena W2 allocated afterwards (other memory) against W1, then both released, the
        # cache emptied and a third one allocated.
        real_alloc = real_ws
        held = {}

        def use(name):
            def ws(device, needed, cap=None):
                if name not in held:
                    held[name] = real_alloc(device, needed, cap)
                return held[name]
            field_ops.scatter_workspace = ws

        def one():
            call = lambda: field_ops.scatter_binned(xs, xs2, offs, P0, 1.0, g_dense, cfg, step, 12196240)   # noqa: E731
            call()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            call(); call(); call()
            e1.record()
            torch.cuda.synchronize()
            return round(e0.elapsed_time(e1) / 3, 2)
        log = []
        use("W1"); log.append(("W1 fresh", one(), held["W1"].data_ptr()))
        from mi3d import sd_standin
        guidance = sd_standin.StableDiffusionStandIn(dev)
        tz = guidance.get_text_embeds()
        img = torch.rand(1, 3, 128, 128, device=dev, requires_grad=True)
        for _ in range(3):
            with torch.autocast("cuda", dtype=torch.float16):
                guidance.train_step(tz, img, guidance_scale=100, t=500)
        torch.cuda.synchronize()
        log.append(("W1 after SD steps (same memory)", one(), held["W1"].data_ptr()))
        use("W2"); log.append(("W2 allocated after the SD steps", one(), held["W2"].data_ptr()))
        use("W1"); log.append(("W1 again", one(), held["W1"].data_ptr()))
        del guidance, tz, img
        held.clear()
        torch.cuda.empty_cache()
        use("W3"); log.append(("W3 after releasing everything", one(), held["W3"].data_ptr()))
        small = [torch.empty(64 << 20, dtype=torch.uint8, device=dev) for _ in range(400)]   # 25 GiB in 64 MiB pieces
        del small[::2]
        held.clear()
        torch.cuda.empty_cache()
        use("W4"); log.append(("W4 after fragmenting 25 GiB into 64 MiB holes", one(), held["W4"].data_ptr()))
        for r in log:
            print(r)
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        json.dump({"samples": n, "arenas": log}, open(a.out, "w"), indent=1)
        return
    if a.placements > 0:
        # Several builds of the library over the SAME sequence of freshly allocated arenas: is a variant less sensitive to
        # where the arena lies than the product?  (Up to three earlier arenas are held while the next one is allocated, so
        # the placements differ.)
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import scatter_ab_libs
        paths = a.libs.split(",")
        libs = [scatter_ab_libs.load(p_) for p_ in paths]
        real_alloc = real_ws
        cur = {}
        field_ops.scatter_workspace = lambda device, needed, cap=None: cur["a"]
        rows, ring = [], []
        for i in range(a.placements):
            cur["a"] = real_alloc(dev, 60 << 30, None)
            ring.append(cur["a"])
            if len(ring) > 3:
                ring.pop(0)
            rec = {"placement": i, "data_ptr": cur["a"].data_ptr()}
            # plain streaming bandwidth of this placement: write (fill) and read (sum) of the whole block
            w64 = cur["a"].view(torch.int64)
            for name, op in (("fill_GBps", lambda: w64.fill_(1)), ("read_GBps", lambda: w64[: w64.numel() // 4].sum())):
                op()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                op(); op()
                e1.record()
                torch.cuda.synchronize()
                nbytes = w64.numel() * 8 * (1.0 if name == "fill_GBps" else 0.25)
                rec[name] = round(nbytes / (e0.elapsed_time(e1) / 2 * 1e-3) / 1e9, 1)
            for p_, lib in zip(paths, libs):
                L._lib = lib
                call = lambda: field_ops.scatter_binned(xs, xs2, offs, P0, 1.0, g_dense, cfg, step, 12196240)   # noqa: E731
                call()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                call(); call(); call()
                e1.record()
                torch.cuda.synchronize()
                rec[os.path.basename(p_)] = round(e0.elapsed_time(e1) / 3, 2)
            rows.append(rec)
            print(rec)
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        json.dump({"samples": n, "libs": paths, "placements": rows}, open(a.out, "w"), indent=1)
        return
    if a.shift:
        # Can better placements be FOUND by shifting where the arena lands - a spacer of k x `--shift-gib` GiB allocated just
        # before it - instead of holding several 56 GiB candidates side by side?
        real_alloc = real_ws
        held = {}

        def ws(device, needed, cap=None):
            if "a" not in held:
                held["a"] = real_alloc(device, needed, cap)
            return held["a"]
        field_ops.scatter_workspace = ws
        field_ops.PLACED_MIN_BYTES = 1 << 62     # (plain per-call path: this tool holds the arena itself)

        def one():
            call = lambda: field_ops.scatter_binned(xs, xs2, offs, P0, 1.0, g_dense, cfg, step, 12196240)   # noqa: E731
            call()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            call(); call(); call()
            e1.record()
            torch.cuda.synchronize()
            return round(e0.elapsed_time(e1) / 3, 2)
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import scatter_ab_libs
        paths = a.libs.split(",")
        libs = [scatter_ab_libs.load(p_) for p_ in paths]
        rows = []
        for rep in range(2):
            for k in range(a.shift):
                held.clear()
                torch.cuda.empty_cache()
                spacer = torch.empty(int(k * a.shift_gib * 2 ** 30) or 1, dtype=torch.uint8, device=dev)
                rec = {"rep": rep, "spacer_GiB": k * a.shift_gib}
                for p_, lib in zip(paths, libs):
                    L._lib = lib
                    rec[os.path.basename(p_)] = one()
                rec["arena_ptr"] = held["a"].data_ptr()
                rows.append(rec)
                print(rec, flush=True)
                del spacer
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        json.dump({"samples": n, "shift": rows}, open(a.out, "w"), indent=1)
        return
    if a.recipes:
        # Round 6, call 5: a re-allocated arena is fast (47.9 ms), slow (55.3) - or, allocated while 12.5 GiB of 64 MiB blocks
        # pin every other hole of a 25 GiB stretch, faster than either (45.6, three processes out of three).  Which
        # conditioning of the allocator does that, and how reliably?  Every variant: condition, allocate the arena, time the
        # dense call, release everything.
        real_alloc = real_ws
        held = {}

        def ws(device, needed, cap=None):
            if "a" not in held:
                held["a"] = real_alloc(device, needed, cap)
            return held["a"]
        field_ops.scatter_workspace = ws

        def one():
            call = lambda: field_ops.scatter_binned(xs, xs2, offs, P0, 1.0, g_dense, cfg, step, 12196240)   # noqa: E731
            call()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            call(); call(); call()
            e1.record()
            torch.cuda.synchronize()
            return round(e0.elapsed_time(e1) / 3, 2)

        def blocks(count, mib):
            return [torch.empty(mib << 20, dtype=torch.uint8, device=dev) for _ in range(count)]

        def variant(name, prepare):
            held.clear()
            torch.cuda.empty_cache()
            keep = prepare()
            torch.cuda.empty_cache()
            t = one()
            held.clear()
            del keep
            torch.cuda.empty_cache()
            return (name, t)

        def every_other(count, mib):
            def prep():
                b = blocks(count, mib)
                del b[::2]
                return b
            return prep
        log = []
        for rep in range(2):
            log.append(variant("plain", lambda: None))
            log.append(variant("400 x 64 MiB, every other one released", every_other(400, 64)))
            log.append(variant("100 x 256 MiB, every other one released", every_other(100, 256)))
            log.append(variant("1600 x 16 MiB, every other one released", every_other(1600, 16)))
            log.append(variant("400 x 64 MiB allocated and ALL released", lambda: (blocks(400, 64), None)[1]))
            log.append(variant("one 25 GiB block held", lambda: blocks(1, 25 * 1024)))
            log.append(variant("200 x 64 MiB held (no holes)", lambda: blocks(200, 64)))
            log.append(variant("plain", lambda: None))
        for r in log:
            print(r)
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        json.dump({"samples": n, "recipes": log}, open(a.out, "w"), indent=1)
        return
    if a.sweep_n > 0:
        # Does the call's time depend on the SAMPLE COUNT?  (Round 6, run 2: the same call took 55.2 ms in every one of 700
        # back-to-back calls of one process and 47.9 ms inside bench.py on the same box minutes later; what differs between
        # processes is n - the march jitter is drawn from the process's RNG stream - and with it the stride between the
        # 13 point blocks of a gradient plane, n x 4 bytes, and the region capacity the plan derives from n.)
        sweep = []
        for k in range(a.sweep_n):
            nk = n - 64 * k * a.sweep_stride
            gk = torch.randn(16, P * nk, 2, device=dev).half()
            xk, x2k = xs[:nk].contiguous(), xs2[:nk].contiguous()
            call = lambda: field_ops.scatter_binned(xk, x2k, offs, P0, 1.0, gk, cfg, step, 12196240)   # noqa: E731
            call()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            call(); call()
            e1.record()
            torch.cuda.synchronize()
            sweep.append({"n": nk, "n_div_64_mod_32": (nk // 64) % 32, "ms": e0.elapsed_time(e1) / 2})
            del gk, xk, x2k
        print("sweep", " ".join(f"{c['n']}:{c['ms']:.1f}" for c in sweep))
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        json.dump({"samples": n, "sweep_dense_no_extra0": sweep}, open(a.out, "w"), indent=1)
        return
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
