"""Kernel A/B micro-benchmarks on the C2-dense view (development aid; bench.py is the contract).  Loads the DEV build
(tools/bin/libmi3d_dev.so, tools/build_dev.py) so kernel variants and launch geometries can be switched at run time:
    python tools/build_dev.py && python tools/kbench.py [--what encode,levels,mlp,scatter] [--out gpurun_out/kbench.json]
Every timing is HIP-event time on the launch stream, averaged over `--iters` launches after one warm-up."""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "make-it-3d_amd")]
os.environ.setdefault("MI3D_LIB", os.path.join(ROOT, "tools", "bin", "libmi3d_dev.so"))

import torch  # noqa: E402

T_ENCODE_VARIANT, T_ENCODE_WGS, T_ENCODE_ONLY_LEVEL, T_EMIT_FINE, T_EMIT_COARSE = 0, 1, 2, 3, 4
T_MLP_WGS, T_REUSE_LEVELS = 7, 10


def timeit(fn, iters=3, warmup=1):
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
    ap.add_argument("--what", default="encode,levels,mlp,scatter")
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--bitfield", default="dense")
    ap.add_argument("--out", default="gpurun_out/kbench.json")
    ap.add_argument("--half-planes", action="store_true", help="binary16 feature / gradient planes (the autocast layout)")
    ap.add_argument("--dev", default="", help="tunables set before anything runs, e.g. 10=1,3=2048 (csrc/mi3d_dev.h)")
    ap.add_argument("--real-census", action="store_true",
                    help="scatter benchmarks: zero the random gradient pairs independently with the per-level non-zero "
                         "fractions bench.py's census measured on a real step (regulariser pass + deferred point 0)")
    a = ap.parse_args()
    what = set(a.what.split(","))
    import raymarching
    from mi3d import _lib as L, field_ops, grid_ops, network, rays as R, sds_step
    lib = L.lib()
    lib.mi3d_dev_set.argtypes = [C.c_int, C.c_int]

    def tune(i, v):
        lib.mi3d_dev_set(i, v)
    for kv in filter(None, a.dev.split(",")):
        tune(*map(int, kv.split("=")))
    dev = torch.device("cuda:0")
    cfg = dict(n_levels=16, base_resolution=16, per_level_scale=1.3819128274917603, log2_hashmap_size=19)
    model = network.NeRFNetwork(sds_step.make_opt()).to(dev)
    sds_step.set_bitfield(model, a.bitfield if a.bitfield == "dense" else float(a.bitfield))
    ro, rd, _ = R.view_rays(128, 128, device=dev)
    ro, rd = ro.view(-1, 3), rd.view(-1, 3)
    nears, fars = raymarching.near_far_from_aabb(ro, rd, model.aabb_train)
    cnt = torch.zeros(2, dtype=torch.int32, device=dev)
    xs, dirs, deltas, rays = raymarching.march_rays_train(ro, rd, 1.0, model.density_bitfield, 1, 128, nears, fars, cnt,
                                                          -1, True, 128, True, 0, 1024)
    xs = xs.contiguous()
    n = xs.shape[0]
    xs2 = (xs + torch.randn_like(xs) * 0.01).contiguous()
    offs, P0 = grid_ops.stencil_offsets(center=True, second=True)
    _, offs_p = grid_ops._offs_arg(offs)
    P = 13
    params = torch.empty(12196240, device=dev).uniform_(-1, 1)
    res = {"samples": n, "evals": n * P}
    # per-level share of non-zero (row, level) gradient pairs of a real C2 step (gpurun_out/r04_1/bench.json: the binary16
    # underflow pattern at loss scale 2-4; scattered per pair, not clustered per sample: 0.98 of the 64-sample tiles and
    # 0.73-0.97 of the samples have a non-zero pair on every level); point 0 carries the SDS pass as well: 0.94
    REAL_NZ = [0.51, 0.75, 0.72, 0.78, 0.74, 0.63, 0.2, 0.63, 0.38, 0.35, 0.6, 0.34, 0.25, 0.35, 0.72, 0.31]

    def gradient_planes(dtype):
        g = torch.randn(16, P * n, 2, device=dev).to(dtype)
        if a.real_census:
            for l in range(16):
                keep = torch.rand(P * n, device=dev) < REAL_NZ[l]
                keep[:n] = torch.rand(n, device=dev) < 0.94
                g[l] *= keep[:, None].to(dtype)
            res["real_census_nonzero_pair_fraction"] = float((g != 0).any(-1).float().mean())
        return g
    feats = torch.empty(16, P * n, 2, device=dev, dtype=torch.float16 if a.half_planes else torch.float32)

    def encode(out=feats):
        L.call("mi3d_grid_encode_points_planes", L.ptr(xs), L.ptr(xs2), n, offs_p, int(P0), P, 1.0, L.ptr(params), 16, 16,
               cfg["per_level_scale"], 19, 2 * 3 ** 0.5 / 1024, L.ptr(out), int(out.dtype == torch.float16), L.stream())

    if "encode" in what:
        tune(T_ENCODE_VARIANT, 0)
        encode()
        ref = feats.clone()
        for v in (0, 1, 3):
            tune(T_ENCODE_VARIANT, v)
            feats.zero_()
            ms = timeit(encode, a.iters)
            res[f"encode_variant{v}_ms"] = ms
            res[f"encode_variant{v}_bitexact"] = bool(torch.equal(feats, ref))
        del ref
        tune(T_ENCODE_VARIANT, -1)
        for w in (2, 3, 4, 6):
            tune(T_ENCODE_WGS, w)
            res[f"encode_wgs{w}_ms"] = timeit(encode, a.iters)
        tune(T_ENCODE_WGS, -1)
    if "encode1" in what:  # the product configuration of the gather, once
        res["encode_ms"] = timeit(encode, 5)
        it = torch.int16 if feats.dtype == torch.float16 else torch.int32   # bit-level checksum per level (A/B across builds)
        res["encode_checksum"] = [int(feats[l].view(it).to(torch.int64).sum()) for l in range(feats.shape[0])]
    if "encode_r05" in what:  # round 5: the x-group evaluation of the fine hashed levels (tunable 18), whole gather and per level
        tune(18, 0)
        encode()
        ref = feats.clone()
        out = {}
        for rep in range(3):
            for v in (0, 1):
                tune(18, v)
                feats.zero_()
                out[f"triple{v}_rep{rep}_ms"] = timeit(encode, a.iters)
                out[f"triple{v}_rep{rep}_bitexact"] = bool(torch.equal(feats, ref))
        for l in range(5, 16):
            tune(T_ENCODE_ONLY_LEVEL, l)
            for v in (0, 1):
                tune(18, v)
                out[f"level{l}_triple{v}_ms"] = timeit(encode, a.iters)
        tune(T_ENCODE_ONLY_LEVEL, -1)
        tune(18, -1)
        del ref
        res["encode_r05"] = out
    if "encode_sweep" in what:  # workgroups per CU for the coarse / fine segments, tiles claimed
        out = {}
        for c in (4, 6, 7):
            for f in (2, 3, 4, 5):
                if f > c:
                    continue
                tune(12, c)
                tune(T_ENCODE_WGS, f)
                out[f"coarse{c}_fine{f}"] = timeit(encode, 3)
        tune(12, -1)
        tune(T_ENCODE_WGS, -1)
        res["encode_ms_by_wgs_per_cu"] = out
    if "encode_xcds" in what:  # when each XCD finished each of its plan segments (dev build's timestamps, 100 MHz)
        import ctypes as C2
        buf = (C2.c_ulonglong * (8 * 17))()
        per = []
        for lds, static in ((-1, 0), (0, 0)):
            tune(16, lds)
            tune(17, static)
            ms = timeit(encode, 3)
            lib.mi3d_dev_encode_times(buf, 1)
            encode()
            torch.cuda.synchronize()
            lib.mi3d_dev_encode_times(buf, 0)
            t0 = min(buf[x * 17] for x in range(8))
            per.append({"lds_levels": lds, "static_tiles": static, "encode_ms": ms, "xcd_end_ms": [
                max(round((buf[x * 17 + i] - t0) / 1e5, 3) for i in range(1, 17)) for x in range(8)]})
        tune(17, -1)
        tune(16, -1)
        res["encode_xcd_timeline"] = per
    if "levels" in what:
        for v in (3,):
            tune(T_ENCODE_VARIANT, v)
            per = []
            for l in range(16):
                tune(T_ENCODE_ONLY_LEVEL, l)
                per.append(timeit(encode, 2))
            res[f"encode_variant{v}_per_level_ms"] = per
        tune(T_ENCODE_ONLY_LEVEL, -1)
        tune(T_ENCODE_VARIANT, -1)
        encode()
    if "levels_wgs" in what:  # do the coarse levels want more waves per SIMD than the fine ones?
        tune(T_ENCODE_VARIANT, 3)
        out = {}
        for w in (2, 3, 4, 5, 6, 8):
            tune(T_ENCODE_WGS, w)
            for l in (0, 2, 4, 5, 6, 7, 8, 9, 11, 13, 15):
                tune(T_ENCODE_ONLY_LEVEL, l)
                out[f"wgs{w}_level{l}"] = timeit(encode, 2)
        tune(T_ENCODE_ONLY_LEVEL, -1)
        tune(T_ENCODE_WGS, -1)
        tune(T_ENCODE_VARIANT, -1)
        res["encode_level_ms_by_wgs_per_cu"] = out
    if "mlp" in what:
        net = model.sigma_net.net
        ws = [t.detach().contiguous() for l in net for t in (l.weight, l.bias)]
        h = torch.empty(P * n, 4, device=dev)

        def fwd(rows=P * n):
            L.call("mi3d_mlp_forward", L.ptr(feats), P * n, int(feats.dtype == torch.float16), rows, *[L.ptr(t) for t in ws],
                   32, 64, 4, 1, L.ptr(h), L.stream())
        res["mlp_fwd_ms"] = timeit(fwd, a.iters)
        dh = torch.randn(P * n, 4, device=dev)
        dplanes = torch.empty(16, P * n, 2, device=dev, dtype=feats.dtype)
        grads = [torch.zeros_like(t) for t in ws]

        def bwd(rows=P * n):
            L.call("mi3d_mlp_backward", L.ptr(feats), P * n, int(feats.dtype == torch.float16), L.ptr(dh), rows,
                   *[L.ptr(t) for t in ws], 32, 64, 4, 1,
                   L.ptr(dplanes), rows, *[L.ptr(g) for g in grads], L.stream())
        for v in (0, 3):   # 0: round 3's kernels (two waves per SIMD, VGPR-form MFMA, transposes); 3: round 2's
            tune(6, v)
            res[f"mlp_fwd_variant{v}_ms"] = timeit(fwd, a.iters)
            for w in (1, 2):
                tune(T_MLP_WGS, w)
                res[f"mlp_bwd_variant{v}_wgs{w}_ms"] = timeit(bwd, a.iters)
            tune(T_MLP_WGS, -1)
        tune(6, -1)
        for w in (2, 3, 4, 5, 6):
            tune(11, w)   # MI3D_T_MLP_FWD_WGS_PER_CU
            res[f"mlp_fwd_wgs{w}_ms"] = timeit(fwd, a.iters)
        tune(11, -1)
        res["mlp_bwd_ms"] = timeit(bwd, a.iters)
        res["mlp_bwd_point0_only_ms"] = timeit(lambda: bwd(n), a.iters)
        res["mlp_bwd_TFLOPs"] = P * n * 25600.0 / res["mlp_bwd_ms"] / 1e9
        res["mlp_bwd_stream_TBps"] = P * n * (128 + 16 + 128) / res["mlp_bwd_ms"] / 1e9
        del dplanes, dh, h
    if "scatter13" in what:  # the 13-point scatter alone, as configured (--dev): for rocprofv3 --kernel-trace --stats
        g = gradient_planes(feats.dtype)
        torch.manual_seed(11)
        g = gradient_planes(feats.dtype)
        res["scatter_fp32_P13_ms"] = timeit(lambda: field_ops.scatter_binned(
            xs, xs2, offs, P0, 1.0, g, cfg, 2 * 3 ** 0.5 / 1024, 12196240), a.iters)
        out = field_ops.scatter_binned(xs, xs2, offs, P0, 1.0, g, cfg, 2 * 3 ** 0.5 / 1024, 12196240)
        res["scatter_fp32_P13_checksum"] = int(out.view(torch.int32).to(torch.int64).sum())   # (A/B across builds)
        del g, out
    if "scatter_ab" in what:  # the 13-point scatter (+ the deferred point-0 pair) under dev settings, dense and real census
        out = {}
        ex = torch.randn(16, n, 2, device=dev).to(feats.dtype)
        for census in (False, True):
            a.real_census = census
            g = gradient_planes(feats.dtype)
            for name, kvs in (("base", {}), ("no_face_pass", {10: 0x10000}), ("merge30", {15: 30}), ("merge58", {15: 58}),
                              ("fine_waves1024", {3: 1024}), ("fine_waves1280", {3: 1280}), ("fine_waves2048", {3: 2048}),
                              ("coarse_waves8192", {4: 8192}), ("base_again", {})):
                for k, v in kvs.items():
                    tune(k, v)
                out[("real_" if census else "dense_") + name] = timeit(lambda: field_ops.scatter_binned(
                    xs, xs2, offs, P0, 1.0, g, cfg, 2 * 3 ** 0.5 / 1024, 12196240, extra0=ex), a.iters)
                for k in kvs:
                    tune(k, -1)
            del g
        res["scatter13_plus_ms"] = out
    if "scatter_r05" in what:  # round 5: the coarse role's run-merged group flush, alone and in the whole scatter, with the level
        # threshold and the fine wave count re-swept around it; wall-clock stamps per configuration so that a clock / power
        # log sampled beside this process can be laid over the timings.  Dev bit 0x20000 of tunable 10 FLIPS the flush
        # against the build's default (MI3D_RUN_MERGE: 0 when profiles/kbench_r05_scatter_run_merge.json was taken - its
        # "run_merge" rows are the merged flush - and 1 since)
        import time
        out, stamps = {}, []
        ex = torch.randn(16, n, 2, device=dev).to(feats.dtype)
        RM = 0x20000
        for census in (False, True):
            a.real_census = census
            g = gradient_planes(feats.dtype)
            tag = "real_" if census else "dense_"
            ref = None
            MF = 0x40000   # flips the masked fused multiply-add of the coarse role's register sums (MI3D_MASK_FMA)
            for name, kvs in (("base", {}), ("run_merge", {10: RM}), ("base_2", {}), ("run_merge_2", {10: RM}),
                              ("mask_fma", {10: MF}), ("base_2b", {}), ("mask_fma_2", {10: MF}), ("base_2c", {}),
                              ("mask_fma_3", {10: MF}),
                              ("coarse_only_mask_fma", {5: 0x007F, 10: MF}),
                              ("dyn_idx", {10: 0x80000}), ("base_2d", {}), ("dyn_idx_2", {10: 0x80000}), ("base_2e", {}),
                              ("dyn_idx_3", {10: 0x80000}),
                              ("coarse_only_base", {5: 0x007F}), ("coarse_only_run_merge", {5: 0x007F, 10: RM}),
                              ("fine_only", {5: 0xFF80}),
                              ("run_merge_merge30", {10: RM, 15: 30}), ("run_merge_merge58", {10: RM, 15: 58}),
                              ("run_merge_fine1280", {10: RM, 3: 1280}), ("run_merge_fine2048", {10: RM, 3: 2048}),
                              ("base_3", {}), ("run_merge_3", {10: RM})):
                for k, v in kvs.items():
                    tune(k, v)
                t0 = time.time()
                out[tag + name] = timeit(lambda: field_ops.scatter_binned(
                    xs, xs2, offs, P0, 1.0, g, cfg, 2 * 3 ** 0.5 / 1024, 12196240, extra0=ex), a.iters)
                stamps.append((tag + name, t0, time.time()))
                if name in ("base", "run_merge", "mask_fma", "dyn_idx"):   # same gradient, up to the fp32 rounding of the register sums
                    got = field_ops.scatter_binned(xs, xs2, offs, P0, 1.0, g, cfg, 2 * 3 ** 0.5 / 1024, 12196240, extra0=ex)
                    if ref is None:
                        ref = got
                    else:
                        out[tag + name + "_max_err_rel"] = float((got - ref).abs().max() / ref.abs().max())
                    del got
                for k in kvs:
                    tune(k, -1)
            del g, ref
        res["scatter_r05_ms"] = out
        res["scatter_r05_stamps"] = stamps
    if "scatter_diag" in what:  # the coarse role alone, with its gather-table atomics switched off (timing only)
        g = gradient_planes(feats.dtype)
        out = {}
        tune(5, 0x00FF)
        for name, flags in (("all", 0), ("no_sum_adds", 0x100), ("no_cas", 0x200), ("neither", 0x300)):
            tune(10, flags)
            out[name] = timeit(lambda: field_ops.scatter_binned(xs, xs2, offs, P0, 1.0, g, cfg, 2 * 3 ** 0.5 / 1024,
                                                                12196240), 2)
        tune(10, -1)
        for l in (0, 2, 4, 5, 7):
            tune(5, 1 << l)
            for name, flags in (("all", 0), ("neither", 0x300)):
                tune(10, flags)
                out[f"level{l}_{name}"] = timeit(lambda: field_ops.scatter_binned(
                    xs, xs2, offs, P0, 1.0, g, cfg, 2 * 3 ** 0.5 / 1024, 12196240), 2)
        tune(10, -1)
        tune(5, -1)
        res["scatter13_dense_coarse_role_ms"] = out
        out = {}
        tune(5, 0xFF00)   # the fine role alone: sorted but not stored / cells, entries and histogram only
        for name, flags in (("all", 0), ("no_region_stores", 0x800), ("through_pass2", 0x4000), ("through_scan", 0x2000),
                            ("pass1_only", 0x1000), ("all_again", 0)):
            tune(10, flags)
            out[name] = timeit(lambda: field_ops.scatter_binned(xs, xs2, offs, P0, 1.0, g, cfg, 2 * 3 ** 0.5 / 1024,
                                                                12196240), 2)
        tune(10, -1)
        tune(5, -1)
        res["scatter13_dense_fine_role_ms"] = out
        del g
    if "scatter_levels" in what:  # the 13-point scatter per role and per level (dev level mask), dense random gradients
        g = gradient_planes(feats.dtype)
        masks = {"all": 0xFFFF, "fine_8_15": 0xFF00, "coarse_0_7": 0x00FF}
        masks.update({f"level{l}": 1 << l for l in range(16)})
        out = {}
        for name, mk in masks.items():
            tune(5, mk)
            out[name] = timeit(lambda: field_ops.scatter_binned(xs, xs2, offs, P0, 1.0, g, cfg, 2 * 3 ** 0.5 / 1024,
                                                                12196240), 2)
        tune(5, -1)
        res["scatter13_dense_ms_by_level_mask"] = out
        del g
    if "scatter_roles" in what:  # each role of the emit alone (dev level mask) against its wave count
        g = gradient_planes(feats.dtype)
        out = {}
        for name, mk, knob, values in (("fine_8_15", 0xFF00, T_EMIT_FINE, (768, 1024, 1536, 2048, 3072)),
                                        ("coarse_0_7", 0x00FF, T_EMIT_COARSE, (2048, 4096, 8192, 16384))):
            tune(5, mk)
            for v in values:
                tune(knob, v)
                out[f"{name}_waves{v}"] = timeit(lambda: field_ops.scatter_binned(xs, xs2, offs, P0, 1.0, g, cfg,
                                                                                  2 * 3 ** 0.5 / 1024, 12196240), 2)
            tune(knob, -1)
        tune(5, -1)
        res["scatter13_dense_role_ms_by_waves"] = out
        del g
    if "scatter" in what:
        step = 2 * 3 ** 0.5 / 1024
        g = gradient_planes(feats.dtype)
        res["scatter_fp32_P13_ms"] = timeit(lambda: field_ops.scatter_binned(
            xs, xs2, offs, P0, 1.0, g, cfg, step, 12196240), a.iters)
        g1 = g[:, :n].contiguous()
        res["planes_dtype"] = str(feats.dtype)
        res["scatter_fp32_P1_ms"] = timeit(lambda: field_ops.scatter_binned(
            xs, None, offs[:1], 1, 1.0, g1, cfg, step, 12196240), a.iters)
        for order in (0, 1):  # csrc/mi3d_dev.h MI3D_T_EMIT_ORDER: the fine role tile-major / level-major
            tune(10, order)
            for fw in (768, 1024, 1536, 2048, 3072):
                tune(T_EMIT_FINE, fw)
                res[f"scatter_fp32_P13_order{order}_fine_waves{fw}_ms"] = timeit(lambda: field_ops.scatter_binned(
                    xs, xs2, offs, P0, 1.0, g, cfg, step, 12196240), a.iters)
        tune(T_EMIT_FINE, -1)
        tune(10, -1)
    print(json.dumps(res, indent=1))
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
