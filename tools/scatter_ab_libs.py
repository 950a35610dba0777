"""A/B of the binned scatter across several PRODUCT-GRADE builds of the library in ONE process (the tools build carries
every variant's code behind run-time switches, which changes the register allocation of all of them: a switch that is
'off' there is not the product).  Variant builds: `python make-it-3d_amd/build.py`-style, e.g.
    python -c "import sys; sys.path.insert(0,'make-it-3d_amd'); import build; build.build(out='tools/bin/libmi3d_dyn1.so', defines=('-DMI3D_DYN_IDX=1',))"
    python tools/scatter_ab_libs.py --libs make-it-3d_amd/csrc/libmi3d.so,tools/bin/libmi3d_dyn1.so --out gpurun_out/scatter_ab_libs.json
The 13-point scatter + deferred point-0 pair of the C2-dense view on dense random binary16 gradients and on a real step's
zero census, the libraries interleaved (A B A B ...) `--rounds` times; every library's gradient against the first one's."""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "make-it-3d_amd")]

import torch  # noqa: E402


def load(path):
    from mi3d import _lib as L
    lib = C.CDLL(os.path.abspath(path))
    for name, args in L._SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes, fn.restype = args, C.c_int
    lib.mi3d_abi_version.restype = C.c_int
    lib.mi3d_hashgrid_levels.restype = C.c_uint32
    lib.mi3d_hashgrid_levels.argtypes = [C.c_uint32, C.c_uint32, C.c_float, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.mi3d_points_rasterize_workspace.restype = C.c_size_t
    lib.mi3d_points_rasterize_workspace.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_float]
    lib.mi3d_last_error_string.restype = C.c_char_p
    lib.mi3d_last_error_string.argtypes = [C.c_int]
    lib.mi3d_grid_scatter_binned_workspace.restype = C.c_size_t
    lib.mi3d_grid_scatter_binned_workspace.argtypes = [C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.c_uint32, C.c_uint32,
                                                       C.c_float, C.c_uint32]
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", required=True)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--out", default="gpurun_out/scatter_ab_libs.json")
    ap.add_argument("--capture", type=int, default=0,
                    help="N > 0: also time the libraries on the gradient planes of a REAL step - the 13-point scatter call (with "
                         "its riding point-0 planes) of the N-th SDS step after the loss scale has settled, captured once with "
                         "the first library (`captured_ms`; the synthetic 'real' census has independent zeros, a real step's "
                         "are clustered by tile, which is what the emit's load balance sees)")
    a = ap.parse_args()
    import raymarching
    from mi3d import _lib as L, field_ops, grid_ops, network, rays as R, sds_step
    paths = a.libs.split(",")
    libs = [load(p) for p in paths]
    L._lib = libs[0]
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
    # per-level share of non-zero pairs of a real C2 step (profiles/bench_r05_c2_dense.json, both passes in one scatter)
    REAL_NZ = [0.72, 0.85, 0.77, 0.8, 0.83, 0.85, 0.76, 0.84, 0.82, 0.77, 0.63, 0.7, 0.55, 0.64, 0.72, 0.59]
    ex = torch.randn(16, n, 2, device=dev).half()
    res = {"libs": paths, "samples": n}
    step = 2 * 3 ** 0.5 / 1024

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

    for census in ("dense", "real"):
        torch.manual_seed(5)
        g = torch.randn(16, P * n, 2, device=dev).half()
        if census == "real":
            for l in range(16):
                keep = torch.rand(P * n, device=dev) < REAL_NZ[l]
                keep[:n] = torch.rand(n, device=dev) < 0.94
                g[l] *= keep[:, None].half()
        call = lambda: field_ops.scatter_binned(xs, xs2, offs, P0, 1.0, g, cfg, step, 12196240, extra0=ex)   # noqa: E731
        ref, times = None, {os.path.basename(p): [] for p in paths}
        for i, (p, lib) in enumerate(zip(paths, libs)):
            L._lib = lib
            out = call()
            if ref is None:
                ref = out
            else:
                res[f"{census}:{os.path.basename(p)}:max_err_rel_vs_first"] = float((out - ref).abs().max() / ref.abs().max())
            del out
        for r in range(a.rounds):
            for p, lib in zip(paths, libs):
                L._lib = lib
                times[os.path.basename(p)].append(timeit(call))
        res[census + "_ms"] = times
        del g, ref
    if a.capture > 0:
        import bench
        from mi3d import sd_standin
        del xs, xs2, ex
        torch.cuda.empty_cache()
        L._lib = libs[0]
        wl = bench.WORKLOADS["c2_dense"]
        opt = sds_step.make_opt(max_steps=wl["max_steps"])
        model, optimizer, scaler = sds_step.build_training_state(opt, dev, seed=0, bitfield=wl["bitfield"], init_scale=65536.0)
        guidance = sd_standin.StableDiffusionStandIn(dev)
        text_z = guidance.get_text_embeds()
        ro, rd, ds = R.view_rays(wl["H"], wl["W"], device=dev)
        torch.manual_seed(1234)

        def sds():
            sds_step.sds_train_step(model, guidance, text_z, optimizer, scaler, ro, rd, ds, wl["H"], wl["W"], opt,
                                    sds_backward="reference", t=bench.T_FIXED)
        good = tries = 0
        while good < 4 and tries < 60:
            before = scaler.get_scale()
            sds()
            good = good + 1 if scaler.get_scale() >= before else 0
            tries += 1
        for _ in range(max(a.capture - 1, 0)):
            sds()
        got = {}
        orig = field_ops.scatter_binned

        def spy(x, x2, offsets, P0_, bound, dplanes, cfg_, step_, n_params, workspace_bytes=None, extra0=None):
            if offsets.shape[0] == 13:
                got.update(x=x, x2=x2, offsets=offsets, P0=P0_, bound=bound, dplanes=dplanes.clone(), cfg=cfg_, step=step_,
                           n_params=n_params, extra0=None if extra0 is None else extra0.clone())
            return orig(x, x2, offsets, P0_, bound, dplanes, cfg_, step_, n_params, workspace_bytes, extra0)
        field_ops.scatter_binned = spy
        sds()
        field_ops.scatter_binned = orig
        del model, optimizer, guidance
        torch.cuda.empty_cache()
        nzp = (got["dplanes"] != 0).any(-1).float().mean().item()
        res["captured"] = {"loss_scale": scaler.get_scale(), "steps_before": tries + a.capture - 1, "nonzero_pair_fraction": nzp,
                           "samples": int(got["x"].shape[0]), "with_point0_planes": got["extra0"] is not None}
        call = lambda: field_ops.scatter_binned(got["x"], got["x2"], got["offsets"], got["P0"], got["bound"], got["dplanes"],   # noqa: E731
                                                got["cfg"], got["step"], got["n_params"], extra0=got["extra0"])
        ref, times = None, {os.path.basename(p): [] for p in paths}
        for p, lib in zip(paths, libs):
            L._lib = lib
            out = call()
            if ref is None:
                ref = out
            else:
                res[f"captured:{os.path.basename(p)}:max_err_rel_vs_first"] = float((out - ref).abs().max() / ref.abs().max())
            del out
        for r in range(a.rounds):
            for p, lib in zip(paths, libs):
                L._lib = lib
                times[os.path.basename(p)].append(timeit(call))
        res["captured_ms"] = times
    print(json.dumps(res, indent=1))
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
