"""A/B of the MLP backward across several builds of the library IN ONE PROCESS (all loaded with ctypes, same inputs,
same box): equality of everything the kernel writes, for every template instance, and timings at the headline size.

    python tools/build_dev.py tools/bin/libmi3d_dev_tr0.so -DMI3D_MLP_LDS_TRANSPOSE=0      # round 3's form: the reference
    python tools/build_dev.py tools/bin/libmi3d_dev_f1l1.so -DMI3D_MLP_BWD_FULL=1 -DMI3D_MLP_BWD_LATE_PREFETCH=1
    python tools/mlp_ab.py --libs tools/bin/libmi3d_dev_tr0.so,tools/bin/libmi3d_dev_f1l1.so --out gpurun_out/mlp_ab.json

The input gradient planes must be EQUAL as numbers to the first library's (a transposition moves values; -0.0 against
+0.0 is the one difference the two ways of turning a tile may leave, and the scatter skips both); the weight gradients
agree to the order their float atomics landed in (1e-4 relative).  Exit code 1 if they do not."""
import argparse
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "make-it-3d_amd"))


def load(path):
    from mi3d import _lib as L
    lib = C.CDLL(os.path.abspath(path))
    for name in ("mi3d_mlp_backward", "mi3d_mlp_forward"):
        fn = getattr(lib, name)
        fn.argtypes, fn.restype = L._SIGNATURES[name], C.c_int
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--a", default="")
    ap.add_argument("--b", default="")
    ap.add_argument("--libs", default="", help="comma-separated libraries, the first one is the reference (instead of --a / --b)")
    ap.add_argument("--rows", type=int, default=13 * 10_878_976)  # the headline's evaluations per step
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--out", default="gpurun_out/mlp_ab.json")
    ap.add_argument("--timing-only", action="store_true",
                    help="the two timed cases only, no equality verdict (for builds whose results are wrong on purpose: "
                         "csrc/field.hip MI3D_MLP_BWD_TIMING_CUT)")
    a = ap.parse_args()
    from mi3d import _lib as L
    paths = a.libs.split(",") if a.libs else [a.a, a.b]
    libs = {os.path.basename(p_): load(p_) for p_ in paths}
    ref = os.path.basename(paths[0])
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(7)
    st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
    res = {"libs": paths, "reference": ref, "cases": []}
    ok = True

    def case(rows, din, hid, layers, half_planes, half_mode, time_it):
        nonlocal ok
        pdt = torch.float16 if half_planes else torch.float32
        feats = (torch.randn(din // 2, rows, 2, device=dev, generator=g) * 0.5).to(pdt)
        dh = torch.randn(rows, 4, device=dev, generator=g)
        dims = [(hid, din)] + ([(hid, hid)] if layers == 3 else []) + [(4, hid)]
        ws = []
        for (o, i) in dims:
            ws += [torch.randn(o, i, device=dev, generator=g) * (2.0 / i) ** 0.5, torch.randn(o, device=dev, generator=g) * 0.1]
        if layers == 2:
            ws = ws[:2] + [None, None] + ws[2:]
        rec = {"rows": rows, "din": din, "hidden": hid, "layers": layers, "half_planes": half_planes, "half_mode": half_mode}
        view = torch.int16 if half_planes else torch.int32
        ref_out, runs = None, {}
        for k, lib in libs.items():
            dplanes = torch.full((din // 2, rows, 2), float("nan"), device=dev, dtype=pdt)
            grads = [torch.zeros_like(t) if t is not None else None for t in ws]

            def run(lib=lib, dplanes=dplanes, grads=grads):
                err = lib.mi3d_mlp_backward(L.ptr(feats), rows, int(half_planes), L.ptr(dh), rows, *[L.ptr(t) for t in ws], din, hid, 4,
                                            int(half_mode), L.ptr(dplanes), rows, *[L.ptr(t) for t in grads], st())
                if err:
                    raise RuntimeError(f"mi3d_mlp_backward: hipError {err}")
            run()
            torch.cuda.synchronize()
            runs[k] = run
            if ref_out is None:
                ref_out = (dplanes, grads)
                rec["dx_nonzero_fraction"] = float((dplanes != 0).float().mean())
                if bool(torch.isnan(dplanes.float()).any()):
                    ok = False
                    rec["reference_left_rows_unwritten"] = True
                continue
            same = bool((dplanes == ref_out[0]).all())
            werr = 0.0
            for x, y in zip(ref_out[1], grads):
                if x is not None:
                    werr = max(werr, float((x - y).abs().max() / x.abs().max().clamp_min(1e-30)))
            rec[k] = {"dx_equal": same, "dx_bit_identical": bool((dplanes.view(view) == ref_out[0].view(view)).all()),
                      "weight_grad_max_rel_diff": werr}
            ok = ok and same and werr < 1e-4
            del dplanes, grads
        if time_it:   # two rounds over the libraries, so that a drift of the chip's clocks shows as a spread, not as a winner
            ms = {k: [] for k in libs}
            for _ in range(2):
                for k, run in runs.items():
                    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
                    ev[0].record()
                    for _ in range(a.iters):
                        run()
                    ev[1].record()
                    torch.cuda.synchronize()
                    ms[k].append(round(ev[0].elapsed_time(ev[1]) / a.iters, 4))
            rec["ms"] = ms
        res["cases"].append(rec)
        print(json.dumps(rec), flush=True)
        del runs, ref_out
        del feats, dh
        torch.cuda.empty_cache()

    # every template instance of k_mlp_bwd_g (hidden 32 / 64 x 2 / 3 layers x plane type), ragged row counts, narrow inputs
    for hid in (() if a.timing_only else (64, 32)):
        for layers in (3, 2):
            for half_planes, half_mode in ((True, True), (False, True), (False, False)):
                for din, rows in ((32, 1_000_003), (14, 50_001)):
                    case(rows, din, hid, layers, half_planes, half_mode, False)
    case(a.rows, 32, 64, 3, True, True, True)   # the headline: 13 points x 10.9 M samples, binary16 planes
    case(a.rows // 13, 32, 64, 3, True, True, True)   # the point-0 pass
    if a.timing_only:
        ok = True
    res["ok"], res["timing_only"] = ok, a.timing_only
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)
    print("MLP_AB", "OK" if ok else "MISMATCH")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
