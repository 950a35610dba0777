"""One C2 view through the product's NeRF kernels only - march, 13-point field (gather, MLP, head), composite, the
two backward passes of the reference's SDS schedule (point 0 only, then all 13 points) - without the diffusion
networks: the command the rocprofv3 kernel-trace / PMC passes of profiles/ are collected on.
    python tools/field_bench.py [--iters 2] [--workload c2_dense]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "make-it-3d_amd")]


def _dev_tunables(spec):
    """--dev 3=2048,4=8192: load the -DMI3D_DEV build (tools/build_dev.py) and set its tunables (csrc/mi3d_dev.h)."""
    import ctypes as C
    os.environ["MI3D_LIB"] = os.path.join(ROOT, "tools", "bin", "libmi3d_dev.so")
    from mi3d import _lib as L
    lib = L.lib()
    lib.mi3d_dev_set.argtypes = [C.c_int, C.c_int]
    for kv in spec.split(","):
        k, v = kv.split("=")
        lib.mi3d_dev_set(int(k), int(v))


import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=2)
    ap.add_argument("--workload", default="c2_dense")
    ap.add_argument("--dev", default="", help="development build tunables, e.g. 3=2048 (emit fine waves)")
    ap.add_argument("--scale", type=float, default=4.0,
                    help="GradScaler loss scale (bench.py settles at 4.0 on this workload: profile at the SAME state)")
    ap.add_argument("--lib", default="", help="load this build of the library (A/B against an older one)")
    a = ap.parse_args()
    if a.lib:
        os.environ["MI3D_LIB"] = a.lib
    if a.dev:
        _dev_tunables(a.dev)
    import bench
    from mi3d import field_ops, rays as R, sds_step
    wl = bench.WORKLOADS[a.workload]
    dev = torch.device("cuda:0")
    opt = sds_step.make_opt(max_steps=wl["max_steps"])
    model, optimizer, scaler = sds_step.build_training_state(opt, dev, seed=0, bitfield=wl["bitfield"], init_scale=a.scale)
    ro, rd, ds = R.view_rays(wl["H"], wl["W"], device=dev)
    for i in range(a.iters):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        optimizer.zero_grad(set_to_none=False)
        with torch.autocast("cuda", dtype=torch.float16, enabled=True):
            out = model.render(ro, rd, depth_scale=ds, bg_color=torch.rand(3, device=dev), perturb=True,
                               force_all_rays=True, **sds_step.render_kwargs(opt))
            img = out["image"]
            img.backward(torch.randn_like(img) * 1e-3, retain_graph=True)     # the SDS pass: reaches point 0 only
            loss = sds_step.regularisers(opt, out, out["weights_sum"].reshape(1, 1, wl["H"], wl["W"]))
        scaler.scale(loss).backward()                                          # the regulariser pass: all 13 points
        torch.cuda.synchronize()
        m = int(model.step_counter[(model.local_step - 1) % 16, 0])
        print(f"iter {i}: {1e3 * (time.perf_counter() - t0):.1f} ms, {m} samples, {13 * m} field evaluations", flush=True)


if __name__ == "__main__":
    main()
