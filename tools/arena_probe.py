"""Does the 13-point scatter's time depend on WHERE its record arena lies?  (Three runs of bench.py's driver command timed
the in-step scatter at 32.0 / 36.1 / 42.3 ms on the same zero census, profiles/bench_r04_c2_dense*.json.)  One big buffer,
the arena placed at different offsets / alignments inside it, the same call timed at each:
    python tools/arena_probe.py [--out gpurun_out/arena_probe.json]"""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "make-it-3d_amd")]

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/arena_probe.json")
    a = ap.parse_args()
    import raymarching
    from mi3d import _lib as L, grid_ops, network, rays as R, sds_step
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
    n = xs.shape[0]
    xs2 = (xs + torch.randn_like(xs) * 0.01).contiguous()
    offs, P0 = grid_ops.stencil_offsets(center=True, second=True)
    _, offs_p = grid_ops._offs_arg(offs)
    P = 13
    nz = [0.71, 0.85, 0.76, 0.81, 0.83, 0.84, 0.76, 0.84, 0.82, 0.76, 0.66, 0.71, 0.62, 0.69, 0.59, 0.54]   # bench_r04_c2_dense
    g = torch.randn(16, P * n, 2, device=dev).to(torch.float16)
    for l in range(16):
        g[l] *= (torch.rand(P * n, device=dev) < nz[l])[:, None].to(torch.float16)
    ex = torch.randn(16, n, 2, device=dev).to(torch.float16)
    grad = torch.zeros(12196240, device=dev)
    gib = 1 << 30
    arena_bytes = 56 * gib
    buf = torch.empty(arena_bytes + 2 * gib, dtype=torch.uint8, device=dev)
    base = buf.data_ptr()
    res = {"buffer_ptr_mod_1GiB": base % gib, "buffer_ptr_mod_2MiB": base % (2 << 20), "ms": {}}

    def call(ptr):
        grad.zero_()
        L.call("mi3d_grid_scatter_binned_plus", L.ptr(xs), L.ptr(xs2), n, offs_p, int(P0), P, 1.0, L.ptr(g), L.ptr(ex), 1,
               16, 16, cfg["per_level_scale"], 19, 2 * 3 ** 0.5 / 1024, C.c_void_p(ptr), C.c_size_t(arena_bytes), L.ptr(grad),
               L.stream())

    def timed(ptr, reps=3):
        call(ptr)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            call(ptr)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    up = lambda p, al: (p + al - 1) // al * al
    places = {"as_allocated": base, "aligned_1GiB": up(base, gib), "aligned_1GiB_plus_2MiB": up(base, gib) + (2 << 20),
              "aligned_2MiB_plus_64KiB": up(base, 2 << 20) + (64 << 10), "plus_4KiB": base + 4096, "plus_256B": base + 256,
              "aligned_1GiB_again": up(base, gib), "as_allocated_again": base}
    for name, ptr in places.items():
        res["ms"][name] = timed(ptr)
    # and the product's way: a fresh torch allocation per call, with ~30 GB of other tensors live (as in a step)
    del buf
    torch.cuda.empty_cache()
    ballast = [torch.empty(9 * gib, dtype=torch.uint8, device=dev) for _ in range(3)]
    fresh = []
    for i in range(4):
        ws = torch.empty(arena_bytes, dtype=torch.uint8, device=dev)
        fresh.append({"ptr_mod_1GiB": ws.data_ptr() % gib, "ms": timed(ws.data_ptr())})
        del ws
        if i == 1:
            ballast.pop()
            torch.cuda.empty_cache()
    res["fresh_allocations_with_ballast"] = fresh
    # physically contiguous memory (hipExtMallocWithFlags, hipDeviceMallocContiguous) against plain hipMalloc, with the
    # ballast still live; torch does not see these bytes
    hip = C.CDLL("libamdhip64.so")
    hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipFree.argtypes = [C.c_void_p]
    torch.cuda.empty_cache()
    direct = []
    for kind, flags in (("hipMalloc", None), ("contiguous", 0x4), ("hipMalloc", None), ("contiguous", 0x4)):
        ptr = C.c_void_p()
        err = hip.hipMalloc(C.byref(ptr), arena_bytes) if flags is None else hip.hipExtMallocWithFlags(C.byref(ptr), arena_bytes, flags)
        if err != 0 or not ptr.value:
            direct.append({"kind": kind, "error": int(err)})
            continue
        direct.append({"kind": kind, "ptr_mod_1GiB": ptr.value % gib, "ms": timed(ptr.value)})
        torch.cuda.synchronize()
        hip.hipFree(ptr)
    res["direct_hip_allocations"] = direct
    print(json.dumps(res, indent=1))
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
