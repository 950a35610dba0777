"""Training march alone (mi3d_march_rays_train through the raymarching op) at C2 dense and C4 pruned.
    [MI3D_LIB=tools/bin/libmi3d_dev_prev.so] python tools/march_bench.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "make-it-3d_amd")]
import torch  # noqa: E402


def main():
    if os.environ.get("MI3D_DEV_SET"):   # e.g. MI3D_DEV_SET=13=16,14=1024 with the dev library
        import ctypes as C
        from mi3d import _lib as L
        L.lib().mi3d_dev_set.argtypes = [C.c_int, C.c_int]
        for kv in os.environ["MI3D_DEV_SET"].split(","):
            L.lib().mi3d_dev_set(*map(int, kv.split("=")))
    import raymarching
    from mi3d import network, rays as R, sds_step
    dev = torch.device("cuda:0")
    out = {"lib": os.path.basename(os.environ.get("MI3D_LIB", "product")), "dev": os.environ.get("MI3D_DEV_SET", "")}
    for name, H, steps, bits in (("c2_dense", 128, 1024, "dense"), ("c4_pruned_view", 256, 2048, 0.5)):
        model = network.NeRFNetwork(sds_step.make_opt(max_steps=steps)).to(dev)
        sds_step.set_bitfield(model, bits)
        ro, rd, _ = R.view_rays(H, H, device=dev)
        ro, rd = ro.view(-1, 3).contiguous(), rd.view(-1, 3).contiguous()
        nears, fars = raymarching.near_far_from_aabb(ro, rd, model.aabb_train)

        def call():
            cnt = torch.zeros(2, dtype=torch.int32, device=dev)
            return raymarching.march_rays_train(ro, rd, 1.0, model.density_bitfield, 1, 128, nears, fars, cnt, -1, True,
                                                128, True, 0, steps)
        xs = call()[0]
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            call()
        e1.record()
        torch.cuda.synchronize()
        out[name] = {"samples": int(xs.shape[0]), "ms_per_call_incl_wrapper": e0.elapsed_time(e1) / 10}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
