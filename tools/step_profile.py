"""Where one C2 SDS step spends its time, kernel by kernel, INCLUDING the torch-side glue (development aid).
    python tools/step_profile.py [--steps 2] [--scale 4.0] [--out gpurun_out/step_profile.txt]
torch.profiler (roctracer) sees every kernel of the process - the C-ABI launches as well as torch's elementwise /
reduction / MIOpen / hipBLASLt kernels - so the table's tail is the "un-named rest" of bench.py's kernels_ms_per_step.
Also times the guidance's U-Net forward eagerly and as a captured hipGraph (launch-bound or not?)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "make-it-3d_amd")]

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--scale", type=float, default=4.0)
    ap.add_argument("--workload", default="c2_dense")
    ap.add_argument("--out", default="gpurun_out/step_profile.txt")
    ap.add_argument("--no-graph", action="store_true")
    a = ap.parse_args()
    import bench
    from mi3d import dp, rays as R, sd_standin, sds_step
    wl = bench.WORKLOADS[a.workload]
    dev = torch.device("cuda:0")
    opt = sds_step.make_opt(max_steps=wl["max_steps"])
    model, optimizer, scaler = sds_step.build_training_state(opt, dev, seed=0, bitfield=wl["bitfield"],
                                                             init_scale=a.scale)
    bucket = dp.FlatGradBucket(model.parameters())
    guidance = sd_standin.StableDiffusionStandIn(dev)
    text_z = guidance.get_text_embeds()
    ro, rd, ds = R.view_rays(wl["H"], wl["W"], device=dev)

    def step():
        bucket.zero()
        sds_step.sds_train_step(model, guidance, text_z, optimizer, scaler, ro, rd, ds, wl["H"], wl["W"], opt,
                                sds_backward="reference", t=500, grad_sync=bucket.all_reduce_mean)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / a.steps * 1e3
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize()
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    lines = [f"un-profiled wall: {wall:.2f} ms/step, loss scale now {scaler.get_scale()}, {a.steps} profiled steps\n"]
    evs = [e for e in prof.key_averages() if e.device_time_total > 0 and e.device_type.name != "CPU"]
    evs.sort(key=lambda e: -e.device_time_total)
    tot = sum(e.device_time_total for e in evs)
    lines.append(f"total device kernel time {tot / a.steps / 1e3:.2f} ms/step over {len(evs)} distinct kernels\n")
    for e in evs[:120]:
        lines.append(f"{e.device_time_total / a.steps / 1e3:9.3f} ms/step  x{e.count / a.steps:7.1f}  {e.key[:150]}\n")
    n_launch = sum(e.count for e in evs) / a.steps
    lines.append(f"kernel launches per step: {n_launch:.0f}\n")
    open(a.out, "w").writelines(lines)
    sys.stdout.writelines(lines[:60])

    if not a.no_graph:   # the U-Net forward of the guidance: eager vs captured hipGraph
        dt = guidance.unet.conv_in.weight.dtype
        x = torch.randn(2, 4, 64, 64, device=dev, dtype=dt)
        t = torch.tensor([500], device=dev)
        ctx = text_z.to(dt)

        def timeit(fn, n=5):
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / n, (time.perf_counter() - t0) / n * 1e3
        with torch.no_grad():
            eager = timeit(lambda: guidance.unet(x, t, encoder_hidden_states=ctx))
            res = [f"unet eager: {eager[0]:.2f} ms device-span, {eager[1]:.2f} ms wall\n"]
            try:
                s = torch.cuda.Stream()
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    for _ in range(2):
                        guidance.unet(x, t, encoder_hidden_states=ctx)
                torch.cuda.current_stream().wait_stream(s)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    y = guidance.unet(x, t, encoder_hidden_states=ctx)
                gr = timeit(g.replay)
                res.append(f"unet hipGraph replay: {gr[0]:.2f} ms device-span, {gr[1]:.2f} ms wall\n")
            except Exception as e:  # noqa: BLE001
                res.append(f"unet graph capture failed: {e!r}\n")
        img = torch.rand(1, 3, 512, 512, device=dev, requires_grad=True)

        def vae():
            lat = guidance.encode_imgs(img)
            lat.backward(torch.randn_like(lat))
        v = timeit(vae)
        res.append(f"vae encode fwd+bwd (fp32 weights, autocast off): {v[0]:.2f} ms\n")

        def vae_ac():
            with torch.autocast("cuda", dtype=torch.float16):
                lat = guidance.encode_imgs(img)
            lat.backward(torch.randn_like(lat))
        v = timeit(vae_ac)
        res.append(f"vae encode fwd+bwd under autocast: {v[0]:.2f} ms\n")
        open(a.out, "a").writelines(res)
        sys.stdout.writelines(res)


if __name__ == "__main__":
    main()
