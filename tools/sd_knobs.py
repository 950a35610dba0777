"""The diffusion half of an SDS step alone - 512 x 512 resize, VAE encode (with graph), U-Net on the CFG pair (hipGraph
replay), SDS gradient, latents.backward down to the rendered image - with each stock-PyTorch knob of mi3d/sd_standin.py
off and on:   python tools/sd_knobs.py [--out gpurun_out/sd_knobs.json]
HIP-event time per call under torch.autocast(float16), mean of 10 after 3 warm-ups; every configuration on one box."""
import argparse
import itertools
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "make-it-3d_amd")]

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/sd_knobs.json")
    a = ap.parse_args()
    from mi3d import sd_standin as S
    dev = torch.device("cuda:0")
    g = S.StableDiffusionStandIn(dev)
    text_z = g.get_text_embeds()
    res = {}
    ref_grad = None
    for gn, vh, vg in ((False, False, False), (True, False, False), (True, False, False)):
        S.GN_SPLIT_STATS, S.VAE_HALF_CACHE, S.VAE_GRAPH = gn, vh, vg
        ts, parts = [], {"encode": [], "unet": [], "backward": []}
        for i in range(13):
            torch.manual_seed(5)
            rgb = torch.rand(1, 3, 128, 128, device=dev, requires_grad=True)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            with torch.autocast("cuda", dtype=torch.float16):
                ev[0].record()
                latents, noise, t, _ = g._encode_view(rgb * 1.0, 500)
                ev[1].record()
                _, eps = g._guided_eps(text_z, latents, noise, t, 10.0)
                with torch.no_grad():
                    grad = torch.nan_to_num((1 - g.alphas[t]) * (eps - noise))
                ev[2].record()
                latents.backward(gradient=grad)
                ev[3].record()
            torch.cuda.synchronize()
            if i >= 3:
                ts.append(ev[0].elapsed_time(ev[3]))
                for k, (x, y) in zip(parts, ((0, 1), (1, 2), (2, 3))):
                    parts[k].append(ev[x].elapsed_time(ev[y]))
        key = f"gn_split={int(gn)},vae_half_cache={int(vh)},vae_graph={int(vg)}"
        while key in res:
            key += "'"
        res[key] = {"ms": sum(ts) / len(ts), **{k: sum(v) / len(v) for k, v in parts.items()}}
        gr = rgb.grad.detach().clone()
        if ref_grad is None:
            ref_grad = gr
        res[key]["image_grad_max_rel_diff_vs_first"] = float((gr - ref_grad).abs().max() / ref_grad.abs().max())
    S.GN_SPLIT_STATS, S.VAE_HALF_CACHE, S.VAE_GRAPH = True, False, False
    # PyTorch's TunableOp: every GEMM shape benchmarked once against rocBLAS / hipBLASLt's candidate solutions (the search is
    # timed: it runs in the first call, eagerly, before the U-Net graph is captured again on a fresh guidance object)
    import time
    try:
        torch.cuda.tunable.enable(True)
        torch.cuda.tunable.tuning_enable(True)
        torch.cuda.tunable.write_file_on_exit(False) if hasattr(torch.cuda.tunable, "write_file_on_exit") else None
        torch.cuda.tunable.set_max_tuning_duration(10)
        torch.cuda.tunable.set_max_tuning_iterations(20)
        g2 = S.StableDiffusionStandIn(dev)
        ts, parts, first = [], {"encode": [], "unet": [], "backward": []}, None
        for i in range(13):
            torch.manual_seed(5)
            rgb = torch.rand(1, 3, 128, 128, device=dev, requires_grad=True)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            t0 = time.perf_counter()
            with torch.autocast("cuda", dtype=torch.float16):
                ev[0].record()
                latents, noise, t, _ = g2._encode_view(rgb * 1.0, 500)
                ev[1].record()
                _, eps = g2._guided_eps(text_z, latents, noise, t, 10.0)
                with torch.no_grad():
                    grad = torch.nan_to_num((1 - g2.alphas[t]) * (eps - noise))
                ev[2].record()
                latents.backward(gradient=grad)
                ev[3].record()
            torch.cuda.synchronize()
            if i == 0:
                first = time.perf_counter() - t0
            if i >= 3:
                ts.append(ev[0].elapsed_time(ev[3]))
                for k, (x, y) in zip(parts, ((0, 1), (1, 2), (2, 3))):
                    parts[k].append(ev[x].elapsed_time(ev[y]))
        res["gn_split=1,tunableop=1"] = {"ms": sum(ts) / len(ts), **{k: sum(v) / len(v) for k, v in parts.items()},
                                         "first_call_s_incl_tuning": first,
                                         "tuned_gemms": len(torch.cuda.tunable.get_results())}
    except Exception as e:  # noqa: BLE001
        res["gn_split=1,tunableop=1"] = {"error": repr(e)}
    finally:
        torch.cuda.tunable.enable(False)
    print(json.dumps(res, indent=1))
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
