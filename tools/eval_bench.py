"""The inference loop of run_cuda (nerf/renderer.py:526-551) on an analytic occupancy, three ways: the reference's round
structure launched from the host (the alive count read back every 8 rounds, round 3's loop), the same rounds as 32 per
replayed hipGraph (round 4), and compact rounds under a row budget (round 5: mi3d.renderer `infer_schedule = "budget"`,
4 rounds per replayed graph) - plus the latter launched from the host.
    python tools/eval_bench.py [--out gpurun_out/eval_bench.json]
Random-weight field (density blob + noise), sphere occupancy 0.5, T_thresh 1e-4, max_steps 1024; wall-clock per render
(synchronised), median of 7 after 2 warm-ups (the first graph render also pays the captures)."""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "make-it-3d_amd")]

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/eval_bench.json")
    a = ap.parse_args()
    from mi3d import rays as R, sds_step
    dev = torch.device("cuda:0")
    opt = sds_step.make_opt(max_steps=1024, fp16=True)
    model, _, _ = sds_step.build_training_state(opt, dev, seed=0, bitfield=0.5)
    with torch.no_grad():
        model.encoder.params.uniform_(-0.5, 0.5)
    model.eval()
    res = {}
    for H in (128, 256, 512):
        ro, rd, ds = R.view_rays(H, H, device=dev)
        outs = {}
        for name, schedule, rounds in (("rounds0", "reference", 0), ("rounds32", "reference", 32),
                                       ("budget_host", "budget", 0), ("budget", "budget", 4)):
            model.infer_schedule = schedule
            model.infer_graph_rounds = rounds
            model.infer_budget_rounds = rounds
            ts = []
            for i in range(9):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
                    out = model.render(ro, rd, depth_scale=ds, bg_color=torch.ones(3, device=dev), perturb=False,
                                       ambient_ratio=1.0, shading="albedo", max_steps=1024)
                torch.cuda.synchronize()
                ts.append(1e3 * (time.perf_counter() - t0))
            outs[name] = out
            res[f"{H}x{H}_{name}"] = {"ms_median": statistics.median(ts[2:]), "ms_first": ts[0], "ms_min": min(ts[2:]),
                                      "stats": dict(model.infer_stats),
                                      "weights_sum_mean": float(out["weights_sum"].mean()),
                                      "peak_mem_GiB": torch.cuda.max_memory_allocated(dev) / 2 ** 30}
        a0, a1, a2 = res[f"{H}x{H}_rounds0"], res[f"{H}x{H}_rounds32"], res[f"{H}x{H}_budget"]
        res[f"{H}x{H}_speedup_graphs_vs_host"] = a0["ms_median"] / a1["ms_median"]
        res[f"{H}x{H}_speedup_budget_vs_reference_rounds"] = a1["ms_median"] / a2["ms_median"]
        res[f"{H}x{H}_budget_vs_reference_rounds_max_abs_diff"] = {
            k: float((outs["budget"][k].float() - outs["rounds32"][k].float()).abs().max())
            for k in ("image", "depth", "weights_sum", "normal")}
        res[f"{H}x{H}_budget_graph_equals_host_launched"] = all(
            bool(torch.equal(outs["budget"][k], outs["budget_host"][k])) for k in ("image", "depth", "weights_sum", "normal"))
    print(json.dumps(res, indent=1))
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
