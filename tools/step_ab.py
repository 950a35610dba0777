"""Whole SDS steps (bench.py's headline configuration: C2 dense, autocast, the reference's two-backward schedule, the SD
stand-in) under several settings of the development build's tunables, interleaved in ONE process on ONE box:
    python tools/build_dev.py && python tools/step_ab.py --configs "base:;merge42:15=42;merge58:15=58" [--rounds 3]
Each config is `name:k=v,k=v` (csrc/mi3d_dev.h indices; empty = product defaults).  The loss scale is settled first
(untimed), then the configs take turns, `--steps` timed steps each per round; reported: ms per step and the scatter's /
gather's / MLP backward's HIP-event time per step, mean over the rounds.  `MI3D_SCATTER_WORKSPACE_GB` applies as usual."""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "make-it-3d_amd")]
os.environ.setdefault("MI3D_LIB", os.path.join(ROOT, "tools", "bin", "libmi3d_dev.so"))

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="base:")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--workload", default="c2_dense")
    ap.add_argument("--anchor", action="store_true",
                    help="measure the FIRST config before and after every other one (A x A y A ...): the scatter gets "
                         "faster as the optimisation proceeds (more gradients underflow), so a config is compared with the "
                         "mean of the two anchor measurements around it (`scatter_vs_anchor`, `ms_vs_anchor`)")
    ap.add_argument("--out", default="gpurun_out/step_ab.json")
    a = ap.parse_args()
    import bench
    from mi3d import _lib as L, field_ops, grid_ops, rays as R, sd_standin, sds_step
    default_cap = field_ops.WORKSPACE_CAP_BYTES
    lib = L.lib()
    lib.mi3d_dev_set.argtypes = [C.c_int, C.c_int]
    configs = []
    for c in a.configs.split(";"):
        name, _, kv = c.partition(":")
        # `cap=<GiB>` is not a library tunable: the Python layer's arena cap (field_ops.WORKSPACE_CAP_BYTES)
        configs.append((name, [(x.split("=")[0] if x.startswith("cap=") else int(x.split("=")[0]), int(x.split("=")[1]))
                               for x in kv.split(",") if x]))
    used = sorted({k for _, kvs in configs for k, _ in kvs if k != "cap"})
    wl = bench.WORKLOADS[a.workload]
    dev = torch.device("cuda:0")
    opt = sds_step.make_opt(max_steps=wl["max_steps"])
    model, optimizer, scaler = sds_step.build_training_state(opt, dev, seed=0, bitfield=wl["bitfield"], init_scale=65536.0)
    guidance = sd_standin.StableDiffusionStandIn(dev)
    text_z = guidance.get_text_embeds()
    ro, rd, ds = R.view_rays(wl["H"], wl["W"], device=dev)
    torch.manual_seed(1234)

    def step():
        sds_step.sds_train_step(model, guidance, text_z, optimizer, scaler, ro, rd, ds, wl["H"], wl["W"], opt,
                                sds_backward="reference", t=bench.T_FIXED)
    good = tries = 0
    while good < 4 and tries < 60:
        before = scaler.get_scale()
        step()
        good = good + 1 if scaler.get_scale() >= before else 0
        tries += 1
    res = {name: {"ms": [], "scatter": [], "encode": [], "mlp_bwd": []} for name, _ in configs}
    order = list(configs)
    if a.anchor and len(configs) > 1:
        order = []
        for c in configs[1:]:
            order += [configs[0], c]
        order.append(configs[0])
    trace = []
    for rnd in range(a.rounds):
        for name, kvs in order:
            for k in used:
                lib.mi3d_dev_set(k, -1)
            field_ops.WORKSPACE_CAP_BYTES = default_cap
            for k, v in kvs:
                if k == "cap":
                    field_ops.WORKSPACE_CAP_BYTES = v << 30
                    torch.cuda.empty_cache()
                else:
                    lib.mi3d_dev_set(k, v)
            step()   # (one untimed step under the new setting: allocator, plans)
            grid_ops.PROFILE = {}
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                step()
            torch.cuda.synchronize()
            dt = 1e3 * (time.perf_counter() - t0) / a.steps
            prof, grid_ops.PROFILE = grid_ops.PROFILE, None
            res[name]["ms"].append(dt)
            trace.append([name, dt, sum(x.elapsed_time(y) for x, y in prof.get("scatter", [])) / a.steps])
            for key in ("scatter", "encode", "mlp_bwd"):
                res[name][key].append(sum(x.elapsed_time(y) for x, y in prof.get(key, [])) / a.steps)
    out = {"loss_scale": scaler.get_scale(), "settle_steps": tries,
           "workspace_cap_GiB": float(os.environ.get("MI3D_SCATTER_WORKSPACE_GB", "56")), "configs": {}}
    for name, kvs in configs:
        r = res[name]
        out["configs"][name] = {"tunables": {str(k): v for k, v in kvs}, **{k: sum(v) / len(v) for k, v in r.items()}, "ms_all": r["ms"],
                                "scatter_all": r["scatter"]}
    if a.anchor and len(configs) > 1:
        base = configs[0][0]
        delta = {}
        for i, (name, ms, sc) in enumerate(trace):
            if name == base or i == 0 or i + 1 >= len(trace) or trace[i - 1][0] != base or trace[i + 1][0] != base:
                continue
            delta.setdefault(name, {"ms": [], "scatter": []})
            delta[name]["ms"].append(ms - 0.5 * (trace[i - 1][1] + trace[i + 1][1]))
            delta[name]["scatter"].append(sc - 0.5 * (trace[i - 1][2] + trace[i + 1][2]))
        for name, d in delta.items():
            out["configs"][name]["ms_vs_anchor"] = sum(d["ms"]) / len(d["ms"])
            out["configs"][name]["scatter_vs_anchor"] = sum(d["scatter"]) / len(d["scatter"])
            out["configs"][name]["scatter_vs_anchor_all"] = d["scatter"]
    out["trace"] = trace
    print(json.dumps(out, indent=1))
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
