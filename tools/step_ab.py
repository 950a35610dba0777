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
    ap.add_argument("--out", default="gpurun_out/step_ab.json")
    a = ap.parse_args()
    import bench
    from mi3d import _lib as L, grid_ops, rays as R, sd_standin, sds_step
    lib = L.lib()
    lib.mi3d_dev_set.argtypes = [C.c_int, C.c_int]
    configs = []
    for c in a.configs.split(";"):
        name, _, kv = c.partition(":")
        configs.append((name, [tuple(map(int, x.split("="))) for x in kv.split(",") if x]))
    used = sorted({k for _, kvs in configs for k, _ in kvs})
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
    for rnd in range(a.rounds):
        for name, kvs in configs:
            for k in used:
                lib.mi3d_dev_set(k, -1)
            for k, v in kvs:
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
            for key in ("scatter", "encode", "mlp_bwd"):
                res[name][key].append(sum(x.elapsed_time(y) for x, y in prof.get(key, [])) / a.steps)
    out = {"loss_scale": scaler.get_scale(), "settle_steps": tries,
           "workspace_cap_GiB": float(os.environ.get("MI3D_SCATTER_WORKSPACE_GB", "56")), "configs": {}}
    for name, kvs in configs:
        r = res[name]
        out["configs"][name] = {"tunables": dict(kvs), **{k: sum(v) / len(v) for k, v in r.items()}, "ms_all": r["ms"],
                                "scatter_all": r["scatter"]}
    print(json.dumps(out, indent=1))
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
