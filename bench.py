#!/usr/bin/env python
"""SDS train-steps/sec of Make-It-3D's coarse stage on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2_dense|c2_pruned|c4_views|c5_refine|small]

One step = one pass of the hot path over one novel view (BASELINE config 2): 128x128 rays, max_steps 1024,
16-level hash grid + 3x64 MLP (fp16 autocast), march -> 13-point field -> composite, SD2-base-shaped U-Net noise
prediction (batch 2, CFG) + VAE encode, SDS gradient + regularisers, backward, clip, Adan step.  Synthetic rays from the
reference's orbit poses, random-init weights, analytic occupancy (dense = all ones: every ray emits ~664 samples,
m ~ 10.9 M samples/step).  Inputs are resident in HBM before the timed region.

`--gpus N` with N > 1 starts N ranks itself (re-exec under torch.distributed.run, one rank per GPU) unless it is
already running inside such a launch (WORLD_SIZE set, as the driver does); a world size that differs from --gpus is
an error.  Every rank renders its own view (phi = 30 + 45 k degrees), the NeRF gradients are averaged with one flat
RCCL all-reduce per step; `value` counts view-steps of all ranks per second (weak scaling).

The HEADLINE (`value`, `ms_per_step`) is the like-for-like configuration: what the reference's main.py gets with the
drop-in packages and NO edits to nerf/sd.py / nerf/utils.py - the reference's two-backward SDS schedule
(`latents.backward(retain_graph=True)` then `scaler.scale(loss).backward()`) - and fp32 gradient contributions in the
hash-grid scatter (tiny-cuda-nn adds fp32 products; the binary16-record mode of round 1 is gone: the 16-byte fp32
pair records of round 2 are faster than it was).  `variants_ms_per_step` adds two schedules that need edits to nerf/sd.py /
nerf/utils.py, each timed on its own steps: the hand-merged single backward, and "overlapped" = the reference's two
backward passes with the U-Net on a second HIP stream under the regulariser pass (mi3d/sds_step.py).

The JSON line also carries
  roofline     : the step's dominant kernel - algorithmic bytes per launch (1024 B per field evaluation
                 gathered, 2048 B read-modify-written by the scatter) over its launch duration measured with HIP events
                 on the launch stream during the timed steps; the other hot kernels under rooflines_other; `traffic` =
                 HBM bytes per launch from the committed rocprofv3 PMC passes of THIS command's --profile-run
                 (profiles/pmc_r05.json; a counter pass wraps the process from outside and costs a run of its own, so it
                 cannot be taken inside the timed process: `traffic_source` says which file and how it was collected);
  cpu_baseline : the reference's own pure-PyTorch renderer (nerf/renderer.py:332-479 `run` + nerf/network_tcnn.py on a
                 torch hash grid; staged sources, oracle/_ref/py) forward + backward on a bounded ray sample on this
                 box's host cores (kind "reference"); the C oracle port when the staged sources are absent ("port");
  reference_shaped_baseline : the SAME step with the reference's own NeRFNetwork / run_cuda Python running unchanged
                 on the drop-in raymarching + tinycudann packages (13 encoder passes, torch MLP, atomic scatter) on
                 this GPU - the zero-change integration route and the only same-GPU anchor the north-star's ">= 10x"
                 can have (the reference's CUDA build cannot run here).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "make-it-3d_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

WORKLOADS = {
    # name: H, W, max_steps, analytic occupancy, views rendered per step and rank
    "c2_dense": dict(H=128, W=128, max_steps=1024, bitfield="dense", views=1),
    "c2_pruned": dict(H=128, W=128, max_steps=1024, bitfield=0.3, views=1),
    # BASELINE config 4: forward render of a batch of 4 views, pruned occupancy, 7 field evaluations per
    # sample (no smoothness pass) - the hash-gather stress; value = view-renders/s, no backward / optimizer
    "c4_views": dict(H=256, W=256, max_steps=2048, bitfield=0.5, views=4, mode="render"),
    # BASELINE config 5: refine stage - textured point cloud rasterised at 512 x 512 (+ 256, 128 for the
    # multi-scale U-Net and a coverage mask), gated U-Net, SD U-Net SDS step, Adam; value = refine-steps/s
    "c5_refine": dict(H=512, W=512, points=500_000, ppp=8, radius_px=2.0, mode="refine"),
    # the same at the reference's own default resolution (main.py --H / --W 800; nerf/utils.py:839-894 renders at opt.H)
    "c5_refine_800": dict(H=800, W=800, points=500_000, ppp=8, radius_px=2.0, mode="refine"),
    "small": dict(H=32, W=32, max_steps=128, bitfield="dense", views=1),
}
HBM_PEAK_GBPS = 8000.0  # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense f16/bf16 MFMA peak, same guide
ENCODE_BYTES_PER_EVAL = 1024   # 16 levels x 8 corners x 2 features x 4 B gathered per field evaluation
SCATTER_BYTES_PER_EVAL = 2048  # the same bytes read-modify-written by the gradient scatter
T_FIXED = 500  # SDS branch under the reference's rule (nerf/sd.py:153: t/1000 <= 0.4 takes the CLIP branch instead)
HEADLINE = ("fp32", "reference")


def ensure_built():
    """libmi3d.so is git-ignored: build it if this checkout has none (hipcc cross-compiles; seconds when cached)."""
    so = os.path.join(ROOT, "make-it-3d_amd", "csrc", "libmi3d.so")
    if not os.path.exists(so):
        import importlib.util
        spec = importlib.util.spec_from_file_location("mi3d_build", os.path.join(ROOT, "make-it-3d_amd", "build.py"))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        m.build()
    return so


def pmc_traffic(kernel, workload, evals):
    """(HBM bytes per launch, the file they come from) of `kernel`: the committed rocprofv3 PMC passes (FETCH_SIZE and
    WRITE_SIZE collected in separate --pmc runs, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950),
    scaled to this run's evaluation count; (None, None) if not collected."""
    for name in ("pmc_r06.json", "pmc_r05.json", "pmc_r04.json", "pmc_r03.json", "pmc_r02.json", "pmc_r01.json"):
        try:
            rec = json.load(open(os.path.join(ROOT, "profiles", name)))[kernel][workload]
            return rec["hbm_bytes_per_eval"] * evals, "profiles/" + name
        except Exception:
            continue
    return None, None


def _c1_network(ref_import, device="cpu"):
    """BASELINE config 1's field on the REFERENCE's classes: L=4 hash grid (16/81/407/2048) + Linear(8,32)-ReLU-
    Linear(32,4), table U(-1e-4, 1e-4), default nn.Linear init, blob 5 / 0.1, seed 0."""
    import torch
    from nerf.network_tcnn import MLP
    from oracle import oracle as O
    from oracle.field_torch import HashGridTorch
    torch.manual_seed(0)
    c1 = O.GridConfig(n_levels=4, per_level_scale=128 ** (1 / 3))
    net = ref_import.reference_network(ref_import.default_opt(cuda_ray=False, lambda_smooth=0.0), "oracle",
                                       num_layers=2, hidden_dim=32)
    net.encoder = HashGridTorch(c1)
    net.sigma_net = MLP(c1.n_output_dims, 4, 32, 2, bias=True)
    return net.to(device)


def cpu_baseline_c1():
    """BASELINE.md section 2, to the letter: the reference's own NeRFRenderer.run (nerf/renderer.py:332-479) on config 1
    WHOLE - 64x64 rays of the orbit pose, 64 samples per ray, no importance pass, L=4 + 2x32 field, fp32, bg ones -
    forward only and forward + backward, 3 warm-up + 10 timed calls each, median."""
    import statistics
    import torch
    from mi3d import rays as R
    from oracle import ref_import
    ref_import.install()
    cores = min(os.cpu_count() or 1, 32)   # beyond a few dozen threads these small gather / index_add ops slow down
    torch.set_num_threads(cores)
    net = _c1_network(ref_import)
    net.train()
    ro, rd, _ = R.view_rays(64, 64)
    N, steps = ro.shape[1], 64
    bg = torch.ones(N, 3)

    def forward():
        return net.run(ro, rd, num_steps=steps, upsample_steps=0, bg_color=bg, perturb=True, ambient_ratio=1.0,
                       shading="albedo")

    def fwd_only():
        with torch.no_grad():
            forward()

    def fwd_bwd():
        net.zero_grad()
        out = forward()
        ((out["image"] ** 2).mean() + out["loss_orient"]).backward()

    def timed(fn, warm=3, n=10, budget=25.0):
        t_start = time.perf_counter()
        for _ in range(warm):
            fn()
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
            if time.perf_counter() - t_start > budget and len(ts) >= 3:
                break
        return statistics.median(ts), len(ts)
    tf, nf = timed(fwd_only)
    tb, nb = timed(fwd_bwd)
    m = N * steps
    return {"value": 1.0 / tb, "unit": "C1 view-steps/s (64x64 render forward + backward, no diffusion)", "cores": cores,
            "kind": "reference",
            "sample": f"BASELINE config 1 whole: {N} rays x {steps} samples ({m} samples, 7 field evaluations each) through "
                      f"the reference's NeRFRenderer.run + NeRFNetwork (L=4 torch hash grid + 2x32 MLP), fp32; median of "
                      f"{nf} forward / {nb} forward+backward calls after 3 warm-ups",
            "forward_ms": 1e3 * tf, "forward_backward_ms": 1e3 * tb,
            "forward_rays_per_s": N / tf, "forward_samples_per_s": m / tf,
            "forward_backward_rays_per_s": N / tb, "forward_backward_samples_per_s": m / tb}


def cpu_baseline_reference(wl, budget_s=25.0, min_rays=128):
    """Labelled extra: the headline workload's own shape (max_steps uniform samples per ray, 13 field evaluations per
    sample with the smoothness term, L=16 + 3x64) through the reference's NeRFRenderer.run + NeRFNetwork on a torch
    hash grid, forward + backward, on >= 256 rays of the same view, extrapolated to the full view."""
    import torch
    from mi3d import rays as R
    from oracle import ref_import
    ref_import.install()
    # threads actually used: torch's intra-op pool; beyond a few dozen threads these small gather / index_add ops slow
    # down (measured on the 256-core GPU box), so the pool is capped and `cores` reports the cap
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    opt = ref_import.default_opt(cuda_ray=False, lambda_smooth=1.0, max_steps=wl["max_steps"])
    torch.manual_seed(0)
    net = ref_import.reference_network(opt, "oracle")
    net.train()
    ro, rd, _ = R.view_rays(wl["H"], wl["W"])
    N = ro.shape[1]
    steps = wl["max_steps"]

    def render(n):
        idx = torch.randperm(N)[:n]
        net.zero_grad()
        out = net.render(ro[:, idx], rd[:, idx], staged=False, num_steps=steps, upsample_steps=0, perturb=True,
                         bg_color=torch.rand(n, 3), ambient_ratio=1.0, shading="albedo")
        loss = (out["image"] ** 2).mean() + out["loss_orient"] + out["loss_smooth"]
        loss.backward()

    n = min(N, min_rays)
    t0 = time.perf_counter()
    render(n)
    dt = time.perf_counter() - t0
    if dt < 0.25 * budget_s and n < N:   # room for a larger sample: the fixed per-pass cost shrinks relative to it
        n = int(min(N, n * 0.6 * budget_s / max(dt, 1e-3)))
        t0 = time.perf_counter()
        render(n)
        dt = time.perf_counter() - t0
    return {"value": (n / N) / dt, "unit": "view-steps/s (render + backward only, no diffusion)", "cores": cores,
            "kind": "reference",
            "sample": f"{n} of {N} rays x {steps} samples x 13 field evaluations of the same view through the "
                      f"reference's NeRFRenderer.run + NeRFNetwork (torch hash grid), forward + backward, {dt:.1f} s, "
                      f"extrapolated to the full view"}


def cpu_baseline_port(wl, budget_s=20.0):
    """Fallback: the C oracle port (march + field + composite, forward only) on a ray sample."""
    import numpy as np
    from mi3d import rays as R
    from oracle import oracle as O
    cores = os.cpu_count() or 1
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    ro, rd, _ = R.view_rays(wl["H"], wl["W"])
    o, d = ro.view(-1, 3).numpy(), rd.view(-1, 3).numpy()
    N = o.shape[0]
    fp = O.FieldParams(O.GridConfig())
    bits = np.full(128 ** 3 // 8, 255, np.uint8)
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)

    def render(idx):
        nears, fars = O.near_far_from_aabb(o[idx], d[idx], aabb)
        xyzs, dirs, deltas, rays = O.march_rays_train(o[idx], d[idx], 1.0, bits, 1, 128, nears, fars, align=128,
                                                      max_steps=wl["max_steps"])
        sig, col, nrm = O.field_forward(xyzs, dirs, fp)
        O.field_normal(xyzs + np.float32(0.01), fp)
        O.composite_rays_train(sig, col, deltas, rays)
        return xyzs.shape[0]

    rng = np.random.default_rng(0)
    n = 32
    t0 = time.perf_counter()
    render(rng.choice(N, n, replace=False))
    dt = time.perf_counter() - t0
    n = int(min(N, max(32, n * budget_s / max(dt, 1e-3))))
    t0 = time.perf_counter()
    m = render(rng.choice(N, n, replace=False))
    dt = time.perf_counter() - t0
    return {"value": (n / N) / dt, "unit": "render-steps/s (forward only)", "cores": cores, "kind": "port",
            "sample": f"{n} of {N} rays ({m} samples, 13 field evaluations each), C oracle march+field+composite, "
                      f"{dt:.1f} s, extrapolated to the full view"}


def log(msg):
    sys.stderr.write(f"[bench {time.strftime('%H:%M:%S')}] {msg}\n")
    sys.stderr.flush()


def product_c1_gpu(dev):
    """BASELINE config 1 on the PRODUCT: the same 64x64 x 64-sample render through mi3d's NeRFRenderer.run (the
    reference's PyTorch sampler path restated) with the L=4 hash grid and the 8 -> 32 -> 4 MLP on the HIP kernels, fp32,
    forward only and forward + backward; HIP events, 3 warm-up + 10 timed calls, median."""
    import statistics
    import numpy as np
    import torch
    from mi3d import network, rays as R, sds_step
    opt = sds_step.make_opt(cuda_ray=False, lambda_smooth=0.0, fp16=False)
    torch.manual_seed(0)
    net = network.NeRFNetwork(opt, num_layers=2, hidden_dim=32, n_levels=4, per_level_scale=float(128 ** (1 / 3))).to(dev)
    net.train()
    ro, rd, _ = R.view_rays(64, 64, device=dev)
    N, steps = ro.shape[1], 64
    bg = torch.ones(N, 3, device=dev)

    def forward():
        return net.run(ro, rd, num_steps=steps, upsample_steps=0, bg_color=bg, perturb=True, ambient_ratio=1.0,
                       shading="albedo")

    def fwd_only():
        with torch.no_grad():
            forward()

    def fwd_bwd():
        net.zero_grad()
        out = forward()
        ((out["image"] ** 2).mean() + out["loss_orient"]).backward()

    def timed(fn):
        for _ in range(3):
            fn()
        ts = []
        for _ in range(10):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        return statistics.median(ts)
    tf, tb = timed(fwd_only), timed(fwd_bwd)
    return {"forward_ms": tf, "forward_backward_ms": tb, "forward_rays_per_s": N / tf * 1e3,
            "forward_backward_rays_per_s": N / tb * 1e3, "forward_backward_samples_per_s": N * steps / tb * 1e3,
            "what": "the same config-1 render (4096 rays x 64 samples, 7 field evaluations per sample) through "
                    "mi3d.renderer.NeRFRenderer.run on this GPU: L=4 hash grid + 8->32->4 MLP on the HIP kernels, fp32"}


def _cpu_leg(kind, workload, hard_limit_s):
    try:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", kind, "--workload",
                              workload], capture_output=True, text=True, timeout=hard_limit_s)
        for ln in reversed(out.stdout.strip().splitlines()):
            if ln.startswith("{"):
                return json.loads(ln)
        log(f"cpu baseline ({kind}) printed no result: {out.stderr[-300:]}")
    except subprocess.TimeoutExpired:
        log(f"cpu baseline ({kind}) exceeded {hard_limit_s} s")
    except Exception as e:
        log(f"cpu baseline ({kind}) failed: {e!r}")
    return None


CPU_LEG_LIMIT_S = 90   # hard wall-clock limit of the CPU child (both legs: ~25 s + ~20 s of CPU work + one torch import)


def cpu_baseline_start(workload):
    """ONE child process with a hard wall-clock limit (the legs must never hold the bench line hostage), started before
    this process builds its model and collected before the headline's warm-up (see main).  Primary: BASELINE config 1 whole through the reference's own renderer;
    extra: the headline workload's shape on a bounded ray sample, extrapolated."""
    kind = "both" if WORKLOADS[workload].get("mode") is None else "c1"
    try:
        return subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", kind, "--workload", workload],
                                stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True), time.perf_counter()
    except Exception as e:  # noqa: BLE001
        log(f"cpu baseline child did not start: {e!r}")
        return None, time.perf_counter()


def cpu_baseline_collect(started, workload):
    """The child's result; the C oracle port (its own short child) only when the reference legs produced nothing."""
    child, t0 = started
    res = None
    if child is not None:
        try:
            out, err = child.communicate(timeout=max(5.0, CPU_LEG_LIMIT_S - (time.perf_counter() - t0)))
            for ln in reversed(out.strip().splitlines()):
                if ln.startswith("{"):
                    res = json.loads(ln)
                    break
            if res is None:
                log(f"cpu baseline child printed no result: {err[-300:]}")
        except subprocess.TimeoutExpired:
            child.kill()
            child.communicate()
            log(f"cpu baseline child exceeded {CPU_LEG_LIMIT_S} s")
        except Exception as e:  # noqa: BLE001
            log(f"cpu baseline child failed: {e!r}")
    if res is None or res.get("value") is None:
        res = _cpu_leg("port", workload, 60) or {"value": None, "error": "every CPU baseline leg failed"}
    return res


def rank_views(rank, views):
    """Which novel views a rank renders every step, and the seed of its random draws (background colour, march jitter,
    smoothness jitter, diffusion noise): rank r takes views r * views ... r * views + views - 1 of the orbit (phi = 30 +
    45 k degrees, mi3d/rays.py:view_rays), so the ranks of one step never render the same view, and seeds differ per rank
    so their noise is independent (independent novel views shard with no data-path exchange)."""
    return [rank * views + v for v in range(views)], 1234 + rank


class ClockSampler:
    """Samples the GPU's shader clock, power and temperature on a background thread while the timed region runs (VERDICT
    round 4, item 1d: the scatter's box-to-box spread - 32 to 42 ms on one zero census - was attributed to power
    management without a measurement).  sysfs only (a few small file reads per sample, no child process): the card is the
    one whose PCI address torch reports for the device; without such a card nothing is sampled.  Summarised into the
    line's `clocks`: what THIS box granted THIS run."""

    def __init__(self, pci_address=None, period=0.25):
        import glob
        self.samples, self.period, self._stop, self._thread = [], period, False, None
        dev = None
        if pci_address:
            for c in sorted(glob.glob("/sys/class/drm/card*/device")):
                if os.path.basename(os.path.realpath(c)).lower() == pci_address.lower() and \
                        os.path.exists(os.path.join(c, "pp_dpm_sclk")):
                    dev = c
                    break
        self.dev = dev
        self.hwmon = (sorted(glob.glob(os.path.join(dev, "hwmon", "hwmon*"))) or [None])[0] if dev else None
        self.source = f"sysfs {pci_address}" if dev else None

    @staticmethod
    def pci_address_of(torch_device):
        """'dddd:bb:dd.0' of a torch CUDA / HIP device, or None if this torch build does not expose it."""
        try:
            import torch
            pr = torch.cuda.get_device_properties(torch_device)
            return f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        except Exception:  # noqa: BLE001
            return None

    @staticmethod
    def _num(text):
        import re
        m = re.search(r"[0-9.]+", str(text))
        return float(m.group(0)) if m else None

    def _sample(self):
        rec = {"t": time.time()}
        try:
            cur = [ln for ln in open(os.path.join(self.dev, "pp_dpm_sclk")).read().splitlines() if ln.strip().endswith("*")]
            if cur:
                rec["sclk_mhz"] = self._num(cur[0].split(":", 1)[1])
            if self.hwmon:
                for name, key, scale in (("power1_average", "power_w", 1e-6), ("power1_input", "power_w", 1e-6),
                                         ("temp1_input", "temp_c", 1e-3)):
                    f = os.path.join(self.hwmon, name)
                    if key not in rec and os.path.exists(f):
                        rec[key] = float(open(f).read().strip()) * scale
        except Exception:  # noqa: BLE001
            pass
        return rec

    def start(self):
        import threading
        if self.dev is None:
            return self

        def loop():
            while not self._stop:
                self.samples.append(self._sample())
                time.sleep(self.period)
        self._stop = False
        self._thread = threading.Thread(target=loop, daemon=True)
        self._thread.start()
        return self

    def stop(self):
        self._stop = True
        if self._thread is not None:
            self._thread.join(timeout=5)

    def summary(self, t0=None, t1=None):
        import statistics
        if self.dev is None:
            return None
        s = [r for r in self.samples if (t0 is None or r["t"] >= t0) and (t1 is None or r["t"] <= t1)]
        out = {"source": self.source, "samples": len(s), "period_s": self.period}
        for key in ("sclk_mhz", "power_w", "temp_c"):
            v = [r[key] for r in s if r.get(key) is not None]
            if v:
                out[key] = {"min": min(v), "median": statistics.median(v), "max": max(v)}
        return out


def gather_rank_records(rec):
    """Every rank's record on every rank (torch.distributed.all_gather_object: works on gloo and on NCCL = RCCL); a list of
    one without a process group."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return [rec]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, rec)
    return out


def summarise_ranks(records, expected_updates, backend, group_world_size, launcher_world_size):
    """What a line measured at N > 1 must say about ITSELF, since nobody else was there: the
    world size the process group really had next to the launcher's, the backend, every rank's own step time (its clock
    from the first timed step to its own device synchronisation, BEFORE the closing barrier), the time its stream spent
    inside the gradient all-reduce (HIP events around the collective: a fast rank waits there for the slowest one), loss
    scale, applied optimizer updates and peak memory per rank - and `valid`: false with a reason when the ranks disagree
    on what they did (an update applied on one rank and skipped on another cannot happen with a summed gradient bucket;
    a different scale or step count means the ranks did not run the same job)."""
    import statistics
    recs = sorted(records, key=lambda r: r["rank"])
    ms = [r["ms_per_step_local"] for r in recs]
    ar = [r["all_reduce_ms_per_step"] for r in recs]
    problems = []
    if group_world_size != launcher_world_size:
        problems.append(f"process group has {group_world_size} ranks, the launcher announced {launcher_world_size}")
    if [r["rank"] for r in recs] != list(range(group_world_size)):
        problems.append(f"records from ranks {[r['rank'] for r in recs]}, expected 0..{group_world_size - 1}")
    for key, what in (("optimizer_steps_applied", "applied optimizer updates"), ("grad_scaler_scale", "loss scale before"),
                      ("grad_scaler_scale_after", "loss scale after"), ("steps", "timed steps")):
        vals = [r[key] for r in recs]
        if len(set(vals)) > 1:
            problems.append(f"ranks disagree on {what}: {vals}")
    if recs and recs[0]["optimizer_steps_applied"] != expected_updates:
        problems.append(f"{expected_updates - recs[0]['optimizer_steps_applied']} timed step(s) skipped the optimizer update")
    return {"backend": backend, "rccl_world_size": group_world_size, "launcher_world_size": launcher_world_size,
            "per_rank": recs,
            "ms_per_step_local": {"min": min(ms), "median": statistics.median(ms), "max": max(ms)},
            "all_reduce_ms_per_step": {"min": min(ar), "median": statistics.median(ar), "max": max(ar)},
            "compute_ms_per_step": {"min": min(m - a for m, a in zip(ms, ar)), "max": max(m - a for m, a in zip(ms, ar))},
            "peak_mem_GiB": [r["peak_mem_GiB"] for r in recs],
            "valid": not problems, "problems": problems}


def spawn_ranks(n):
    """`python bench.py --gpus N` outside a launcher: become the launcher (one rank per GPU on this node)."""
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c2_dense", choices=list(WORKLOADS))
    ap.add_argument("--variant-steps", type=int, default=4, help="timed steps of each non-headline variant (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-reference-shaped", action="store_true")
    ap.add_argument("--init-scale", type=float, default=65536.0,
                    help="GradScaler initial loss scale (the reference's: torch's default, nerf/utils.py:309)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the RCCL process group even at world size 1 (under torch.distributed.run): the "
                         "gradient all-reduce, the state broadcast and the occupancy broadcast then really execute")
    ap.add_argument("--cpu-baseline-only", default=None, choices=["c1", "reference", "both", "port"], help=argparse.SUPPRESS)
    ap.add_argument("--reference-shaped-only", action="store_true",
                    help="time ONLY the reference-shaped leg (the reference's NeRFNetwork / run_cuda Python on the drop-in "
                         "packages) - the command a rocprofv3 --kernel-trace of that route wraps")
    ap.add_argument("--all-variants", action="store_true",
                    help="also time the t=300 branch with the CLIP weights left trainable (the reference's wasted backward "
                         "into the towers) at the full --variant-steps; the default run gives it 2 steps")
    ap.add_argument("--profile-run", action="store_true",
                    help="the run rocprofv3 wraps (profiles/README.md): the headline's settle + warm-up + timed steps and "
                         "NOTHING after them - no variants, no census / dense-gradient extras, no baselines - so the last "
                         "steps/steps_run_total of every kernel's dispatches in the trace ARE the timed steps "
                         "(tools/trace_sum.py --tail)")
    ap.add_argument("--no-clock-log", action="store_true",
                    help="do not sample the GPU's clocks / power on a background thread during the timed region")
    ap.add_argument("--refresh-every", type=int, default=16,
                    help="update_extra_state interval inside the timed loop (nerf/utils.py:970-972; 0 = never)")
    args = ap.parse_args()
    if args.profile_run:
        args.variant_steps, args.no_cpu_baseline, args.no_reference_shaped = 0, True, True

    if args.cpu_baseline_only:   # child process of cpu_baseline(): host cores only, no GPU
        wl = WORKLOADS[args.workload]
        if args.cpu_baseline_only in ("c1", "reference", "both"):
            from oracle import ref_import
            if not ref_import.available():
                raise SystemExit("reference sources not staged")
            if args.cpu_baseline_only == "reference":
                res = cpu_baseline_reference(wl)
            else:
                res = cpu_baseline_c1()
                print(json.dumps(res), flush=True)   # (the primary leg is on record even if the extra is cut off)
                if args.cpu_baseline_only == "both":
                    try:
                        res["headline_shape_extrapolated"] = cpu_baseline_reference(wl)
                    except Exception as e:  # noqa: BLE001
                        res["headline_shape_extrapolated"] = {"value": None, "error": repr(e)}
            print(json.dumps(res))
        else:
            print(json.dumps(cpu_baseline_port(wl)))
        return

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args.gpus)
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        raise SystemExit(f"bench invalid: --gpus {args.gpus} but WORLD_SIZE is {world}")
    ensure_built()
    # The CPU baseline works on host cores in ONE child process while this process builds the model, lets MIOpen pick its
    # kernels and settles the loss scale (all untimed); the headline's warm-up does not start before the child is done, so
    # no timed figure - headline, variants, reference-shaped leg - shares the host with it.  (Round 5 ran two children at
    # the END of the run, one after the other: 198 s of a 454 s driver run.  Round 6's first try ran the child beside
    # the variants: a 32-thread host job next to the eager VAE's ~900 launches per step inflated them by 10-40 %.)
    cpu_child = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.reference_shaped_only:
        cpu_child = cpu_baseline_start(args.workload)

    import torch
    import torch.distributed as dist
    if world > 1 or args.force_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    wl = WORKLOADS[args.workload]
    if wl.get("mode") == "refine":
        return bench_refine(args, wl, dev, rank, world)
    views = wl["views"]

    from mi3d import dp, field_ops as field_ops_mod, grid_ops, rays as R, sd_standin, sds_step
    opt = sds_step.make_opt(max_steps=wl["max_steps"])
    # GradScaler: the reference constructs it at 65536 (nerf/utils.py:309).  On this workload the normal-smoothness
    # regulariser back-propagates through safe_normalize of finite differences that fp16 rounds to exactly zero
    # (gradient x 1e10), so every step overflows binary16 until the scaler has halved itself down to ~1; a skipped step
    # does NO optimizer work, and the loss scale decides how many binary16 gradients underflow to exact zeros (which
    # the scatter skips - as it always did).  So the bench lets the scaler SETTLE first, exactly as a training run
    # would: untimed steps from the reference's initial scale until several consecutive steps apply their update, and
    # it asserts below that every timed step really applied its Adan update at that scale.
    model, optimizer, scaler = sds_step.build_training_state(opt, dev, seed=0, bitfield=wl["bitfield"],
                                                             init_scale=args.init_scale)
    dp.broadcast_module_state(model)
    bucket = dp.FlatGradBucket(model.parameters())
    guidance = sd_standin.StableDiffusionStandIn(dev)
    text_z = guidance.get_text_embeds()
    view_ids, seed = rank_views(rank, views)
    view_rays = [R.view_rays(wl["H"], wl["W"], view=v, device=dev) for v in view_ids]
    t_fixed = T_FIXED  # an int: the guidance decides its branch on the host without a device sync
    torch.manual_seed(seed)

    render_only = wl.get("mode") == "render"
    if render_only:
        opt.lambda_smooth = 0.0

    if args.reference_shaped_only:   # the command `rocprofv3 --kernel-trace` wraps for that route (profiles/README.md)
        # (at loss scale 2.0: where the headline's scaler settles on this workload - the reference's 65536 would spend the
        # leg on overflowing steps)
        res = reference_shaped(model, guidance, text_z, opt, view_rays[0], wl, t_fixed, dev, 2.0, steps=max(1, args.steps),
                               mark=True)
        if rank == 0:
            print(json.dumps(res))
        return

    def make_step(the_model, the_optimizer, the_scaler, schedule, sync):
        def render_step():
            for ro, rd, ds in view_rays:
                with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16, enabled=opt.fp16):
                    the_model.render(ro, rd, depth_scale=ds, bg_color=torch.rand(3, device=dev), staged=False,
                                     perturb=True, ambient_ratio=1.0, shading="albedo", force_all_rays=True,
                                     **sds_step.render_kwargs(opt))
        if render_only:
            return render_step

        clock = {"i": 0, "refreshes": 0}

        def refresh():
            # nerf/utils.py:970-972: every update_extra_interval (16) steps, before the step, under autocast.  Rank 0
            # evaluates the 128^3 jittered cell centres (2.1 M field evaluations per cascade), EMA-maxes the grid,
            # thresholds and repacks the bitfield; the other ranks receive grid + bitfield (mi3d/dp.py:sync_occupancy,
            # SURVEY 8(e)).  A random-weight field has near-constant density, so what the refresh computes is all-on or
            # blob-only depending on the seed: the workload's ANALYTIC occupancy is put back afterwards
            # (a 262 144-byte fill - not something the reference does, and the only part of this that is not).
            with grid_ops.phase("grid_refresh"):
                if rank == 0:
                    with torch.autocast("cuda", dtype=torch.float16, enabled=opt.fp16):
                        the_model.update_extra_state()
                dp.sync_occupancy(the_model)
                sds_step.set_bitfield(the_model, wl["bitfield"])
            clock["refreshes"] += 1

        def step():
            # a batch of `views` views is `views` consecutive single-view passes (the renderer "assumes B == 1",
            # renderer.py:482), each a full training step as the reference's loader (batch_size 1) would issue them
            for ro, rd, ds in view_rays:
                if args.refresh_every > 0 and clock["i"] % args.refresh_every == 0:
                    refresh()
                clock["i"] += 1
                # (the step's own optimizer.zero_grad clears the gradients, which ARE the flat bucket's views)
                sds_step.sds_train_step(the_model, guidance, text_z, the_optimizer, the_scaler, ro, rd, ds, wl["H"],
                                        wl["W"], opt, sds_backward=schedule, t=step_extra.get("t", t_fixed), grad_sync=sync,
                                        **{k: v for k, v in step_extra.items() if k != "t"})
        step.clock = clock
        return step

    # phase timers (HIP events on the launch stream) around the two PyTorch-side phases
    _sds, _opt_step = guidance._predict_noise, optimizer.step  # VAE encode + U-Net: shared by both backward schedules

    def timed(kind, fn):
        def wrapper(*a, **k):
            box = []
            grid_ops._timed(kind, lambda: box.append(fn(*a, **k)), 1)
            return box[0]
        return wrapper
    guidance._predict_noise = timed("sd_guidance", _sds)
    optimizer.step = timed("optimizer", _opt_step)
    # the gradient all-reduce between HIP events on the launch stream (a rank that finished its backward early WAITS in this
    # collective for the slowest rank: its per-rank spread is the skew the step time hides)
    grad_sync = timed("all_reduce", bucket.all_reduce_mean) if dist.is_initialized() else bucket.all_reduce_mean
    step_extra = {}     # extra arguments of sds_train_step (the denoise + CLIP variant sets them)

    def run(records, schedule, steps, warmup):
        step = make_step(model, optimizer, scaler, schedule, grad_sync)
        if not render_only:
            step.clock["i"] = 1   # warm-up steps do not refresh the grid ...
        for _ in range(warmup):
            step()
        if not render_only:
            step.clock["i"] = 0   # ... the first timed step does (global_step 0 of nerf/utils.py:970), then every 16th
            step.clock["refreshes"] = 0
        grid_ops.PROFILE = {"scatter": [], "encode": []}
        torch.cuda.reset_peak_memory_stats(dev)
        scale0 = scaler.get_scale() if not render_only else None
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()
        if args.profile_run:   # a marker dispatch (`spin_kernel`) either side of the timed region: tools/trace_sum.py --window
            torch.cuda._sleep(1000)
            torch.cuda.synchronize()
        sampler = None if args.no_clock_log else ClockSampler(ClockSampler.pci_address_of(dev)).start()
        w0 = time.time()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        local = time.perf_counter() - t0      # this rank's own clock, before it waits for the others
        w1 = time.time()
        if args.profile_run:
            torch.cuda._sleep(1000)
            torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
        elapsed = time.perf_counter() - t0
        clocks = None
        if sampler is not None:   # (after the clock has stopped: joining the sampler's thread may take a period)
            sampler.stop()
            clocks = sampler.summary(w0, w1)
        prof, grid_ops.PROFILE = grid_ops.PROFILE, None
        if dist.is_initialized():
            tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        info = {"applied": len(prof.get("optimizer", [])), "scale_before": scale0,
                "scale_after": scaler.get_scale() if not render_only else None,
                "refreshes": 0 if render_only else step.clock["refreshes"],
                "peak_mem_GiB": torch.cuda.max_memory_allocated(dev) / 2 ** 30,
                "local_ms_per_step": 1e3 * local / steps, "clocks": clocks,
                "all_reduce_ms_per_step": sum(a.elapsed_time(b) for a, b in prof.get("all_reduce", [])) / steps}
        return elapsed, prof, info

    log(f"{args.workload}: timing {args.steps} steps of the headline variant {HEADLINE} on {world} GPU(s)")
    # settle the loss scale (untimed; before the W warm-up steps of the contract)
    settle = make_step(model, optimizer, scaler, HEADLINE[1], grad_sync)
    good, tries = 0, 0
    settle_s = []
    while not render_only and opt.fp16 and good < 4 and tries < 60:
        before = scaler.get_scale()
        t_s = time.perf_counter()
        settle()
        good = good + 1 if scaler.get_scale() >= before else 0   # (get_scale() synchronises: the step has finished)
        settle_s.append(round(time.perf_counter() - t_s, 2))
        tries += 1
    if settle_s:
        log(f"settle steps took {settle_s} s (the first carries MIOpen's kernel selection and the U-Net graph capture; an "
            f"overflowing step hands its inf / NaN gradients to float atomics, as the reference does)")
    if not render_only and opt.fp16:
        # timed AT the scale the scaler settled at (round 2 timed two halvings below it: fewer binary16 gradients survive
        # there, so the scatter did less work than a training run's).  A timed step may still overflow - GradScaler
        # then skips that optimizer update and halves, exactly as in training; the line reports how many updates were
        # applied and the scale before / after instead of aborting.
        log(f"loss scale: 4 consecutive steps applied at {scaler.get_scale()} after {tries} untimed steps; timing there")
    # ---- untimed, BEFORE the timed region (so that a --profile-run trace ends with the timed steps): which gradient pairs
    # one step's scatter actually carries
    census = None
    if not render_only:
        grid_ops.CENSUS = []
        make_step(model, optimizer, scaler, HEADLINE[1], grad_sync)()
        torch.cuda.synchronize()
        raw, grid_ops.CENSUS = grid_ops.CENSUS, None
        census = {}
        for c in raw:
            nz = [int(v) for v in c["nonzero_pairs_per_level"].tolist()]
            census[c["P_active"]] = {
                "rows": c["rows"], "with_deferred_point0": bool(c.get("with_deferred_point0", False)),
                "nonzero_pair_fraction_per_level": [v / c["rows"] for v in nz],
                "nonzero_pair_fraction": sum(nz) / (len(nz) * c["rows"]),
                # how the zeros cluster (what a lane / tile compaction in the emit could drop): per level, the share of
                # SAMPLES with a non-zero pair on any of their stencil points, the share of 64-sample tiles with any,
                # the share of samples that are non-zero on any level at all, and the non-zero share per stencil point
                "samples_with_any_nonzero_point_per_level": [float(v) for v in c["samples_with_any_nonzero_point_per_level"].tolist()],
                "tiles64_with_any_nonzero_per_level": [float(v) for v in c["tiles64_with_any_nonzero_per_level"].tolist()],
                "samples_nonzero_on_any_level": float(c["samples_nonzero_on_any_level"]),
                "nonzero_fraction_per_point": [float(v) for v in c["nonzero_fraction_per_point"].tolist()]}
        del raw
        torch.cuda.empty_cache()
    cpu_result = None
    if cpu_child is not None:
        t_w = time.perf_counter()
        cpu_result = cpu_baseline_collect(cpu_child, args.workload)
        log(f"cpu baseline: child finished (waited {time.perf_counter() - t_w:.1f} s for it before the timed region)")
    elapsed, prof, info = run(*HEADLINE, args.steps, args.warmup)
    log(f"headline: {1e3 * elapsed / args.steps:.1f} ms/step, {info['applied']} of {args.steps * views} updates applied, "
        f"loss scale {info['scale_before']} -> {info['scale_after']}, peak memory {info['peak_mem_GiB']:.1f} GiB")
    # A timed step that overflowed did NO optimizer work and halved the scale: the region is timed again (at most twice)
    # at the scale the scaler has moved to; a line whose timed steps still did not all apply their update says so
    # (`valid`: false) instead of passing a figure that includes skipped updates.
    retimed = 0
    while (not render_only and opt.fp16 and info["applied"] != args.steps * views and retimed < 2
           and not args.profile_run):
        retimed += 1
        log(f"a timed step skipped its update (GradScaler overflow): timing the region again at {scaler.get_scale()}")
        elapsed, prof, info = run(*HEADLINE, args.steps, args.warmup)
        log(f"headline (re-timed): {1e3 * elapsed / args.steps:.1f} ms/step, {info['applied']} of "
            f"{args.steps * views} updates applied, loss scale {info['scale_before']} -> {info['scale_after']}")
    ranks = None
    if dist.is_initialized() and not render_only:
        ranks = summarise_ranks(
            gather_rank_records({"rank": rank, "steps": args.steps, "ms_per_step_local": info["local_ms_per_step"],
                                 "all_reduce_ms_per_step": info["all_reduce_ms_per_step"],
                                 "optimizer_steps_applied": info["applied"], "grad_scaler_scale": info["scale_before"],
                                 "grad_scaler_scale_after": info["scale_after"], "peak_mem_GiB": info["peak_mem_GiB"],
                                 "views": view_ids, "device": torch.cuda.get_device_name(dev)}),
            args.steps * views, dist.get_backend(), dist.get_world_size(), world)
    steps_run_total = tries + 1 + (1 + retimed) * (args.warmup + args.steps)   # settle + census + warm-up + timed
    variants = {f"records={HEADLINE[0]},sds_backward={HEADLINE[1]}": 1e3 * elapsed / args.steps}
    dense_step, clip_step = None, None
    if args.variant_steps > 0 and not render_only:
        e, _, _ = run("fp32", "single", args.variant_steps, 1)
        variants["records=fp32,sds_backward=single"] = 1e3 * e / args.variant_steps
        e, _, _ = run("fp32", "overlapped", args.variant_steps, 1)
        variants["records=fp32,sds_backward=overlapped"] = 1e3 * e / args.variant_steps
        # ---- the guidance's OTHER branch (nerf/sd.py:153-159): for t/1000 <= 0.4 - t ~ U{200..600}: 201 of 401 draws - a
        # novel-view step injects NO SDS gradient; it takes one DDIM step, decodes the latents (VAE decoder, no_grad), runs
        # two CLIP image towers + one text tower on the denoised image and adds 10 x those similarities - constants w.r.t.
        # the NeRF - to the regularisers.  Timed here at t = 300 with full-size stand-ins (VAE decoder ~49 M
        # parameters, ViT-B/16 + text tower ~150 M); the NeRF side is then ONE backward pass (the 13-point regulariser
        # pass, nothing parked), the VAE encoder still runs forward (its latents are noised and denoised) but never backward.
        if True:
            try:
                ref_rgb = torch.rand(1, 3, 512, 512, device=dev)
                if guidance.vae_decoder is None:
                    with torch.random.fork_rng(devices=[dev]):
                        torch.manual_seed(7)
                        guidance.vae_decoder = sd_standin.VAEDecoderSD().to(dev)
                    for prm in guidance.vae_decoder.parameters():
                        prm.requires_grad_(False)
                with torch.random.fork_rng(devices=[dev]):
                    torch.manual_seed(8)
                    clip_model = sd_standin.CLIPStandIn().to(dev).half()
                for frozen, label in ((True, "t300_clip_branch"), (False, "t300_clip_branch_clip_weights_unfrozen")):
                    # (the reference freezes the guidance's parameters only, nerf/utils.py:280-281: its CLIP model keeps
                    # requires_grad and scaler.scale(loss).backward() walks back into the CLIP towers for nothing - the
                    # second figure; the first is the same step with CLIP frozen)
                    for prm in clip_model.parameters():
                        prm.requires_grad_(not frozen)
                    step_extra.update(t=300, clip_model=clip_model, ref_rgb=ref_rgb, ref_text="a toy")
                    # (the unfrozen figure is context - what the reference wastes - and gets 2 steps unless --all-variants)
                    vs = args.variant_steps if (frozen or args.all_variants) else min(2, args.variant_steps)
                    try:   # (two warm-up steps: the CLIP towers' weight-gradient kernels are first used here)
                        e, cprof, cinfo = run("fp32", "reference", vs, 2)
                    finally:
                        step_extra.clear()
                        clip_model.zero_grad(set_to_none=True)
                    variants[label] = 1e3 * e / vs
                    if frozen:
                        clip_step = {"ms_per_step": variants[label], "t": 300,
                                     "optimizer_steps_applied": cinfo["applied"],
                                     "optimizer_steps_attempted": args.variant_steps * views,
                                     "scatter_ms_per_step": sum(a.elapsed_time(b) for a, b in cprof.get("scatter", [])) / args.variant_steps,
                                     "phases_ms_per_step": {k[6:]: sum(a.elapsed_time(b) for a, b in v) / args.variant_steps
                                                            for k, v in cprof.items() if k.startswith("phase:")}}
                del clip_model
                guidance.vae_decoder = None
                torch.cuda.empty_cache()
            except Exception as ex:  # noqa: BLE001 - an extra: it must not take the line down
                log(f"denoise + CLIP branch variant failed: {ex!r}")
                step_extra.clear()
        # the headline schedule with the scatter's zero-skip DEFEATED: every gradient pair that binary16 underflowed to
        # an exact zero is replaced by +-2^-24 (the smallest binary16 subnormal) before the scatter, so the emit and the
        # reduce process all 141 M x 16 pairs.  The replacement is one extra pass over the gradient planes, timed on its
        # own (`densify`) and subtracted: what is reported is the step a dense-gradient workload would cost.
        grid_ops.DENSIFY = True
        try:
            e, dprof, dinfo = run("fp32", "reference", args.variant_steps, 1)
        finally:
            grid_ops.DENSIFY = False
        dens_ms = sum(a.elapsed_time(b) for a, b in dprof.get("densify", [])) / args.variant_steps
        variants["dense_gradients"] = 1e3 * e / args.variant_steps - dens_ms
        dense_step = {"ms_per_step": variants["dense_gradients"], "densify_pass_ms_subtracted": dens_ms,
                      "scatter_ms_per_step": sum(a.elapsed_time(b) for a, b in dprof.get("scatter", [])) / args.variant_steps,
                      "optimizer_steps_applied": dinfo["applied"], "optimizer_steps_attempted": args.variant_steps * views}

    # ---- untimed: the 13-point scatter on dense random gradients
    dense = None
    if not render_only and not args.profile_run and rank == 0 and world == 1:
        dense = scatter_on_dense_gradients(model, view_rays[0], opt, dev)
    m = int(model.step_counter[(model.local_step - 1) % 16, 0].item())
    line = None
    if rank == 0:
        ms = {k: [a.elapsed_time(b) for a, b in v] for k, v in prof.items() if not k.endswith("_evals")}
        evals = {k[:-6]: v for k, v in prof.items() if k.endswith("_evals")}

        def roof(kernel, key, per_eval, bound, peak, unit, note, pmc_key=None, work_of=None):
            """achieved = algorithmic work of the timed launches / their summed HIP-event duration.  work_of(evals of
            one launch) overrides evals x per_eval (the scatter counts only the pairs it does not skip)."""
            t = sum(ms.get(key, [])) * 1e-3
            ev = evals.get(key, [])
            work = sum(work_of(e) for e in ev) if work_of else sum(ev) * per_eval
            a = work / t / (1e9 if bound == "hbm" else 1e12) if t > 0 else 0.0
            n_launch = max(1, len(ms.get(key, [])))
            traffic, source = None, None
            if pmc_key is not None:
                traffic, source = pmc_traffic(pmc_key, args.workload, sum(ev) / n_launch)
            return {"kernel": kernel, "bound": bound, "achieved": a, "peak": peak, "unit": unit, "frac": a / peak,
                    "traffic": traffic,
                    "traffic_source": (source + ": rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes around `bench.py "
                                       "--profile-run` (separate runs, averaged over the timed steps' launches, FETCH_SIZE "
                                       "x 2, the emit's WRITE_SIZE divided by its sector calibration), scaled by this run's "
                                       "evaluations per launch - not collected in this process") if source else None,
                    "launches": len(ms.get(key, [])), "avg_launch_ms": 1e3 * t / n_launch,
                    "ms_per_step": 1e3 * t / args.steps,
                    "algorithmic_work_per_launch": work / n_launch, "note": note}

        n_last = m + (128 - m % 128)   # rows of the last view (march pads past the next multiple of 128)

        def scatter_work(ev):
            """2048 B per evaluation = 128 B per (evaluation, level) gradient pair - counted only for the pairs that are
            not exactly zero (the census of one untimed step at the same state; the emit skips the others)."""
            P_act = 1 if ev < 2 * n_last else 13
            c = (census or {}).get(P_act)
            return ev * SCATTER_BYTES_PER_EVAL * (c["nonzero_pair_fraction"] if c else 1.0)

        roofs = [
            roof("k_grid_encode_planes (13-point hash-grid gather, levels tied to XCDs, 16-byte pair loads, non-temporal "
                 "plane stores) + k_grid_encode_planes_lds (levels 0-1 from LDS, ~0.8 ms of it); csrc/hashgrid.hip", "encode", ENCODE_BYTES_PER_EVAL, "hbm", HBM_PEAK_GBPS, "GB/s",
                 "algorithmic bytes = 1024 B per field evaluation; the tables are L2-resident per XCD, so the binding "
                 "limit is the L1 line-lookup rate for divergent gathers, not HBM - traffic shows how few bytes reach it",
                 "k_grid_encode_planes"),
            roof("grid gradient scatter: one mi3d_grid_scatter_binned_plus call = (k_bin_emit + k_bin_reduce) x slices "
                 "(records through HBM, no global atomics)", "scatter", SCATTER_BYTES_PER_EVAL, "hbm", HBM_PEAK_GBPS,
                 "GB/s", "a 'launch' here is one scatter call (kernel-trace: sum the k_bin_emit and k_bin_reduce rows of a "
                 "step: profiles/kernel_stats_r06_bench_steps.csv, collected on THIS command's own timed steps with "
                 "--profile-run); algorithmic bytes = 2048 B per evaluation read-modify-write of the table, counted ONLY for "
                 "the (evaluation, level) gradient pairs that are not exactly zero (grad_nonzero_pair_fraction: binary16 "
                 "gradients underflow, and adding a zero is what the reference's atomics would do); dense-gradient figures: "
                 "scatter_dense_gradients (the call alone) and dense_gradient_step (a whole step).  ONE call per step: the "
                 "SDS pass (nerf/sd.py:171) reaches stencil point 0 only, its gradient planes are parked and ride along in "
                 "the regulariser pass's 13-point scatter (mi3d.grid_ops.DEFER_POINT0)",
                 "scatter_binned", work_of=scatter_work),
            roof("k_mlp_fwd_g<F16, 2, 3, binary16 planes> (csrc/field.hip)", "mlp_fwd", 12800.0, "mfma", MFMA_F16_PEAK_TFLOPS, "TFLOP/s",
                 "streams 64 B of binary16 planes in + 16 B out per evaluation (fp32 planes without autocast: 128 B)", "k_mlp_forward"),
            roof("k_mlp_bwd_g<F16, 2, 3, binary16 planes, full width> (recompute + dgrad + wgrad in one kernel; the tiles of the weight "
                 "gradients turned round through LDS, ds_read_b64_tr_b16; csrc/field.hip)", "mlp_bwd",
                 25600.0, "mfma", MFMA_F16_PEAK_TFLOPS, "TFLOP/s",
                 "80 B read + 64 B written per evaluation with binary16 planes", "k_mlp_backward"),
        ]
        roofs.sort(key=lambda r: -r["ms_per_step"])
        line = {
            "metric": ("forward view-renders/sec (march + 7-point field + composite), BASELINE config 4" if render_only
                       else "SDS train-steps/sec (NeRF render+SD U-Net fwd+bwd) @128x128"),
            "value": world * views * args.steps / elapsed, "unit": "view-renders/s" if render_only else "view-steps/s",
            "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("f32 hash grid + f16 MFMA MLP (torch.autocast semantics)" if render_only else
                      "f32 hash grid (fp32 gather, fp32 gradient contributions accumulated in 64-bit fixed point) + f16 "
                      "MFMA MLP (torch.autocast semantics; the planes between gather, MLP and scatter hold the binary16 "
                      "values autocast rounds them to), f16 U-Net"),
            "data": "synthetic (reference orbit rays, random-init weights, analytic occupancy)",
            "config": ({"workload": f"{args.workload}: {wl['H']}x{wl['W']} rays, max_steps {wl['max_steps']}, L=16 hash "
                                    f"grid + 3x64 MLP, forward render only (no diffusion, no backward), occupancy "
                                    f"{wl['bitfield']}, {m} samples in the last view x 7 field evaluations, {views} "
                                    f"view(s) per step",
                        "views_per_step": world * views,
                        "rays_per_s": world * views * wl["H"] * wl["W"] * args.steps / elapsed,
                        "samples_per_s": world * sum(evals.get("encode", [])) / 7.0 / elapsed,
                        "parallelism": f"dp{world} (independent views)"} if render_only else
                       {"workload": f"{args.workload}: {wl['H']}x{wl['W']} rays, max_steps {wl['max_steps']}, "
                                    f"L=16 hash grid + 3x64 MLP, SD2-base-shaped U-Net SDS step (t={T_FIXED}: the SDS "
                                    f"branch of nerf/sd.py:153), occupancy {wl['bitfield']}, {m} samples/view x 13 "
                                    f"field evaluations, {views} view(s) per step",
                        "views_per_step": world * views, "sds_backward": HEADLINE[1], "gradient_records": HEADLINE[0],
                        "optimizer_steps_applied": info["applied"], "optimizer_steps_attempted": args.steps * views,
                        "grad_scaler_scale": info["scale_before"], "grad_scaler_scale_after": info["scale_after"],
                        "grid_refreshes_in_timed_region": info["refreshes"],
                        "steps_run_total": steps_run_total, "timed_region_repeats": retimed,
                        "parallelism": f"dp{world} (one view per GPU, flat {bucket.nbytes / 1e6:.1f} MB grad "
                                       f"all-reduce)"}),
            "variants_ms_per_step": variants,
            "roofline": roofs[0],
            "rooflines_other": roofs[1:],
            "kernels_ms_per_step": {k: sum(v) / args.steps for k, v in ms.items() if not k.startswith("phase:")},
            # exclusive phases of the step (HIP events on the launch stream): their sum is the step
            "phases_ms_per_step": {k[6:]: sum(v) / args.steps for k, v in ms.items() if k.startswith("phase:")},
            "peak_mem_GiB": info["peak_mem_GiB"],
            # which of the candidate blocks the persistent record arena was placed on, and what the scatter cost on each
            # (mi3d/field_ops.py: the emit's time depends on the arena's physical placement, 44.5-55.3 ms for one call)
            "scatter_arena_placement": list(field_ops_mod.PLACEMENT_LOG),
            # shader clock / board power / temperature sampled on a background thread during the timed region (ClockSampler)
            "clocks": info.get("clocks"),
        }
        line["phases_ms_per_step"]["unattributed"] = line["ms_per_step"] - sum(line["phases_ms_per_step"].values())
        if census is not None:
            line["grad_nonzero_pair_fraction"] = {
                ("sds_pass_point0" if k == 1 else
                 (f"both_passes_in_one_scatter_{k}_points" if v.get("with_deferred_point0") else
                  f"regulariser_pass_{k}_points")): v for k, v in sorted(census.items())}
        if dense is not None:
            line["scatter_dense_gradients"] = dense
        if dense_step is not None:
            line["dense_gradient_step"] = dense_step
        if clip_step is not None:
            # on novel views past diff_iters the guidance draws t ~ U{200..600}; t <= 400 (201 of 401 draws)
            # takes the denoise + CLIP branch.  The blend is what an average such step costs; `value` stays the SDS branch
            # (t = 500), which is the configuration BASELINE.json's metric names.
            p_clip = 201.0 / 401.0
            blend = p_clip * clip_step["ms_per_step"] + (1 - p_clip) * line["ms_per_step"]
            line["guidance_branches"] = {
                "sds_branch_t500_ms_per_step": line["ms_per_step"], "denoise_clip_branch_t300": clip_step,
                "share_of_novel_view_steps_on_the_clip_branch": p_clip,
                "blended_ms_per_step": blend, "blended_view_steps_per_s": world * views * 1e3 / blend,
                "note": "nerf/sd.py:153-159: t/1000 <= 0.4 takes one DDIM step + VAE decode + 2 CLIP image + 1 text "
                        "forward on the denoised image and injects no SDS gradient (the NeRF sees the regulariser pass "
                        "only); full-size random-weight stand-ins (mi3d/sd_standin.py)"}
        if ranks is not None:
            line["ranks"] = ranks
            if not ranks["valid"]:
                line["valid"] = False
                line["invalid_reason"] = "; ".join(ranks["problems"])
        if not render_only and opt.fp16 and line.get("valid", True):
            line["valid"] = info["applied"] == args.steps * views
            if not line["valid"]:
                line["invalid_reason"] = (f"{args.steps * views - info['applied']} timed step(s) skipped the optimizer "
                                          f"update (GradScaler overflow) even after {retimed} re-timing(s)")

    # ---- baselines (rank 0 of a single-GPU run only; never part of the timed region above)
    if rank == 0 and world == 1:
        if not args.no_reference_shaped and not render_only:
            log("reference-shaped baseline (reference Python on the drop-in packages)")
            try:
                line["reference_shaped_baseline"] = reference_shaped(model, guidance, text_z, opt, view_rays[0], wl,
                                                                     t_fixed, dev, scaler.get_scale())
            except Exception as e:
                line["reference_shaped_baseline"] = {"value": None, "error": repr(e)}
            if line["reference_shaped_baseline"].get("value"):
                line["speedup_vs_reference_shaped"] = line["value"] / line["reference_shaped_baseline"]["value"]
        if cpu_result is not None:
            line["cpu_baseline"] = cpu_result
            try:
                line["cpu_baseline"]["product_c1_gpu"] = product_c1_gpu(dev)
                fb = line["cpu_baseline"].get("forward_backward_ms")
                if fb:
                    line["cpu_baseline"]["product_c1_gpu"]["speedup_vs_cpu_forward_backward"] = \
                        fb / line["cpu_baseline"]["product_c1_gpu"]["forward_backward_ms"]
            except Exception as e:  # noqa: BLE001
                line["cpu_baseline"]["product_c1_gpu"] = {"error": repr(e)}
        log("done")
    if rank == 0:
        if dist.is_initialized():
            line["collectives"] = dict(dp.STATS, backend=dist.get_backend(), world_size=world)
        print(json.dumps(line))
    if dist.is_initialized():
        dist.destroy_process_group()


def scatter_on_dense_gradients(model, rays, opt, dev, reps=2):
    """The 13-point scatter of this view on DENSE random binary16 gradient planes (no pair is zero): the figure to hold
    against the real step's, whose work depends on how many binary16 gradients underflowed."""
    import math
    import raymarching
    import torch
    from mi3d import field_ops, grid_ops
    ro, rd, _ = rays
    ro, rd = ro.view(-1, 3).contiguous(), rd.view(-1, 3).contiguous()
    nears, fars = raymarching.near_far_from_aabb(ro, rd, model.aabb_train)
    cnt = torch.zeros(2, dtype=torch.int32, device=dev)
    xs, _, _, _ = raymarching.march_rays_train(ro, rd, model.bound, model.density_bitfield, model.cascade,
                                               model.grid_size, nears, fars, cnt, -1, True, 128, True, opt.dt_gamma,
                                               opt.max_steps)
    xs = xs.contiguous()
    x2 = (xs + torch.randn_like(xs) * 1e-2).contiguous()
    offs, P0 = grid_ops.stencil_offsets(center=True, second=True)
    n, P, cfg = xs.shape[0], offs.shape[0], model.encoder.cfg
    g = torch.randn(cfg["n_levels"], P * n, 2, device=dev).to(torch.float16)
    step = 2 * math.sqrt(3) / opt.max_steps

    def call():
        return field_ops.scatter_binned(xs, x2, offs, P0, float(model.bound), g, cfg, step, model.encoder.params.numel())
    call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / reps * 1e-3
    a = n * P * SCATTER_BYTES_PER_EVAL / t / 1e9
    return {"ms": 1e3 * t, "evaluations": n * P, "achieved": a, "unit": "GB/s", "peak": HBM_PEAK_GBPS,
            "frac": a / HBM_PEAK_GBPS, "what": "13-point mi3d_grid_scatter_binned on dense random binary16 gradients"}


def bench_refine(args, wl, dev, rank, world):
    """BASELINE config 5: one refine iteration (nerf/utils.py:839-894) on a synthetic textured point cloud."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from mi3d import rays as R, refine, sd_standin
    torch.manual_seed(rank)
    P, H, W = wl["points"], wl["H"], wl["W"]
    d = torch.randn(P, 3, device=dev)
    points = (d / d.norm(dim=-1, keepdim=True) * 0.35 * (1 + 0.05 * torch.randn(P, 1, device=dev))).contiguous()
    colour = torch.nn.Parameter(torch.rand(P, 3, device=dev))
    feat = torch.nn.Parameter(torch.randn(P, 16, device=dev))
    origin = colour.detach().clone()
    unet = refine.UNet(num_input_channels=19).to(dev).train()
    params = {"colour": colour, "feat": feat}
    optimizer = torch.optim.Adam([{"params": [colour, feat], "lr": 1e-3}, {"params": unet.parameters(), "lr": 1e-3}],
                                 betas=(0.9, 0.99), eps=1e-15)
    guidance = sd_standin.StableDiffusionStandIn(dev)
    text_z = guidance.get_text_embeds()
    t_fixed = T_FIXED  # an int: the guidance decides its branch on the host without a device sync
    w2c = torch.linalg.inv(R.orbit_pose(1.25, 80.0, 30.0 + 45.0 * rank, device=dev)[0])
    focal = 1.0 / (2 * np.tan(np.radians(20) / 2))
    radius = wl["radius_px"] / H * 2.0

    def step():
        refine.refine_train_step(unet, params, optimizer, guidance, text_z, points, w2c, focal, H, W, radius, wl["ppp"],
                                 origin, guidance_scale=5.0, t=t_fixed)
    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    # the point renderer alone (4 rasterise + composite passes of one iteration), HIP events on the launch stream
    feats = torch.cat((colour, feat), -1).detach()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    reps = 5
    for _ in range(reps):
        for sc in (1, 2, 4):
            refine.render_point(points, feats, H // sc, W // sc, refine.intrinsics(focal, H // sc, W // sc, dev), w2c,
                                (H // sc, W // sc), radius, wl["ppp"])
        refine.render_point(points, torch.ones_like(points), H, W, refine.intrinsics(focal, H, W, dev), w2c, (H, W),
                            radius, wl["ppp"])
    e1.record()
    torch.cuda.synchronize()
    render_ms = e0.elapsed_time(e1) / reps
    px = sum((H // s) * (W // s) for s in (1, 2, 4)) + H * W
    alg = 4 * P * 12 + px * wl["ppp"] * 16 + (px - H * W) * 19 * 4 + H * W * 3 * 4   # points in, idx+dists out and back in, images out
    if rank == 0:
        a = alg / (render_ms * 1e-3) / 1e9
        print(json.dumps({
            "metric": "refine-stage SDS train-steps/sec (point rasterise + U-Net + SD U-Net), BASELINE config 5",
            "value": world * args.steps / elapsed, "unit": "refine-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32 point renderer + f32 gated U-Net, f16 SD U-Net",
            "data": "synthetic (500 k points on a noisy sphere shell, random features / weights)",
            "config": {"workload": f"{args.workload}: {P} points, {H}x{W} (+ /2, /4, + mask pass), radius {wl['radius_px']} px, "
                                   f"{wl['ppp']} points per pixel, gated U-Net 19->3, SD2-base-shaped U-Net SDS (t={T_FIXED})",
                       "parallelism": f"dp{world} (independent views)"},
            "roofline": {"kernel": "point renderer: k_raster_count/scan/fill/tiles + k_points_composite_fwd x 4 passes "
                                   "(csrc/raster.hip)", "bound": "hbm", "achieved": a, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": a / HBM_PEAK_GBPS, "traffic": None, "avg_launch_ms": render_ms,
                         "note": "latency / atomics bound at this size (a few MB per pass): reported for completeness"},
            "render_ms_per_step": render_ms}))
    if world > 1:
        dist.destroy_process_group()


def reference_shaped(model, guidance, text_z, opt, rays, wl, t_fixed, dev, init_scale, steps=1, mark=False):
    """The reference's own NeRFNetwork + NeRFRenderer.run_cuda Python (staged, oracle/_ref/py) on the drop-in
    raymarching / tinycudann packages: same weights, same step (two-backward schedule, fp16 autocast, Adan)."""
    import torch
    from mi3d import optim, sds_step
    from oracle import ref_import
    if not ref_import.available():
        return {"value": None, "error": "reference sources not staged (oracle/build_ref.py)"}
    ref_model = ref_import.reference_network(opt, "dropin").to(dev)
    ref_model.load_state_dict(model.state_dict())
    ref_model.train()
    torch.cuda.empty_cache()
    optimizer = optim.Adan(ref_model.get_params(5 * opt.lr), eps=1e-8, weight_decay=2e-5, max_grad_norm=5.0)
    scaler = torch.amp.GradScaler("cuda", enabled=opt.fp16, init_scale=float(init_scale))
    ro, rd, ds = rays

    def step():
        sds_step.sds_train_step(ref_model, guidance, text_z, optimizer, scaler, ro, rd, ds, wl["H"], wl["W"], opt,
                                sds_backward="reference", t=t_fixed)
    t0 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    first = time.perf_counter() - t0
    log(f"reference-shaped warm-up step: {first:.2f} s")
    steps = 1 if first > 15 else steps   # bounded: the leg must not stretch the default run
    if mark:   # marker dispatches either side of the timed steps (tools/trace_sum.py --window spin_kernel)
        torch.cuda._sleep(1000)
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    if mark:
        torch.cuda._sleep(1000)
        torch.cuda.synchronize()
    peak = torch.cuda.max_memory_allocated(dev) / 2 ** 30
    del ref_model, optimizer
    torch.cuda.empty_cache()
    return {"value": 1.0 / dt, "unit": "view-steps/s", "ms_per_step": 1e3 * dt, "steps": steps,
            "what": "reference nerf/network_tcnn.py + nerf/renderer.py:481-583 unchanged on the drop-in raymarching + "
                    "tinycudann packages (13 hash-grid passes, torch nn.Linear MLP under autocast, atomic scatter), same "
                    "SD stand-in, two-backward schedule, Adan", "peak_mem_GiB": peak}


if __name__ == "__main__":
    main()
