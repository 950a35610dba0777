"""The field evaluation - stencil-aware hash-grid encode + matrix-core MLP (+ the field head) - as ONE autograd node.

forward : features = encode(all P stencil points of every sample) -> h = MLP(features) -> head      (3 launches)
backward: head backward -> MLP backward (feature gradient written as level-major planes, weight gradients reduced
          on chip) -> hash-grid gradient scatter through the binned record path (csrc/hashgrid.hip)

Rows are POINT-MAJOR everywhere (row = p*n + s): the rows of one stencil point are contiguous, so a backward pass
that reaches only the first P' points of the stencil runs over that prefix and nothing else.  That is what makes the
reference's two-backward SDS schedule (nerf/sd.py:171 then nerf/utils.py:983) cheap here: its first backward carries a
gradient for sigma and albedo only (point 0: 1/13 of the rows), its second one for everything; `_Field` reads which
of its outputs received a gradient (None = none) and sizes the MLP backward and the scatter accordingly.

The per-layer composition (grid_ops.encode_points + mlp_ops.fused_mlp) stays available and computes the same thing.
"""
import ctypes as C
import os
import threading

import numpy as np
import torch
from torch.autograd import Function

from . import _lib as L
from . import grid_ops
from . import mlp_ops

# Upper bound of the scatter's record arena, bytes.  The arena holds every corner contribution of a slice of samples, so
# the samples are processed in the fewest equal slices that fit (mi3d_grid_scatter_binned: csrc/hashgrid.hip slice_for).  It is a plain torch allocation made per call: the caching allocator hands the same block
# back every step, it is stream-safe, and torch.cuda.empty_cache() releases it.
WORKSPACE_CAP_BYTES = int(float(os.environ.get("MI3D_SCATTER_WORKSPACE_GB", "33")) * (1 << 30))


# ---- where the record arena LIES matters (round 6, DESIGN.md 3.2: "the placement of the arena") -------------------------
# The emit appends to ~885 000 open (wave, bin) regions at once; what leaves the L2s is a stream of partial 64-byte
# writes, and how well HBM takes that stream depends on the PHYSICAL placement of the arena: the same call, same data,
# same clocks, took 44.5 to 55.3 ms on differently placed 56 GiB blocks of one process (k_bin_emit 13.7 to 18.7 ms per
# slice, k_bin_reduce - streaming reads - 8.9 ms on all of them; TCC_EA0_WRREQ_STALL 205 M against 320 M, address
# translation misses and plain fill / read bandwidth equal: profiles/scatter_placement_r06.json).  That was the
# "bimodal" dense call of round 5 and most of the step-time spread between boxes: torch's caching allocator hands the arena
# back and re-allocates it as memory pressure comes and goes, and every re-allocation is a new draw.
# So an arena of PLACED_MIN_BYTES and more is allocated ONCE per device, kept for the life of the process (never returned
# to the caching allocator: torch.cuda.empty_cache() does not move it; release_scatter_arena() does), and chosen: up to
# PLACEMENT_TRIALS candidate blocks are allocated side by side, the caller's own scatter is timed on each with synthetic
# dense gradients, the fastest is kept and the others are returned to the driver (~1 s, once).
PLACED_MIN_BYTES = 8 << 30
# candidate blocks: 1 = take the first block as it comes; unset = 7 for an arena below 40 GiB, 4 above (of six runs at 20-33 GiB
# one was offered no fast block among four candidates, none of the runs at 56 GiB: profiles/bench_r06_arena_cap_claimed_tiles.json)
PLACEMENT_TRIALS = int(os.environ.get("MI3D_SCATTER_PLACEMENT_TRIALS", "0"))
_ARENA_LOCK = threading.RLock()   # (autograd runs backward passes on worker threads, one per device)
_ARENAS = {}          # device index -> the persistent arena (uint8 tensor)
_ARENA_LAST_USE = {}  # device index -> event behind the last scatter that used it (torch's allocator no longer orders them)
PLACEMENT_LOG = []    # one record per calibration: candidates' ms, the one kept (bench.py reports it)


def release_scatter_arena(device=None):
    """Give the persistent record arena(s) back to the caching allocator (the next large scatter places a new one)."""
    with _ARENA_LOCK:
        for k in [k for k in _ARENAS if device is None or k == torch.device(device).index]:
            del _ARENAS[k]
            _ARENA_LAST_USE.pop(k, None)


def _alloc(want):
    """torch.empty(want) bytes on the current device; halved after a failure, None below 64 MiB of a failed request."""
    failed = False
    while want > 0 and not (failed and want < (64 << 20)):
        try:
            return torch.empty(want, dtype=torch.uint8, device="cuda")
        except torch.cuda.OutOfMemoryError:
            failed = True
            want //= 2
    return None


def scatter_workspace(device, needed, cap=None, trial=None):
    """uint8 scratch of min(needed, cap, 80 % of what the device can still give) bytes.  A small request is honoured as it
    is from torch's caching allocator (a 2000-sample call needs 40 MiB and gets 40 MiB: it takes the record path like a
    large one); only after an allocation FAILURE is the request halved, and below 64 MiB of a failed request nothing
    useful can be had: None, and the scatter takes its all-atomic path.  Requests above 1 GiB are whole GiB.  Requests of
    PLACED_MIN_BYTES and more are served from the device's persistent, placed arena (see above); `trial(ws)` - optional -
    runs the caller's scatter on a candidate block for the placement choice."""
    cap = WORKSPACE_CAP_BYTES if cap is None else int(cap)
    gib = 1 << 30
    device = torch.device(device)
    held = _ARENAS.get(device.index)
    free, _ = torch.cuda.mem_get_info(device)
    cached = torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device)  # torch can re-use this itself
    room = free + cached + (held.numel() if held is not None else 0)
    want = min((int(needed) + gib - 1) // gib * gib if needed > gib else int(needed), cap, int(0.8 * room))
    with torch.cuda.device(device), _ARENA_LOCK:
        if want < PLACED_MIN_BYTES:
            return _alloc(want)
        if held is not None and held.numel() >= want:
            return held[:want]
        if held is not None:                       # a larger one is needed: the old one goes first
            del _ARENAS[device.index], held
        n_try = 1
        trials = PLACEMENT_TRIALS if PLACEMENT_TRIALS > 0 else (7 if want < (40 << 30) else 4)
        if trial is not None and trials > 1:
            torch.cuda.empty_cache()               # (cached blocks would only shrink the room for candidates)
            free, _ = torch.cuda.mem_get_info(device)
            n_try = max(1, min(trials, int(0.9 * free) // want))
        # candidates SPREAD over the free memory (a spacer in front of each): blocks allocated back to back sit next to
        # each other, and whole stretches of the device memory are slow for this write stream - at 32 GiB the first three of
        # six back-to-back candidates timed 61-62 ms, the next three 52-54 (profiles/bench_r06_arena_cap_curve.json)
        gap = 0
        if n_try > 1:
            gap = max(0, (int(0.9 * free) - n_try * want) // n_try) >> 28 << 28
        cands, times, spacers = [], [], []
        for _ in range(n_try):
            if gap:
                spacers.append(_alloc(gap))
            c = _alloc(want)
            if c is None or (cands and c.numel() < cands[0].numel()):
                break
            cands.append(c)
        del spacers
        if not cands:
            return None
        if len(cands) > 1:
            for c in cands:
                trial(c)                            # (warm: first touch of the block)
                torch.cuda.synchronize(device)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                trial(c)
                e1.record()
                torch.cuda.synchronize(device)
                times.append(e0.elapsed_time(e1))
            best = min(range(len(cands)), key=lambda i: times[i])
            PLACEMENT_LOG.append({"device": device.index, "bytes": cands[0].numel(), "candidates_ms": [round(t, 2) for t in times],
                                  "kept": best})
            keep = cands[best]
            del cands, c
            _ARENAS[device.index] = keep
            torch.cuda.empty_cache()               # the others go back to the driver, not into torch's cache
        else:
            _ARENAS[device.index] = cands[0]
        held = _ARENAS[device.index]
        return held[:want] if held.numel() >= want else held


def scatter_binned(x, x2, offsets, P0, bound, dplanes, cfg, step, n_params, workspace_bytes=None, extra0=None):
    """grad_params [n_params] from level-major feature-gradient planes [L][P*n][2], rows point-major (C ABI:
    mi3d_grid_scatter_binned_plus).  workspace_bytes: None = scratch sized by scatter_workspace; 0 = force the all-atomic
    path.  extra0: a second set of planes [L][n][2] for stencil point 0 (same dtype), added to point 0's pairs."""
    offs, offs_p = grid_ops._offs_arg(offsets)
    P, n = offs.shape[0], x.shape[0]
    if dplanes.dtype not in (torch.float32, torch.float16) or not dplanes.is_cuda or not dplanes.is_contiguous():
        raise L.Mi3dError("dplanes must be a contiguous float32 / float16 GPU tensor")
    if dplanes.numel() != cfg["n_levels"] * n * P * 2:
        raise L.Mi3dError(f"dplanes has {dplanes.numel()} elements, expected [L={cfg['n_levels']}][{P * n}][2]")
    if extra0 is not None and (extra0.dtype != dplanes.dtype or not extra0.is_cuda or not extra0.is_contiguous()
                               or extra0.numel() != cfg["n_levels"] * n * 2 or extra0.device != dplanes.device):
        raise L.Mi3dError(f"extra0 must be contiguous {dplanes.dtype} planes [L={cfg['n_levels']}][{n}][2] on the same GPU")
    grad = torch.zeros(n_params, dtype=torch.float32, device=x.device)
    lib = L.lib()

    def launch(ws, planes, out):
        L.call("mi3d_grid_scatter_binned_plus", L.ptr(x), L.ptr(x2), n, offs_p, int(P0), P, float(bound), L.ptr(planes),
               L.ptr(extra0), int(dplanes.dtype == torch.float16), cfg["n_levels"], cfg["base_resolution"],
               cfg["per_level_scale"], cfg["log2_hashmap_size"], float(step),
               L.ptr(ws), C.c_size_t(ws.numel() if ws is not None else 0), L.ptr(out), L.stream(x))

    st = {}   # (not an attribute of `trial`: a function that names itself is a reference cycle - it would keep the gradient
              #  planes and the arena view alive until the cyclic collector runs)

    def trial(cand):
        """This very scatter on a candidate block, with synthetic dense gradients (the real ones may be full of inf / NaN
        while the loss scale settles - those bypass the records) into a scratch gradient."""
        if "planes" not in st:
            st["planes"] = torch.empty_like(dplanes).uniform_(-1.0, 1.0)
            st["out"] = torch.zeros_like(grad)
        with L.on(x):
            launch(cand, st["planes"], st["out"])

    ws = None
    if workspace_bytes != 0 and n > 0:
        needed = lib.mi3d_grid_scatter_binned_workspace(n, P, float(bound), float(step), cfg["n_levels"],
                                                        cfg["base_resolution"], cfg["per_level_scale"],
                                                        cfg["log2_hashmap_size"])
        if needed:
            try:
                ws = scatter_workspace(x.device, needed, workspace_bytes, trial=trial)
            except TypeError:     # (a spy with the three-argument signature: tests/conftest.py)
                ws = scatter_workspace(x.device, needed, workspace_bytes)
        st.clear()
    with L.on(x):
        placed = ws is not None and ws.numel() >= PLACED_MIN_BYTES
        if placed:   # the persistent arena is shared by every large scatter of the device: order them across streams
            cur = torch.cuda.current_stream(x.device)
            last = _ARENA_LAST_USE.get(x.device.index)
            if last is not None:
                cur.wait_event(last)
        grid_ops._timed("scatter", lambda: launch(ws, dplanes, grad), n * P)
        if placed:
            _ARENA_LAST_USE[x.device.index] = cur.record_event()
    return grad


def _forward_encode_mlp(params, ws, x, x2, offs, offs_p, P0, bound, cfg, half_mode, step=0.0, count=None):
    """feats [L][P*n][2] and h [P*n, 4] (point-major rows).  Under torch.autocast(float16) the planes hold binary16
    pairs: the first nn.Linear of the reference rounds its input to binary16 there, so the MLP output is bit-identical
    and every pass over the planes moves half the bytes."""
    P, n = offs.shape[0], x.shape[0]
    pdt = torch.float16 if half_mode else torch.float32
    feats = torch.empty(cfg["n_levels"], P * n, 2, dtype=pdt, device=x.device)
    # `count`: an int32 device scalar (the inference loop's row count) - samples beyond it are skipped by every kernel
    grid_ops._timed("encode", lambda: L.call(
        "mi3d_grid_encode_points_planes_counted", L.ptr(x), L.ptr(x2), n, L.ptr(count), offs_p, int(P0), P, float(bound),
        L.ptr(params), cfg["n_levels"], cfg["base_resolution"], cfg["per_level_scale"], cfg["log2_hashmap_size"],
        float(step), L.ptr(feats), int(bool(half_mode)), L.stream(x)), n * P)
    dims = (ws[0].shape[1], ws[0].shape[0], ws[4].shape[0])
    h = torch.empty(P * n, dims[2], dtype=torch.float32, device=x.device)
    grid_ops._timed("mlp_fwd", lambda: L.call(
        "mi3d_mlp_forward_counted", L.ptr(feats), P * n, int(bool(half_mode)), P * n, L.ptr(count), max(n, 1),
        *[L.ptr(t) for t in ws], *dims, int(half_mode), L.ptr(h), L.stream(x)), n * P)
    return feats, h, dims


def _last_pass_through_graph():
    """True when the backward pass now running frees the graph behind it (no retain_graph): whatever this node saved for
    backward is dead once this call returns.  False whenever that cannot be established."""
    try:
        return not bool(torch._C._autograd._get_current_graph_task_keep_graph())
    except Exception:  # noqa: BLE001 - the private probe is gone, or no graph task is running
        return False


def _backward_mlp(dh, feats, ws, dims, x, cfg, half_mode, P_active, in_place=False):
    """(dplanes [L][P_active*n][2], [dW1, db1, dW2, db2, dW3, db3]) from dh [P_active*n, 4]: the MLP backward over the
    first P_active points of the stencil only.  `in_place`: the gradient planes are written OVER the feature planes (the
    caller knows them dead after this pass) - a wave reads a tile's 32 rows of every plane into registers before it stores
    the same rows' gradients and no other wave touches those rows, so the kernel is indifferent; 9.05 GB less at the step's
    memory peak.  Only for whole passes over whole tiles (rows == plane rows, a multiple of 32: a partial tile's idle lanes
    re-read the last row, which another wave may have overwritten by then)."""
    n = x.shape[0]
    rows, plane_rows = P_active * n, feats.shape[1]
    if in_place and rows == plane_rows and rows % 32 == 0 and feats.is_contiguous():
        dplanes = feats
    else:
        # binary16 gradient planes under autocast: what the reference's binary16 dgrad GEMM hands the encoder's backward
        dplanes = torch.empty(cfg["n_levels"], rows, 2, dtype=feats.dtype, device=x.device)
    grads = [None if t is None else torch.zeros_like(t) for t in ws]
    grid_ops._timed("mlp_bwd", lambda: L.call(
        "mi3d_mlp_backward", L.ptr(feats), plane_rows, int(feats.dtype == torch.float16), L.ptr(dh), rows,
        *[L.ptr(t) for t in ws], *dims, int(half_mode),
        L.ptr(dplanes), rows, *[L.ptr(g) for g in grads], L.stream(x)), rows)
    return dplanes, grads


def _scatter_planes(dplanes, x, x2, offs, P0, bound, cfg, step, n_params, P_active, extra0=None):
    n = x.shape[0]
    rows = P_active * n
    if grid_ops.CENSUS is not None:  # bench.py, one untimed step: which gradient pairs are non-zero, and how they cluster
        nz = (dplanes != 0).any(-1).view(dplanes.shape[0], P_active, n)          # [L, P', n]
        if extra0 is not None:
            nz[:, 0] |= (extra0 != 0).any(-1)
        per_sample = nz.any(1)                                                    # [L, n]: any stencil point of the sample
        tiles = per_sample[:, :n - n % 64].view(per_sample.shape[0], -1, 64).any(-1)
        grid_ops.CENSUS.append({
            "P_active": int(P_active), "rows": int(rows), "with_deferred_point0": extra0 is not None,
            "nonzero_pairs_per_level": nz.sum((1, 2)),
            "samples_with_any_nonzero_point_per_level": per_sample.float().mean(1),
            "tiles64_with_any_nonzero_per_level": tiles.float().mean(1),
            "samples_nonzero_on_any_level": per_sample.any(0).float().mean(),
            "nonzero_fraction_per_point": nz.float().mean((0, 2))})
        del nz, per_sample, tiles
    if grid_ops.DENSIFY:     # bench.py, `dense_gradients` variant: defeat the emit's zero skip
        tiny = 2.0 ** -24 if dplanes.dtype == torch.float16 else 1e-30
        grid_ops._timed("densify", lambda: dplanes.masked_fill_(dplanes == 0, tiny), rows)
    return scatter_binned(x, x2 if P_active > P0 else None, offs[:P_active], min(P0, P_active), bound, dplanes, cfg, step,
                          n_params, extra0=extra0)


def _backward_mlp_scatter(dh, feats, ws, dims, x, x2, offs, P0, bound, cfg, step, n_params, half_mode, P_active):
    """(grad_params, [dW1, db1, dW2, db2, dW3, db3]) from dh [P_active*n, 4]: the MLP backward and the scatter run over
    the first P_active points of the stencil only."""
    dplanes, grads = _backward_mlp(dh, feats, ws, dims, x, cfg, half_mode, P_active)
    return _scatter_planes(dplanes, x, x2, offs, P0, bound, cfg, step, n_params, P_active), grads


# ---- the point-0 pass of a two-backward schedule, deferred (grid_ops.DEFER_POINT0) --------------------------------------

def _accumulates_into_grad(param):
    """True when the backward pass now running will ACCUMULATE into param.grad (Tensor.backward), False when it hands
    the gradient back to a caller (torch.autograd.grad: the engine refuses the question for a leaf then) or cannot say."""
    try:
        node = param.__dict__.get("_mi3d_acc_node")
        if node is None:
            with torch.enable_grad():
                node = param.expand_as(param).grad_fn.next_functions[0][0]
            param.__dict__["_mi3d_acc_node"] = node
        return bool(torch._C._will_engine_execute_node(node))
    except AttributeError as e:   # the private probe is gone from this torch build: say so once, then scatter every pass
        _warn_once(f"torch._C._will_engine_execute_node unavailable ({e}): the point-0 deferral is off, every backward "
                   f"pass scatters on its own (about +10 ms per C2 step)")
        return False
    except Exception:  # noqa: BLE001 - any doubt (e.g. torch.autograd.grad: the engine refuses the question): scatter now
        return False


_WARNED = set()


def _warn_once(msg):
    if msg not in _WARNED:
        _WARNED.add(msg)
        import warnings
        warnings.warn(msg, RuntimeWarning, stacklevel=3)


def _may_defer(param, P, P_active):
    """Park this pass's point-0 planes instead of scattering them?  Only when every one of these holds: the switch is on;
    the table is a GridParameter (its `.grad` completes the scatter for any reader); the pass stops at point 0 of a wider
    stencil; it runs under retain_graph=True, i.e. the caller announced another pass through this forward (nerf/sd.py:171
    does); and the engine is accumulating into `.grad` rather than returning gradients."""
    if not (grid_ops.deferral_allowed() and isinstance(param, grid_ops.GridParameter) and P_active == 1 and P > 1):
        return False   # (deferral_allowed: off under a process group whose gradient sync is not mi3d.dp's - torch DDP)
    if getattr(param, "_post_accumulate_grad_hooks", None):
        # FSDP and any register_post_accumulate_grad_hook user read the gradient right after this pass: they would miss
        # what is parked - scatter now (grid_ops.DEFER_POINT0's note)
        return False
    try:
        keep = torch._C._autograd._get_current_graph_task_keep_graph()
    except AttributeError as e:
        _warn_once(f"torch._C._autograd._get_current_graph_task_keep_graph unavailable ({e}): the point-0 deferral is "
                   f"off, every backward pass scatters on its own (about +10 ms per C2 step)")
        return False
    except Exception:  # noqa: BLE001
        return False
    return bool(keep) and _accumulates_into_grad(param)


def _park(param, key, item):
    pend = param.__dict__.setdefault("_mi3d_pending", [])
    older = [it for it in pend if it["key"] == key]
    if older:                       # a second point-0 pass through the same forward: complete the first one now
        flush_pending(param, key)
    if len(pend) >= 2:              # planes of forwards nobody came back to: complete them rather than pile them up
        flush_pending(param)
    item["key"] = key
    item["stream"] = torch.cuda.current_stream(item["x"].device)
    item["event"] = item["stream"].record_event()
    pend.append(item)


def _find_parked(param, key):
    """The item an earlier pass through the forward `key` parked on `param` (None if none).  It STAYS parked until the
    scatter that takes its planes along has been issued (`_consume_parked`): if that scatter raises - a dtype / shape
    check, an out-of-memory - the first pass's gradient is still there for the next reader of `.grad`."""
    pend = None if param is None else param.__dict__.get("_mi3d_pending")
    for it in pend or ():
        if it["key"] == key:
            cur = torch.cuda.current_stream(it["x"].device)
            cur.wait_event(it["event"])
            if it.get("stream") is not None and it["stream"] != cur:
                # parked on another stream: the caching allocator must not hand the block back to that stream's
                # allocations while this one still reads it
                it["dplanes"].record_stream(cur)
            return it
    return None


def _consume_parked(param, item):
    pend = None if param is None else param.__dict__.get("_mi3d_pending")
    if pend and item is not None:
        pend[:] = [it for it in pend if it is not item]


@torch.no_grad()
def flush_pending(param, key=None):
    """Scatter what is parked on `param` (all of it, or the forward `key`'s) and accumulate it into the real `.grad`."""
    pend = param.__dict__.get("_mi3d_pending")
    if not pend:
        return
    items = [it for it in pend if key is None or it["key"] == key]
    for it in items:
        x = it["x"]
        with L.on(x):
            cur = torch.cuda.current_stream(x.device)
            cur.wait_event(it["event"])
            if it.get("stream") is not None and it["stream"] != cur:
                it["dplanes"].record_stream(cur)
            g = _scatter_planes(it["dplanes"], x, None, it["offs"], it["P0"], it["bound"], it["cfg"], it["step"],
                                it["n_params"], 1)
        pend[:] = [other for other in pend if other is not it]   # only once its scatter has been issued
        real = torch.Tensor.grad.__get__(param)
        if real is None:
            torch.Tensor.grad.__set__(param, g.view_as(param))
        else:
            real.add_(g.view_as(real))      # in place: `.grad` may be a view into an all-reduce bucket


class _FieldStencil(Function):
    """h [P*n, 4] = MLP(encode(stencil)); the head stays outside (any P, e.g. the 6-point finite_difference_normal)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, params, W1, b1, W2, b2, W3, b3, x, x2, offsets, P0, bound, cfg, step, half_mode):
        x = L.dev_f32(x.contiguous().view(-1, 3), "x", 3)
        if x2 is not None:
            x2 = L.dev_f32(x2.contiguous().view(-1, 3), "x2", 3)
        params = L.dev_f32(params, "params")
        ws = mlp_ops._weights((W1, b1, W2, b2, W3, b3))
        offs, offs_p = grid_ops._offs_arg(offsets)
        with L.on(x):
            feats, h, dims = _forward_encode_mlp(params, ws, x, x2, offs, offs_p, P0, bound, cfg, half_mode, step)
        ctx.save_for_backward(x, x2 if x2 is not None else x, feats, *ws)
        ctx.meta = (offs, int(P0), float(bound), cfg, float(step), x2 is not None, params.numel(), dims,
                    int(half_mode))
        return h

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dh):
        x, x2, feats, *ws = ctx.saved_tensors
        offs, P0, bound, cfg, step, has_x2, n_params, dims, half_mode = ctx.meta
        dh = L.dev_f32(dh.float().contiguous(), "dh", dims[2])
        with L.on(x):
            gp, grads = _backward_mlp_scatter(dh, feats, ws, dims, x, x2 if has_x2 else None, offs, P0, bound, cfg,
                                              step, n_params, half_mode, offs.shape[0])
        return (gp, *grads, None, None, None, None, None, None, None, None)


def _half_mode(half_mode):
    if half_mode is None:
        return torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.float16
    return bool(half_mode)


def field_stencil(params, layers, x, offsets, cfg, bound=1.0, x2=None, P0=None, step=0.0, half_mode=None):
    """h [P*n, 4] (row = point*n + sample): the MLP output at clamp(base + offsets[p]) for every sample, differentiable
    w.r.t. the hash table and the MLP weights.  `layers`: the two or three nn.Linear modules of sigma_net."""
    P = np.asarray(offsets).reshape(-1, 3).shape[0]
    if P0 is None:
        P0 = P
    return _FieldStencil.apply(params, *mlp_ops.layer_args(layers), x, x2, offsets, P0, bound, cfg, step,
                               _half_mode(half_mode))


def _head_forward(h, x, x2, offs, offs_p, bound, blob_density, blob_radius, epsilon, count=None, half_mode=False):
    """`half_mode`: under torch.autocast(float16) the reference's `torch.sigmoid(h[..., 1:])` (network_tcnn.py:110) takes
    the binary16 MLP output and RETURNS binary16 - its albedo carries 11 significant bits into the compositor.  The albedo
    here keeps its fp32 buffer but holds those rounded values (sigma does not change: `h[..., 0] + gaussian(x)` promotes
    to fp32 and trunc_exp casts to fp32, activation.py:7).  Round 5 left the albedo unrounded: that was the whole of the
    4e-5 / 8.4e-5 image difference to the reference route under autocast (profiles/headline_parity_r05.json)."""
    P, n, dev = offs.shape[0], x.shape[0], x.device
    alloc = torch.empty if count is None else torch.zeros   # rows beyond a device count are not written: keep them defined
    sigma = alloc(n, dtype=torch.float32, device=dev)
    albedo = alloc(n, 3, dtype=torch.float32, device=dev)
    normal = alloc(n, 3, dtype=torch.float32, device=dev)
    normal2 = alloc(n, 3, dtype=torch.float32, device=dev) if P == 13 else None
    grid_ops._timed("head_fwd", lambda: L.call(
        "mi3d_field_head_forward_counted", L.ptr(h), L.ptr(x), L.ptr(x2), n, L.ptr(count), offs_p, P, float(bound),
        float(blob_density), float(blob_radius), float(epsilon), L.ptr(sigma), L.ptr(albedo), L.ptr(normal),
        L.ptr(normal2), L.stream(x)), n)
    if half_mode:
        albedo.copy_(albedo.to(torch.float16))
    return sigma, albedo, normal, normal2


def _head_backward(h, x, x2, offs, bound, blob_density, blob_radius, epsilon, grads, P_active):
    """dh [P_active*n, 4] from the upstream gradients (dsigma, dalbedo, dnormal, dnormal2), any of them None."""
    _, offs_p = grid_ops._offs_arg(offs)
    n, P = x.shape[0], offs.shape[0]
    g = [None if t is None else L.dev_f32(t.float().contiguous(), "grad") for t in grads]
    g += [None] * (4 - len(g))
    dh = torch.empty(P_active * n, 4, dtype=torch.float32, device=x.device)
    grid_ops._timed("head_bwd", lambda: L.call(
        "mi3d_field_head_backward", L.ptr(h), L.ptr(x), L.ptr(x2), n, offs_p, P, int(P_active), float(bound),
        float(blob_density), float(blob_radius), float(epsilon), L.ptr(g[0]), L.ptr(g[1]), L.ptr(g[2]), L.ptr(g[3]),
        L.ptr(dh), L.stream(x)), n)
    return dh


def _active_points(P, grads):
    """How far into the stencil the upstream gradient reaches: 1 (sigma / albedo only), 7 (normal), 13 (normal2)."""
    if P == 13 and len(grads) > 3 and grads[3] is not None:
        return 13
    if len(grads) > 2 and grads[2] is not None:
        return 7
    return 1


class _FieldHead(Function):
    """sigma, albedo, normal(x), normal(x2) from h [P*n, 4] in one elementwise kernel per direction (C ABI Part 5)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, h, x, x2, offsets, bound, blob_density, blob_radius, epsilon):
        offs, offs_p = grid_ops._offs_arg(offsets)
        P = offs.shape[0]
        x = L.dev_f32(x.contiguous().view(-1, 3), "x", 3)
        n = x.shape[0]
        h = L.dev_f32(h.contiguous().view(P * n, 4), "h", 4)
        if x2 is not None:
            x2 = L.dev_f32(x2.contiguous().view(-1, 3), "x2", 3)
        with L.on(x):
            sigma, albedo, normal, normal2 = _head_forward(h, x, x2, offs, offs_p, bound, blob_density, blob_radius,
                                                           epsilon)
        ctx.save_for_backward(h, x, x2 if x2 is not None else x)
        ctx.meta = (offs, float(bound), float(blob_density), float(blob_radius), float(epsilon), x2 is not None)
        ctx.set_materialize_grads(False)
        if normal2 is None:
            return sigma, albedo, normal
        return sigma, albedo, normal, normal2

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, *grads):
        h, x, x2 = ctx.saved_tensors
        offs, bound, blob_density, blob_radius, epsilon, has_x2 = ctx.meta
        P, n = offs.shape[0], x.shape[0]
        P_active = _active_points(P, grads)
        with L.on(x):
            dh = _head_backward(h, x, x2 if has_x2 else None, offs, bound, blob_density, blob_radius, epsilon, grads,
                                P_active)
        if P_active < P:  # autograd wants the full [P*n, 4]: the points beyond the active prefix carry zeros
            dh = torch.cat([dh, dh.new_zeros((P - P_active) * n, 4)], 0)
        return dh, None, None, None, None, None, None, None


def field_head(h, x, offsets, bound, blob_density, blob_radius, x2=None, epsilon=grid_ops.EPS):
    """(sigma [n], albedo [n,3], normal [n,3], normal_jitter [n,3] or None) from h [P*n, 4]; P must be 7 or 13."""
    out = _FieldHead.apply(h, x, x2, offsets, bound, blob_density, blob_radius, epsilon)
    return out if len(out) == 4 else (*out, None)


class _Field(Function):
    """encode -> MLP -> head as one node (P = 7 or 13).  Its backward looks at WHICH outputs received a gradient and
    runs head backward, MLP backward and scatter over the reached prefix of the stencil only (module docstring)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, params, W1, b1, W2, b2, W3, b3, x, x2, offsets, P0, bound, cfg, step, half_mode, blob_density,
                blob_radius, epsilon):
        x = L.dev_f32(x.contiguous().view(-1, 3), "x", 3)
        if x2 is not None:
            x2 = L.dev_f32(x2.contiguous().view(-1, 3), "x2", 3)
        params = L.dev_f32(params, "params")
        ws = mlp_ops._weights((W1, b1, W2, b2, W3, b3))
        offs, offs_p = grid_ops._offs_arg(offsets)
        if offs.shape[0] not in (7, 13):
            raise L.Mi3dError("the fused field takes the 7- or 13-point stencil")
        with L.on(x):
            feats, h, dims = _forward_encode_mlp(params, ws, x, x2, offs, offs_p, P0, bound, cfg, half_mode, step)
            sigma, albedo, normal, normal2 = _head_forward(h, x, x2, offs, offs_p, bound, blob_density, blob_radius,
                                                           epsilon, half_mode=half_mode)
        ctx.save_for_backward(x, x2 if x2 is not None else x, feats, h, *ws)
        ctx.meta = (offs, int(P0), float(bound), cfg, float(step), x2 is not None, params.numel(), dims,
                    int(half_mode), float(blob_density), float(blob_radius), float(epsilon))
        # the table as the caller's parameter object (custom_fwd leaves an fp32 tensor as it is): where a point-0 pass
        # parks its planes for the pass that follows it through this forward (grid_ops.DEFER_POINT0)
        ctx.param_ref = params if isinstance(params, grid_ops.GridParameter) else None
        ctx.forward_key = object()
        ctx.set_materialize_grads(False)
        if normal2 is None:
            return sigma, albedo, normal
        return sigma, albedo, normal, normal2

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, *grads):
        x, x2, feats, h, *ws = ctx.saved_tensors
        (offs, P0, bound, cfg, step, has_x2, n_params, dims, half_mode, blob_density, blob_radius,
         epsilon) = ctx.meta
        none = (None,) * 18
        if all(g is None for g in grads):
            return none
        P_active = _active_points(offs.shape[0], grads)
        param = getattr(ctx, "param_ref", None)
        with L.on(x):
            dh = _head_backward(h, x, x2 if has_x2 else None, offs, bound, blob_density, blob_radius, epsilon, grads,
                                P_active)
            # the last pass through this forward (no retain_graph) over the whole stencil: the gradient planes take the
            # feature planes' place (grid_ops.INPLACE_GRAD_PLANES) - both were live at the step's memory peak
            dplanes, wg = _backward_mlp(dh, feats, ws, dims, x, cfg, half_mode, P_active,
                                        in_place=grid_ops.INPLACE_GRAD_PLANES and P_active == offs.shape[0]
                                        and _last_pass_through_graph())
            del dh
            if _may_defer(param, offs.shape[0], P_active):
                # nerf/sd.py:171's pass: nothing is scattered now - the planes wait for the pass that follows (or for
                # the first reader of encoder.params.grad)
                _park(param, ctx.forward_key, dict(dplanes=dplanes, x=x, offs=offs, P0=P0, bound=bound, cfg=cfg,
                                                   step=step, n_params=n_params))
                gp = None
            else:
                parked = _find_parked(param, ctx.forward_key) if P_active > 1 else None
                gp = _scatter_planes(dplanes, x, x2 if has_x2 else None, offs, P0, bound, cfg, step, n_params, P_active,
                                     extra0=None if parked is None else parked["dplanes"])
                _consume_parked(param, parked)   # (after the call: a scatter that raised leaves the planes parked)
        return (gp, *wg, *none[:11])


@torch.no_grad()
def field_rows(params, layers, x, offsets, cfg, bound, blob_density, blob_radius, count, step=0.0, half_mode=None,
               epsilon=grid_ops.EPS):
    """Forward only (sigma, albedo, normal) of the 7-point stencil for the rows below the DEVICE-side `count` (int32[1]):
    the inference loop's field call (renderer.py:546) - gather, MLP and head skip the rows the round does not use, so a
    stale host-side upper bound of the row count costs launches of empty tiles, not field evaluations."""
    x = L.dev_f32(x.contiguous().view(-1, 3).float(), "x", 3)
    params = L.dev_f32(params.detach(), "params")
    ws = mlp_ops._weights([None if t is None else t.detach() for t in mlp_ops.layer_args(layers)])
    offs, offs_p = grid_ops._offs_arg(offsets)
    if offs.shape[0] != 7:
        raise L.Mi3dError("field_rows takes the 7-point stencil")
    count = L.dev_typed(count, "count", torch.int32)
    with L.on(x):
        hm = _half_mode(half_mode)
        _, h, _ = _forward_encode_mlp(params, ws, x, None, offs, offs_p, 7, bound, cfg, hm, step, count)
        sigma, albedo, normal, _ = _head_forward(h, x, None, offs, offs_p, bound, blob_density, blob_radius, epsilon, count,
                                                 half_mode=hm)
    return sigma, albedo, normal


def field(params, layers, x, offsets, cfg, bound, blob_density, blob_radius, x2=None, P0=None, step=0.0,
          half_mode=None, epsilon=grid_ops.EPS):
    """(sigma [n], albedo [n,3], normal [n,3], normal_jitter [n,3] or None) of the 7- / 13-point stencil, one node."""
    P = np.asarray(offsets).reshape(-1, 3).shape[0]
    if P0 is None:
        P0 = P
    out = _Field.apply(params, *mlp_ops.layer_args(layers), x, x2, offsets, P0, bound, cfg, step,
                       _half_mode(half_mode), blob_density, blob_radius, epsilon)
    return out if len(out) == 4 else (*out, None)
