"""CPU: mi3d.grid_ops.GridParameter - the hash table's parameter class (`tinycudann.Encoding.params`).  It IS an
nn.Parameter (module registration, state_dict key, .to(), deepcopy, optimizers, GradScaler all see a parameter); the one
thing it adds is that READING `.grad` first completes whatever a backward pass parked on it (grid_ops.DEFER_POINT0,
mi3d/field_ops.py), and that ASSIGNING `.grad` drops it.  The scatter itself needs the GPU (tests/test_sds_step_gpu.py);
here the completion hook is observed through a stand-in."""
import copy

import torch


def test_grid_parameter_is_a_parameter_with_a_completing_grad(monkeypatch):
    import tinycudann as tcnn
    from mi3d import field_ops, grid_ops
    enc = tcnn.Encoding(3, {"otype": "HashGrid", "n_levels": 4, "n_features_per_level": 2, "log2_hashmap_size": 12,
                            "base_resolution": 4, "per_level_scale": 1.5})
    p = enc.params
    assert isinstance(p, torch.nn.Parameter) and isinstance(p, grid_ops.GridParameter) and p.requires_grad
    assert list(enc.state_dict().keys()) == ["params"] and [n for n, _ in enc.named_parameters()] == ["params"]
    # autograd accumulates through the C++ accessor; Python readers go through the property
    (p * 2).sum().backward()
    assert torch.equal(p.grad, torch.full_like(p, 2.0))
    # something parked: the first read completes it (stand-in: adds 1 to the real accumulator), later reads do not
    calls = []

    def fake_flush(param, key=None):
        calls.append(len(param.__dict__["_mi3d_pending"]))
        param.__dict__["_mi3d_pending"].clear()
        torch.Tensor.grad.__get__(param).add_(1.0)
    monkeypatch.setattr(field_ops, "flush_pending", fake_flush)
    p.__dict__["_mi3d_pending"] = [{"key": object()}]
    assert torch.equal(p.grad, torch.full_like(p, 3.0)) and calls == [1]
    assert torch.equal(p.grad, torch.full_like(p, 3.0)) and calls == [1]
    # the readers a training step has: clip_grad_norm_, an optimizer, zero_grad - all complete first
    p.__dict__["_mi3d_pending"].append({"key": object()})
    torch.nn.utils.clip_grad_norm_([p], max_norm=1e9)
    assert calls == [1, 1] and not p.__dict__["_mi3d_pending"]
    opt = torch.optim.SGD([p], lr=0.5)
    before = p.detach().clone()
    p.__dict__["_mi3d_pending"].append({"key": object()})
    opt.step()
    assert calls == [1, 1, 1] and torch.allclose(p.detach(), before - 0.5 * 5.0)
    # assigning .grad replaces what was accumulated - and drops what was parked
    p.__dict__["_mi3d_pending"].append({"key": object()})
    p.grad = None
    assert not p.__dict__["_mi3d_pending"] and p.grad is None and calls == [1, 1, 1]
    # module plumbing keeps the class
    enc64 = copy.deepcopy(enc).double()
    assert isinstance(enc64.params, grid_ops.GridParameter) and enc64.params.dtype == torch.float64
    enc.load_state_dict(enc.state_dict())
    assert isinstance(enc.params, grid_ops.GridParameter)


def test_grid_parameter_pickles_without_its_per_process_state(monkeypatch):
    """torch.save(model) / mp.spawn / copy.copy after a deferred backward pass: the parked planes (HIP events) and the
    cached AccumulateGrad node in the parameter's __dict__ must not reach the pickle - what is parked is completed first,
    and the parameter comes back as a GridParameter."""
    import io
    import pickle
    import threading
    from mi3d import field_ops, grid_ops
    p = grid_ops.GridParameter(torch.arange(6, dtype=torch.float32))
    (p * 3).sum().backward()
    flushed = []

    def fake_flush(param, key=None):
        flushed.append(len(param.__dict__["_mi3d_pending"]))
        param.__dict__["_mi3d_pending"].clear()
    monkeypatch.setattr(field_ops, "flush_pending", fake_flush)
    p.__dict__["_mi3d_pending"] = [{"key": object(), "event": threading.Lock()}]   # (a lock does not pickle either)
    p.__dict__["_mi3d_acc_node"] = threading.Lock()
    q = pickle.loads(pickle.dumps(p))
    assert flushed == [1] and isinstance(q, grid_ops.GridParameter) and q.requires_grad and torch.equal(q, p)
    assert "_mi3d_pending" not in q.__dict__ and "_mi3d_acc_node" not in q.__dict__
    buf = io.BytesIO()
    mod = torch.nn.Module()
    mod.params = p
    torch.save(mod, buf)
    buf.seek(0)
    back = torch.load(buf, weights_only=False)
    assert isinstance(back.params, grid_ops.GridParameter) and torch.equal(back.params, p)
    # the state_dict route stores plain tensors, as ever
    sd = pickle.loads(pickle.dumps(mod.state_dict()))
    assert torch.equal(sd["params"], p.detach())


def test_deferral_is_off_for_parameters_with_post_accumulate_grad_hooks():
    """torch DDP / FSDP read the gradient from C++ right after a backward pass (post-accumulate-grad hooks): a pass whose
    planes were parked would be missing there, so such a parameter is never deferred."""
    from mi3d import field_ops, grid_ops
    p = grid_ops.GridParameter(torch.zeros(4))
    assert grid_ops.DEFER_POINT0
    p.register_post_accumulate_grad_hook(lambda t: None)
    assert field_ops._may_defer(p, 13, 1) is False


def test_deferral_is_off_under_a_process_group_unless_the_python_bucket_syncs(tmp_path):
    """torch DDP's reducer hooks the AccumulateGrad node in C++ - no Python-visible attribute gives it away (ADVICE round 5:
    the post-accumulate-grad test above does NOT cover it).  So under ANY initialised process group nothing is parked,
    unless the gradient sync is mi3d.dp.FlatGradBucket, which reads `.grad` through GridParameter.grad and says so."""
    import torch.distributed as dist
    from mi3d import dp, grid_ops
    assert grid_ops.deferral_allowed()
    was = grid_ops.PYTHON_GRAD_SYNC
    dist.init_process_group("gloo", init_method=f"file://{tmp_path}/pg", rank=0, world_size=1)
    try:
        grid_ops.PYTHON_GRAD_SYNC = False
        assert not grid_ops.deferral_allowed()                       # e.g. the reference trainer's DDP wrap
        p = grid_ops.GridParameter(torch.zeros(8))
        dp.FlatGradBucket([p])
        assert grid_ops.PYTHON_GRAD_SYNC and grid_ops.deferral_allowed()
    finally:
        dist.destroy_process_group()
        grid_ops.PYTHON_GRAD_SYNC = was
    assert grid_ops.deferral_allowed()


def test_eval_round_budget_and_schedule_defaults():
    """mi3d.renderer's compact eval rounds (DESIGN.md 3.5'): the row budget a round gets - 16 rows per ray, at least 2^18,
    at most max(2^21, 2 N) - and the attributes a user may set."""
    import types
    from mi3d.renderer import NeRFRenderer
    opt = types.SimpleNamespace(bound=1.0, cuda_ray=False, min_near=0.1, density_thresh=10.0, bg_radius=-1)
    r = NeRFRenderer(opt)
    assert r.infer_schedule == "budget" and r.infer_budget_rounds % 2 == 0 and r.infer_graph_rounds % 2 == 0
    assert [r.budget_rows(n) for n in (1024, 128 * 128, 256 * 256, 512 * 512, 1024 * 1024, 2048 * 2048)] == \
        [1 << 18, 1 << 18, 1 << 20, 1 << 21, 1 << 21, 2 * 2048 * 2048]
    r.infer_budget_rows = 5000
    assert r.budget_rows(128 * 128) == 5000
    r.infer_budget_rows = 0
    assert r.budget_rows(7) == 1
