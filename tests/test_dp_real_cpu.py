"""CPU, world_size 2, gloo: the data-parallel step on the REAL parameter set of the field -
mi3d.network.NeRFNetwork's 12 196 240-entry hash table + the 6 532 MLP parameters in ONE flat bucket, per-rank views and
seeds as bench.py deals them, the all-reduce between backward and the global-norm clip, an identical Adan step on every
rank, and a GradScaler overflow on ONE rank skipping the update on BOTH (the sum carries the inf to every rank).

The field's kernels do not run on the CPU: the per-rank gradients are seeded synthetic tensors written into the bucket's
views (what backward would have accumulated there); everything after backward is the product's own code path
(dp.FlatGradBucket, clip_grad_norm_, GradScaler, mi3d.optim.Adan's torch-op route - the arithmetic the HIP kernel is
pinned to by tests/test_headline_parity_gpu.py)."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path.insert(0, ROOT)
        import bench
        from mi3d import dp, optim, rays as R, sds_step
        from mi3d.network import NeRFNetwork
        torch.set_num_threads(4)
        opt = sds_step.make_opt()
        torch.manual_seed(100 + rank)                    # different initial weights per rank, on purpose
        model = NeRFNetwork(opt)
        with torch.no_grad():
            model.density_bitfield.fill_(rank + 1)
        dp.broadcast_module_state(model)
        bucket = dp.FlatGradBucket(model.parameters())
        optimizer = optim.Adan(model.get_params(5 * opt.lr), eps=1e-8, weight_decay=2e-5, max_grad_norm=5.0)
        scaler = torch.amp.GradScaler("cpu", init_scale=4.0, growth_interval=1000)
        views, seed = bench.rank_views(rank, 1)
        ro, rd, _ = R.view_rays(16, 16, view=views[0])
        rec = {"numel": bucket.flat.numel(), "nbytes": bucket.nbytes, "views": views, "seed": seed,
               "ray_dir0": rd[0, 0].clone(), "w0": model.sigma_net.net[0].weight.detach().clone(),
               "bits0": int(model.density_bitfield[0])}
        gen = torch.Generator().manual_seed(seed)

        def one_step(poison):
            optimizer.zero_grad(set_to_none=False)
            scaler.scale(torch.zeros(1))                 # (initialises the scaler's state, as scale(loss) does)
            for p in model.parameters():                 # "backward": rank-dependent gradients into the bucket's views
                p.grad.copy_(torch.randn(p.shape, generator=gen) * 1e-3 * scaler.get_scale())
            if poison:
                model.encoder.params.grad[12345] = float("inf")
            local = bucket.flat[:64].clone()
            bucket.all_reduce_mean()
            reduced = bucket.flat[:64].clone()
            torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=10)
            before = model.encoder.params.detach()[:4096].clone()
            scaler.step(optimizer)
            scaler.update()
            moved = bool((model.encoder.params.detach()[:4096] != before).any())
            return local, reduced, moved, scaler.get_scale()

        rec["step1"] = one_step(poison=False)
        rec["step2"] = one_step(poison=(rank == 1))      # overflow on ONE rank
        rec["step3"] = one_step(poison=False)
        rec["table_sum"] = float(model.encoder.params.detach().double().sum())
        rec["table_head"] = model.encoder.params.detach()[:256].clone()
        rec["w_after"] = model.sigma_net.net[0].weight.detach().clone()
        out[rank] = rec
    finally:
        dist.destroy_process_group()


def test_real_parameter_bucket_world2():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    a, b = out[0], out[1]
    # the all-reduce unit: 6 098 120 table entries x 2 features + the 3 x 64 MLP (32*64+64 + 64*64+64 + 64*4+4)
    assert a["numel"] == b["numel"] == 12_196_240 + 6_532 == 12_202_772 and a["nbytes"] == 4 * 12_202_772
    # views and seeds differ per rank (bench.rank_views), and so do the rays
    assert a["views"] == [0] and b["views"] == [1] and a["seed"] != b["seed"]
    assert not torch.allclose(a["ray_dir0"], b["ray_dir0"])
    # identical initial state from rank 0
    assert torch.equal(a["w0"], b["w0"]) and a["bits0"] == b["bits0"] == 1
    # step 1: local gradients differ, the reduced bucket is their mean on both ranks, both ranks update
    la, ra, ma, sa = a["step1"]
    lb, rb, mb, sb = b["step1"]
    assert not torch.equal(la, lb) and torch.equal(ra, rb) and torch.allclose(ra, (la + lb) / 2, rtol=1e-6, atol=1e-9)
    assert ma and mb and sa == sb == 4.0
    # step 2: an inf on rank 1 only -> the summed bucket is non-finite on BOTH -> neither rank updates, both halve
    _, ra2, ma2, sa2 = a["step2"]
    _, rb2, mb2, sb2 = b["step2"]
    assert not ma2 and not mb2 and sa2 == sb2 == 2.0
    # step 3: both update again; the parameters stayed identical across the ranks throughout
    assert a["step3"][2] and b["step3"][2]
    assert a["table_sum"] == b["table_sum"] and torch.equal(a["table_head"], b["table_head"])
    assert torch.equal(a["w_after"], b["w_after"]) and not torch.equal(a["w_after"], a["w0"])


def test_launcher_for_eight_gpus(monkeypatch):
    """`python bench.py --gpus 8` outside a launcher: eight ranks on one node over 127.0.0.1, dmabuf IPC kept on."""
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    seen = {}
    monkeypatch.setattr(bench.subprocess, "call", lambda cmd, env=None: seen.update(cmd=cmd, env=env) or 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5"])
    try:
        bench.spawn_ranks(8)
    except SystemExit as e:
        assert e.code == 0
    cmd = seen["cmd"]
    assert "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # every rank of an 8-GPU step renders its own view with its own seed
    views = [bench.rank_views(r, 1) for r in range(8)]
    assert sorted(v[0][0] for v in views) == list(range(8)) and len({v[1] for v in views}) == 8
