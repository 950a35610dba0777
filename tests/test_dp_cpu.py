"""CPU, world_size 2, gloo: the data-parallel path of the SDS loop (mi3d/dp.py) - one flat in-place all-reduce of the
parameter gradients per step, identical initial state on every rank, occupancy broadcast."""
import os
import socket
import types

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


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
        from mi3d import dp
        torch.manual_seed(100 + rank)  # different init per rank on purpose
        net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
        net.register_buffer("density_bitfield", torch.full((64,), rank + 1, dtype=torch.uint8))
        dp.broadcast_module_state(net)
        w_after = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
        bucket = dp.FlatGradBucket(net.parameters())
        assert bucket.nbytes == sum(p.numel() for p in net.parameters()) * 4
        # every .grad is a view into the one flat buffer
        off = 0
        for p in net.parameters():
            assert p.grad.data_ptr() == bucket.flat.data_ptr() + off * 4
            off += p.numel()
        bucket.zero()
        x = torch.full((4, 6), float(rank + 1))
        net(x).pow(2).sum().backward()  # rank-dependent gradients, accumulated in place into the bucket
        local = bucket.flat.clone()
        # something replaces a .grad tensor (zero_grad(set_to_none=True) then a fresh backward): folded back in
        first = next(net.parameters())
        first.grad = first.grad.clone()
        bucket.all_reduce_mean()
        assert first.grad.data_ptr() == bucket.flat.data_ptr()
        # occupancy sync
        model = types.SimpleNamespace(density_bitfield=torch.full((32,), 7 * (rank + 1), dtype=torch.uint8),
                                      density_grid=torch.full((1, 16), float(rank)), mean_density=float(rank + 3))
        dp.sync_occupancy(model)
        out[rank] = dict(w=w_after, local=local, reduced=bucket.flat.clone(), bits=net.density_bitfield.clone(),
                         occ=(model.density_bitfield.clone(), model.density_grid.clone(), model.mean_density))
    finally:
        dist.destroy_process_group()


def test_flat_bucket_allreduce_world2():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    a, b = out[0], out[1]
    assert torch.equal(a["w"], b["w"])                      # broadcast_module_state: same weights ...
    assert torch.equal(a["bits"], b["bits"]) and int(a["bits"][0]) == 1  # ... and buffers, from rank 0
    assert not torch.equal(a["local"], b["local"])          # local gradients differ
    mean = (a["local"] + b["local"]) / 2
    assert torch.allclose(a["reduced"], mean, rtol=1e-6, atol=1e-7)
    assert torch.equal(a["reduced"], b["reduced"])          # identical on every rank -> identical Adan clip + step
    for r in (a, b):
        bits, grid, md = r["occ"]
        assert int(bits[0]) == 7 and float(grid[0, 0]) == 0.0 and md == 3.0


def test_bucket_is_a_noop_without_process_group():
    from mi3d import dp
    net = torch.nn.Linear(3, 2)
    bucket = dp.FlatGradBucket(net.parameters())
    net(torch.ones(1, 3)).sum().backward()
    before = bucket.flat.clone()
    bucket.all_reduce_mean()
    assert torch.equal(before, bucket.flat) and before.abs().sum() > 0
    with pytest.raises(ValueError):
        dp.FlatGradBucket([])
