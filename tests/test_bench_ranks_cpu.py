"""CPU, gloo, world size 2: what bench.py says about a line measured on N > 1 ranks - every
rank's record gathered on every rank, the process group's own world size and backend next to the launcher's, per-rank step
time / all-reduce time / loss scale / applied updates / peak memory, and `valid: false` with a reason when the ranks
disagree on what they did.  The records here are stubs (no GPU in this container): the function under test is the one
bench.py calls with the real ones (tests/test_rccl_gpu.py runs that on the RCCL backend at world size 1)."""
import json
import os
import socket
import sys

import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _record(rank, applied=5, scale=2.0):
    return {"rank": rank, "steps": 5, "ms_per_step_local": 100.0 + 3.0 * rank, "all_reduce_ms_per_step": 4.0 - 3.0 * rank,
            "optimizer_steps_applied": applied, "grad_scaler_scale": scale, "grad_scaler_scale_after": scale,
            "peak_mem_GiB": 80.0 + rank, "views": [rank], "device": "stub"}


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path.insert(0, ROOT)
        import bench
        agree = bench.summarise_ranks(bench.gather_rank_records(_record(rank)), 5, dist.get_backend(),
                                      dist.get_world_size(), world)
        # rank 1 skipped an update the others applied, and sits on another loss scale
        split = bench.summarise_ranks(bench.gather_rank_records(_record(rank, applied=5 - rank, scale=2.0 + rank)), 5,
                                      dist.get_backend(), dist.get_world_size(), world)
        json.dump({"agree": agree, "split": split}, open(f"{out}.{rank}", "w"))
    finally:
        dist.destroy_process_group()


def test_rank_records_are_gathered_and_judged_on_every_rank(tmp_path):
    out = str(tmp_path / "ranks")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    res = [json.load(open(f"{out}.{r}")) for r in range(2)]
    assert res[0] == res[1]                                  # every rank holds the same summary
    a, s = res[0]["agree"], res[0]["split"]
    assert a["valid"] and not a["problems"] and a["backend"] == "gloo"
    assert a["rccl_world_size"] == a["launcher_world_size"] == 2 and [r["rank"] for r in a["per_rank"]] == [0, 1]
    assert a["ms_per_step_local"] == {"min": 100.0, "median": 101.5, "max": 103.0}
    assert a["all_reduce_ms_per_step"] == {"min": 1.0, "median": 2.5, "max": 4.0}   # the faster rank waits longer
    assert a["compute_ms_per_step"] == {"min": 96.0, "max": 102.0} and a["peak_mem_GiB"] == [80.0, 81.0]
    assert not s["valid"] and any("applied optimizer updates" in p for p in s["problems"])
    assert any("loss scale" in p for p in s["problems"])


def test_a_single_process_summary_and_the_launcher_mismatch():
    sys.path.insert(0, ROOT)
    import bench
    one = bench.summarise_ranks(bench.gather_rank_records(_record(0)), 5, "nccl", 1, 1)
    assert one["valid"] and one["rccl_world_size"] == 1 and len(one["per_rank"]) == 1
    skipped = bench.summarise_ranks([_record(0, applied=4)], 5, "nccl", 1, 1)
    assert not skipped["valid"] and "skipped the optimizer update" in skipped["problems"][0]
    wrong = bench.summarise_ranks([_record(0)], 5, "nccl", 1, 8)     # the launcher said 8, the group has 1
    assert not wrong["valid"] and "launcher announced 8" in wrong["problems"][0]
