"""The multi-GPU path's RCCL calls on the ONE GPU a test box has: bench.py under torch.distributed.run
with --nproc-per-node=1 and the NCCL (= RCCL on ROCm) backend forced - process-group init on the device, the broadcast of
the module state, the flat 48.8 MB-bucket gradient all-reduce of every step, the occupancy broadcast after a grid refresh.
No scaling claim: it only proves that the first 8-GPU launch cannot die in calls that never ran anywhere."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_small_under_torchrun_with_nccl(cuda):
    port = 29500 + os.getpid() % 2000
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", "small", "--steps", "3",
           "--warmup", "1", "--variant-steps", "0", "--no-cpu-baseline", "--no-reference-shaped", "--force-dist"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")][-1])
    col = line["collectives"]
    assert col["backend"] == "nccl" and col["world_size"] == 1
    # one all-reduce per step: settle steps + 1 warm-up + 3 timed + the census step; broadcasts: module state + the
    # occupancy sync (bitfield, grid, mean density) of the timed region's grid refresh
    assert col["all_reduce"] >= 5, col
    assert col["broadcast"] >= 3 + 3, col
    assert line["n_gpus"] == 1 and line["value"] > 0
    assert line["config"]["grid_refreshes_in_timed_region"] >= 1
    # what a line measured under a process group says about itself (bench.summarise_ranks): the group's own world size and
    # backend, this rank's clock, the time its stream spent in the gradient all-reduce (HIP events), scale, updates, memory
    rk = line["ranks"]
    assert rk["backend"] == "nccl" and rk["rccl_world_size"] == rk["launcher_world_size"] == 1, rk
    assert rk["valid"] or all("skipped the optimizer update" in p for p in rk["problems"]), rk   # (a GradScaler overflow)
    r0 = rk["per_rank"][0]
    assert r0["rank"] == 0 and 0 <= r0["optimizer_steps_applied"] <= 3 and r0["ms_per_step_local"] > 0
    assert 0 < r0["all_reduce_ms_per_step"] < r0["ms_per_step_local"] and r0["peak_mem_GiB"] > 0
