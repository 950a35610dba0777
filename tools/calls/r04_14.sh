#!/bin/bash
# round 4, GPU call 14: the full-size step under torch.distributed.run with the RCCL process group (world size 1): the
# state broadcast, the per-step 48.8 MB bucket all-reduce and the occupancy broadcast execute at BASELINE size
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r04_14
mkdir -p $O
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 3 --force-dist --variant-steps 0 --no-cpu-baseline --no-reference-shaped > $O/bench_force_dist.json 2> $O/bench_force_dist.err
echo "rc=$?" >> $O/bench_force_dist.err
tail -3 $O/bench_force_dist.err
python - <<'P'
import json
d=json.loads([l for l in open("gpurun_out/r04_14/bench_force_dist.json") if l.startswith("{")][-1])
print(round(d["ms_per_step"],1), d.get("collectives"), d["phases_ms_per_step"]["sync_clip_optimizer"])
P
