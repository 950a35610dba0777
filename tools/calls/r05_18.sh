#!/bin/bash
# round 5, GPU call 18: the other BASELINE workloads and the RCCL world-1 step at the final HEAD
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_18
mkdir -p $O
for W in c2_pruned c4_views c5_refine c5_refine_800; do
  timeout 400 python bench.py --workload $W --steps 10 --warmup 3 --variant-steps 0 --no-cpu-baseline --no-reference-shaped > $O/bench_$W.json 2> $O/bench_$W.err
  echo "$W rc=$?"
done
PORT=$((29500 + $$ % 2000))
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $PORT bench.py --gpus 1 --steps 10 --warmup 3 --variant-steps 0 --no-cpu-baseline --no-reference-shaped --force-dist > $O/bench_c2_dense_rccl_world1.json 2> $O/bench_rccl.err
echo "rccl rc=$?"
python - <<'P'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05_18/bench_*.json")):
    b = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
    print(f.split("/")[-1], round(b["ms_per_step"], 2), round(b["value"], 3), b.get("valid"), b.get("peak_mem_GiB"), b.get("kernels_ms_per_step", {}).get("scatter"))
    if "ranks" in b:
        r = b["ranks"]; print("  ranks:", r["backend"], r["rccl_world_size"], r["valid"], r["all_reduce_ms_per_step"]["max"], r["ms_per_step_local"]["max"])
P
