#!/bin/bash
# round 5, GPU call 8 (call 7 again: its step times included the clock sampler thread's join): the driver's bench command at the round's kernels, the other BASELINE workloads (+ the refine stage at
# the reference's default 800 x 800), the full-size step under the RCCL process group at world size 1, the eval bench
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_8
mkdir -p $O
timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c2_dense.json 2> $O/bench.err
echo "bench rc=$?" >> $O/bench.err; tail -2 $O/bench.err
for W in c2_pruned c4_views c5_refine c5_refine_800; do
  timeout 400 python bench.py --workload $W --steps 10 --warmup 3 --variant-steps 0 --no-cpu-baseline --no-reference-shaped > $O/bench_$W.json 2> $O/bench_$W.err
  echo "$W rc=$?"
done
PORT=$((29500 + $$ % 2000))
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $PORT bench.py --gpus 1 --steps 10 --warmup 3 --variant-steps 0 --no-cpu-baseline --no-reference-shaped --force-dist > $O/bench_c2_dense_rccl_world1.json 2> $O/bench_rccl.err
echo "rccl rc=$?"
timeout 300 python tools/eval_bench.py --out $O/eval_bench.json > /dev/null 2> $O/eval_bench.err
python - <<'P'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05_8/bench_*.json")):
    try:
        b = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
        print(f.split("/")[-1], round(b["ms_per_step"], 2), round(b["value"], 3), b.get("valid"), b.get("clocks"), b.get("peak_mem_GiB"))
        if "ranks" in b:
            r = b["ranks"]; print("  ranks:", r["backend"], r["rccl_world_size"], r["valid"], r["all_reduce_ms_per_step"], r["ms_per_step_local"])
    except Exception as e:
        print(f, "unreadable", e)
e = json.load(open("gpurun_out/r05_8/eval_bench.json"))
print({k: round(v["ms_median"], 2) for k, v in e.items() if isinstance(v, dict) and "ms_median" in v})
P
