#!/bin/bash
# round 6, GPU call 49: the other BASELINE workloads, the RCCL world-1 run and the eval loop at the shipped defaults
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_49
mkdir -p $O
for W in c2_pruned c4_views c5_refine c5_refine_800; do
  timeout 600 python bench.py --workload $W --steps 10 --warmup 3 --variant-steps 0 --no-cpu-baseline --no-reference-shaped > $O/bench_$W.json 2> $O/bench_$W.err
  python - <<PY
import json
d=json.loads(open('$O/bench_$W.json').read().strip().splitlines()[-1])
print('$W', round(d['value'],3), d['unit'], round(d['ms_per_step'],2), 'ms/step', d.get('valid'), d.get('peak_mem_GiB'), d.get('scatter_arena_placement'))
PY
done
HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 10 --warmup 3 --force-dist --variant-steps 0 --no-cpu-baseline --no-reference-shaped > $O/bench_rccl_world1.json 2> $O/bench_rccl_world1.err
python - <<PY
import json
d=json.loads(open('$O/bench_rccl_world1.json').read().strip().splitlines()[-1])
print('rccl world 1', round(d['ms_per_step'],2), d['valid'], d['collectives'], {k:d['ranks'][k] for k in ('valid','backend','world_size_group')} if 'ranks' in d else None)
PY
timeout 600 python tools/eval_bench.py 2>&1 | tail -6
