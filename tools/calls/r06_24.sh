#!/bin/bash
# round 6, GPU call 24: default arena cap 30 GiB (4 slices at C2), 6 candidates spread over the free memory: three bench
# processes (headline only), then 56 GiB / 4 the same way for comparison on the same box
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_24
mkdir -p $O
for CFG in "30 6 a" "30 6 b" "30 6 c" "56 4 a" "30 6 d"; do
  set -- $CFG
  MI3D_SCATTER_WORKSPACE_GB=$1 MI3D_SCATTER_PLACEMENT_TRIALS=$2 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --variant-steps 0 --no-cpu-baseline --no-reference-shaped > $O/bench_cap$1_$3.json 2> $O/bench.err
  python - <<PY
import json
d=json.loads(open('$O/bench_cap$1_$3.json').read().strip().splitlines()[-1])
print('cap $1 GiB, $2 trials:', round(d['ms_per_step'],2), 'ms/step, scatter', round(d['kernels_ms_per_step']['scatter'],2), 'peak', round(d['peak_mem_GiB'],1), 'GiB', d['scatter_arena_placement'], 'dense', round(d['scatter_dense_gradients']['ms'],2))
PY
done
