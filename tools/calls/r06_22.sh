#!/bin/bash
# round 6, GPU call 22: the other end of the arena cap <-> step time curve: 112 GiB (one slice at C2) and 80 GiB
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_22
mkdir -p $O
for G in 112 80 56; do
  MI3D_SCATTER_WORKSPACE_GB=$G timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --variant-steps 0 --no-cpu-baseline --no-reference-shaped > $O/bench_cap$G.json 2> $O/bench_cap$G.err
  python - <<PY
import json
d=json.loads(open('$O/bench_cap$G.json').read().strip().splitlines()[-1])
print('cap $G GiB:', round(d['ms_per_step'],2), 'ms/step, scatter', round(d['kernels_ms_per_step']['scatter'],2), 'peak', round(d['peak_mem_GiB'],1), 'GiB', d['scatter_arena_placement'], 'dense', round(d['scatter_dense_gradients']['ms'],2))
PY
done
