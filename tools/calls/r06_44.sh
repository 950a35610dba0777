#!/bin/bash
# round 6, GPU call 44: the arena cap with claimed tiles and the fewest-slices rule: 56 (2 slices) / 33 (3) / 25 (4) / 20 (5) GiB,
# the driver's command without its side legs, one box
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_44
mkdir -p $O
for GB in 56 33 25 20 56 25; do
  MI3D_SCATTER_WORKSPACE_GB=$GB timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-reference-shaped --variant-steps 0 > $O/bench_$GB.json 2> $O/bench_$GB.err
  python - <<PY
import json
d=json.loads(open('$O/bench_$GB.json').read().strip().splitlines()[-1])
print('$GB GiB', round(d['ms_per_step'],2), 'scatter', round(d['kernels_ms_per_step']['scatter'],2), 'dense', round(d['scatter_dense_gradients']['ms'],2), 'peak', round(d['peak_mem_GiB'],1), d['scatter_arena_placement'][0]['candidates_ms'], 'valid', d['valid'], 'encode', round(d['kernels_ms_per_step']['encode'],2), 'sd', round(d['kernels_ms_per_step']['sd_guidance'],2))
PY
  cp $O/bench_$GB.json $O/bench_${GB}_$(date +%s).json
done
