#!/bin/bash
# round 6, GPU calls 51-53: the driver's command (side legs off) at the shipped defaults, one run per call = one per box
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_51
mkdir -p $O
T=$(date +%s)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-reference-shaped --variant-steps 0 > $O/bench_$T.json 2> $O/bench_$T.err
python - <<PY
import json
d=json.loads(open('$O/bench_$T.json').read().strip().splitlines()[-1])
print(round(d['ms_per_step'],2), 'scatter', round(d['kernels_ms_per_step']['scatter'],2), 'dense', round(d['scatter_dense_gradients']['ms'],2), 'peak', round(d['peak_mem_GiB'],1), d['scatter_arena_placement'][0]['candidates_ms'], 'valid', d['valid'], {k:round(v,2) for k,v in d['kernels_ms_per_step'].items()})
PY
