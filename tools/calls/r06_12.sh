#!/bin/bash
# round 6, GPU call 12: after removing the reference cycle in scatter_binned's trial closure: the memory test, the whole
# GPU suite, one bench run (peak memory back to ~81 GiB?)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_12
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
cp gpurun_out/headline_parity.json $O/ 2>/dev/null
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print(round(d['ms_per_step'],2), 'scatter', round(d['kernels_ms_per_step']['scatter'],2), 'dense', round(d['scatter_dense_gradients']['ms'],2), 'dense step', round(d['dense_gradient_step']['ms_per_step'],1), 'peak', round(d['peak_mem_GiB'],1), d['scatter_arena_placement'], 'valid', d['valid'])
PY
