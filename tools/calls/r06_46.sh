#!/bin/bash
# round 6, GPU call 46: gradient planes written over the feature planes in the last pass + 33 GiB default arena (3 slices, 7 placement
# candidates): the new tests, the field / headline-parity / step tests, the driver's command twice
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_46
mkdir -p $O
timeout 900 python -m pytest tests/test_field_gpu.py tests/test_sds_step_gpu.py tests/test_headline_parity_gpu.py tests/test_mlp_gpu.py -m gpu -x -q 2>&1 | tail -4
for i in 1 2; do
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-reference-shaped --variant-steps 0 > $O/bench_$i.json 2> $O/bench_$i.err
python - <<PY
import json
d=json.loads(open('$O/bench_$i.json').read().strip().splitlines()[-1])
print(round(d['ms_per_step'],2), 'scatter', round(d['kernels_ms_per_step']['scatter'],2), 'dense', round(d['scatter_dense_gradients']['ms'],2), 'peak', round(d['peak_mem_GiB'],1), d['scatter_arena_placement'][0]['candidates_ms'], 'valid', d['valid'], 'mlp_bwd', round(d['kernels_ms_per_step']['mlp_bwd'],2))
PY
done
