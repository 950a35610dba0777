#!/bin/bash
# round 6, GPU call 11: the persistent, placed record arena (mi3d/field_ops.py): scatter tests, then the driver's bench
# command three times (three processes: does the in-step scatter still move from run to run?)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_11
mkdir -p $O
timeout 900 python -m pytest tests/test_grid_points_gpu.py tests/test_sds_step_gpu.py tests/test_headline_parity_gpu.py -q -x 2>&1 | tail -4
for i in 1 2 3; do
  S=$(date +%s)
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_$i.json 2> $O/bench_$i.err
  echo "bench $i wall $(( $(date +%s) - S )) s"
  python - <<PY
import json
d=json.loads(open('$O/bench_$i.json').read().strip().splitlines()[-1])
print(round(d['ms_per_step'],2), 'scatter', round(d['kernels_ms_per_step']['scatter'],2), 'dense', round(d['scatter_dense_gradients']['ms'],2), 'dense step', round(d['dense_gradient_step']['ms_per_step'],1), round(d['dense_gradient_step']['scatter_ms_per_step'],1), 'peak', round(d['peak_mem_GiB'],1), d['scatter_arena_placement'], 'refshaped', round(d['reference_shaped_baseline']['ms_per_step']), {k:round(v,1) for k,v in d['variants_ms_per_step'].items()})
PY
done
