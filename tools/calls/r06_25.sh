#!/bin/bash
# round 6, GPU call 25: at the HEAD - smoke, the whole GPU suite, the driver's bench command
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_25
mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
S=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
echo "suite wall $(( $(date +%s) - S )) s"
cp gpurun_out/headline_parity.json $O/ 2>/dev/null
S=$(date +%s)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
echo "bench wall $(( $(date +%s) - S )) s"
grep "bench " $O/bench.err | tail -12
python - <<PY
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print(round(d['ms_per_step'],2), 'scatter', round(d['kernels_ms_per_step']['scatter'],2), 'dense', round(d['scatter_dense_gradients']['ms'],2), 'dense step', round(d['dense_gradient_step']['ms_per_step'],1), 'peak', round(d['peak_mem_GiB'],1), d['scatter_arena_placement'], 'valid', d['valid'], 'roof', round(d['roofline']['frac'],3), 'refshaped', round(d['reference_shaped_baseline']['ms_per_step']), {k:round(v,1) for k,v in d['variants_ms_per_step'].items()}, 'cpu', d['cpu_baseline'].get('forward_backward_ms'))
PY
