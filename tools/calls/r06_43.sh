#!/bin/bash
# round 6, GPU call 43: at the HEAD - smoke, the whole GPU suite, the driver's bench command at 56 GiB and (fast) at 33 GiB
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_43
mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
S=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
echo "suite wall $(( $(date +%s) - S )) s"
cp gpurun_out/headline_parity.json $O/ 2>/dev/null
S=$(date +%s)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
echo "bench wall $(( $(date +%s) - S )) s"
MI3D_SCATTER_WORKSPACE_GB=33 timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-reference-shaped --variant-steps 0 > $O/bench_33.json 2> $O/bench_33.err
for f in bench bench_33; do
python - <<PY
import json
d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1])
print('$f', round(d['ms_per_step'],2), 'scatter', round(d['kernels_ms_per_step']['scatter'],2), 'dense', round(d['scatter_dense_gradients']['ms'],2), 'dense step', round(d.get('dense_gradient_step',{}).get('ms_per_step',0),1), 'peak', round(d['peak_mem_GiB'],1), d['scatter_arena_placement'], 'valid', d['valid'], 'roof', round(d['roofline']['frac'],3), {k:round(v,1) for k,v in d.get('variants_ms_per_step',{}).items()})
print({k:round(v,2) for k,v in d['kernels_ms_per_step'].items()})
PY
done
