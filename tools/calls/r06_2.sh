#!/bin/bash
# round 6, GPU call 2: the whole GPU suite on the new build (ABI 5, albedo rounding, autopatch, ADVICE fixes); the dense
# scatter back to back for 40 s with clocks (item 1d); the reference-shaped route's weight-gradient GEMMs under torch's two
# BLAS back ends and the route itself with rocBLAS preferred; the driver's bench command, timed.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_2
mkdir -p $O
S=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
echo "suite wall $(( $(date +%s) - S )) s"
cp gpurun_out/headline_parity.json $O/ 2>/dev/null
timeout 300 python tools/scatter_bimodal.py --rounds 3 --sustain-s 40 --out $O/scatter_bimodal.json 2>&1 | tail -8
timeout 300 python tools/wgrad_gemm_probe.py --out $O/wgrad_gemm_probe.json 2>&1 | tail -40
TORCH_BLAS_PREFER_HIPBLASLT=0 timeout 300 python bench.py --reference-shaped-only --steps 2 2> $O/refshaped_rocblas.err | tail -c 400 | tee $O/refshaped_rocblas.json
S=$(date +%s)
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
echo "bench wall $(( $(date +%s) - S )) s"
grep "bench " $O/bench.err | tail -20
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_2/bench.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ('value','ms_per_step','valid','peak_mem_GiB','variants_ms_per_step','kernels_ms_per_step')})
print('ref-shaped', d.get('reference_shaped_baseline',{}).get('ms_per_step'), 'dense', d.get('scatter_dense_gradients',{}).get('ms'), d.get('dense_gradient_step'), 'cpu', {k:v for k,v in d.get('cpu_baseline',{}).items() if k in ('value','forward_ms','forward_backward_ms','cores')})
PY
