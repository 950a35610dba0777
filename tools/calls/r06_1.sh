#!/bin/bash
# round 6, GPU call 1: smoke; kernel trace of ONE reference-shaped step (where do its 1.8 s go);
# the headline parity tests with the new autocast CPU-oracle subset and the binary16-rounded albedo; the dense scatter's
# bimodality (item 1d) under a kernel trace; the driver's bench command with the shortened extras (item 7), timed.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_1
mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/trace_refshaped -o rs --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --reference-shaped-only --steps 1 > $GRAFT_REPO_ROOT/$O/refshaped.json 2> $GRAFT_REPO_ROOT/$O/refshaped.err )
tail -c 600 $O/refshaped.json; tail -3 $O/refshaped.err
python tools/trace_sum.py $O/trace_refshaped --window spin_kernel --steps 1 --out $O/kernel_stats_r06_reference_shaped.csv 2>&1 | head -40
timeout 900 python -m pytest tests/test_headline_parity_gpu.py -q -x -k "autocast or fp32" 2>&1 | tail -5
cp gpurun_out/headline_parity.json $O/ 2>/dev/null
( cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/trace_bimodal -o bm --output-format csv -- python $GRAFT_REPO_ROOT/tools/scatter_bimodal.py --rounds 6 --out $GRAFT_REPO_ROOT/$O/scatter_bimodal.json 2>&1 | tail -6 )
python tools/scatter_bimodal.py --per-dispatch $O/trace_bimodal > $O/scatter_bimodal_dispatches.json 2>&1; head -c 1500 $O/scatter_bimodal_dispatches.json
rm -rf $O/trace_bimodal $O/trace_refshaped/*/*agent* 2>/dev/null
S=$(date +%s)
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
echo "bench wall $(( $(date +%s) - S )) s"
grep "bench " $O/bench.err | tail -20
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_1/bench.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ('value','ms_per_step','valid','peak_mem_GiB','variants_ms_per_step','kernels_ms_per_step')})
print('ref-shaped', d.get('reference_shaped_baseline',{}).get('ms_per_step'), 'dense', d.get('scatter_dense_gradients',{}).get('ms'), 'cpu', {k:v for k,v in d.get('cpu_baseline',{}).items() if k in ('value','forward_ms','forward_backward_ms','cores')})
PY
