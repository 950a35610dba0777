#!/bin/bash
# round 5, GPU call 22: the whole GPU suite, smoke and the driver's bench command at the final HEAD (MLP backward: the output
# gradient's tile through LDS)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_22
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
cp gpurun_out/headline_parity.json $O/ 2>/dev/null
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
echo "smoke rc=$?" >> $O/smoke.log; tail -2 $O/smoke.log
timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c2_dense.json 2> $O/bench.err
echo "bench rc=$?" >> $O/bench.err; tail -2 $O/bench.err
python - <<'P'
import json
b = json.load(open("gpurun_out/r05_22/bench_c2_dense.json"))
print(b["ms_per_step"], b["kernels_ms_per_step"], b.get("valid"), b["roofline"]["frac"], b.get("clocks"), b["config"]["steps_run_total"])
print(b["variants_ms_per_step"]); print(b["scatter_dense_gradients"]["ms"])
P
