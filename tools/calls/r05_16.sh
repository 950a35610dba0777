#!/bin/bash
# round 5, GPU call 16: the final HEAD (the reduce's loads in flight, 4 per lane): whole GPU suite, smoke, the driver's bench
# command, the kernel trace of bench.py's own timed steps
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_16
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
cp gpurun_out/headline_parity.json $O/ 2>/dev/null
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
echo "smoke rc=$?" >> $O/smoke.log; tail -2 $O/smoke.log
timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c2_dense.json 2> $O/bench.err
echo "bench rc=$?" >> $O/bench.err; tail -2 $O/bench.err
timeout 600 rocprofv3 --kernel-trace -f csv -d $O/prof_trace -o b -- python bench.py --profile-run --steps 5 --warmup 3 > $O/profile_run.json 2> $O/profile_run.err
python tools/trace_sum.py $O/prof_trace --window spin_kernel --steps 5 --out $O/kernel_stats_bench_steps.csv > $O/trace_sum.txt 2>> $O/profile_run.err
find $O/prof_trace -name "*kernel_trace.csv" -delete
grep -i "k_mlp\|k_bin\|k_grid" $O/kernel_stats_bench_steps.csv | head; python - <<'P'
import json
b = json.load(open("gpurun_out/r05_16/bench_c2_dense.json"))
print(b["ms_per_step"], b["kernels_ms_per_step"], b.get("valid"), b["roofline"]["frac"], b.get("clocks"))
print(b["variants_ms_per_step"]); print(b["scatter_dense_gradients"]["ms"], b["config"]["steps_run_total"])
p = json.load(open("gpurun_out/r05_16/profile_run.json")); print(p["ms_per_step"], p["kernels_ms_per_step"])
P
