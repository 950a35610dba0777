#!/bin/bash
# round 4, GPU call 1: the new headline-mode parity tests, a bench line with the new variants / census, and the
# kernel trace of bench.py's OWN timed steps
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r04_1
mkdir -p $O
timeout 1500 python -m pytest tests/test_headline_parity_gpu.py tests/test_grid_points_gpu.py -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
cp gpurun_out/headline_parity.json $O/ 2>/dev/null
timeout 900 python bench.py --steps 5 --warmup 2 --variant-steps 2 --no-cpu-baseline --no-reference-shaped > $O/bench.json 2> $O/bench.err
echo "bench rc=$?" >> $O/bench.err
timeout 900 rocprofv3 --kernel-trace -f csv -d $O/prof -o b -- python bench.py --profile-run --steps 3 --warmup 2 > $O/profile_run.json 2> $O/profile_run.err
echo "profile rc=$?" >> $O/profile_run.err
python tools/trace_sum.py $O/prof --window spin_kernel --steps 3 --out $O/kernel_stats_bench_steps.csv > /dev/null 2>> $O/profile_run.err
find $O/prof -name "*kernel_trace.csv" -size +20M -delete
tail -5 $O/pytest.log
