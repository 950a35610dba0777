#!/bin/bash
# round 4, GPU call 11: kernel traces of the other BASELINE workloads' own timed steps
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04_11
mkdir -p $O
for W in c4_views c2_pruned; do
  timeout 600 rocprofv3 --kernel-trace -f csv -d $O/prof_$W -o b -- python bench.py --workload $W --profile-run --steps 3 --warmup 2 > $O/profile_run_$W.json 2> $O/profile_run_$W.err
  python tools/trace_sum.py $O/prof_$W --window spin_kernel --steps 3 --out $O/kernel_stats_${W}.csv > /dev/null 2>> $O/profile_run_$W.err
  find $O/prof_$W -name "*kernel_trace.csv" -delete
done
head -12 $O/kernel_stats_c4_views.csv; head -8 $O/kernel_stats_c2_pruned.csv
