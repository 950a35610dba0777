#!/bin/bash
# round 6, GPU call 16: the kernel trace and the PMC passes of bench.py's OWN timed steps at round 6 (placed arena;
# group flush), SQ counters of the emit / reduce (is the emit bound by VALU issue?), clocks beside the trace run
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_16
mkdir -p $O
timeout 600 rocprofv3 --kernel-trace -f csv -d $O/prof_trace -o b -- python bench.py --profile-run --steps 5 --warmup 3 > $O/profile_run.json 2> $O/profile_run.err
python tools/trace_sum.py $O/prof_trace --window spin_kernel --steps 5 --out $O/kernel_stats_bench_steps.csv > $O/trace_sum.txt 2>> $O/profile_run.err
find $O/prof_trace -name "*kernel_trace.csv" -delete
RX='k_bin_|k_grid_|k_mlp_|k_head_|k_march_|k_composite_|k_adan|k_sumsq'
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --kernel-include-regex "$RX" -f csv -d $O/pmc_$C -o p -- python bench.py --profile-run --steps 3 --warmup 2 > $O/pmc_$C.json 2> $O/pmc_$C.err
done
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-include-regex "$RX" -f csv -d $O/pmc_SQ -o p -- python bench.py --profile-run --steps 3 --warmup 2 > $O/pmc_SQ.json 2> $O/pmc_SQ.err
EV=$(python -c "import json;d=json.load(open('$O/profile_run.json'));import re;print(int(re.search(r'(\d+) samples/view',d['config']['workload']).group(1))*13)")
TT=$(python -c "import json;d=json.load(open('$O/pmc_FETCH_SIZE.json'));print(d['config']['steps_run_total'])")
python tools/pmc_summarise.py $O/pmc_r06.json c2_dense $EV --tail 3/$TT $(find $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ -name "*counter_collection.csv") > $O/pmc_summary.txt 2>&1
find $O -name "*counter_collection.csv" -size +30M -delete
cat $O/trace_sum.txt | head -30; tail -5 $O/pmc_summary.txt
