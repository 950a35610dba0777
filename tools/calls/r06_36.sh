#!/bin/bash
# round 6, GPU call 36: the kernel trace and the PMC passes of bench.py's OWN timed steps at the claimed-tile emit
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_36
mkdir -p $O
timeout 600 rocprofv3 --kernel-trace -f csv -d $O/prof_trace -o b -- python bench.py --profile-run --steps 5 --warmup 3 > $O/profile_run.json 2> $O/profile_run.err
python tools/trace_sum.py $O/prof_trace --window spin_kernel --steps 5 --out $O/kernel_stats_bench_steps.csv > $O/trace_sum.txt 2>> $O/profile_run.err
find $O/prof_trace -name "*kernel_trace.csv" -delete
export MI3D_SCATTER_PLACEMENT_TRIALS=1
RX='k_bin_|k_grid_|k_mlp_|k_head_|k_march_|k_composite_|k_adan|k_sumsq'
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --kernel-include-regex "$RX" -f csv -d $O/pmc_$C -o p -- python bench.py --profile-run --steps 3 --warmup 2 > $O/pmc_$C.json 2> $O/pmc_$C.err
done
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-include-regex "$RX" -f csv -d $O/pmc_SQ -o p -- python bench.py --profile-run --steps 3 --warmup 2 > $O/pmc_SQ.json 2> $O/pmc_SQ.err
EV=$(python -c "import json;d=json.load(open('$O/pmc_SQ.json'));import re;print(int(re.search(r'(\d+) samples/view',d['config']['workload']).group(1))*13)")
TT=$(python -c "import json;d=json.load(open('$O/pmc_FETCH_SIZE.json'));print(d['config']['steps_run_total'])")
python tools/pmc_summarise.py $O/pmc_r06.json c2_dense $EV --tail 3/$TT $(find $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ -name "*counter_collection.csv") > $O/pmc_summary.txt 2>&1
find $O -name "*counter_collection.csv" -size +30M -delete
head -30 $O/trace_sum.txt
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06_36/pmc_r06.json'))
for k,v in d.items():
    w=v.get('c2_dense',{})
    print(k, {kk:(round(vv,3) if isinstance(vv,float) else vv) for kk,vv in w.items() if kk in ('hbm_bytes_per_eval','mfma_busy_frac','lds_conflict_frac','fetch_bytes_per_launch_corrected_x2','write_bytes_per_launch')})
PY
python -c "
import json
d=json.loads(open('$O/profile_run.json').read().strip().splitlines()[-1])
print(round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['kernels_ms_per_step'].items()}, d.get('roofline'))"
