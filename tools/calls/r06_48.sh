#!/bin/bash
# round 6, GPU call 48: at the shipped defaults (33 GiB arena, 3 slices, in-place gradient planes) - smoke, the whole GPU suite,
# the driver's bench command, then the kernel trace and the PMC passes of bench.py's own timed steps
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_48
mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
S=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
echo "suite wall $(( $(date +%s) - S )) s"
cp gpurun_out/headline_parity.json $O/ 2>/dev/null
S=$(date +%s)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
echo "bench wall $(( $(date +%s) - S )) s"
python - <<PY
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print(round(d['ms_per_step'],2), 'scatter', round(d['kernels_ms_per_step']['scatter'],2), 'dense', round(d['scatter_dense_gradients']['ms'],2), 'dense step', round(d['dense_gradient_step']['ms_per_step'],1), 'peak', round(d['peak_mem_GiB'],1), d['scatter_arena_placement'], 'valid', d['valid'], 'roof', round(d['roofline']['frac'],3), 'refshaped', round(d['reference_shaped_baseline']['ms_per_step']), {k:round(v,1) for k,v in d['variants_ms_per_step'].items()}, 'cpu', d['cpu_baseline'].get('forward_backward_ms'))
print({k:round(v,2) for k,v in d['kernels_ms_per_step'].items()})
PY
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
head -12 $O/trace_sum.txt
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06_48/pmc_r06.json'))
for k in ('k_bin_emit','k_bin_reduce','k_grid_encode_planes','k_mlp_backward','k_mlp_forward','scatter_binned'):
    w=d[k].get('c2_dense',{})
    print(k, {kk:(round(vv,3) if isinstance(vv,float) else vv) for kk,vv in w.items() if kk in ('hbm_bytes_per_eval','mfma_busy_frac','lds_conflict_frac','fetch_bytes_per_launch_corrected_x2','write_bytes_per_launch')})
d=json.loads(open('gpurun_out/r06_48/profile_run.json').read().strip().splitlines()[-1])
print(round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['kernels_ms_per_step'].items()}, round(d['roofline']['frac'],3), d['peak_mem_GiB'])
PY
