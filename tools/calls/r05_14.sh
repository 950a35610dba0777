#!/bin/bash
# round 5, GPU call 14: the final kernels (relative register indexing in the emit): whole GPU suite, smoke, the driver's bench
# command, the kernel trace and the counter passes of bench.py's own timed steps, SQ counters of the emit / reduce
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_14
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
RX='k_bin_|k_grid_|k_mlp_|k_head_|k_march_|k_composite_|k_adan|k_sumsq'
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --kernel-include-regex "$RX" -f csv -d $O/pmc_$C -o p -- python bench.py --profile-run --steps 3 --warmup 2 > $O/pmc_$C.json 2> $O/pmc_$C.err
done
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-include-regex "$RX" -f csv -d $O/pmc_SQ -o p -- python bench.py --profile-run --steps 3 --warmup 2 > $O/pmc_SQ.json 2> $O/pmc_SQ.err
EV=$(python -c "import json;d=json.load(open('$O/profile_run.json'));import re;print(int(re.search(r'(\d+) samples/view',d['config']['workload']).group(1))*13)")
TT=$(python -c "import json;d=json.load(open('$O/pmc_FETCH_SIZE.json'));print(d['config']['steps_run_total'])")
python tools/pmc_summarise.py $O/pmc_r05.json c2_dense $EV --tail 3/$TT $(find $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ -name "*counter_collection.csv") > $O/pmc_summary.txt 2>&1
find $O -name "*counter_collection.csv" -size +30M -delete
for C in dense real; do
  X=""; [ $C = real ] && X="--real-census"
  timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU --kernel-include-regex "k_bin_" -f csv -d $O/pmcsq_$C -o p -- python tools/kbench.py --what scatter13 --half-planes $X --iters 3 --out $O/k_$C.json > /dev/null 2> $O/pmcsq_$C.err
done
python - <<'P'
import csv, glob, json, collections
out = {}
for c in ("dense", "real"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/r05_14/pmcsq_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = "k_bin_emit" if "k_bin_emit" in r["Kernel_Name"] else "k_bin_reduce"
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        m = {n: sum(v) / len(v) for n, v in d.items()}
        wc = m.get("SQ_WAVE_CYCLES", 0) or 1
        out[f"{c}:{k}"] = {"launches": len(next(iter(d.values()))), "SQ_WAVE_CYCLES_per_launch": m.get("SQ_WAVE_CYCLES"),
                           "SQ_INSTS_VALU_per_launch": m.get("SQ_INSTS_VALU"), "SQ_INSTS_LDS_per_launch": m.get("SQ_INSTS_LDS"),
                           "per_wave_cycle": {n: round(v / wc, 4) for n, v in m.items() if n != "SQ_WAVE_CYCLES"}}
json.dump(out, open("gpurun_out/r05_14/pmc_emit_reduce.json", "w"), indent=1)
for k, v in out.items():
    print(k, v["SQ_WAVE_CYCLES_per_launch"], v["SQ_INSTS_VALU_per_launch"], v["per_wave_cycle"])
P
find $O -name "*counter_collection.csv" -size +30M -delete
grep -i "k_mlp\|k_bin\|k_grid" $O/kernel_stats_bench_steps.csv | head; tail -4 $O/pmc_summary.txt; python - <<'P'
import json
b = json.load(open("gpurun_out/r05_14/bench_c2_dense.json"))
print(b["ms_per_step"], b["kernels_ms_per_step"], b.get("valid"), b["roofline"]["frac"], b.get("clocks"))
print(b["variants_ms_per_step"]); print(b["scatter_dense_gradients"]["ms"], b["config"]["steps_run_total"])
p = json.load(open("gpurun_out/r05_14/profile_run.json")); print(p["ms_per_step"], p["kernels_ms_per_step"])
P
