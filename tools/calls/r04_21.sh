#!/bin/bash
# round 4, GPU call 21: kernel trace of bench.py's own timed steps at the final kernels (the MLP backward changed after the
# committed trace) and the matrix-core / LDS counters of the MLP kernels in the same command
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04_21
mkdir -p $O
timeout 400 rocprofv3 --kernel-trace -f csv -d $O/prof_trace -o b -- python bench.py --profile-run --steps 5 --warmup 3 > $O/profile_run.json 2> $O/profile_run.err
python tools/trace_sum.py $O/prof_trace --window spin_kernel --steps 5 --out $O/kernel_stats_bench_steps.csv > $O/trace_sum.txt 2>> $O/profile_run.err
find $O/prof_trace -name "*kernel_trace.csv" -delete
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS --kernel-include-regex "k_mlp_" -f csv -d $O/pmc_SQ -o p -- python bench.py --profile-run --steps 3 --warmup 2 > $O/pmc_SQ.json 2> $O/pmc_SQ.err
python - <<'P'
import csv, glob, json, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/r04_21/pmc_SQ/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = "k_mlp_bwd_g" if "k_mlp_bwd_g" in r["Kernel_Name"] else ("k_mlp_fwd_g" if "k_mlp_fwd_g" in r["Kernel_Name"] else r["Kernel_Name"][:40])
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, d in acc.items():
    n = len(next(iter(d.values())))
    # the timed steps are the LAST launches: 3 steps x (2 backward launches | 1 forward launch)
    tail = 6 if "bwd" in k else 3
    m = {c: sum(v[-tail:]) / len(v[-tail:]) for c, v in d.items()}
    out[k] = {"launches_seen": n, "averaged_over_last": tail, "counters_per_launch": m}
    if m.get("GRBM_GUI_ACTIVE"):
        out[k]["mfma_busy_frac"] = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (m["GRBM_GUI_ACTIVE"] / 8.0 * 256 * 4)   # as tools/pmc_summarise.py
    if m.get("SQ_WAVE_CYCLES"):
        out[k]["issuing_share_of_wave_cycles"] = m.get("SQ_ACTIVE_INST_ANY", 0) / m["SQ_WAVE_CYCLES"]
        out[k]["lds_issue_stall_share_of_wave_cycles"] = m.get("SQ_WAIT_INST_LDS", 0) / m["SQ_WAVE_CYCLES"]
    if m.get("SQ_LDS_IDX_ACTIVE"):
        out[k]["lds_conflict_share_of_lds_active"] = m.get("SQ_LDS_BANK_CONFLICT", 0) / m["SQ_LDS_IDX_ACTIVE"]
json.dump(out, open("gpurun_out/r04_21/pmc_mlp_sq.json", "w"), indent=1)
print(json.dumps(out, indent=1))
P
find $O -name "*counter_collection.csv" -size +30M -delete
grep -i "k_mlp\|k_bin" $O/kernel_stats_bench_steps.csv | head; python -c "
import json; b=json.load(open('$O/profile_run.json')); print(b['ms_per_step'], b.get('kernels_ms_per_step'))"
