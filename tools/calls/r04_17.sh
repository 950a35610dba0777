#!/bin/bash
# round 4, GPU call 17: SQ counters of the emit / reduce at the final kernels (dense and real census)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04_17
mkdir -p $O
for C in dense real; do
  X=""; [ $C = real ] && X="--real-census"
  timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU --kernel-include-regex "k_bin_" -f csv -d $O/pmc_$C -o p -- python tools/kbench.py --what scatter13 --half-planes $X --iters 3 --out $O/k_$C.json > /dev/null 2> $O/pmc_$C.err
done
python - <<'P'
import csv, glob, json, collections
out = {}
for c in ("dense", "real"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/r04_17/pmc_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = "k_bin_emit" if "k_bin_emit" in r["Kernel_Name"] else "k_bin_reduce"
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        m = {n: sum(v) / len(v) for n, v in d.items()}
        wc = m.get("SQ_WAVE_CYCLES", 0) or 1
        out[f"{c}:{k}"] = {"launches": len(next(iter(d.values()))), "SQ_WAVE_CYCLES_per_launch": m.get("SQ_WAVE_CYCLES"),
                           "per_wave_cycle": {n: round(v / wc, 4) for n, v in m.items() if n != "SQ_WAVE_CYCLES"}}
json.dump(out, open("gpurun_out/r04_17/pmc_emit_reduce.json", "w"), indent=1)
print(json.dumps(out, indent=1))
P
