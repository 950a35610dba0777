#!/bin/bash
# round 4, GPU call 7: WRITE_SIZE calibration for the emit's store patterns, the other BASELINE workloads, eval loop
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04_7
mkdir -p $O
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d $O/wcal -o w -- tools/bin/write_calib > $O/write_calib.txt 2> $O/write_calib.err
python - <<'P' >> $O/write_calib.txt 2>&1
import csv, glob
rows = {}
for f in glob.glob("gpurun_out/r04_7/wcal/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "WRITE_SIZE":
            rows.setdefault(r["Kernel_Name"].split("(")[0], []).append(float(r["Counter_Value"]))
for k, v in rows.items():
    print(f"WRITE_SIZE {k}: {sum(v)/len(v):.1f} KB per launch ({len(v)} launch(es))")
P
for W in c2_pruned c4_views c5_refine; do
  timeout 600 python bench.py --workload $W --steps 10 --warmup 3 --variant-steps 0 --no-cpu-baseline --no-reference-shaped > $O/bench_$W.json 2> $O/bench_$W.err
done
timeout 600 python tools/eval_bench.py --out $O/eval_bench.json > /dev/null 2> $O/eval_bench.err
cat $O/write_calib.txt
