#!/bin/bash
# round 6, GPU call 9: address-translation counters of k_bin_emit on 8 freshly placed arenas (is the placement effect the TLB?)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_9
mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -i "utcl\|tlb\|translation" | head -40 > $O/counters_available.txt; wc -l $O/counters_available.txt; head -30 $O/counters_available.txt
for C in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "TCP_UTCL1_REQUEST_sum TCP_UTCL1_PERMISSION_MISS_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WR_UNCACHED_32B_sum"; do
  T=$(echo $C | tr ' ' '_')
  ( cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-include-regex "k_bin_emit" -f csv -d $GRAFT_REPO_ROOT/$O/pmc_$T -o p -- python $GRAFT_REPO_ROOT/tools/scatter_bimodal.py --placements 8 --libs $GRAFT_REPO_ROOT/make-it-3d_amd/csrc/libmi3d.so --out $GRAFT_REPO_ROOT/$O/placements_$T.json 2>&1 | grep "^{" | cut -c1-160 )
done
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/r06_9/pmc_*/**/*counter_collection.csv', recursive=True)):
    rows=list(csv.DictReader(open(f)))
    by=collections.defaultdict(list)
    for r in rows: by[r['Counter_Name']].append((int(r['Dispatch_Id']), float(r['Counter_Value'])))
    for k,v in by.items():
        v.sort()
        print(k, len(v), [round(x[1]/1e6,2) for x in v][:64])
PY
