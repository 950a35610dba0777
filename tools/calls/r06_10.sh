#!/bin/bash
# round 6, GPU call 10: the emit's write requests / stalls per placement, product (tile-major) against the level-major variant
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_10
mkdir -p $O
for C in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_WRITEBACK_sum TCC_REQ_sum"; do
  T=$(echo $C | tr ' ' '_')
  ( cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-include-regex "k_bin_emit" -f csv -d $GRAFT_REPO_ROOT/$O/pmc_$T -o p -- python $GRAFT_REPO_ROOT/tools/scatter_bimodal.py --placements 4 --libs $GRAFT_REPO_ROOT/make-it-3d_amd/csrc/libmi3d.so,$GRAFT_REPO_ROOT/tools/bin/libmi3d_level_major.so --out $GRAFT_REPO_ROOT/$O/placements_$T.json 2>&1 | grep "^{" | cut -c1-200 )
done
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/r06_10/pmc_*/**/*counter_collection.csv', recursive=True)):
    rows=list(csv.DictReader(open(f)))
    by=collections.defaultdict(list)
    for r in rows: by[r['Counter_Name']].append((int(r['Dispatch_Id']), float(r['Counter_Value'])))
    for k,v in by.items():
        v.sort()
        # per placement: 4 calls x 2 slices for lib A, then the same for lib B
        vals=[x[1]/1e6 for x in v]
        print(k, len(vals), 'per placement [tile-major mean, level-major mean]:', [(round(sum(vals[i*16:i*16+8])/8,1), round(sum(vals[i*16+8:i*16+16])/8,1)) for i in range(len(vals)//16)])
PY
