#!/bin/bash
# round 6, GPU call 14: what limits the gather on the fine hashed levels: the real level against a
# probe that issues its exact address stream without the arithmetic (tools/gather_probe.{hip,py}), then the same command
# under two --pmc passes (TCP -> TCC read requests and pending-stall cycles; TCC hits / misses)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_14
mkdir -p $O
timeout 600 python tools/gather_probe.py --levels 7,9,11,13,15 --out $O/gather_probe.json 2>&1 | grep -v amdgpu.ids | tail -8
for C in "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN2_sum"; do
  T=$(echo $C | tr ' ' '_')
  ( cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-include-regex "k_grid_encode_planes|k_probe" -f csv -d $GRAFT_REPO_ROOT/$O/pmc_$T -o p -- python $GRAFT_REPO_ROOT/tools/gather_probe.py --levels 11,15 --wgs-per-cu 3 --out $GRAFT_REPO_ROOT/$O/gp_$T.json > /dev/null 2>&1 )
done
python - <<'PY'
import csv, glob, collections, json
out={}
for f in sorted(glob.glob('gpurun_out/r06_14/pmc_*/**/*counter_collection.csv', recursive=True)):
    rows=list(csv.DictReader(open(f)))
    by=collections.defaultdict(list)
    for r in rows:
        k='probe' if 'k_probe' in r['Kernel_Name'] else ('real_lds' if 'planes_lds' in r['Kernel_Name'] else 'real')
        by[(r['Counter_Name'],k)].append((int(r['Dispatch_Id']), float(r['Counter_Value'])))
    for (c,k),v in sorted(by.items()):
        v.sort()
        out[f"{c}:{k}"]=[round(x[1]/1e6,2) for x in v]
        print(c,k,len(v),[round(x[1]/1e6,1) for x in v][:40])
json.dump(out,open('gpurun_out/r06_14/gather_probe_pmc.json','w'),indent=1)
PY
