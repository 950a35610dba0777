#!/bin/bash
# round 6, GPU call 30: tiles claimed from a counter against tiles dealt statically, x coarse waves, product-grade builds in one
# process, on synthetic censuses AND on the captured gradient planes of a real step; 56 GiB (2 slices) and 33 GiB (3 slices)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_30
mkdir -p $O
LIBS=make-it-3d_amd/csrc/libmi3d.so,tools/bin/libmi3d_static3072.so,tools/bin/libmi3d_claim1536.so,tools/bin/libmi3d_claim16384.so,tools/bin/libmi3d_static16384.so
for GB in 56 33; do
  MI3D_SCATTER_WORKSPACE_GB=$GB timeout 1200 python tools/scatter_ab_libs.py --libs $LIBS --rounds 3 --capture 8 --out $O/scatter_ab_libs_claim_${GB}GiB.json > $O/log_$GB.txt 2>&1
  python - <<PY
import json
d=json.load(open('$O/scatter_ab_libs_claim_${GB}GiB.json'))
print('$GB GiB', d.get('captured'))
for c in ('dense_ms','real_ms','captured_ms'):
    print(c, {k.replace('libmi3d','').replace('.so',''):round(min(v),2) for k,v in d[c].items()})
print({k:v for k,v in d.items() if 'err' in k})
PY
  tail -2 $O/log_$GB.txt
done
