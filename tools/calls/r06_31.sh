#!/bin/bash
# round 6, GPU call 31: claimed tiles - how many waves per role (fine 1280 / 1536 / 1792; coarse 2304 / 3072 / 4608 / 6144),
# product-grade builds in one process, synthetic censuses and a captured real step
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_31
mkdir -p $O
LIBS=make-it-3d_amd/csrc/libmi3d.so,tools/bin/libmi3d_claim_fw1280.so,tools/bin/libmi3d_claim_fw1792.so,tools/bin/libmi3d_claim_cw2304.so,tools/bin/libmi3d_claim_cw4608.so,tools/bin/libmi3d_claim_cw6144.so
for GB in 56; do
  MI3D_SCATTER_WORKSPACE_GB=$GB timeout 1200 python tools/scatter_ab_libs.py --libs $LIBS --rounds 3 --capture 8 --out $O/scatter_ab_libs_claim_waves_${GB}GiB.json > $O/log_$GB.txt 2>&1
  python - <<PY
import json
d=json.load(open('$O/scatter_ab_libs_claim_waves_${GB}GiB.json'))
print('$GB GiB', d.get('captured'))
for c in ('dense_ms','real_ms','captured_ms'):
    print(c, {k.replace('libmi3d','').replace('.so',''):round(min(v),2) for k,v in d[c].items()})
PY
  tail -2 $O/log_$GB.txt
done
