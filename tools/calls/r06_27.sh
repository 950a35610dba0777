#!/bin/bash
# round 6, GPU call 27: emitting waves of the coarse role, the small end (4096 / 3072 / 2048 / 1536 / 1024 / 512 against the
# product's 16384), product-grade builds, on a placed arena of 56 GiB (2 slices) and of 30 GiB (4 slices)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_27
mkdir -p $O
LIBS=make-it-3d_amd/csrc/libmi3d.so,tools/bin/libmi3d_cw4096.so,tools/bin/libmi3d_cw3072.so,tools/bin/libmi3d_cw2048.so,tools/bin/libmi3d_cw1536.so,tools/bin/libmi3d_cw1024.so,tools/bin/libmi3d_cw512.so
for GB in 56 30; do
  MI3D_SCATTER_WORKSPACE_GB=$GB timeout 900 python tools/scatter_ab_libs.py --libs $LIBS --rounds 3 --out $O/scatter_ab_libs_coarse_waves_${GB}GiB.json > $O/log_$GB.txt 2>&1
  python - <<PY
import json
d=json.load(open('$O/scatter_ab_libs_coarse_waves_${GB}GiB.json'))
print('$GB GiB')
for c in ('dense_ms','real_ms'):
    print(c, {k.replace('libmi3d','').replace('.so',''):round(min(v),2) for k,v in d[c].items()})
PY
done
