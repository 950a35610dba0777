#!/bin/bash
# round 6, GPU call 34: the reduce's record loads software-pipelined (next batch issued before this batch's LDS atomics), batches
# of 4 (product) / 2 / 8 records per lane, against the unpipelined loop; product-grade builds in one process
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_34
mkdir -p $O
LIBS=make-it-3d_amd/csrc/libmi3d.so,tools/bin/libmi3d_nopipe.so,tools/bin/libmi3d_pipe_u2.so,tools/bin/libmi3d_pipe_u8.so
timeout 1200 python tools/scatter_ab_libs.py --libs $LIBS --rounds 3 --capture 8 --out $O/scatter_ab_libs_reduce_pipe.json > $O/log.txt 2>&1
python - <<PY
import json
d=json.load(open('$O/scatter_ab_libs_reduce_pipe.json'))
for c in ('dense_ms','real_ms','captured_ms'):
    print(c, {k.replace('libmi3d','').replace('.so',''):round(min(v),2) for k,v in d[c].items()})
print({k:v for k,v in d.items() if 'err' in k})
PY
tail -2 $O/log.txt
