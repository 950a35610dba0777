#!/bin/bash
# round 6, GPU call 15: the new arena test; the emit with perfectly sequential stores (timing-only build MI3D_TIMING_SEQ_FLUSH:
# the ceiling of what re-organising the stores could buy), product-grade builds in one process, under a kernel trace
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_15
mkdir -p $O
timeout 300 python -m pytest tests/test_grid_points_gpu.py -q -x -k "persistent_placed" 2>&1 | tail -15
( cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/trace -o t --output-format csv -- python $GRAFT_REPO_ROOT/tools/scatter_ab_libs.py --libs $GRAFT_REPO_ROOT/make-it-3d_amd/csrc/libmi3d.so,$GRAFT_REPO_ROOT/tools/bin/libmi3d_seq_flush.so --rounds 3 --out $GRAFT_REPO_ROOT/$O/scatter_ab_libs_seq_flush.json 2>&1 | grep -A12 dense_ms | head -40 )
python tools/scatter_bimodal.py --per-dispatch $O/trace > $O/seq_flush_dispatches.json 2>&1
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06_15/seq_flush_dispatches.json'))
e=d['emit_ms']; r=d['reduce_ms']
print('emit dispatches', len(e)); print([round(x,2) for x in e]); print([round(x,2) for x in r])
PY
rm -rf $O/trace
