#!/bin/bash
# round 6, GPU call 4: the dense scatter with / without / with an all-zero deferred pair, before and after SD stand-in
# steps in the same process, under a kernel trace (item 1d: which difference is the code path, which the process)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_4
mkdir -p $O
( cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/trace -o mx --output-format csv -- python $GRAFT_REPO_ROOT/tools/scatter_bimodal.py --matrix --out $GRAFT_REPO_ROOT/$O/scatter_matrix.json 2>&1 | grep -v "^W\|^E" | tail -5 )
python tools/scatter_bimodal.py --per-dispatch $O/trace > $O/scatter_matrix_dispatches.json 2>&1
rm -rf $O/trace
timeout 300 python tools/scatter_bimodal.py --matrix --out $O/scatter_matrix_untraced.json 2>&1 | tail -3
