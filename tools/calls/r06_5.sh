#!/bin/bash
# round 6, GPU call 5: the dense scatter with explicitly held record arenas (item 1d: placement or process state?), three
# processes in a row on one box
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_5
mkdir -p $O
for i in 1 2 3; do timeout 300 python tools/scatter_bimodal.py --arenas --out $O/scatter_arenas_$i.json 2>&1 | grep "^('W"; echo; done
