#!/bin/bash
# round 6, GPU call 6: which conditioning of the allocator gives the record arena its fastest placement (item 1d)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_6
mkdir -p $O
for i in 1 2; do timeout 600 python tools/scatter_bimodal.py --recipes --out $O/scatter_recipes_$i.json 2>&1 | grep "^('"; echo; done
