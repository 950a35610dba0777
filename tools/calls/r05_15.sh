#!/bin/bash
# round 5, GPU call 15: the reduce with its eight record loads really in flight (unconditional loads from a clamped index)
# against the product of call 14, product-grade builds in one process
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_15
mkdir -p $O
timeout 500 python tools/scatter_ab_libs.py --libs tools/bin/libmi3d_base.so,tools/bin/libmi3d_reduce_loads.so,tools/bin/libmi3d_reduce_u16.so,tools/bin/libmi3d_reduce_u4.so --rounds 3 --out $O/scatter_ab_libs.json 2>&1 | tail -30
