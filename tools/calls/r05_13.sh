#!/bin/bash
# round 5, GPU call 13: product-grade variant builds in ONE process (tools/scatter_ab_libs.py): the product, the product with
# relative register indexing of a tile's gradient pairs (-DMI3D_DYN_IDX=1), the product without the masked fma
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_13
mkdir -p $O
timeout 500 python tools/scatter_ab_libs.py --libs tools/bin/libmi3d_base.so,tools/bin/libmi3d_dyn1.so,tools/bin/libmi3d_nofma.so --rounds 3 --out $O/scatter_ab_libs.json 2>&1 | tail -40
