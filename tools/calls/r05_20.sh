#!/bin/bash
# round 5, GPU call 20: hashed fine levels with the record's rank taken from the histogram atomic's return value (no cursor
# pass: -DMI3D_RANK=1) against the product, product-grade builds in one process; scatter parity on the variant
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_20
mkdir -p $O
timeout 500 python tools/scatter_ab_libs.py --libs tools/bin/libmi3d_base.so,tools/bin/libmi3d_rank.so --rounds 3 --out $O/scatter_ab_libs.json 2>&1 | tail -30
MI3D_LIB=$PWD/tools/bin/libmi3d_rank.so timeout 300 python -m pytest tests/test_grid_points_gpu.py -q -x 2>&1 | tail -3
