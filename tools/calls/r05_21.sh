#!/bin/bash
# round 5, GPU call 21: the MLP backward with the 4-wide output gradient's tile turned round through LDS too
# (-DMI3D_MLP_DO_LDS=1: 42 MFMAs and 72 packed converts per tile instead of 43 / 80) against the product: every template
# instance's outputs and the timings at the headline size, both libraries in one process (tools/mlp_ab.py)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_21
mkdir -p $O
timeout 400 python tools/mlp_ab.py --libs tools/bin/libmi3d_base.so,tools/bin/libmi3d_do_lds.so --out $O/mlp_ab.json 2>&1 | tail -8 | cut -c1-600
