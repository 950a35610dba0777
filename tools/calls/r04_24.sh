#!/bin/bash
# round 4, GPU call 24: what groups of the MLP backward's vector instructions cost (timing-only builds:
#   python tools/build_dev.py tools/bin/libmi3d_dev_cut1.so -DMI3D_MLP_BWD_TIMING_CUT=1   # no bias sums
#   python tools/build_dev.py tools/bin/libmi3d_dev_cut3.so -DMI3D_MLP_BWD_TIMING_CUT=3   # ... and no gradient masks)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04_24
B=tools/bin
timeout 70 python tools/mlp_ab.py --timing-only --libs $B/libmi3d_dev_f1l1.so,$B/libmi3d_dev_cut1.so,$B/libmi3d_dev_cut3.so,$B/libmi3d_dev_tr0.so --out gpurun_out/r04_24/mlp_cuts.json 2>&1 | tail -4
