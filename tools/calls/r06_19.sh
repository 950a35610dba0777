#!/bin/bash
# round 6, GPU call 19: padding behind the fine levels' blocks of the arena (their 4 GiB + 24 MB stride against the 4 GiB
# period of the placement effect): product and three paddings over the same 1 GiB shift sweep
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_19
mkdir -p $O
timeout 1200 python tools/scatter_bimodal.py --shift 12 --shift-gib 1 --libs make-it-3d_amd/csrc/libmi3d.so,tools/bin/libmi3d_pad64.so,tools/bin/libmi3d_pad455.so,tools/bin/libmi3d_pad1100.so --out $O/scatter_shift_pads.json 2>&1 | grep "^{" | sed 's/libmi3d//g; s/.so//g'
