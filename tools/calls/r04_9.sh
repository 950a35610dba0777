#!/bin/bash
# round 4, GPU call 9: the VAE encoder's forward + backward as captured graphs (A/B)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04_9
mkdir -p $O
timeout 400 python tools/sd_knobs.py --out $O/sd_knobs.json > $O/sd_knobs.out 2> $O/sd_knobs.err
tail -30 $O/sd_knobs.out; tail -5 $O/sd_knobs.err
