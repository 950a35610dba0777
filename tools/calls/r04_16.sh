#!/bin/bash
# round 4, GPU call 16: does the scatter's time depend on where its arena lies?  (twice, two processes)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04_16
mkdir -p $O
for r in 1 2; do timeout 400 python tools/arena_probe.py --out $O/arena_probe_$r.json > $O/probe_$r.out 2> $O/probe_$r.err; done
cat $O/probe_1.out; cat $O/probe_2.out | head -30; tail -3 $O/probe_1.err
