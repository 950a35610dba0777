#!/bin/bash
# round 4, GPU call 8: the whole GPU suite, smoke and the driver's bench command at HEAD
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04_8
mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
echo "smoke rc=$?" >> $O/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c2_dense.json 2> $O/bench.err
echo "bench rc=$?" >> $O/bench.err
tail -4 $O/pytest.log; tail -2 $O/smoke.log; tail -3 $O/bench.err
