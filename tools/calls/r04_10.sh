#!/bin/bash
# round 4, GPU call 10: the new full-size parity tests (C4 view, eval loop at C2 size), emitting-wave counts dense / real
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04_10
mkdir -p $O
timeout 900 python -m pytest tests/test_headline_parity_gpu.py -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
cp gpurun_out/headline_parity.json $O/ 2>/dev/null
timeout 600 python tools/kbench.py --what scatter_ab --half-planes --iters 3 --out $O/kbench_scatter_ab.json > /dev/null 2> $O/kbench.err
tail -6 $O/pytest.log
