#!/bin/bash
# round 4, GPU call 13: the whole GPU suite + smoke at HEAD
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04_13
mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
echo "smoke rc=$?" >> $O/smoke.log
tail -4 $O/pytest.log; tail -2 $O/smoke.log
