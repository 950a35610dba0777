#!/bin/bash
# round 5, GPU call 2: the whole GPU suite (no -x: every failure at once) after the round counter fix
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_2
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -25 $O/pytest.log
cp gpurun_out/headline_parity.json $O/ 2>/dev/null
