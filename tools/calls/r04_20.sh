#!/bin/bash
# round 4, GPU call 20: the MLP backward after its instruction diet (whole-register sums / masks; an instance for the full
# input width with one lane address per tile; the next tile's rows requested late).  A/B of the variants in one process
# against the round-3 form (call 19 found a wrong bias-gradient sum that way: __builtin_bit_cast of a vector element); then the
# product library - full-width instance + late prefetch - runs the whole suite, smoke and the driver's bench command.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04_20
mkdir -p $O
B=tools/bin
timeout 500 python tools/mlp_ab.py --libs $B/libmi3d_dev_tr0.so,$B/libmi3d_dev_tr1_head.so,$B/libmi3d_dev_f0l0.so,$B/libmi3d_dev_f1l1.so --out $O/mlp_ab.json > $O/mlp_ab.log 2>&1
rc=$?
echo "mlp_ab rc=$rc" >> $O/mlp_ab.log
tail -4 $O/mlp_ab.log
if [ $rc -ne 0 ]; then echo "A/B not clean: stopping here"; grep -v '"dx_equal": true' $O/mlp_ab.log | head; exit 1; fi
timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1
prc=$?
echo "pytest rc=$prc" >> $O/pytest.log
tail -4 $O/pytest.log
if [ $prc -ne 0 ]; then grep -n "Error\|FAILED\|assert" $O/pytest.log | head -20; exit 1; fi
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
echo "smoke rc=$?" >> $O/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c2_dense.json 2> $O/bench.err
echo "bench rc=$?" >> $O/bench.err
tail -2 $O/smoke.log; tail -3 $O/bench.err; python - <<'P'
import json
b = json.load(open("gpurun_out/r04_20/bench_c2_dense.json"))
print(b["ms_per_step"], b.get("kernels_ms_per_step"), b.get("valid"))
P
