#!/bin/bash
# round 4, GPU call 18: the MLP backward with its tile transposes through LDS (ds_read_b64_tr_b16) instead of identity
# products.  Probe of the instruction, A/B against the round-3 form in one process, then - only if both are clean - the
# whole GPU suite, smoke and the driver's bench command on the product library built that way.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04_18
mkdir -p $O
timeout 60 tools/bin/tr_probe > $O/tr_probe.txt 2>&1
echo "probe rc=$?" >> $O/tr_probe.txt
cat $O/tr_probe.txt
timeout 400 python tools/mlp_ab.py --a tools/bin/libmi3d_dev_tr0.so --b tools/bin/libmi3d_dev_tr1.so --out $O/mlp_ab.json > $O/mlp_ab.log 2>&1
rc=$?
echo "mlp_ab rc=$rc" >> $O/mlp_ab.log
tail -8 $O/mlp_ab.log
if [ $rc -ne 0 ] || ! grep -q "PROBE OK" $O/tr_probe.txt; then echo "A/B not clean: stopping here"; exit 1; fi
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
b = json.load(open("gpurun_out/r04_18/bench_c2_dense.json"))
print(b["ms_per_step"], b.get("kernels_ms_per_step"), b.get("valid"))
P
