#!/bin/bash
# round 4, GPU call 15: the flush's per-chunk overflow vote + hoisted record parts: tests (incl. forced region overflow), A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04_15
mkdir -p $O
timeout 900 python -m pytest tests/test_grid_points_gpu.py tests/test_fullsize_gpu.py tests/test_sds_step_gpu.py tests/test_headline_parity_gpu.py -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
for rep in 1 2 3; do
for V in dev_prev dev; do
  MI3D_LIB=$PWD/tools/bin/libmi3d_$V.so timeout 300 python tools/kbench.py --what scatter13 --half-planes --iters 4 --out $O/k_${V}_dense_$rep.json > /dev/null 2>> $O/kbench.err
  MI3D_LIB=$PWD/tools/bin/libmi3d_$V.so timeout 300 python tools/kbench.py --what scatter13 --half-planes --real-census --iters 4 --out $O/k_${V}_real_$rep.json > /dev/null 2>> $O/kbench.err
done
done
tail -3 $O/pytest.log
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04_15/k_*.json")):
    d=json.load(open(f)); print(f.split("/")[-1], round(d["scatter_fp32_P13_ms"],2))
P
