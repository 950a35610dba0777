set -x
cd /root/repo
export TMPDIR=/tmp
python -m pytest tests/test_fullsize_gpu.py tests/test_reference_glue_gpu.py tests/test_rccl_gpu.py tests/test_grid_points_gpu.py tests/test_hashgrid_gpu.py -x -q 2>&1 | tail -25 > gpurun_out/c6_tests.log
tail -5 gpurun_out/c6_tests.log
python tools/kbench.py --what encode --half-planes --out gpurun_out/kb_encode_r03c.json > gpurun_out/kb_encode_r03c.log 2>&1
for cw in 3 4 5 8; do python tools/kbench.py --what encode1 --half-planes --dev 12=$cw --out gpurun_out/kb_enc_cw$cw.json 2>&1 | grep encode_ms; done
# un-profiled A/B of the scatter: previous library vs this one, 56 and 120 GiB caps
for cap in 56 120; do for v in old new; do
  if [ $v = old ]; then export MI3D_LIB=/root/repo/tools/bin/libmi3d_dev_oldscatter.so; else export MI3D_LIB=/root/repo/tools/bin/libmi3d_dev.so; fi
  MI3D_SCATTER_WORKSPACE_GB=$cap python tools/kbench.py --what scatter13 --half-planes --iters 4 --out gpurun_out/kb_ab_${v}_$cap.json 2>&1 | grep scatter_fp32
done; done
