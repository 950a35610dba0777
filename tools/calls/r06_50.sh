#!/bin/bash
# round 6, GPU call 50: the point-0 pass deferred in FRONT of the MLP (grid_ops.DEFER_AT = "dh"): the step / field / parity tests
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_sds_step_gpu.py tests/test_field_gpu.py tests/test_headline_parity_gpu.py tests/test_reference_glue_gpu.py -m gpu -q 2>&1 | tail -40
