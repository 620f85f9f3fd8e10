#!/bin/bash
# round 6, call 48: the optimizer tests after the tolerance note (test_adamw_f32_is_torch_adamw: 2.4e-6 on the no-SLP build)
O=$GRAFT_REPO_ROOT/gpurun_out/r6bq; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_optim_gpu.py -q -m gpu > $O/tests.log 2>&1; grep -E "passed|failed" $O/tests.log
