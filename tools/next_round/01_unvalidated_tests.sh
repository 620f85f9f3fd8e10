#!/bin/bash
# gpurun --timeout 300 -- 'bash tools/next_round/01_unvalidated_tests.sh'
O=gpurun_out/next1; mkdir -p $O
SG_TEST_UNVALIDATED=1 timeout 200 python -m pytest tests/test_backward_gpu.py -q -m gpu 2>&1 | tee $O/backward.log | tail -n 30
SG_TEST_UNVALIDATED=1 timeout 120 python -m pytest tests/test_unet_gpu.py -q -m gpu -k "split_graphs" 2>&1 | tee $O/split.log | tail -n 5
