#!/bin/bash
# round 3, call 16: what is left of the GPU budget on the UNet / drop-in parity tests of the final sources (verbose: a test that
# finished before the limit is on record)
O=$GRAFT_REPO_ROOT/gpurun_out/r3c16; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 420 python -m pytest tests/test_unet_gpu.py tests/test_dropin_gpu.py -v --no-header -p no:cacheprovider --maxfail=5 --deselect tests/test_unet_gpu.py::test_config5_fp8_attention_vs_oracle_golden_96x96_r5 2>&1 | grep -E "PASSED|FAILED|ERROR|passed|failed" | tee $O/unet_dropin_tests.log | tail -n 45
