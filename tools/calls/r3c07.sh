#!/bin/bash
# round 3, call 7: ref-split timing experiment; config-5 golden tests (fp16 vs oracle, fp8 deviation); RCCL all-reduce with one rank; PNDM at 50 steps
O=$GRAFT_REPO_ROOT/gpurun_out/r3c07; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python tools/exp_ref_split.py 2>&1 | grep -v amdgpu.ids | tee $O/exp_ref_split.txt
timeout 900 python -m pytest tests/test_unet_gpu.py -q -m gpu --no-header -p no:cacheprovider -s -k "config5 or pndm" 2>&1 | tee $O/config5.log | grep -E "config 5|PNDM|multi-image|auto-reg|passed|failed|Error|assert" | tail -n 14
timeout 300 python -m pytest tests/test_optim_gpu.py -q -m gpu --no-header -p no:cacheprovider -k "rccl" 2>&1 | tail -n 3
