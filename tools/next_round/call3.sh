#!/bin/bash
O=gpurun_out/r2c3; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu --no-header -p no:cacheprovider -x 2>&1 | tee $O/kernels.log | tail -n 25
timeout 900 python -m pytest tests/test_unet_gpu.py tests/test_dropin_gpu.py -q -m gpu --no-header -p no:cacheprovider -x -k "not full_depth and not ref_ahead and not config5" 2>&1 | tee $O/unet.log | tail -n 25
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
timeout 300 $B > $O/bench.log 2>&1; grep -o '"ms_per_step": [0-9.]*' $O/bench.log || tail -n 5 $O/bench.log
timeout 300 python tools/profile_step.py > $O/per_shape.txt 2>&1; head -5 $O/per_shape.txt
