#!/bin/bash
O=gpurun_out/r2c9; mkdir -p $O
timeout 600 python -m pytest tests/test_unet_gpu.py -q -m gpu --no-header -p no:cacheprovider -s -k "config5" 2>&1 | tee $O/config5.log | grep -v "^$" | tail -n 12
B="python bench.py --steps 8 --warmup 2 --no-cpu-baseline --config5-shape"
timeout 400 $B > $O/bench_c5_f16.json 2>$O/err1.log; grep -o '"ms_per_step": [0-9.]*' $O/bench_c5_f16.json || tail -n 5 $O/err1.log
timeout 400 $B --fp8-attention > $O/bench_c5_f8.json 2>$O/err2.log; grep -o '"ms_per_step": [0-9.]*' $O/bench_c5_f8.json || tail -n 5 $O/err2.log
