#!/bin/bash
O=gpurun_out/r2c7; mkdir -p $O
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -m gpu --no-header -p no:cacheprovider -s -k "fp8" 2>&1 | tee $O/f8test.log | grep -v "^$" | tail -n 25
timeout 300 python tools/bench_attn_f8.py 2>&1 | tee $O/f8bench.txt | tail -n 6
timeout 300 python tools/anatomy.py > $O/anatomy.txt 2>&1; cut -c1-250 $O/anatomy.txt
