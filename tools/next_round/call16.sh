#!/bin/bash
O=gpurun_out/r2c16; mkdir -p $O
timeout 200 python -m pytest tests/test_kernels_gpu.py -q -m gpu --no-header -p no:cacheprovider -x -k "pingpong" 2>&1 | tail -n 15
SG_PINGPONG=1 timeout 300 python tools/exp_feed.py > $O/feed_pp.log 2>&1; grep -A6 "auto cold" $O/feed_pp.log | cut -c1-150 | head -90
