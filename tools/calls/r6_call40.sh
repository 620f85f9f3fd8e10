#!/bin/bash
# round 6, call 40: eight waves of 128x64 on a 2-stage ring (mma_fat_kernel): parity, per-shape timing vs the shipped plan, contract A/B with fat_m
O=$GRAFT_REPO_ROOT/gpurun_out/r6bi; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "fat_wave or tile_waves or every_tile_every_mode or conv3x3" > $O/tests.log 2>&1; grep -E "passed|failed|Error|assert" $O/tests.log | head -20
timeout 600 python tools/bench_fat.py > $O/bench_fat.txt 2>&1; cat $O/bench_fat.txt
for i in 1 2; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loop > $O/base_$i.json 2>$O/err.txt; cut -c100-200 $O/base_$i.json
  SG_FAT_M=40000 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loop > $O/fat_$i.json 2>$O/err_fat.txt; cut -c100-200 $O/fat_$i.json
done
tail -3 $O/err_fat.txt
