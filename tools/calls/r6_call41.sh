#!/bin/bash
# round 6, call 41: contract step with the 512x128 / 128x64-per-wave tile for the convolutions of >= 20 000 rows (fat_m), same-box A/B
O=$GRAFT_REPO_ROOT/gpurun_out/r6bj; mkdir -p $O
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loop > $O/base_$i.json 2>$O/err.txt; cut -c100-200 $O/base_$i.json
  SG_DEV_OPTIONS=1 SG_FAT_M=20000 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loop > $O/fat_$i.json 2>$O/err_fat.txt; cut -c100-200 $O/fat_$i.json
done
grep "development options" $O/err_fat.txt
timeout 600 python -m pytest tests/test_unet_gpu.py -x -q -m gpu -k "golden_32x32 or bit_reproducible" > $O/tests.log 2>&1; grep -E "passed|failed" $O/tests.log
