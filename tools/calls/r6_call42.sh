#!/bin/bash
# round 6, call 42: how long the reference pass's workgroups hold a CU vs the step: its launches of >= 20 000 rows on smaller tiles (big_m / big_bm / big_bn), same-box A/B
O=$GRAFT_REPO_ROOT/gpurun_out/r6bk; mkdir -p $O
cd $GRAFT_REPO_ROOT
run() { name=$1; shift; env "$@" SG_DEV_OPTIONS=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loop > $O/$name.json 2>$O/err_$name.txt; echo "$name $(python -c "import json;print(json.load(open('$O/$name.json'))['ms_per_step'])")"; }
for i in 1 2; do
  run base_$i SG_NOP=1
  run t128x128_$i SG_BIG_M=20000 SG_BIG_BM=128 SG_BIG_BN=128
  run t256x64_$i SG_BIG_M=20000 SG_BIG_BM=256 SG_BIG_BN=64
  run t128x64_$i SG_BIG_M=20000 SG_BIG_BM=128 SG_BIG_BN=64
done
grep "development options" $O/err_t128x64_2.txt
