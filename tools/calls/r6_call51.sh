#!/bin/bash
# round 6, call 51: K slices of the main pass's 16x16 / 8x8 convolutions (M <= 768) capped at 2 .. 12: convolution + the GroupNorm that sums the slices, against the step
# (the development option split_cap / split_cap_m existed only for this call: storygen_amd/csrc/gemm_conv.hip choose_plan skipped s > cap for M <= split_cap_m; result in profiles/r06bs_*)
O=$GRAFT_REPO_ROOT/gpurun_out/r6bs; mkdir -p $O
cd $GRAFT_REPO_ROOT
run() { name=$1; shift; env "$@" SG_DEV_OPTIONS=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loop > $O/$name.json 2>$O/err_$name.txt; echo "$name $(python -c "import json;print(json.load(open('$O/$name.json'))['ms_per_step'])")"; }
for i in 1 2; do
  run base_$i SG_NOP=1
  for c in 2 4 6 8 12; do run cap${c}_$i SG_SPLIT_CAP=$c SG_SPLIT_CAP_M=800; done
done
