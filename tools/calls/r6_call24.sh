#!/bin/bash
# round 6, call 24: the CONTRACT configuration (64x64, R = 3, G = 5) repeated: every repeat of 10 steps must be bit-identical (shipped defaults),
# and the same with every kind on the latency kernel
O=$GRAFT_REPO_ROOT/gpurun_out/r6v; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1200 python tools/exp_determinism.py reps=20 only=one-graph hw=64 G=5 R=3 steps=10 > $O/det_contract_default.txt 2>&1; echo "contract config, shipped defaults: $(grep -c bit-identical $O/det_contract_default.txt) of 20"
SG_LAT_MASK=63 timeout 1200 python tools/exp_determinism.py reps=20 only=one-graph hw=64 G=5 R=3 steps=10 > $O/det_contract_mask63.txt 2>&1; echo "contract config, pairs on the latency kernel: $(grep -c bit-identical $O/det_contract_mask63.txt) of 20"
