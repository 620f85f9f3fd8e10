#!/bin/bash
# round 6, call 31: backward attention with compile-time stages / full-tile path / VGPR-form MFMA: parity, per-shape A/B, train step A/B
O=$GRAFT_REPO_ROOT/gpurun_out/r6ba; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_backward_gpu.py -x -q -m gpu 2>&1 | tail -n 3 | tee $O/tests.txt
timeout 300 python tools/bench_attn_bwd.py storygen_amd/lib/libstorygen_hip_prev.so > $O/bwd_prev.txt 2>&1; cat $O/bwd_prev.txt
timeout 300 python tools/bench_attn_bwd.py > $O/bwd_new.txt 2>&1; cat $O/bwd_new.txt
timeout 600 python tools/ab_lib.py storygen_amd/lib/libstorygen_hip_prev.so --train-step --steps 5 --warmup 2 > $O/train_prev.json 2>$O/err1.txt; cut -c1-330 $O/train_prev.json
timeout 600 python bench.py --train-step --steps 5 --warmup 2 > $O/train_new.json 2>$O/err2.txt; cut -c1-330 $O/train_new.json
