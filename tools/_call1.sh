#!/bin/bash
# one-off measurement script (GN wide / split-K reduce / 4-stage ring A/B)
O=gpurun_out/r1z; mkdir -p $O
timeout 120 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu > $O/pytest_k.log 2>&1; echo "kernels default rc=$?"
SG_STAGES=4 timeout 120 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm or conv" > $O/pytest_s4.log 2>&1; echo "kernels S4 rc=$?"
timeout 60 python tools/bench_norm.py > $O/norm_wide.log 2>&1
SG_GN_WIDE=0 timeout 60 python tools/bench_norm.py > $O/norm_old.log 2>&1
SG_GN_FUSED_MAX=0 timeout 60 python tools/bench_norm.py > $O/norm_nofused.log 2>&1
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
timeout 120 $B > $O/bench_new.log 2>&1
SG_GN_WIDE=0 timeout 120 $B > $O/bench_oldgn.log 2>&1
SG_GN_FUSED_MAX=0 timeout 120 $B > $O/bench_nofused.log 2>&1
SG_STAGES=4 timeout 120 $B > $O/bench_s4.log 2>&1
timeout 120 $B > $O/bench_new2.log 2>&1
timeout 120 python tools/profile_step.py > $O/per_shape.log 2>&1
tail -2 $O/pytest_k.log $O/pytest_s4.log
for f in new oldgn nofused s4 new2; do echo -n "$f: "; grep -o '"ms_per_step": [0-9.]*' $O/bench_$f.log; done
