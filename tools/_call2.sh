#!/bin/bash
O=gpurun_out/r2a; mkdir -p $O
timeout 120 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu > $O/pytest_k.log 2>&1; echo "kernels rc=$?"
for v in 1 2 3; do SG_ATTN_D160=$v timeout 60 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attention" > $O/pytest_a$v.log 2>&1; echo "attn v$v rc=$?"; done
for v in 0 1 2 3; do SG_ATTN_D160=$v timeout 60 python tools/bench_norm.py --attn > $O/attn160_v$v.log 2>&1; done
timeout 60 python tools/bench_norm.py > $O/norm_wide.log 2>&1
SG_GN_WIDE=0 timeout 60 python tools/bench_norm.py > $O/norm_old.log 2>&1
SG_GN_FUSED_MAX=0 timeout 60 python tools/bench_norm.py > $O/norm_nofused.log 2>&1
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
timeout 120 $B > $O/bench_new.log 2>&1
SG_GN_FUSED_MAX=0 timeout 120 $B > $O/bench_nofused.log 2>&1
SG_GN_FUSED_MAX=10240 timeout 120 $B > $O/bench_f10k.log 2>&1
SG_ATTN_D160=1 timeout 120 $B > $O/bench_a1.log 2>&1
SG_ATTN_D160=3 timeout 120 $B > $O/bench_a3.log 2>&1
timeout 120 $B > $O/bench_new2.log 2>&1
timeout 120 python tools/profile_step.py > $O/per_shape.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/trace.log 2>&1
cd $GRAFT_REPO_ROOT
for f in $O/pytest_k.log $O/pytest_a1.log $O/pytest_a2.log $O/pytest_a3.log; do tail -n 1 $f; done
for f in new nofused f10k a1 a3 new2; do echo -n "$f: "; grep -o '"ms_per_step": [0-9.]*' $O/bench_$f.log; done
ls -la $O/trace | head
