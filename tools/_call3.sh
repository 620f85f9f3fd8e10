#!/bin/bash
O=gpurun_out/r2b; mkdir -p $O
SG_ATTN_PRIO=1 SG_ATTN_D80=1 timeout 60 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attention" > $O/pytest_a1.log 2>&1; echo "attn prio+d80=1 rc=$?"
SG_ATTN_D80=2 timeout 60 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attention" > $O/pytest_a2.log 2>&1; echo "attn d80=2 rc=$?"
timeout 60 python tools/bench_norm.py --attn > $O/attn_def.log 2>&1
SG_ATTN_PRIO=1 SG_ATTN_D80=1 timeout 60 python tools/bench_norm.py --attn > $O/attn_prio_d80v1.log 2>&1
SG_ATTN_D80=2 timeout 60 python tools/bench_norm.py --attn > $O/attn_d80v2.log 2>&1
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
timeout 120 $B > $O/bench_def.log 2>&1
SG_ATTN_PRIO=1 timeout 120 $B > $O/bench_prio.log 2>&1
for f in $O/pytest_a1.log $O/pytest_a2.log; do tail -n 1 $f; done
for f in def prio; do echo -n "$f: "; grep -o '"ms_per_step": [0-9.]*' $O/bench_$f.log; done
paste <(grep attn_ $O/attn_def.log) <(grep attn_ $O/attn_prio_d80v1.log | awk '{print $5,$6}') <(grep attn_ $O/attn_d80v2.log | awk '{print $5,$6}')
