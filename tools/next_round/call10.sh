#!/bin/bash
O=gpurun_out/r2c10; mkdir -p $O
timeout 200 python -m pytest tests/test_kernels_gpu.py -q -m gpu --no-header -p no:cacheprovider -s -k "fp8" 2>&1 | grep "rel-L2\|passed\|failed" | tail -n 8
timeout 200 python tools/bench_attn_f8.py 2>&1 | grep -v amdgpu | tee $O/f8bench.txt
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
run() { n=$1; shift; env "$@" timeout 300 $B > $O/bench_$n.log 2>&1; echo -n "$n: "; grep -o '"ms_per_step": [0-9.]*' $O/bench_$n.log || tail -n 3 $O/bench_$n.log; }
run default A=1
run t128x64 SG_TILE=128,64
run t128x64_notable SG_TILE=128,64 SG_NO_TILE_TABLE=1
run t128x128_s2 SG_TILE=128,128 SG_STAGES=2
run t64x64_s2 SG_TILE=64,64 SG_STAGES=2
run t256x64_s2 SG_TILE=256,64 SG_STAGES=2
run s2 SG_STAGES=2
run default2 A=1
