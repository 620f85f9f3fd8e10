#!/bin/bash
# round 6, call 39: the scalar-base form of the LDS-DMA instruction in the GEMM / conv mainloop (variant _saddr, inline asm): kernel tests, chain cost, contract A/B
O=$GRAFT_REPO_ROOT/gpurun_out/r6bh; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python tools/pytest_with_lib.py storygen_amd/lib/libstorygen_hip_saddr.so tests/test_kernels_gpu.py -x -q -m gpu > $O/tests_saddr.log 2>&1; grep -E "passed|failed" $O/tests_saddr.log
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu > $O/tests_base.log 2>&1; grep -E "passed|failed" $O/tests_base.log
for i in 1 2; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loop > $O/base_$i.json 2>$O/err.txt; cut -c100-200 $O/base_$i.json
  timeout 300 python tools/ab_lib.py storygen_amd/lib/libstorygen_hip_saddr.so --steps 20 --warmup 5 --no-cpu-baseline --no-loop > $O/saddr_$i.json 2>$O/err.txt; cut -c100-200 $O/saddr_$i.json
done
