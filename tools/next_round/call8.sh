#!/bin/bash
O=gpurun_out/r2c8; mkdir -p $O
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -m gpu --no-header -p no:cacheprovider -k "spread" 2>&1 | tail -n 5
for sp in 0 1 2; do SG_SPREAD=$sp timeout 200 python tools/anatomy.py > $O/anatomy_sp$sp.txt 2>&1; cut -c1-215 $O/anatomy_sp$sp.txt | grep -v amdgpu; done
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
for v in sp0:"--spread 0" sp1:"--spread 1" sp2:"--spread 2" sp0b:"--spread 0" sp1b:"--spread 1" sp2b:"--spread 2"; do
  n=${v%%:*}; f=${v#*:}
  timeout 300 $B $f > $O/bench_$n.log 2>&1; echo -n "$n: "; grep -o '"ms_per_step": [0-9.]*' $O/bench_$n.log || tail -n 5 $O/bench_$n.log
done
