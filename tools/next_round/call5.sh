#!/bin/bash
O=gpurun_out/r2c5; mkdir -p $O
timeout 900 python -m pytest tests/test_unet_gpu.py tests/test_dropin_gpu.py -q -m gpu --no-header -p no:cacheprovider -s -k "full_depth or processor or rccl or sees_new" 2>&1 | tee $O/tests.log | grep -v "^$" | tail -n 30
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
for v in pairs:"" nopairs:"--no-gemm-pairs" pairs2:"" nopairs2:"--no-gemm-pairs"; do
  n=${v%%:*}; f=${v#*:}
  timeout 300 $B $f > $O/bench_$n.log 2>&1; echo -n "$n: "; grep -o '"ms_per_step": [0-9.]*' $O/bench_$n.log || tail -n 5 $O/bench_$n.log
done
timeout 600 python bench.py --train-step --steps 3 --warmup 1 > $O/train.log 2>&1; tail -n 3 $O/train.log
