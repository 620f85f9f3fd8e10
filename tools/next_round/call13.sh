#!/bin/bash
O=gpurun_out/r2c13; mkdir -p $O
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -m gpu --no-header -p no:cacheprovider -k "producer_epilogues or gemm or conv3x3 or groupnorm" 2>&1 | tail -n 25
timeout 300 python -m pytest tests/test_backward_gpu.py -q -m gpu --no-header -p no:cacheprovider -k "graph_replay" 2>&1 | tail -n 6
timeout 900 python -m pytest tests/test_unet_gpu.py -q -m gpu --no-header -p no:cacheprovider -s -k "full_depth or passes_vs_oracle or single_pass or graph_replay or both_stages" 2>&1 | grep -v "^$" | tail -n 14
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
for v in gn:"" nogn:"--no-gn-epilogue" gn2:"" nogn2:"--no-gn-epilogue"; do
  n=${v%%:*}; f=${v#*:}
  timeout 300 $B $f > $O/bench_$n.log 2>&1; echo -n "$n: "; grep -o '"ms_per_step": [0-9.]*' $O/bench_$n.log || tail -n 5 $O/bench_$n.log
done
timeout 300 python tools/profile_step.py > $O/per_shape.txt 2>&1; grep -A4 "bandwidth-class" $O/per_shape.txt
