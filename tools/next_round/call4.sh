#!/bin/bash
O=gpurun_out/r2c4; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu --no-header -p no:cacheprovider -k "conv or gemm or split" 2>&1 | tee $O/kernels.log | tail -n 40
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
for v in patch:"" nopatch:"--no-conv-patch" patch2:"" ; do
  n=${v%%:*}; f=${v#*:}
  timeout 300 $B $f > $O/bench_$n.log 2>&1; echo -n "$n: "; grep -o '"ms_per_step": [0-9.]*' $O/bench_$n.log || tail -n 5 $O/bench_$n.log
done
timeout 300 python tools/profile_step.py > $O/per_shape.txt 2>&1; head -4 $O/per_shape.txt
timeout 300 python tools/exp_feed.py conv > $O/feed_conv.log 2>&1; grep "auto cold" $O/feed_conv.log
