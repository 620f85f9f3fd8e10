#!/bin/bash
O=gpurun_out/r2c15; mkdir -p $O
B="python bench.py --steps 8 --warmup 2 --no-cpu-baseline --config5-shape"
for v in f16:"" f8:"--fp8-attention" f16b:"" f8b:"--fp8-attention"; do
  n=${v%%:*}; f=${v#*:}
  timeout 400 $B $f > $O/bench_c5_$n.json 2>$O/err_$n.log; echo -n "$n: "; grep -o '"ms_per_step": [0-9.]*' $O/bench_c5_$n.json || tail -n 5 $O/err_$n.log
done
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_default.json 2>/dev/null; grep -o '"ms_per_step": [0-9.]*' $O/bench_default.json; grep -o '"hbm_families": {[^}]*}[^}]*}[^}]*}' $O/bench_default.json | head -c 600
