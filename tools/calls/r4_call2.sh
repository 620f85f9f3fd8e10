#!/bin/bash
# round 4, call 2: fused feed-forward kernel + batched split-K pre-pass of the GroupNorm: kernel tests, full-depth parity, A/B lines
set -u
O=gpurun_out/r4b; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "fused_geglu or deferred or fast_path or short_kv or extreme" > $O/kernel_tests.log 2>&1; echo "kernel tests rc=$?" >> $O/summary.txt
timeout 900 python -m pytest tests/test_unet_gpu.py -q -x -k "full_depth and multi-image or single_pass or denoise_steps" > $O/unet_tests.log 2>&1; echo "unet tests rc=$?" >> $O/summary.txt
for v in default:"" noff:"--no-ff-fused" nosplitkgn:"--no-splitk-in-gn" default2:""; do
  n=${v%%:*}; a=${v#*:}
  timeout 300 python bench.py --no-cpu-baseline --steps 20 $a > $O/bench_$n.json 2> $O/bench_$n.err
done
timeout 400 python tools/profile_step.py > $O/per_shape.txt 2>&1
for f in $O/bench_*.json; do python - "$f" <<'PY' >> $O/summary.txt
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d["roofline"]
    print(sys.argv[1], d["ms_per_step"], "ms", {k:(v["launches"],round(v["ms"],3),round(v["tflops"],1)) for k,v in r["families"].items()}, {k:(v["launches"],v["ms"]) for k,v in r["hbm_families"].items()})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
tail -n 5 $O/kernel_tests.log $O/unet_tests.log; cat $O/summary.txt
