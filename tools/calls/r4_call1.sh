#!/bin/bash
# round 4, call 1: kernel tests of the new attention / deferred split-K paths, the UNet-level parity tests that exercise them, A/B bench lines
set -u
O=gpurun_out/r4a; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "attention or deferred or groupnorm or copy_rows" > $O/kernel_tests.log 2>&1; echo "kernel tests rc=$?" >> $O/summary.txt
timeout 900 python -m pytest tests/test_unet_gpu.py -q -x -k "single_pass or denoise_steps or dedup or distinct_prev or graph_replay" > $O/unet_tests.log 2>&1; echo "unet tests rc=$?" >> $O/summary.txt
for v in "default:" "general:SG_ATTN_D40_GENERAL=1" "lean:SG_ATTN_LEAN=1"; do
  n=${v%%:*}; e=${v#*:}
  env SG_DEV_OPTIONS=1 $e timeout 300 python bench.py --no-cpu-baseline --steps 20 > $O/bench_$n.json 2> $O/bench_$n.err
done
timeout 300 python bench.py --no-cpu-baseline --steps 20 --no-short-rows > $O/bench_noshort.json 2> $O/bench_noshort.err
timeout 300 python bench.py --no-cpu-baseline --steps 20 --no-splitk-in-gn > $O/bench_nosplitkgn.json 2> $O/bench_nosplitkgn.err
timeout 300 python bench.py --no-cpu-baseline --steps 20 > $O/bench_default2.json 2> $O/bench_default2.err
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
tail -5 $O/kernel_tests.log $O/unet_tests.log; cat $O/summary.txt
