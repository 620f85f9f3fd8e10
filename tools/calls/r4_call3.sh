#!/bin/bash
# round 4, call 3: anatomy of the fused feed-forward kernel and its schedule variants
set -u
O=gpurun_out/r4c; mkdir -p $O
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -k "fused_geglu" > $O/kernel_tests.log 2>&1; echo "kernel tests rc=$?" >> $O/summary.txt
timeout 300 python tools/anatomy_ff.py > $O/anatomy_ff.txt 2>&1
for var in 0 1 2 3; do
  SG_DEV_OPTIONS=1 SG_FF_VARIANT=$var timeout 300 python bench.py --no-cpu-baseline --steps 20 > $O/bench_var$var.json 2> $O/bench_var$var.err
done
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
tail -n 3 $O/kernel_tests.log; cat $O/anatomy_ff.txt; cat $O/summary.txt
