#!/bin/bash
# round 4, call 9: GroupNorm merge after the half-wave second stage (+ the apply kernel's floor: one partial per channel); time-embedding tables A/B
set -u
O=gpurun_out/r4i; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "groupnorm" -x > $O/gn_tests.log 2>&1; echo "gn tests rc=$?" > $O/summary.txt
timeout 900 python -m pytest tests/test_unet_gpu.py -q -m gpu -k "tabulated or graph_replay or two_samples or ref_ahead" -x > $O/unet_tests.log 2>&1; echo "unet tests rc=$?" >> $O/summary.txt
timeout 300 python tools/bench_norm.py --pstats > $O/bench_norm_pstats.txt 2>&1
for i in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --steps 20 > $O/bench_tables_$i.json 2> $O/bench_tables_$i.err
  timeout 300 python bench.py --no-cpu-baseline --steps 20 --no-time-tables > $O/bench_notables_$i.json 2> $O/bench_notables_$i.err
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
tail -n 5 $O/gn_tests.log; tail -n 15 $O/unet_tests.log; cat $O/bench_norm_pstats.txt; cat $O/summary.txt
