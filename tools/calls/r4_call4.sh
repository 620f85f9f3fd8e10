#!/bin/bash
# round 4, call 4: the whole GPU suite with per-test durations (the driver gives it 1200 s), GroupNorm thread-count A/B, split-K-in-GroupNorm A/B
set -u
O=gpurun_out/r4d; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu --durations=45 > $O/gpu_suite.log 2>&1; echo "gpu suite rc=$?" >> $O/summary.txt
for v in default:"" gnnt1024:"SG_GN_FUSED_NT=1024" ; do
  n=${v%%:*}; e=${v#*:}
  env SG_DEV_OPTIONS=1 $e timeout 300 python bench.py --no-cpu-baseline --steps 20 > $O/bench_$n.json 2> $O/bench_$n.err
done
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
tail -n 60 $O/gpu_suite.log; cat $O/summary.txt
