#!/bin/bash
# round 4, call 10: wide GroupNorm as ONE round of workgroups (256 / B chunks per sample) against the 64-chunk cap of rounds 1-3
set -u
O=gpurun_out/r4j; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "groupnorm" -x > $O/gn_tests.log 2>&1; echo "gn tests rc=$?" > $O/summary.txt
timeout 300 python tools/bench_norm.py --pstats > $O/bench_norm_pstats.txt 2>&1
SG_GN_CHUNKS=64 timeout 300 python tools/bench_norm.py --pstats > $O/bench_norm_pstats_chunks64.txt 2>&1
for i in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --steps 20 > $O/bench_default_$i.json 2> $O/bench_default_$i.err
  SG_DEV_OPTIONS=1 SG_GN_CHUNKS=64 timeout 300 python bench.py --no-cpu-baseline --steps 20 > $O/bench_chunks64_$i.json 2> $O/bench_chunks64_$i.err
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
tail -n 5 $O/gn_tests.log; paste $O/bench_norm_pstats.txt $O/bench_norm_pstats_chunks64.txt | cut -c1-150; cat $O/summary.txt
