#!/bin/bash
# round 4, call 8: channel-coalesced merge of the producers' GroupNorm statistics (gn_merge) against the round-3 gather, kernel level and whole step
set -u
O=gpurun_out/r4h; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "groupnorm" -x > $O/gn_tests.log 2>&1; echo "gn tests rc=$?" > $O/summary.txt
timeout 300 python tools/bench_norm.py --pstats > $O/bench_norm_pstats.txt 2>&1
for i in 1 2; do
  SG_DEV_OPTIONS=1 SG_GN_MERGE=1 timeout 300 python bench.py --no-cpu-baseline --steps 20 > $O/bench_coal_$i.json 2> $O/bench_coal_$i.err
  SG_DEV_OPTIONS=1 SG_GN_MERGE=0 timeout 300 python bench.py --no-cpu-baseline --steps 20 > $O/bench_gather_$i.json 2> $O/bench_gather_$i.err
done
SG_DEV_OPTIONS=1 SG_GN_CHUNKS=128 timeout 300 python bench.py --no-cpu-baseline --steps 20 > $O/bench_coal_chunks128.json 2> $O/bench_coal_chunks128.err
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
tail -n 5 $O/gn_tests.log; cat $O/bench_norm_pstats.txt; cat $O/summary.txt
