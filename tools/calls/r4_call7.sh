#!/bin/bash
# round 4, call 7: non-temporal epilogue traffic (SG_EPI_NT build of gemm_conv.hip) against the default library, alternating on one box; attn_sub2
set -u
O=gpurun_out/r4g; mkdir -p $O
NT=storygen_amd/lib/libstorygen_hip_nt.so
for i in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --steps 20 > $O/bench_default_$i.json 2> $O/bench_default_$i.err
  timeout 300 python tools/ab_lib.py $NT --no-cpu-baseline --steps 20 > $O/bench_nt_$i.json 2> $O/bench_nt_$i.err
done
SG_DEV_OPTIONS=1 SG_ATTN_SUB2=1 timeout 300 python bench.py --no-cpu-baseline --steps 20 > $O/bench_sub2.json 2> $O/bench_sub2.err
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
cat $O/summary.txt
