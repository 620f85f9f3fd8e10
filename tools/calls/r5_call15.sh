#!/bin/bash
# round 5, call 15: the attention / GroupNorm instantiation options re-measured under the group schedule (all exist since rounds 1 - 4)
O=$GRAFT_REPO_ROOT/gpurun_out/r5o; mkdir -p $O
cd $GRAFT_REPO_ROOT
run() { n=$1; shift; env SG_DEV_OPTIONS=1 "$@" timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 10 > $O/bench_${n}_$RANDOM.json 2>> $O/bench.err; }
run default SG_NOOP=1
run d80_2x3 SG_ATTN_D80=2
run d80_auto SG_ATTN_D80=0
run sub2 SG_ATTN_SUB2=1
run lean SG_ATTN_LEAN=1
run gnchunks64 SG_GN_CHUNKS=64
run default SG_NOOP=1
for f in $O/bench_*.json; do python - "$f" <<'PY' >> $O/summary.txt
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d["roofline"]
    print(sys.argv[1].split('/')[-1], d["ms_per_step"], "ms", {k:(round(v["ms"],2),round(v["tflops"])) for k,v in r["families"].items() if k.startswith("attention")}, r["hbm_families"]["groupnorm"]["ms"])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
cat $O/summary.txt; tail -3 $O/bench.err
