#!/bin/bash
# round 5, call 13: hidden-split fused feed-forward as its own instantiation — A/B, then the closing measurements again on
# these (final) kernel sources: whole GPU suite, smoke, contract line, PMC traffic + algorithmic bytes
O=$GRAFT_REPO_ROOT/gpurun_out/r5m; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 200 python __graft_entry__.py smoke 2>&1 | tail -n 1 | tee $O/smoke.txt
timeout 1500 python -m pytest tests -q -m gpu --no-header -p no:cacheprovider --maxfail=20 2>&1 | tee $O/gpu_tests.log | tail -n 6
for i in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 10 > $O/bench_split_$i.json 2>> $O/bench.err
  timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 10 --no-ff-split > $O/bench_nosplit_$i.json 2>> $O/bench.err
done
for f in $O/bench_*.json; do python - "$f" <<'PY' >> $O/summary.txt
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d["roofline"]
    print(sys.argv[1].split('/')[-1], d["ms_per_step"], "ms", {k:(v["launches"],round(v["ms"],3),round(v["tflops"],1)) for k,v in r["families"].items() if k in ("gemm", "ff_fused")})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
cat $O/summary.txt
timeout 600 python bench.py --steps 20 --warmup 5 --dump-algorithmic $O/algorithmic.json > $O/bench.json 2>$O/bench_contract.err; cut -c1-260 $O/bench.json
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 5 --no-cpu-baseline"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f -o p -- $CMD > $O/f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/w -o p -- $CMD > $O/w.log 2>&1
cd $GRAFT_REPO_ROOT
F=$(find $O/f -name "*counter_collection.csv" | head -1); W=$(find $O/w -name "*counter_collection.csv" | head -1)
python tools/traffic_from_pmc.py $F $W "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) --kernel-trace -- python bench.py --steps 5 --warmup 5 --no-cpu-baseline (ref_ahead 5); algorithmic bytes: bench.py --dump-algorithmic of the same build; MI355X; round 5, FINAL sources (taps innermost, grouped tiles, key-split D = 160 attention, hidden-split fused feed-forward); $(date -u +%F)" $O/algorithmic.json > $O/traffic.json; head -c 300 $O/traffic.json
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete
