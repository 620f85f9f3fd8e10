#!/bin/bash
# round 5, call 11: the shared head's input check, the drop-in pipeline tests on the final tree, and three round-4 switches re-measured under
# the group schedule (2-stage ring, unfused feed-forward, split-K reduction outside the GroupNorm)
O=$GRAFT_REPO_ROOT/gpurun_out/r5k; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_unet_gpu.py tests/test_dropin_gpu.py -q -m gpu -x -k "shared_cfg or pipeline or dropin or guard" > $O/tests.log 2>&1; echo "tests rc=$?" > $O/summary.txt; tail -2 $O/tests.log >> $O/summary.txt
run() { n=$1; shift; "$@" > $O/bench_${n}_$RANDOM.json 2>> $O/bench.err; }
run default timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 10
run stages2 env SG_DEV_OPTIONS=1 SG_PIPE_STAGES=2 timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 10
run noff timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 10 --no-ff-fused
run nosplitgn timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 10 --no-splitk-in-gn
run default timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 10
for f in $O/bench_*.json; do python - "$f" <<'PY' >> $O/summary.txt
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d["roofline"]
    print(sys.argv[1].split('/')[-1], d["ms_per_step"], "ms", {k:(v["launches"],round(v["ms"],2)) for k,v in r["families"].items() if k in ("gemm","conv3x3","ff_fused")}, r["hbm_families"].get("groupnorm",{}).get("ms"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
cat $O/summary.txt
