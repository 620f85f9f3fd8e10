#!/bin/bash
# round 6, call 14: reference engine on an fp16 stream by default: GPU suite, bench A/B on one box, whole-loop line with the memoised step table
O=$GRAFT_REPO_ROOT/gpurun_out/r6n; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log | cut -c1-200; grep -n FAILED $O/gpu_tests.log | head
for i in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$i.json 2>$O/bench.err; cut -c1-200 $O/bench_$i.json
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loop --ref-fp32-stream > $O/bench_ref_fp32_$i.json 2>$O/bench.err; cut -c1-200 $O/bench_ref_fp32_$i.json
done
python - <<'PY'
import json
for f in ("bench_1","bench_2"):
    d=json.load(open(f"gpurun_out/r6n/{f}.json"))
    print({k:d[k] for k in ("ms_per_step","loop_50_steps_ms","loop")})
PY
