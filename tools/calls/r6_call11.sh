#!/bin/bash
# round 6, call 11: pairs off the latency kernel (lat_mask 62 default): determinism over 40 repeats per schedule, then the GPU suite and the bench A/B
O=$GRAFT_REPO_ROOT/gpurun_out/r6k; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python tools/exp_determinism.py reps=40 only=one-graph > $O/det_one_graph.txt 2>&1; echo "one graph: $(grep -c bit-identical $O/det_one_graph.txt) of 40"
timeout 900 python tools/exp_determinism.py reps=20 only=split+priority > $O/det_split_prio.txt 2>&1; echo "split+priority: $(grep -c bit-identical $O/det_split_prio.txt) of 20"
timeout 2400 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; tail -4 $O/gpu_tests.log
for i in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$i.json 2>$O/bench.err; cut -c1-200 $O/bench_$i.json
SG_DEV_OPTIONS=1 SG_LAT_MASK=63 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loop > $O/bench_pairs_lat_$i.json 2>$O/bench.err; cut -c1-200 $O/bench_pairs_lat_$i.json
done
python - <<'PY'
import json
d=json.load(open("gpurun_out/r6k/bench_1.json"))
print({k:d[k] for k in ("ms_per_step","loop_50_steps_ms","loop")})
PY
