#!/bin/bash
# round 5, call 4: EXPERIMENT, second half — the batched reference pass as a hipGraph replayed on a CU-masked stream (a linear graph
# runs on its launch stream's queue) beside the main-pass graphs
O=$GRAFT_REPO_ROOT/gpurun_out/r5d; mkdir -p $O
cd $GRAFT_REPO_ROOT
for v in "" "--split-graphs" "--ref-cus 128" "--ref-cus 96" "--ref-cus 160" "--ref-cus 192" "--ref-cus 128 --ref-cu-layout spread" "--ref-cus 64" "--ref-cus 128 --stream-priority" ""; do
  n=$(echo "$v" | tr -d ' -'); 
  timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 10 $v > $O/bench_${n:-default}_$RANDOM.json 2>> $O/bench.err
done
for f in $O/bench_*.json; do python - "$f" <<'PY' >> $O/summary.txt
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    c=d["config"]
    print(sys.argv[1].split('/')[-1], d["ms_per_step"], "ms", "G", c["ref_ahead"], "split", c["split_graphs"], "ref_cus", c["ref_pass_on_cus"], "eager", c["ref_pass_eager"], "prio", c["stream_priority"], "finite", d["latents_finite"])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
cat $O/summary.txt; tail -3 $O/bench.err
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "guard" > $O/kernel_tests.log 2>&1; echo "kernel guard tests rc=$?" >> $O/summary.txt
tail -2 $O/summary.txt
