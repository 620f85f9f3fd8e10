#!/bin/bash
# round 5, call 5: the whole GPU suite on the round-5 tree (group schedule default in bench + pipeline, shared CFG head, LayerNorm-fold
# guard, batched reference pass in training), smoke, and quick A/Bs of the remaining schedule switches at G = 5
O=$GRAFT_REPO_ROOT/gpurun_out/r5e; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu -x > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?" > $O/summary.txt; tail -4 $O/gpu_tests.log >> $O/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt >> $O/summary.txt
for v in "" "--attn-pair" "--no-gemm-pairs" "--ref-ahead 10" ""; do
  n=$(echo "$v" | tr -d ' -'); 
  timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 10 $v > $O/bench_${n:-default}_$RANDOM.json 2>> $O/bench.err
done
for f in $O/bench_*.json; do python - "$f" <<'PY' >> $O/summary.txt
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    c=d["config"]
    print(sys.argv[1].split('/')[-1], d["ms_per_step"], "ms", "G", c["ref_ahead"], "attn_pair", c["paired_text_image_attention"], "gemm_pairs", c["paired_gemm_launches"])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
cat $O/summary.txt
