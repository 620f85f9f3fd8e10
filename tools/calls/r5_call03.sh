#!/bin/bash
# round 5, call 3: EXPERIMENT — the batched reference pass launched eagerly on a CU-masked stream (hipExtStreamCreateWithCUMask)
# beside the main-pass graphs; training step with the batched early-exit reference pass; guard tests after the test fix
O=$GRAFT_REPO_ROOT/gpurun_out/r5c; mkdir -p $O
cd $GRAFT_REPO_ROOT
for v in "" "--ref-cus 256" "--ref-cus 128" "--ref-cus 128 --ref-cu-layout spread" "--ref-cus 96" "--ref-cus 160" "--ref-cus 64 --ref-cu-layout spread" "--split-graphs" ""; do
  n=$(echo "$v" | tr -d ' -'); 
  timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 10 $v > $O/bench_${n:-default}_$RANDOM.json 2>> $O/bench.err
done
for f in $O/bench_*.json; do python - "$f" <<'PY' >> $O/summary.txt
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], d["ms_per_step"], "ms", "G", d["config"]["ref_ahead"], "split", d["config"]["split_graphs"], "ref_cus", d["config"]["ref_pass_eager_on_cus"], "tflop", d["tflop_per_step_executed"], "finite", d["latents_finite"])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
cat $O/summary.txt; tail -5 $O/bench.err
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "guard" > $O/kernel_tests.log 2>&1; echo "kernel guard tests rc=$?" >> $O/summary.txt
timeout 600 python -m pytest tests/test_unet_gpu.py -q -m gpu -x -s -k "guard" > $O/unet_tests.log 2>&1; echo "unet guard tests rc=$?" >> $O/summary.txt; grep "stream offset" $O/unet_tests.log >> $O/summary.txt
timeout 900 python -m pytest tests/test_backward_gpu.py -q -m gpu -x -s > $O/backward_tests.log 2>&1; echo "backward tests rc=$?" >> $O/summary.txt; tail -3 $O/backward_tests.log >> $O/summary.txt
timeout 600 python bench.py --train-step --steps 5 --warmup 2 > $O/train_none.json 2> $O/train.err; python - <<'PY' >> $O/summary.txt
import json
try:
    d=json.loads(open("gpurun_out/r5c/train_none.json").read().strip().splitlines()[-1]); print("train step", d["ms_per_step"], "ms", d["roofline"]["frac"], d["tflop_per_step_executed"], {k:(v["launches"],v["ms"]) for k,v in d["roofline"]["families"].items()})
except Exception as e: print("train FAILED", e)
PY
tail -12 $O/summary.txt
