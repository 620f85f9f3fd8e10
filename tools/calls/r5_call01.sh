#!/bin/bash
# round 5, call 1: the one-graph look-ahead schedule (ref_ahead = G as ONE hipGraph per group): parity against the step-by-step
# schedule and the reference's 50-step goldens, then same-box A/B of G = 1 / 2 / 5 / 10 and a per-shape table at G = 5
O=$GRAFT_REPO_ROOT/gpurun_out/r5a; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_unet_gpu.py -q -m gpu -x -s -k "ref_ahead or group_schedule or full_depth or split_graphs" > $O/unet_tests.log 2>&1; echo "unet tests rc=$?" > $O/summary.txt
tail -5 $O/unet_tests.log >> $O/summary.txt
for G in 1 5 2 10 1 5; do
  timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 10 --ref-ahead $G > $O/bench_G${G}_$RANDOM.json 2> $O/bench_G$G.err
done
for f in $O/bench_*.json; do python - "$f" <<'PY' >> $O/summary.txt
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d["roofline"]
    print(sys.argv[1].split('/')[-1], d["ms_per_step"], "ms", "G", d["config"]["ref_ahead"], "tflop", d["tflop_per_step_executed"], {k:(v["launches"],round(v["ms"],3),round(v["tflops"],1)) for k,v in r["families"].items()}, {k:(v["launches"],v["ms"]) for k,v in r["hbm_families"].items()})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
timeout 300 python tools/profile_step.py --ref-ahead 5 > $O/per_shape_G5.txt 2>&1
cat $O/summary.txt
