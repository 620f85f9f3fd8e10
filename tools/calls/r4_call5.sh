#!/bin/bash
# round 4, call 5: re-run what call 4 found broken (N = 2 noise indexing, GroupNorm geometry), 2-stage ring A/B, schedule probe
set -u
O=gpurun_out/r4e; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "groupnorm or deferred or two_stage or attention" > $O/kernel_tests.log 2>&1; echo "kernel tests rc=$?" >> $O/summary.txt
timeout 900 python -m pytest tests/test_unet_gpu.py -q -k "two_samples or single_pass or denoise_steps or full_depth" > $O/unet_tests.log 2>&1; echo "unet tests rc=$?" >> $O/summary.txt
for v in default:"" stages2:"SG_PIPE_STAGES=2" ; do
  n=${v%%:*}; e=${v#*:}
  env SG_DEV_OPTIONS=1 $e timeout 300 python bench.py --no-cpu-baseline --steps 20 > $O/bench_$n.json 2> $O/bench_$n.err
done
timeout 300 python bench.py --no-cpu-baseline --steps 20 --no-splitk-in-gn > $O/bench_nosplitkgn.json 2> $O/bench_nosplitkgn.err
env SG_DEV_OPTIONS=1 SG_PIPE_STAGES=2 timeout 300 python bench.py --no-cpu-baseline --steps 20 > $O/bench_stages2b.json 2> $O/bench_stages2b.err
timeout 300 python bench.py --no-cpu-baseline --steps 20 > $O/bench_default2.json 2> $O/bench_default2.err
timeout 300 python tools/probe_schedule.py > $O/probe_schedule.txt 2>&1
timeout 400 python tools/profile_step.py > $O/per_shape.txt 2>&1
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
tail -n 6 $O/kernel_tests.log $O/unet_tests.log; cat $O/probe_schedule.txt; cat $O/summary.txt
