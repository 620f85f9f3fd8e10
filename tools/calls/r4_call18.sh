#!/bin/bash
# round 4, call 18: refill spread over the k-steps on the tiles of <= 4 waves (default) against the burst everywhere (-DSG_PIPE_BURST);
# then the tile table re-tuned on this build (tools/tune_tiles.py) and the step with it
set -u
O=gpurun_out/r4q; mkdir -p $O
ALT=storygen_amd/lib/libstorygen_hip_burst.so
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "gemm or conv or pair or ring or fold" -x > $O/kernel_tests.log 2>&1; echo "kernel tests rc=$?" > $O/summary.txt
for i in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --steps 20 > $O/bench_default_$i.json 2> $O/bench_default_$i.err
  timeout 300 python tools/ab_lib.py $ALT --no-cpu-baseline --steps 20 > $O/bench_burst_$i.json 2> $O/bench_burst_$i.err
done
cp storygen_amd/tuning/mi355x_tiles.json $O/tiles_before.json
timeout 400 python tools/tune_tiles.py > $O/tune.log 2>&1
cp storygen_amd/tuning/mi355x_tiles.json $O/tiles_after.json
for i in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --steps 20 > $O/bench_retuned_$i.json 2> $O/bench_retuned_$i.err
done
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
tail -n 3 $O/kernel_tests.log; tail -n 4 $O/tune.log; cat $O/summary.txt
