#!/bin/bash
# round 4, call 14: GELU of the GEMM epilogues by Abramowitz-Stegun 7.1.26 (13 VALU) instead of ocml erff (~32, both branches), against the
# previous build (libstorygen_hip_erff.so); D = 40 attention unrolled only
set -u
O=gpurun_out/r4m; mkdir -p $O
PREV=storygen_amd/lib/libstorygen_hip_erff.so
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "geglu or feed_forward or attention or gemm" -x > $O/kernel_tests.log 2>&1; echo "kernel tests rc=$?" > $O/summary.txt
timeout 900 python -m pytest tests/test_unet_gpu.py -q -m gpu -k "single_pass or denoise_steps or full_depth" -x > $O/unet_tests.log 2>&1; echo "unet tests rc=$?" >> $O/summary.txt
timeout 900 python -m pytest tests/test_encoders_gpu.py tests/test_backward_gpu.py -q -m gpu -x > $O/enc_bwd_tests.log 2>&1; echo "encoder/backward tests rc=$?" >> $O/summary.txt
for i in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --steps 20 > $O/bench_as_$i.json 2> $O/bench_as_$i.err
  timeout 300 python tools/ab_lib.py $PREV --no-cpu-baseline --steps 20 > $O/bench_erff_$i.json 2> $O/bench_erff_$i.err
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
tail -n 4 $O/kernel_tests.log; tail -n 4 $O/unet_tests.log; tail -n 4 $O/enc_bwd_tests.log; cat $O/summary.txt
