#!/bin/bash
# round 4, call 15: GEMM / conv mainloop with the ring stage as a compile-time constant (S copies of the slab, per-stage LDS offset registers)
# against the same sources built with -DSG_PIPE_RT_STAGE (run-time stage, rounds 1-4): kernel tests, tile microbench, whole step
set -u
O=gpurun_out/r4n; mkdir -p $O
RT=storygen_amd/lib/libstorygen_hip_rtstage.so
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "gemm or conv or geglu or layernorm_fold or pair or ring" -x > $O/kernel_tests.log 2>&1; echo "kernel tests rc=$?" > $O/summary.txt
timeout 300 python tools/bench_gemm.py > $O/bench_gemm_unrolled.txt 2>&1
timeout 300 python - > $O/bench_gemm_rtstage.txt 2>&1 <<'PY'
import os, sys, runpy
sys.path.insert(0, os.getcwd())
from storygen_amd import _lib
_lib.LIB_PATH = os.path.abspath("storygen_amd/lib/libstorygen_hip_rtstage.so")
sys.argv = ["tools/bench_gemm.py"]
runpy.run_path("tools/bench_gemm.py", run_name="__main__")
PY
for i in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --steps 20 > $O/bench_unrolled_$i.json 2> $O/bench_unrolled_$i.err
  timeout 300 python tools/ab_lib.py $RT --no-cpu-baseline --steps 20 > $O/bench_rtstage_$i.json 2> $O/bench_rtstage_$i.err
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
tail -n 4 $O/kernel_tests.log; paste $O/bench_gemm_unrolled.txt $O/bench_gemm_rtstage.txt | cut -c1-200 | head -70; cat $O/summary.txt
