#!/bin/bash
# round 4, call 13: attention tile loop with the ring stage as a compile-time constant (S copies of the body) + scalar-select DMA sources,
# against the same sources built with -DSG_ATTN_RT_STAGE (run-time stage, rounds 1-4), kernel level and whole step
set -u
O=gpurun_out/r4l; mkdir -p $O
RT=storygen_amd/lib/libstorygen_hip_rtstage.so
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "attention or attn" -x > $O/attn_tests.log 2>&1; echo "attention tests rc=$?" > $O/summary.txt
timeout 300 python tools/bench_norm.py --attn > $O/attn_micro_unrolled.txt 2>&1
timeout 300 python - > $O/attn_micro_rtstage.txt 2>&1 <<'PY'
import os, sys, runpy
sys.path.insert(0, os.getcwd())
from storygen_amd import _lib
_lib.LIB_PATH = os.path.abspath("storygen_amd/lib/libstorygen_hip_rtstage.so")
sys.argv = ["tools/bench_norm.py", "--attn"]
runpy.run_path("tools/bench_norm.py", run_name="__main__")
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
tail -n 5 $O/attn_tests.log; paste $O/attn_micro_unrolled.txt $O/attn_micro_rtstage.txt | cut -c1-160; cat $O/summary.txt
