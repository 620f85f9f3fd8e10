#!/bin/bash
# round 4, call 19: 8-wave 256x128 tile with the refill burst of the second wave of every SIMD two k-steps later (-DSG_PIPE_STAGGER) against the default
set -u
O=gpurun_out/r4r; mkdir -p $O
ALT=storygen_amd/lib/libstorygen_hip_alt.so
timeout 300 python tools/bench_gemm.py > $O/bench_gemm_default.txt 2>&1
timeout 300 python - > $O/bench_gemm_alt.txt 2>&1 <<'PY'
import os, sys, runpy
sys.path.insert(0, os.getcwd())
from storygen_amd import _lib
_lib.LIB_PATH = os.path.abspath("storygen_amd/lib/libstorygen_hip_alt.so")
sys.argv = ["tools/bench_gemm.py"]
runpy.run_path("tools/bench_gemm.py", run_name="__main__")
PY
for i in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --steps 20 > $O/bench_default_$i.json 2> $O/bench_default_$i.err
  timeout 300 python tools/ab_lib.py $ALT --no-cpu-baseline --steps 20 > $O/bench_alt_$i.json 2> $O/bench_alt_$i.err
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
python - <<'PY'
import re
a=open('gpurun_out/r4r/bench_gemm_default.txt').read().splitlines()
b=open('gpurun_out/r4r/bench_gemm_alt.txt').read().splitlines()
for x,y in zip(a,b):
    m1=re.findall(r'([\d.]+)\|',x); m2=re.findall(r'([\d.]+)\|',y)
    if m1 and m2:
        print(f"{x[:26]:26s} auto: default {m1[0]:>6s} vs alt {m2[0]:>6s} us ({(float(m2[0])/float(m1[0])-1)*100:+.1f}%)   256x128: {m1[1]} vs {m2[1]}")
PY
cat $O/summary.txt
