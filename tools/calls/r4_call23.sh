#!/bin/bash
# round 4, call 23: operand base of every slab from running counters (channel block, tap column, tap row, offset) against the
# previous build (libstorygen_hip_prev.so: multiply-shift division + 64-bit products per slab); PMC passes on the new build in the same call
O=$GRAFT_REPO_ROOT/gpurun_out/r4v; mkdir -p $O
cd $GRAFT_REPO_ROOT
ALT=storygen_amd/lib/libstorygen_hip_prev.so
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "gemm or conv or pair or ring or fold or geglu" -x > $O/kernel_tests.log 2>&1; echo "kernel tests rc=$?" > $O/summary.txt
timeout 600 python -m pytest tests/test_unet_gpu.py -q -m gpu -k "single_pass or denoise_steps" -x > $O/unet_tests.log 2>&1; echo "unet tests rc=$?" >> $O/summary.txt
timeout 300 python tools/bench_gemm.py > $O/bench_gemm_new.txt 2>&1
timeout 300 python - > $O/bench_gemm_prev.txt 2>&1 <<'PY'
import os, sys, runpy
sys.path.insert(0, os.getcwd())
from storygen_amd import _lib
_lib.LIB_PATH = os.path.abspath("storygen_amd/lib/libstorygen_hip_prev.so")
sys.argv = ["tools/bench_gemm.py"]
runpy.run_path("tools/bench_gemm.py", run_name="__main__")
PY
for i in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --steps 20 > $O/bench_new_$i.json 2> $O/bench_new_$i.err
  timeout 300 python tools/ab_lib.py $ALT --no-cpu-baseline --steps 20 > $O/bench_prev_$i.json 2> $O/bench_prev_$i.err
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
a=open('gpurun_out/r4v/bench_gemm_new.txt').read().splitlines()
b=open('gpurun_out/r4v/bench_gemm_prev.txt').read().splitlines()
for x,y in zip(a,b):
    m1=re.findall(r'([\d.]+)\|',x); m2=re.findall(r'([\d.]+)\|',y)
    if m1 and m2:
        print(f"{x[:26]:26s} auto: new {m1[0]:>6s} vs prev {m2[0]:>6s} us ({(float(m1[0])/float(m2[0])-1)*100:+.1f}%)   256x128: {m1[1]} vs {m2[1]}  128x128: {m1[2]} vs {m2[2]}  256x64: {m1[3]} vs {m2[3]}")
PY
cat $O/summary.txt
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f -o p -- $CMD > $O/f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/w -o p -- $CMD > $O/w.log 2>&1
cd $GRAFT_REPO_ROOT
F=$(find $O/f -name "*counter_collection.csv" | head -1); W=$(find $O/w -name "*counter_collection.csv" | head -1)
python tools/traffic_from_pmc.py $F $W "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) --kernel-trace -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline; MI355X; round 4, final sources (call 23); $(date -u +%F)" > $O/traffic.json; head -c 300 $O/traffic.json
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
