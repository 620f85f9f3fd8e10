#!/bin/bash
# round 3, call 2: anatomy of the new prologue / epilogue, isolated tile sweep, true kernel durations of a step (rocprofv3)
O=$GRAFT_REPO_ROOT/gpurun_out/r3c02; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -m gpu --no-header -p no:cacheprovider -x -k "register" 2>&1 | tail -n 3
timeout 300 python tools/anatomy.py > $O/anatomy.txt 2>&1; tail -n 40 $O/anatomy.txt
timeout 400 python tools/bench_gemm.py > $O/bench_gemm.txt 2>&1; cat $O/bench_gemm.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/kt.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r3c02"
f=glob.glob(O+"/kt/**/*kernel_stats.csv", recursive=True)
if f:
    rows=list(csv.DictReader(open(f[0])))
    open(O+"/kernel_stats.csv","w").write(open(f[0]).read())
    for r in rows[:40]:
        print(r["Name"][:80], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"])
PY
find $O -name "*kernel_trace.csv" -delete
