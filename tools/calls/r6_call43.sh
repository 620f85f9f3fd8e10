#!/bin/bash
# round 6, call 43: is the generalised epilogue / byte-offset mainloop (current gemm_conv.hip) as fast as the one of the r06bz closing run?  Same box, alternating.
O=$GRAFT_REPO_ROOT/gpurun_out/r6bl; mkdir -p $O
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loop > $O/cur_$i.json 2>$O/err.txt; echo "current $(python -c "import json;print(json.load(open('$O/cur_$i.json'))['ms_per_step'])")"
  timeout 300 python tools/ab_lib.py storygen_amd/lib/libstorygen_hip_oldgc.so --steps 20 --warmup 5 --no-cpu-baseline --no-loop > $O/old_$i.json 2>$O/err_old.txt; echo "old gemm_conv $(python -c "import json;print(json.load(open('$O/old_$i.json'))['ms_per_step'])")"
done
tail -2 $O/err_old.txt
