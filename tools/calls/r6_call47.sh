#!/bin/bash
# round 6, call 47: what the 0.5 % of the no-SLP library is: A = every file without SLP (current), B = gemm_conv / attention_bwd / ff_fused only, C = the packed r06cz library
O=$GRAFT_REPO_ROOT/gpurun_out/r6bp; mkdir -p $O
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loop > $O/A_$i.json 2>$O/err.txt; echo "A all no-SLP $(python -c "import json;print(json.load(open('$O/A_$i.json'))['ms_per_step'])")"
  timeout 300 python tools/ab_lib.py storygen_amd/lib/libstorygen_hip_B.so --steps 20 --warmup 5 --no-cpu-baseline --no-loop > $O/B_$i.json 2>$O/err.txt; echo "B three files   $(python -c "import json;print(json.load(open('$O/B_$i.json'))['ms_per_step'])")"
  timeout 300 python tools/ab_lib.py storygen_amd/lib/libstorygen_hip_slp.so --steps 20 --warmup 5 --no-cpu-baseline --no-loop > $O/C_$i.json 2>$O/err.txt; echo "C packed        $(python -c "import json;print(json.load(open('$O/C_$i.json'))['ms_per_step'])")"
done
