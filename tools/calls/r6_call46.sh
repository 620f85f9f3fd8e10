#!/bin/bash
# round 6, call 46: the whole library without SLP vectorisation (no packed fp32 VALU): kernel + UNet suites, run-to-run identity with EVERYTHING on the
# latency kernel (lat_mask 63, pairs included), contract step vs the packed build (libstorygen_hip_slp.so = the r06cz library) on one box
O=$GRAFT_REPO_ROOT/gpurun_out/r6bo; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_unet_gpu.py tests/test_backward_gpu.py -x -q -m gpu > $O/tests.log 2>&1; grep -E "passed|failed" $O/tests.log
SG_LAT_MASK=63 timeout 900 python tools/exp_determinism.py only=one-graph reps=30 > $O/det_mask63.txt 2>&1; echo "lat_mask 63 (paired launches on the latency kernel too), no-SLP build: $(grep -c bit-identical $O/det_mask63.txt) of 30 bit-identical"
SG_LAT_MASK=63 timeout 900 python tools/exp_determinism.py only=one-graph reps=30 nopairs > $O/det_mask63_nopairs.txt 2>&1; echo "... every pair as two plain launches free to take it: $(grep -c bit-identical $O/det_mask63_nopairs.txt) of 30 bit-identical"
SG_LAT_MASK=63 timeout 900 python tools/exp_determinism.py only=one-graph reps=30 lib=storygen_amd/lib/libstorygen_hip_slp.so > $O/det_mask63_slp.txt 2>&1; echo "lat_mask 63, packed build: $(grep -c bit-identical $O/det_mask63_slp.txt) of 30 bit-identical"
for i in 1 2 3; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loop > $O/noslp_$i.json 2>$O/err.txt; echo "no-SLP $(python -c "import json;print(json.load(open('$O/noslp_$i.json'))['ms_per_step'])")"
  timeout 300 python tools/ab_lib.py storygen_amd/lib/libstorygen_hip_slp.so --steps 20 --warmup 5 --no-cpu-baseline --no-loop > $O/slp_$i.json 2>$O/err_slp.txt; echo "packed $(python -c "import json;print(json.load(open('$O/slp_$i.json'))['ms_per_step'])")"
done
