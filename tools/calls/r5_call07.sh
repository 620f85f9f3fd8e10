#!/bin/bash
# round 5, call 7: convolution K order with the taps innermost (channel-block-major) + grouped tile order — parity of every GEMM / conv
# kernel test, the UNet parity file, then a same-box A/B against the previous commit's library and PMC traffic of the new build
O=$GRAFT_REPO_ROOT/gpurun_out/r5g; mkdir -p $O
cd $GRAFT_REPO_ROOT
ALT=storygen_amd/lib/libstorygen_hip_prev.so
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x > $O/kernel_tests.log 2>&1; echo "kernel tests rc=$?" > $O/summary.txt; tail -2 $O/kernel_tests.log >> $O/summary.txt
timeout 900 python -m pytest tests/test_unet_gpu.py -q -m gpu -x -s -k "full_depth or unet_passes or denoise_steps or shared_cfg or two_samples" > $O/unet_tests.log 2>&1; echo "unet tests rc=$?" >> $O/summary.txt; tail -2 $O/unet_tests.log >> $O/summary.txt; grep "latent rel-L2 at steps" $O/unet_tests.log >> $O/summary.txt
timeout 600 python -m pytest tests/test_backward_gpu.py -q -m gpu -x > $O/backward_tests.log 2>&1; echo "backward tests rc=$?" >> $O/summary.txt
cat $O/summary.txt
for i in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 10 > $O/bench_new_$i.json 2> $O/bench_new_$i.err
  timeout 300 python tools/ab_lib.py $ALT --no-cpu-baseline --steps 20 --warmup 10 > $O/bench_prev_$i.json 2> $O/bench_prev_$i.err
done
for f in $O/bench_*.json; do python - "$f" <<'PY' >> $O/summary.txt
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d["roofline"]
    print(sys.argv[1].split('/')[-1], d["ms_per_step"], "ms", {k:(v["launches"],round(v["ms"],3),round(v["tflops"],1)) for k,v in r["families"].items()}, {k:(v["launches"],v["ms"]) for k,v in r["hbm_families"].items()})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
tail -4 $O/summary.txt
timeout 300 python tools/profile_step.py --ref-ahead 5 > $O/per_shape.txt 2>&1
timeout 300 python bench.py --steps 5 --warmup 5 --no-cpu-baseline --dump-algorithmic $O/algorithmic.json > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 5 --no-cpu-baseline"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f -o p -- $CMD > $O/f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/w -o p -- $CMD > $O/w.log 2>&1
cd $GRAFT_REPO_ROOT
F=$(find $O/f -name "*counter_collection.csv" | head -1); W=$(find $O/w -name "*counter_collection.csv" | head -1)
python tools/traffic_from_pmc.py $F $W "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) --kernel-trace -- python bench.py --steps 5 --warmup 5 --no-cpu-baseline (ref_ahead 5); algorithmic bytes: bench.py --dump-algorithmic of the same build; MI355X; round 5, taps-innermost convolution + grouped tile order; $(date -u +%F)" $O/algorithmic.json > $O/traffic.json; python - <<'PY'
import json
t=json.load(open("gpurun_out/r5g/traffic.json"))
print(t["kernels"]["mma_pipe_kernel (gemm + conv3x3)"])
for c in t["top_traffic_classes"][:12]: print({k:v for k,v in c.items() if k!="shapes"})
print(t["total_hbm_gbytes_all_kernels"])
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete
