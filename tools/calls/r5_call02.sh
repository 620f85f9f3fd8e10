#!/bin/bash
# round 5, call 2: LayerNorm-fold guard + shared CFG head parity, same-box A/B of the shared head, tile table for the batch-20 shapes
# of the group schedule, SQ counter passes on the hot kernels
O=$GRAFT_REPO_ROOT/gpurun_out/r5b; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -s -k "fold or guard or gemm" > $O/kernel_tests.log 2>&1; echo "kernel tests rc=$?" > $O/summary.txt
tail -3 $O/kernel_tests.log >> $O/summary.txt; grep "LayerNorm fold, |mean|" $O/kernel_tests.log >> $O/summary.txt
timeout 900 python -m pytest tests/test_unet_gpu.py -q -m gpu -x -s -k "shared_cfg or guard or group_schedule or full_depth or tabulated or unet_passes" > $O/unet_tests.log 2>&1; echo "unet tests rc=$?" >> $O/summary.txt
tail -3 $O/unet_tests.log >> $O/summary.txt; grep "shared head\|stream offset\|ref_ahead=" $O/unet_tests.log >> $O/summary.txt
for v in "" "--no-shared-head" "" "--no-shared-head"; do
  timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 10 $v > $O/bench_head${v}_$RANDOM.json 2>> $O/bench.err
done
timeout 600 python tools/tune_tiles.py --ref-ahead 5 > $O/tune_tiles.log 2>&1; tail -2 $O/tune_tiles.log >> $O/summary.txt
cp storygen_amd/tuning/mi355x_tiles.json $O/mi355x_tiles.json
for i in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 10 > $O/bench_tuned_$i.json 2>> $O/bench.err
done
for f in $O/bench_*.json; do python - "$f" <<'PY' >> $O/summary.txt
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d["roofline"]
    print(sys.argv[1].split('/')[-1], d["ms_per_step"], "ms", "G", d["config"]["ref_ahead"], "head", d["config"]["shared_cfg_head_of_main_pass"], "tflop", d["tflop_per_step_executed"], {k:(v["launches"],round(v["ms"],3),round(v["tflops"],1)) for k,v in r["families"].items()}, {k:(v["launches"],v["ms"]) for k,v in r["hbm_families"].items()})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
cat $O/summary.txt
cd /tmp && export TMPDIR=/tmp
C1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
C2="SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
timeout 300 rocprofv3 --pmc $C1 --kernel-trace --output-format csv -d $O/p1 -o p -- python $GRAFT_REPO_ROOT/tools/pmc_kernels.py attn conv gemm ff > $O/p1.log 2>&1
timeout 300 rocprofv3 --pmc $C2 --kernel-trace --output-format csv -d $O/p2 -o p -- python $GRAFT_REPO_ROOT/tools/pmc_kernels.py attn conv gemm ff > $O/p2.log 2>&1
cd $GRAFT_REPO_ROOT
for n in 1 2; do
  F=$(find $O/p$n -name "*counter_collection.csv" | head -1); T=$(find $O/p$n -name "*kernel_trace.csv" | head -1)
  python tools/pmc_table.py $F $T > $O/pmc_pass$n.txt 2>&1
done
head -50 $O/pmc_pass1.txt; head -50 $O/pmc_pass2.txt
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete
