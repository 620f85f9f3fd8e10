#!/bin/bash
# conflict-free LDS reads in the GEMM / conv epilogue: parity of every tile x epilogue, PMC conflict counters, full-depth parity,
# contract bench line, PMC traffic of this build
O=$GRAFT_REPO_ROOT/gpurun_out/r2c23; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/test_kernels_gpu.py -q -m gpu --no-header -p no:cacheprovider -x -k "gemm or conv or geglu or stats" 2>&1 | tee $O/kernel_tests.log | tail -n 4
grep -q " failed" $O/kernel_tests.log && exit 1
timeout 300 python -m pytest tests/test_unet_gpu.py -q -m gpu --no-header -p no:cacheprovider -x -k "full_depth" -rP 2>&1 | tee $O/full_depth.log | grep "latent rel-L2\|passed\|failed"
timeout 200 python bench.py --no-cpu-baseline > $O/bench.json 2>/dev/null; cut -c1-260 $O/bench.json
cd /tmp && export TMPDIR=/tmp
C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/sq -o p -- python $GRAFT_REPO_ROOT/tools/pmc_kernels.py conv gemm > $O/sq.log 2>&1
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
timeout 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f -o p -- $CMD > $O/f.log 2>&1
timeout 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/w -o p -- $CMD > $O/w.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_table.py $(find $O/sq -name "*counter_collection.csv" | head -1) $(find $O/sq -name "*kernel_trace.csv" | head -1) | tee $O/sq.txt | cut -c1-230
python tools/traffic_from_pmc.py $O/f/p_counter_collection.csv $O/w/p_counter_collection.csv "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) --kernel-trace -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline; MI355X; round 2, conflict-free epilogue reads; $(date -u +%F)" > $O/traffic.json; head -4 $O/traffic.json
find $O/f $O/w -name "*kernel_trace.csv" -delete; find $O/f $O/w -name "*counter_collection.csv" -delete
