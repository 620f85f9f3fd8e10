#!/bin/bash
# refreshed per-kernel PMC table (one SQ pass) of the round-2 kernels, default mainloop and ping-pong mainloop
O=$GRAFT_REPO_ROOT/gpurun_out/r2c22; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/sq -o p -- python $GRAFT_REPO_ROOT/tools/pmc_kernels.py attn conv gemm > $O/sq.log 2>&1
SG_PINGPONG=1 timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pp -o p -- python $GRAFT_REPO_ROOT/tools/pmc_kernels.py conv > $O/pp.log 2>&1
cd $GRAFT_REPO_ROOT
for d in sq pp; do f=$(find $O/$d -name "*counter_collection.csv" | head -1); t=$(find $O/$d -name "*kernel_trace.csv" | head -1); echo "== $d"; python tools/pmc_table.py $f $t | tee $O/$d.txt; done
tail -3 $O/sq.log
