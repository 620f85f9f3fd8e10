#!/bin/bash
# round 6, call 7: is the rare run-to-run difference of the 5-step loop new?  30 repeats of the one-graph schedule on the round-5 tree and on this tree
# (with and without the latency kernel / the merged GEMM)
O=$GRAFT_REPO_ROOT/gpurun_out/r6g; mkdir -p $O
cd $GRAFT_REPO_ROOT/.old_r5 && timeout 900 python tools/exp_determinism.py reps=30 only=one-graph > $O/det_round5.txt 2>&1; grep -c "bit-identical" $O/det_round5.txt; grep -v "bit-identical" $O/det_round5.txt | tail -5
cd $GRAFT_REPO_ROOT
timeout 900 python tools/exp_determinism.py reps=30 only=one-graph > $O/det_now.txt 2>&1; grep -c "bit-identical" $O/det_now.txt; grep -v "bit-identical" $O/det_now.txt | tail -5
timeout 900 python tools/exp_determinism.py reps=30 only=one-graph nolat > $O/det_nolat.txt 2>&1; grep -c "bit-identical" $O/det_nolat.txt; grep -v "bit-identical" $O/det_nolat.txt | tail -5
timeout 900 python tools/exp_determinism.py reps=30 only=one-graph nomerge > $O/det_nomerge.txt 2>&1; grep -c "bit-identical" $O/det_nomerge.txt; grep -v "bit-identical" $O/det_nomerge.txt | tail -5
timeout 900 python tools/exp_determinism.py reps=30 only=eager > $O/det_eager.txt 2>&1; grep -c "bit-identical" $O/det_eager.txt; grep -v "bit-identical" $O/det_eager.txt | tail -5
