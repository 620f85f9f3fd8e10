#!/bin/bash
# round 6, call 45: the run-to-run difference of the columns-are-tokens fold on the 4-stage latency kernel loses the d_row term in ONE half of a packed fp32 add
# for lanes 48-63 (call 44).  Same experiment on gemm_conv.hip built WITHOUT SLP vectorisation (no v_pk_*_f32 in the epilogue): does it go away?
O=$GRAFT_REPO_ROOT/gpurun_out/r6bn; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python tools/exp_determinism.py only=one-graph reps=40 vt=both:down_blocks.1.attentions.0 > $O/slp.txt 2>&1; echo "shipped build (SLP): $(grep -c bit-identical $O/slp.txt) of 40 bit-identical"
timeout 900 python tools/exp_determinism.py only=one-graph reps=40 vt=both:down_blocks.1.attentions.0 lib=storygen_amd/lib/libstorygen_hip_gcnoslp.so > $O/noslp.txt 2>&1; echo "no-SLP build: $(grep -c bit-identical $O/noslp.txt) of 40 bit-identical"
timeout 900 python tools/exp_determinism.py only=one-graph reps=40 vt=both:down_blocks.1.attentions.0 > $O/slp2.txt 2>&1; echo "shipped build (SLP), again: $(grep -c bit-identical $O/slp2.txt) of 40 bit-identical"
timeout 900 python tools/exp_determinism.py only=one-graph reps=40 vt=both:down_blocks.1.attentions.0 lib=storygen_amd/lib/libstorygen_hip_gcnoslp.so > $O/noslp2.txt 2>&1; echo "no-SLP build, again: $(grep -c bit-identical $O/noslp2.txt) of 40 bit-identical"
