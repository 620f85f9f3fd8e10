#!/bin/bash
O=gpurun_out/r2c2; mkdir -p $O
timeout 600 python -m pytest tests/test_backward_gpu.py tests/test_dropin_gpu.py -q -m gpu --no-header -p no:cacheprovider -x 2>&1 | tee $O/tests1.log | tail -n 30
timeout 600 python -m pytest tests/test_unet_gpu.py -q -m gpu --no-header -p no:cacheprovider -k "stage_no or pndm or split_graphs" 2>&1 | tee $O/tests2.log | tail -n 30
for S in 3 4 2; do
  SG_STAGES=$S timeout 300 python tools/exp_feed.py > $O/feed_s$S.log 2>&1
  grep -c "auto cold" $O/feed_s$S.log
done
grep "auto cold" $O/feed_s3.log
