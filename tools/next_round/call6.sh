#!/bin/bash
O=gpurun_out/r2c6; mkdir -p $O
timeout 300 python tools/anatomy.py > $O/anatomy.txt 2>&1; cat $O/anatomy.txt | cut -c1-260
