#!/bin/bash
# round 3, GPU job 30: soak -- 300 handle lifetimes over 5 geometries, then a 600-block observation
O=gpurun_out/r03; mkdir -p $O
timeout 1500 python tools/soak.py 900 100 2>&1 | grep -v amdgpu.ids | tee $O/soak.txt
