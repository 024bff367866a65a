#!/bin/bash
# round 3, GPU job 24: DM wide kernel with scalar DMA addressing against the committed one
O=gpurun_out/r03; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "dedisperse or dm_kernel or pulse" > $O/gputest24.log 2>&1; echo "pytest rc $?"; tail -2 $O/gputest24.log
for r in 1 2; do
echo "== old"; DSABF_LIB_PATH=variants/dm0/libdsabf.so python tools/dm_ab.py 2>&1 | grep -v amdgpu.ids | cut -c1-200 | tee -a $O/dm_ab_scalar.txt
echo "== new"; python tools/dm_ab.py 2>&1 | grep -v amdgpu.ids | cut -c1-200 | tee -a $O/dm_ab_scalar.txt
done
