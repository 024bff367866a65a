#!/bin/bash
# round 3, GPU job 49: scheduling strategies for the DM kernels
O=gpurun_out/r03; mkdir -p $O
echo "== max-ilp"; python tools/dm_ab.py 2>&1 | grep -v amdgpu.ids | cut -c1-175 | tee -a $O/dm_ab_sched.txt
for s in iterative-ilp iterative-maxocc; do echo "== $s"; DSABF_LIB_PATH=variants/dm_$s/libdsabf.so python tools/dm_ab.py 2>&1 | grep -v amdgpu.ids | cut -c1-175 | tee -a $O/dm_ab_sched.txt; done
echo "== max-ilp"; python tools/dm_ab.py 2>&1 | grep -v amdgpu.ids | cut -c1-175 | tee -a $O/dm_ab_sched.txt
