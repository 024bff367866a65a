#!/bin/bash
# round 3, GPU job 35: DM wide kernel with a deeper window ring (4 / 5 LDS buffers: DMA 3 / 4 channels ahead)
O=gpurun_out/r03; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "dedisperse or dm_kernel or pulse" 2>&1 | tail -2
for v in nb4 nb5; do DSABF_LIB_PATH=variants/$v/libdsabf.so timeout 900 python -m pytest tests -m gpu -x -q -k "dedisperse or dm_kernel or pulse" 2>&1 | tail -1; done
for r in 1 2; do
echo "== nbuf 3"; python tools/dm_ab.py 2>&1 | grep -v amdgpu.ids | cut -c1-175 | tee -a $O/dm_ab_nbuf.txt
echo "== nbuf 4"; DSABF_LIB_PATH=variants/nb4/libdsabf.so python tools/dm_ab.py 2>&1 | grep -v amdgpu.ids | cut -c1-175 | tee -a $O/dm_ab_nbuf.txt
echo "== nbuf 5"; DSABF_LIB_PATH=variants/nb5/libdsabf.so python tools/dm_ab.py 2>&1 | grep -v amdgpu.ids | cut -c1-175 | tee -a $O/dm_ab_nbuf.txt
done
