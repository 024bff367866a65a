#!/bin/bash
O=gpurun_out/r03; mkdir -p $O
for i in 1 2; do
echo "== product (32 times x 64 beams per tile)"; timeout 600 python tools/dm_ab.py 2>&1 | grep -v amdgpu.ids
echo "== variant dwtb16 (16 times x 128 beams per tile)"; DSABF_LIB_PATH=variants/dwtb16/libdsabf.so timeout 600 python tools/dm_ab.py 2>&1 | grep -v amdgpu.ids
done > $O/dm_ab_waves.txt
cat $O/dm_ab_waves.txt
