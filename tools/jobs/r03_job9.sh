#!/bin/bash
O=gpurun_out/r03; mkdir -p $O
for v in product dwx_NODMA dwx_NOCOMP dwx_NOBAR; do
  echo "== $v"; if [ $v = product ]; then unset DSABF_LIB_PATH; else export DSABF_LIB_PATH=variants/$v/libdsabf.so; fi
  timeout 300 python tools/dm_ab.py 2>&1 | grep -v amdgpu.ids | head -2 | cut -c1-200
done > $O/dm_ablate.txt 2>&1
cat $O/dm_ablate.txt
