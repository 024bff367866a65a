#!/bin/bash
# round 3, GPU job 31: one-k-step pair kernel, 512 beams: 8 slots per wave at 3 waves per SIMD against 4 slots at 4
O=gpurun_out/r03; mkdir -p $O
for d in canonical contracted; do
python tools/ab_libs.py --workload c3 --n-beams 512 --units 64 --paired 1 --detect $d --rounds 5 ns4=product ns8=variants/ns8k1/libdsabf.so ns8t=variants/ns8k1/libdsabf.so,DSABF_TSPLIT=12 2>&1 | tee -a $O/ab_k1_ns8.txt
done
