#!/bin/bash
# round 3, GPU job 29: C5 general kernel with 8 slots per wave at ONE wave per SIMD (368 registers, accumulators in AGPRs)
O=gpurun_out/r03; mkdir -p $O
python tools/ab_libs.py --workload c5 --paired 0 --rounds 5 w8=product g8=variants/g8/libdsabf.so g8t2=variants/g8/libdsabf.so,DSABF_TSPLIT=2 g8t4=variants/g8/libdsabf.so,DSABF_TSPLIT=4 2>&1 | tee -a $O/ab_c5_general_ns8.txt
