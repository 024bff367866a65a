#!/bin/bash
# round 3, GPU job 46: further scheduler options on top of iterative-maxocc for the 8-wave general kernel
O=gpurun_out/r03; mkdir -p $O
python tools/ab_libs.py --workload c5 --paired 0 --rounds 5 product=product trackers=variants/g_trk/libdsabf.so bottomup=variants/g_bu/libdsabf.so bias0=variants/g_bias/libdsabf.so 2>&1 | tee -a $O/ab_sched_more2.txt
