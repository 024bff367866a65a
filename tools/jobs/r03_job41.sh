#!/bin/bash
# round 3, GPU job 41: other scheduler options for the 8-slot pair kernel
O=gpurun_out/r03; mkdir -p $O
python tools/ab_libs.py --workload c5 --paired 1 --rounds 5 product=product trackers=variants/s8_trk/libdsabf.so bottomup=variants/s8_bu/libdsabf.so bias0=variants/s8_bias/libdsabf.so 2>&1 | tee -a $O/ab_s8_sched.txt
