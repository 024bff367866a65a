#!/bin/bash
# round 3, GPU job 33: ABLATION (timing only): the per-chunk workgroup barrier removed -- how much of the time is the rendezvous?
O=gpurun_out/r03; mkdir -p $O
python tools/ab_libs.py --workload c3 --paired 1 --rounds 5 base=product nobar=variants/nobar/libdsabf.so 2>&1 | tee -a $O/ab_nobar.txt
python tools/ab_libs.py --workload c3 --paired 0 --rounds 3 base=product nobar=variants/nobar/libdsabf.so 2>&1 | tee -a $O/ab_nobar.txt
python tools/ab_libs.py --workload c5 --paired 1 --rounds 3 base=product nobar=variants/nobar/libdsabf.so 2>&1 | tee -a $O/ab_nobar.txt
