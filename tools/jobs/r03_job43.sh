#!/bin/bash
# round 3, GPU job 43: the remaining GCN scheduling strategies (iterative-minreg, iterative-maxocc) on the 8-slot, 8-wave and C3 kernels
O=gpurun_out/r03; mkdir -p $O
python tools/ab_libs.py --workload c5 --paired 1 --rounds 3 product=product minreg=variants/s_iterative-minreg/libdsabf.so maxocc=variants/s_iterative-maxocc/libdsabf.so 2>&1 | tee -a $O/ab_sched_more.txt
python tools/ab_libs.py --workload c5 --paired 0 --rounds 3 product=product minreg=variants/s_iterative-minreg/libdsabf.so maxocc=variants/s_iterative-maxocc/libdsabf.so 2>&1 | tee -a $O/ab_sched_more.txt
python tools/ab_libs.py --workload c3 --paired 1 --rounds 3 product=product minreg=variants/s_iterative-minreg/libdsabf.so maxocc=variants/s_iterative-maxocc/libdsabf.so 2>&1 | tee -a $O/ab_sched_more.txt
python tools/ab_libs.py --workload c3 --paired 0 --rounds 3 product=product minreg=variants/s_iterative-minreg/libdsabf.so maxocc=variants/s_iterative-maxocc/libdsabf.so 2>&1 | tee -a $O/ab_sched_more.txt
