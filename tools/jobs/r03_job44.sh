#!/bin/bash
# round 3, GPU job 44: 8-wave kernels: max-ilp vs iterative-ilp (product) vs iterative-maxocc
O=gpurun_out/r03; mkdir -p $O
python tools/ab_libs.py --workload c5 --paired 0 --rounds 5 max-ilp=variants/s_max-ilp/libdsabf.so iterative-ilp=product iterative-maxocc=variants/s_iterative-maxocc/libdsabf.so 2>&1 | tee -a $O/ab_sched_more.txt
python tools/ab_libs.py --workload c5 --paired 1 --rounds 5 max-ilp=variants/s_max-ilp/libdsabf.so,DSABF_COL_TILES=4 iterative-ilp=product,DSABF_COL_TILES=4 iterative-maxocc=variants/s_iterative-maxocc/libdsabf.so,DSABF_COL_TILES=4 2>&1 | tee -a $O/ab_sched_more.txt
python tools/ab_libs.py --workload c5 --n-freq 128 --paired 0 --rounds 5 max-ilp=variants/s_max-ilp/libdsabf.so iterative-ilp=product iterative-maxocc=variants/s_iterative-maxocc/libdsabf.so 2>&1 | tee -a $O/ab_sched_more.txt
python tools/ab_libs.py --workload c5 --paired 0 --detect contracted --rounds 3 max-ilp=variants/s_max-ilp/libdsabf.so iterative-ilp=product iterative-maxocc=variants/s_iterative-maxocc/libdsabf.so 2>&1 | tee -a $O/ab_sched_more.txt
