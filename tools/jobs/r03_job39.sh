#!/bin/bash
# round 3, GPU job 39: iterative-ilp for the wide translation units: the pair kernels in them (8 slots; 8 waves)
O=gpurun_out/r03; mkdir -p $O
python tools/ab_libs.py --workload c5 --paired 1 --rounds 5 max-ilp=product iterative-ilp=variants/s_iter/libdsabf.so 2>&1 | tee -a $O/ab_sched_general.txt
python tools/ab_libs.py --workload c5 --paired 1 --detect contracted --rounds 5 max-ilp=product iterative-ilp=variants/s_iter/libdsabf.so 2>&1 | tee -a $O/ab_sched_general.txt
python tools/ab_libs.py --workload c5 --paired 1 --rounds 3 max-ilp=product,DSABF_COL_TILES=4 iterative-ilp=variants/s_iter/libdsabf.so,DSABF_COL_TILES=4 2>&1 | tee -a $O/ab_sched_general.txt
python tools/ab_libs.py --workload c5 --n-freq 128 --paired 0 --rounds 5 max-ilp=product iterative-ilp=variants/s_iter/libdsabf.so 2>&1 | tee -a $O/ab_sched_general.txt
