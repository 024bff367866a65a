#!/bin/bash
# round 3, GPU job 38: iterative-ilp scheduling for the GENERAL kernels: C3 (canonical, contracted, calibrated), C5, C5 with the plain launch
O=gpurun_out/r03; mkdir -p $O
for d in canonical contracted; do
python tools/ab_libs.py --workload c3 --paired 0 --detect $d --rounds 5 max-ilp=product iterative-ilp=variants/s_iter/libdsabf.so 2>&1 | tee -a $O/ab_sched_general.txt
done
python tools/ab_libs.py --workload c3 --paired 0 --weights calibrated --rounds 5 max-ilp=product iterative-ilp=variants/s_iter/libdsabf.so 2>&1 | tee -a $O/ab_sched_general.txt
python tools/ab_libs.py --workload c5 --paired 0 --detect contracted --rounds 5 max-ilp=product iterative-ilp=variants/s_iter/libdsabf.so 2>&1 | tee -a $O/ab_sched_general.txt
python tools/ab_libs.py --workload c5 --paired 0 --rounds 3 max-ilp=product,DSABF_WG_WAVES=4 iterative-ilp=variants/s_iter/libdsabf.so,DSABF_WG_WAVES=4 2>&1 | tee -a $O/ab_sched_general.txt
python tools/ab_libs.py --workload c3 --paired 1 --rounds 5 max-ilp=product iterative-ilp=variants/s_iter/libdsabf.so 2>&1 | tee -a $O/ab_sched_general.txt
