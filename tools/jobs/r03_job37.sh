#!/bin/bash
# round 3, GPU job 37: LLVM scheduling strategies for the C5 kernels (round 2 tested them on C3 only)
O=gpurun_out/r03; mkdir -p $O
for p in 1 0; do
python tools/ab_libs.py --workload c5 --paired $p --rounds 5 max-ilp=product iterative-ilp=variants/s_iterative-ilp/libdsabf.so max-memory-clause=variants/s_max-memory-clause/libdsabf.so 2>&1 | tee -a $O/ab_c5_sched.txt
done
