#!/bin/bash
# round 3, GPU job 40: per-file scheduling strategy shipped (8-wave kernels: iterative-ilp) -- whole suite, fuzz, C5 A/B against max-ilp
O=gpurun_out/r03; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q > $O/gputest40.log 2>&1; echo "pytest rc $?" | tee -a $O/gputest40.log
tail -3 $O/gputest40.log
FUZZ_WIDE=1 SEED=81 CASES=200 timeout 1200 python tools/fuzz_long.py 2>&1 | tail -1 | tee -a $O/fuzz_sched.txt
for p in 0 1; do
python tools/ab_libs.py --workload c5 --paired $p --rounds 5 w8-max-ilp=variants/w8maxilp/libdsabf.so product=product 2>&1 | tee -a $O/ab_sched_final.txt
done
python tools/ab_libs.py --workload c5 --paired 1 --rounds 3 w8-max-ilp=variants/w8maxilp/libdsabf.so,DSABF_COL_TILES=4 product=product,DSABF_COL_TILES=4 2>&1 | tee -a $O/ab_sched_final.txt
