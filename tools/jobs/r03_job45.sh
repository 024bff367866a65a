#!/bin/bash
# round 3, GPU job 45: per-family scheduling strategies shipped (8-wave general: iterative-maxocc, 8-wave pair: iterative-ilp)
O=gpurun_out/r03; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q > $O/gputest45.log 2>&1; echo "pytest rc $?" | tee -a $O/gputest45.log
tail -3 $O/gputest45.log
FUZZ_WIDE=1 SEED=91 CASES=250 timeout 1200 python tools/fuzz_long.py 2>&1 | tail -1 | tee -a $O/fuzz_sched2.txt
python tools/ab_libs.py --workload c5 --paired 0 --rounds 5 all-max-ilp=variants/allmaxilp/libdsabf.so product=product 2>&1 | tee -a $O/ab_sched_final2.txt
python tools/ab_libs.py --workload c5 --paired 1 --rounds 3 all-max-ilp=variants/allmaxilp/libdsabf.so,DSABF_COL_TILES=4 product=product,DSABF_COL_TILES=4 2>&1 | tee -a $O/ab_sched_final2.txt
python tools/ab_libs.py --workload c5 --paired 0 --weights calibrated --rounds 3 all-max-ilp=variants/allmaxilp/libdsabf.so product=product 2>&1 | tee -a $O/ab_sched_final2.txt
