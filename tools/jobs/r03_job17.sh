#!/bin/bash
# round 3, GPU job 17: the general kernel at 4 resident workgroups per CU (fused_min_waves) -- suite, fuzz, time-split sweep
O=gpurun_out/r03; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gputest17.log 2>&1; echo "pytest rc $?" | tee -a $O/gputest17.log; tail -3 $O/gputest17.log
SEED=404 CASES=600 timeout 1200 python tools/fuzz_long.py > $O/fuzz_long2.txt 2>&1; tail -2 $O/fuzz_long2.txt
timeout 900 python tools/ab_libs.py --workload c3 --paired 0 --rounds 3 default=product ts24=product,DSABF_TSPLIT=24 ts28=product,DSABF_TSPLIT=28 ts32=product,DSABF_TSPLIT=32 ts20=product,DSABF_TSPLIT=20 > $O/ab_c3_general_occ4_tsplit.txt 2>&1; cat $O/ab_c3_general_occ4_tsplit.txt | cut -c1-140
