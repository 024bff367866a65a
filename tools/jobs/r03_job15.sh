#!/bin/bash
# round 3, GPU job 15: 8-wave workgroups with their launch shape -- whole GPU suite, then C5 and the C5 rank shard
O=gpurun_out/r03; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gputest15.log 2>&1; echo "pytest rc $?" | tee -a $O/gputest15.log
tail -3 $O/gputest15.log
for p in 1 0; do
python tools/ab_libs.py --workload c5 --paired $p --rounds 5 w4=product,DSABF_WG_WAVES=4 w8=product 2>&1 | tee -a $O/ab_c5_w8_final.txt
python tools/ab_libs.py --workload c5 --n-freq 128 --paired $p --rounds 5 w4=product,DSABF_WG_WAVES=4 w8=product w8t1=product,DSABF_TSPLIT=1 w8t2=product,DSABF_TSPLIT=2 w8t8=product,DSABF_TSPLIT=8 2>&1 | tee -a $O/ab_c5_w8_final.txt
done
