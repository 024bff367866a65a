#!/bin/bash
# round 3, GPU job 1: the suite on the 3-fragment general image, then A/B of builds / time splits
O=gpurun_out/r03; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/gputest1.log 2>&1; echo "pytest rc $?" | tee -a $O/gputest1.log
G4=variants/gen4/libdsabf.so; W2=variants/w2n8/libdsabf.so
timeout 900 python tools/ab_libs.py --workload c3 --paired 0 --rounds 3 gen3=product gen4=$G4 w2n8=$W2 w2n8_ts13=$W2,DSABF_TSPLIT=13 \
   ts21=product,DSABF_TSPLIT=21 ts24=product,DSABF_TSPLIT=24 ts27=product,DSABF_TSPLIT=27 ts30=product,DSABF_TSPLIT=30 ts36=product,DSABF_TSPLIT=36 > $O/ab_c3_general.txt 2>&1
timeout 600 python tools/ab_libs.py --workload c3 --paired 1 --rounds 3 p=product w2n8=$W2 ts24=product,DSABF_TSPLIT=24 ts28=product,DSABF_TSPLIT=28 ts32=product,DSABF_TSPLIT=32 > $O/ab_c3_paired.txt 2>&1
timeout 600 python tools/ab_libs.py --workload c5 --paired 0 --rounds 2 gen3=product gen4=$G4 > $O/ab_c5_general.txt 2>&1
timeout 600 python tools/ab_libs.py --workload c3 --paired 0 --detect contracted --rounds 2 gen3=product gen4=$G4 w2n8=$W2 > $O/ab_c3_general_contracted.txt 2>&1
cat $O/ab_*.txt
tail -3 $O/gputest1.log
