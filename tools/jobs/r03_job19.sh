#!/bin/bash
# round 3, GPU job 19: 8 output slots per wave (two-k-step pair kernels) -- parity, whole suite, fuzz, C5 A/B
O=gpurun_out/r03; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q > $O/gputest19.log 2>&1; echo "pytest rc $?" | tee -a $O/gputest19.log
tail -3 $O/gputest19.log
for s in 51 52; do FUZZ_WIDE=1 SEED=$s CASES=120 timeout 900 python tools/fuzz_long.py 2>&1 | tail -1 | tee -a $O/fuzz_ns8.txt; done
for d in canonical contracted fast; do
python tools/ab_libs.py --workload c5 --paired 1 --detect $d --rounds 5 plain=product,DSABF_WG_WAVES=4,DSABF_COL_TILES=4 w8=product,DSABF_COL_TILES=4 slots8=product 2>&1 | tee -a $O/ab_c5_ns8_final.txt
done
python tools/ab_libs.py --workload c5 --n-freq 128 --paired 1 --rounds 5 plain=product,DSABF_WG_WAVES=4,DSABF_COL_TILES=4 w8=product,DSABF_COL_TILES=4 slots8=product s8t2=product,DSABF_TSPLIT=2 s8t8=product,DSABF_TSPLIT=8 2>&1 | tee -a $O/ab_c5_ns8_final.txt
