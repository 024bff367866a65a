#!/bin/bash
# round 3, GPU job 14: 8-wave workgroups of the two-k-step classes -- parity, then C5 A/B incl. time splits
O=gpurun_out/r03; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_round3.py tests/test_gpu_parity.py -m gpu -x -q -k "eight_wave or wide_antenna or dsa100 or fuzz or many_chunks or beam_groups" > $O/gputest14.log 2>&1; echo "pytest rc $?" | tee -a $O/gputest14.log
tail -3 $O/gputest14.log
for p in 1 0; do
python tools/ab_libs.py --workload c5 --paired $p --rounds 5 w4=product,DSABF_WG_WAVES=4 w8=product w8t1=product,DSABF_TSPLIT=1 w8t3=product,DSABF_TSPLIT=3 w8t4=product,DSABF_TSPLIT=4 2>&1 | tee -a $O/ab_c5_w8.txt
done
python tools/ab_libs.py --workload c5 --paired 1 --detect contracted --rounds 5 w4=product,DSABF_WG_WAVES=4 w8=product 2>&1 | tee -a $O/ab_c5_w8.txt
