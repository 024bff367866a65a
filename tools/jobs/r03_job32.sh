#!/bin/bash
# round 3, GPU job 32: C5 with row-tile t8+1's LDS fragments requested one tile early (2 waves per SIMD: less latency cover than C3)
O=gpurun_out/r03; mkdir -p $O
for d in canonical contracted; do
python tools/ab_libs.py --workload c5 --paired 1 --detect $d --rounds 5 base=product ahead=variants/fa5/libdsabf.so 2>&1 | tee -a $O/ab_c5_fragahead.txt
done
python tools/ab_libs.py --workload c5 --paired 0 --rounds 5 base=product ahead=variants/fa5/libdsabf.so 2>&1 | tee -a $O/ab_c5_fragahead.txt
python tools/ab_libs.py --workload c5 --n-freq 128 --paired 1 --rounds 5 base=product ahead=variants/fa5/libdsabf.so 2>&1 | tee -a $O/ab_c5_fragahead.txt
