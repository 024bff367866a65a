#!/bin/bash
# round 3, GPU job 20: C3 with row-tile t8+1's LDS fragments requested one tile early (now that the pair kernel is held to 128 VGPRs)
O=gpurun_out/r03; mkdir -p $O
for d in canonical contracted; do
python tools/ab_libs.py --workload c3 --paired 1 --detect $d --rounds 5 base=product ahead=variants/fa/libdsabf.so 2>&1 | tee -a $O/ab_c3_fragahead.txt
done
python tools/ab_libs.py --workload c3 --paired 0 --rounds 3 base=product ahead=variants/fa/libdsabf.so 2>&1 | tee -a $O/ab_c3_fragahead.txt
