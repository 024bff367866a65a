#!/bin/bash
# round 3, GPU job 34: ABLATION (timing only, wrong results): every MFMA chain starts from the inline constant 0 instead of the
# seed registers -- does the C-operand register read cost issue time?
O=gpurun_out/r03; mkdir -p $O
python tools/ab_libs.py --workload c3 --paired 1 --rounds 5 base=product noseed=variants/noseed/libdsabf.so 2>&1 | tee -a $O/ab_noseed.txt
python tools/ab_libs.py --workload c3 --paired 0 --rounds 5 base=product noseed=variants/noseed/libdsabf.so 2>&1 | tee -a $O/ab_noseed.txt
