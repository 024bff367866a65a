#!/bin/bash
# round 3, GPU job 22: C3 pair kernel with the +-P2 / +-P4 of the conjugate pairs chained on the MFMA pipe (5 / 6 MFMAs per pair tile)
O=gpurun_out/r03; mkdir -p $O
for d in canonical contracted fast; do
python tools/ab_libs.py --workload c3 --paired 1 --detect $d --rounds 5 base=product pm5=variants/pm5/libdsabf.so pm6=variants/pm6/libdsabf.so 2>&1 | tee -a $O/ab_c3_pairmfma.txt
done
