#!/bin/bash
# round 3, GPU job 27: fuzz on the final build -- fused kernel over the whole contract, wide launches, DM kernels
O=gpurun_out/r03; mkdir -p $O
for s in 61 62; do SEED=$s CASES=300 timeout 1200 python tools/fuzz_long.py 2>&1 | tail -1 | tee -a $O/fuzz_final.txt; done
for s in 63 64; do FUZZ_WIDE=1 SEED=$s CASES=150 timeout 1200 python tools/fuzz_long.py 2>&1 | tail -1 | tee -a $O/fuzz_final.txt; done
for s in 71 72 73; do SEED=$s CASES=100 timeout 900 python tools/fuzz_dm.py 2>&1 | tail -1 | tee -a $O/fuzz_final.txt; done
