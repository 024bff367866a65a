#!/bin/bash
# round 3, GPU job 17: fuzz biased to the 8-wave workgroups + the whole contract again on the final build
O=gpurun_out/r03; mkdir -p $O
for s in 31 32 33; do FUZZ_WIDE=1 SEED=$s CASES=120 timeout 900 python tools/fuzz_long.py 2>&1 | tail -3 | tee -a $O/fuzz_w8.txt; done
for s in 41 42; do SEED=$s CASES=200 timeout 900 python tools/fuzz_long.py 2>&1 | tail -3 | tee -a $O/fuzz_w8.txt; done
