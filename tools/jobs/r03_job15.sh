#!/bin/bash
O=gpurun_out/r03; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "dedisperse or dm or pulse or burst or gather_detected" > $O/gputest15.log 2>&1; echo "pytest rc $?" | tee -a $O/gputest15.log; tail -2 $O/gputest15.log
SEED=11 CASES=300 timeout 900 python tools/fuzz_dm.py 2>&1 | tail -2
timeout 600 python tools/dm_ab.py 2>&1 | grep -v amdgpu.ids > $O/dm_ab.txt; cat $O/dm_ab.txt | cut -c1-220
