#!/bin/bash
# round 3, GPU job 10: the whole suite on the final build, then the r03 profile refresh
O=gpurun_out/r03; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gputest10.log 2>&1; echo "pytest rc $?" | tee -a $O/gputest10.log
tail -4 $O/gputest10.log
bash tools/refresh_profiles_r03.sh > $O/refresh.log 2>&1
ls gpurun_out/r03p
