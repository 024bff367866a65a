#!/bin/bash
# round 3, GPU job 13: final build -- whole GPU suite, smoke, then the complete r03 profile refresh
O=gpurun_out/r03; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gputest13.log 2>&1; echo "pytest rc $?" | tee -a $O/gputest13.log
tail -3 $O/gputest13.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/refresh_profiles_r03.sh > $O/refresh.log 2>&1
ls gpurun_out/r03p | wc -l
