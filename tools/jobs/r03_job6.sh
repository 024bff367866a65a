#!/bin/bash
O=gpurun_out/r03; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "dedisperse or dm or pulse or burst or gather_detected" > $O/gputest6.log 2>&1; echo "pytest rc $?" | tee -a $O/gputest6.log
timeout 600 python tools/dm_ab.py > $O/dm_ab.txt 2>&1; cat $O/dm_ab.txt
bash tools/dm_pmc.sh r03/dm_pmc > $O/dm_pmc.log 2>&1
cat gpurun_out/r03/dm_pmc/summary.txt
rm -rf gpurun_out/r03/dm_pmc/pass*
