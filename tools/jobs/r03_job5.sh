#!/bin/bash
# round 3, GPU job 5: counters of the shared-window DM kernel
O=gpurun_out/r03; mkdir -p $O
bash tools/dm_pmc.sh r03/dm_pmc > $O/dm_pmc.log 2>&1
cat gpurun_out/r03/dm_pmc/summary.txt
rm -rf gpurun_out/r03/dm_pmc/pass*
