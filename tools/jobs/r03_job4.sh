#!/bin/bash
# round 3, GPU job 4: the shared-window DM kernel -- parity tests, A/B against the per-thread-window kernel
O=gpurun_out/r03; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "dedisperse or dm or pulse or burst or gather_detected" > $O/gputest4.log 2>&1; echo "pytest rc $?" | tee -a $O/gputest4.log
tail -15 $O/gputest4.log
timeout 600 python tools/dm_ab.py > $O/dm_ab.txt 2>&1; cat $O/dm_ab.txt
