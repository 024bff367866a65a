#!/bin/bash
O=gpurun_out/r03; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "dedisperse or dm or pulse or burst or gather_detected" > $O/gputest8.log 2>&1; echo "pytest rc $?" | tee -a $O/gputest8.log; tail -3 $O/gputest8.log
bash tools/jobs/r03_job7.sh
