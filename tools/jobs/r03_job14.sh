#!/bin/bash
O=gpurun_out/r03; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_multirank.py -m gpu -x -q > $O/gputest14.log 2>&1; echo "pytest rc $?" | tee -a $O/gputest14.log; tail -3 $O/gputest14.log
