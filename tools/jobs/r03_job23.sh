#!/bin/bash
# round 3, GPU job 23: the launch switches now live in the handle -- the tests that use them, then the whole suite
O=gpurun_out/r03; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q > $O/gputest23.log 2>&1; echo "pytest rc $?" | tee -a $O/gputest23.log
tail -3 $O/gputest23.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
