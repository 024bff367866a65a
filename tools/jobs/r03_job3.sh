#!/bin/bash
# round 3, GPU job 3: the whole GPU suite, the DEBUG flow in both launch patterns (pinned buffers allocated before the timer),
# the launch-size record with the 8-queue aggregate
O=gpurun_out/r03; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gputest3.log 2>&1; echo "pytest rc $?" | tee -a $O/gputest3.log
tail -4 $O/gputest3.log
C=tests/golden/config
for i in 1 2 3; do
  for mode in "" "-u"; do
    echo "== beam $mode"; dsabeamformer_amd/beam $mode -p $C/linear_positions.txt -d $C/linear_directions.txt -s $C/linear_source_directions_1024.txt -o /tmp/data$mode.py 2>&1 | grep -E "Observation ran|Time per data chunk|datarate"
  done
done > $O/debug_flow_patterns.txt 2>&1
cmp /tmp/data.py /tmp/data-u.py && echo "data.py identical" >> $O/debug_flow_patterns.txt
cat $O/debug_flow_patterns.txt
timeout 900 python bench.py --steps 100 --no-cpu-baseline > $O/bench3.json 2> $O/bench3.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03/bench3.json').read().strip().splitlines()[-1])
print(json.dumps(d.get('launch_size'),indent=1)[:2500])
PY
