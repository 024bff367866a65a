#!/bin/bash
# round 3, GPU job 2: the new round-3 tests, the DEBUG flow in both launch patterns, a bench run with the new records
O=gpurun_out/r03; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_round3.py -m gpu -x -q -s > $O/gputest_round3.log 2>&1; echo "pytest rc $?" | tee -a $O/gputest_round3.log
tail -5 $O/gputest_round3.log
C=tests/golden/config
for i in 1 2 3; do
  for mode in "" "-u"; do
    echo "== beam $mode"; dsabeamformer_amd/beam $mode -p $C/linear_positions.txt -d $C/linear_directions.txt -s $C/linear_source_directions_1024.txt -o /tmp/data$mode.py 2>&1 | grep -E "Observation ran|Time per data chunk|datarate"
  done
done > $O/debug_flow_patterns.txt 2>&1
cmp /tmp/data.py /tmp/data-u.py && echo "data.py identical" >> $O/debug_flow_patterns.txt
cat $O/debug_flow_patterns.txt
timeout 900 python bench.py --steps 100 > $O/bench2.json 2> $O/bench2.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03/bench2.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], d['roofline']['kernel_ms_avg'])
for k in ('general_kernel','calibrated_weights','calibrated_weights_contracted','contracted_detect_mode','contracted_detect_mode_general_kernel','fast_detect_mode','c5_shard'):
    v=d.get(k,{}); print(k, v.get('kernel_ms_avg'), v.get('frac'), v.get('error'))
print(json.dumps(d.get('launch_size'),indent=1)[:1500])
print(d.get('cpu_baseline'))
PY
