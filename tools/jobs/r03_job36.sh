#!/bin/bash
# round 3, GPU job 36: the driver's own bench command, timed
O=gpurun_out/r03; mkdir -p $O
time python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "rc $?"

python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r03/bench_driver.json") if l.startswith("{")][-1])
r=d["roofline"]; print(d["value"], d["ms_per_step"], r["frac"], r["peak_measured"], r["frac_of_measured_peak"], d["cpu_baseline"]["value"], sorted(d.keys()))
PY
