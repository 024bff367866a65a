#!/bin/bash
# round 3, GPU job 25: the measured MFMA peak in the bench line
O=gpurun_out/r03; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_round3.py -m gpu -x -q -k "mfma_peak" 2>&1 | tail -3
python bench.py --no-extras --no-cpu-baseline > $O/bench25.json 2> $O/bench25.err; echo "bench rc $?"
python bench.py --workload c5 --units 16 --no-extras --no-cpu-baseline > $O/bench25_c5.json 2>> $O/bench25.err; echo "bench rc $?"
python - <<'PY'
import json
for f in ("gpurun_out/r03/bench25.json","gpurun_out/r03/bench25_c5.json"):
    d=json.loads([l for l in open(f) if l.startswith("{")][-1]); r=d["roofline"]
    print(f, r["frac"], r.get("peak_measured"), r.get("frac_of_measured_peak"), r.get("peak_measured_note"))
PY
