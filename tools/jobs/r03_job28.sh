#!/bin/bash
# round 3, GPU job 28: C2 line with the measured streaming peak (bench + rocprof stats of the same command)
R=$PWD; O=$R/gpurun_out/r03p; mkdir -p $O
python bench.py --workload c2 --no-cpu-baseline > $O/r03_c2_bench.json 2> $O/r03_c2_bench.err; echo "bench rc $?"
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c2 -- python3 $R/bench.py --workload c2 --units 128 --steps 50 --warmup 10 --no-cpu-baseline --no-extras > $O/prof_c2.log 2>&1
find $O/prof_c2 -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c "grep -E 'Name|fused16' {} > $O/r03_c2_kernel_stats.csv"
rm -rf $O/prof_c2 $O/*.log; cd $R
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r03p/r03_c2_bench.json") if l.startswith("{")][-1]); r=d["roofline"]
print(r["frac"], r["kernel_ms_avg"], r.get("peak_measured"), r.get("frac_of_measured_peak"), r.get("peak_measured_note"))
PY
cat $O/r03_c2_kernel_stats.csv | cut -c1-150
