#!/bin/bash
# GPU box, repo root: the round-2 artifacts of the other two workloads -- C5 (DSA100 scale-up, whole band on one GPU) and C2
# (DEBUG geometry, HBM-write-bound): bench line, rocprofv3 kernel stats, separate PMC passes.
set -x
R=$PWD; mkdir -p gpurun_out
python bench.py --workload c5 --units 16 --no-cpu-baseline > gpurun_out/r02_c5_bench.json 2> gpurun_out/r02_c5_bench.err
python bench.py --workload c2 --no-cpu-baseline > gpurun_out/r02_c2_bench.json 2> gpurun_out/r02_c2_bench.err
cd /tmp; export TMPDIR=/tmp
for WL in c5 c2; do
  U=$([ $WL = c5 ] && echo 16 || echo 128)
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$WL -- python3 $R/bench.py --workload $WL --units $U --steps 50 --warmup 10 --no-cpu-baseline --no-extras > $R/gpurun_out/prof_$WL.log 2>&1
  find $R/gpurun_out/prof_$WL -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c "grep -E 'Name|fused16' {} > $R/gpurun_out/r02_${WL}_kernel_stats.csv"
  rm -rf $R/gpurun_out/prof_$WL
  (cd $R && bash tools/pmc.sh pmc_$WL --workload $WL --units $U > /dev/null 2>&1 && cp gpurun_out/pmc_$WL/summary.txt gpurun_out/r02_${WL}_pmc_summary.txt)
done
cd $R; ls -la gpurun_out/r02_c5_* gpurun_out/r02_c2_*
