#!/bin/bash
# GPU box, repo root: regenerate the profiles/r02_* artifacts into gpurun_out/ (bench line, rocprofv3 kernel stats and
# separate PMC passes for the conjugate-pair AND the general kernel, launch-size sweep, interleaved mode A/B).
set -x
R=$PWD; mkdir -p gpurun_out
python bench.py > gpurun_out/r02_c3_bench.json 2> gpurun_out/r02_c3_bench.err
python tools/launch_size.py > gpurun_out/r02_launch_size.txt 2>&1
python tools/ab_modes.py 5 > gpurun_out/r02_ab_modes.txt 2>&1
cd /tmp; export TMPDIR=/tmp
for V in paired general; do
  if [ $V = general ]; then export DSABF_PAIRED=0; else unset DSABF_PAIRED; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$V -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extras > $R/gpurun_out/prof_$V.log 2>&1
  find $R/gpurun_out/prof_$V -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c "grep -E 'Name|fused16|relayout|pair_check' {} > $R/gpurun_out/r02_c3_${V}_kernel_stats.csv"
  rm -rf $R/gpurun_out/prof_$V
  (cd $R && bash tools/pmc.sh pmc_$V > /dev/null 2>&1 && cp gpurun_out/pmc_$V/summary.txt gpurun_out/r02_c3_${V}_pmc_summary.txt)
done
unset DSABF_PAIRED
cd $R
ls -la gpurun_out/r02_*
