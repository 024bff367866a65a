#!/bin/bash
# GPU box, repo root: regenerate the profiles/r04_* artifacts into gpurun_out/r04p/ (round 4: + the generic kernel, + fuzz runs)
# previously: (bench lines, rocprofv3 kernel stats and
# separate PMC passes for the conjugate-pair AND the general kernel at C3, C5 and C2, stage kernels, interleaved mode A/B).
set -x
R=$PWD; O=$R/gpurun_out/r04p; mkdir -p $O
python bench.py > $O/r04_c3_bench.json 2> $O/r04_c3_bench.err
python bench.py --workload c5 --units 16 --no-cpu-baseline > $O/r04_c5_bench.json 2> $O/r04_c5_bench.err
python bench.py --workload c2 --no-cpu-baseline > $O/r04_c2_bench.json 2> $O/r04_c2_bench.err
python tools/ab_modes.py 5 > $O/r04_ab_modes.txt 2>&1
python tools/bench_stages.py > $O/r04_stage_kernels.json 2> /dev/null
cd /tmp; export TMPDIR=/tmp
for V in paired general; do
  if [ $V = general ]; then export DSABF_PAIRED=0; else unset DSABF_PAIRED; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$V -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extras > $O/prof_$V.log 2>&1
  find $O/prof_$V -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c "grep -E 'Name|fused16|relayout|pair_check' {} > $O/r04_c3_${V}_kernel_stats.csv"
  rm -rf $O/prof_$V
  (cd $R && bash tools/pmc.sh r04p/pmc_$V > /dev/null 2>&1 && cp $O/pmc_$V/summary.txt $O/r04_c3_${V}_pmc_summary.txt; rm -rf $O/pmc_$V)
done
unset DSABF_PAIRED
for WL in c5 c2; do
  U=$([ $WL = c5 ] && echo 16 || echo 128)
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$WL -- python3 $R/bench.py --workload $WL --units $U --steps 50 --warmup 10 --no-cpu-baseline --no-extras > $O/prof_$WL.log 2>&1
  find $O/prof_$WL -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c "grep -E 'Name|fused16' {} > $O/r04_${WL}_kernel_stats.csv"
  rm -rf $O/prof_$WL
  (cd $R && bash tools/pmc.sh r04p/pmc_$WL --workload $WL --units $U > /dev/null 2>&1 && cp $O/pmc_$WL/summary.txt $O/r04_${WL}_pmc_summary.txt; rm -rf $O/pmc_$WL)
done
export DSABF_PAIRED=0   # C5 general kernel (a calibrated DSA100): stats + counters
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c5g -- python3 $R/bench.py --workload c5 --units 16 --steps 50 --warmup 10 --no-cpu-baseline --no-extras > $O/prof_c5g.log 2>&1
find $O/prof_c5g -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c "grep -E 'Name|fused16' {} > $O/r04_c5_general_kernel_stats.csv"
rm -rf $O/prof_c5g
(cd $R && bash tools/pmc.sh r04p/pmc_c5g --workload c5 --units 16 > /dev/null 2>&1 && cp $O/pmc_c5g/summary.txt $O/r04_c5_general_pmc_summary.txt; rm -rf $O/pmc_c5g)
unset DSABF_PAIRED
# the DM-trial dedispersion: kernel trace + counters
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_dm -- python3 $R/tools/dm_one.py > $O/prof_dm.log 2>&1
find $O/prof_dm -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c "grep -E 'Name|dedisperse' {} > $O/r04_dm_kernel_stats.csv"
rm -rf $O/prof_dm
(cd $R && bash tools/dm_pmc.sh r04p/pmc_dm > /dev/null 2>&1 && cp $O/pmc_dm/summary.txt $O/r04_dm_pmc_summary.txt; rm -rf $O/pmc_dm)
# fusedg_kernel at 256 antennas (4 k-steps), 1 GiB launch: kernel trace + counters; the antenna / window sweep
(cd $R && export DSABF_DEEP=0 && bash tools/generic_pmc.sh r04p/pmc_g256 256 16 32 > /dev/null 2>&1 && cp $O/pmc_g256/summary.txt $O/r04_generic256_pmc_summary.txt; rm -rf $O/pmc_g256)
# the deep class of fused16_kernel on the same shape (the default routing)
(cd $R && bash tools/generic_pmc.sh r04p/pmc_d256 256 16 32 > /dev/null 2>&1 && cp $O/pmc_d256/summary.txt $O/r04_deep256_pmc_summary.txt; rm -rf $O/pmc_d256)
cd $R
python tools/generic_perf.py > $O/r04_generic_perf.txt 2>/dev/null
# parity at length on the final build: the specialised classes, the wide launches, the generic kernel, the DM kernels
SEED=41 CASES=600 python tools/fuzz_long.py > $O/r04_fuzz_long.txt 2>&1
SEED=42 CASES=400 FUZZ_WIDE=1 python tools/fuzz_long.py >> $O/r04_fuzz_long.txt 2>&1
SEED=43 CASES=600 FUZZ_GENERIC=1 python tools/fuzz_long.py > $O/r04_fuzz_generic.txt 2>&1
SEED=45 CASES=400 FUZZ_DEEP=1 python tools/fuzz_long.py >> $O/r04_fuzz_generic.txt 2>&1
SEED=44 CASES=300 python tools/fuzz_dm.py > $O/r04_fuzz_dm.txt 2>&1
python -m pytest tests -m gpu -q 2>&1 | tail -4 > $O/r04_gputest_tail.txt
cd $R; rm -f $O/*.log; ls -la $O
