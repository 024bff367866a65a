#!/bin/bash
# GPU box, repo root: regenerate the profiles/r06_* artifacts into gpurun_out/r06p/ (round 6).  Run it LAST, on the final kernels: every
# counter summary carries the key of the build it was taken from (kernel build id + instantiation + launch, tools/pmc.sh), and
# bench.py quotes a summary beside its timings only when the key is its own run's.
set -x
R=$PWD; O=$R/gpurun_out/r06p; mkdir -p $O
python bench.py > $O/r06_c3_bench.json 2> $O/r06_c3_bench.err
python bench.py --workload c5 --units 16 --no-cpu-baseline > $O/r06_c5_bench.json 2> $O/r06_c5_bench.err
python bench.py --workload c2 --no-cpu-baseline > $O/r06_c2_bench.json 2> $O/r06_c2_bench.err
cd /tmp; export TMPDIR=/tmp
for V in paired general; do
  if [ $V = general ]; then export DSABF_PAIRED=0; else unset DSABF_PAIRED; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$V -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extras > $O/prof_$V.log 2>&1
  find $O/prof_$V -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c "grep -E 'Name|fused16|relayout|pair_check' {} > $O/r06_c3_${V}_kernel_stats.csv"
  rm -rf $O/prof_$V
  (cd $R && bash tools/pmc.sh r06p/pmc_$V > /dev/null 2>&1 && cp $O/pmc_$V/summary.txt $O/r06_c3_${V}_pmc_summary.txt; rm -rf $O/pmc_$V)
done
unset DSABF_PAIRED
for WL in c5 c2; do
  U=$([ $WL = c5 ] && echo 16 || echo 128)
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$WL -- python3 $R/bench.py --workload $WL --units $U --steps 50 --warmup 10 --no-cpu-baseline --no-extras > $O/prof_$WL.log 2>&1
  find $O/prof_$WL -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c "grep -E 'Name|fused16' {} > $O/r06_${WL}_kernel_stats.csv"
  rm -rf $O/prof_$WL
  (cd $R && bash tools/pmc.sh r06p/pmc_$WL --workload $WL --units $U > /dev/null 2>&1 && cp $O/pmc_$WL/summary.txt $O/r06_${WL}_pmc_summary.txt; rm -rf $O/pmc_$WL)
done
export DSABF_PAIRED=0   # C5 general kernel (a calibrated DSA100): stats + counters
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c5g -- python3 $R/bench.py --workload c5 --units 16 --steps 50 --warmup 10 --no-cpu-baseline --no-extras > $O/prof_c5g.log 2>&1
find $O/prof_c5g -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c "grep -E 'Name|fused16' {} > $O/r06_c5_general_kernel_stats.csv"
rm -rf $O/prof_c5g
(cd $R && bash tools/pmc.sh r06p/pmc_c5g --workload c5 --units 16 > /dev/null 2>&1 && cp $O/pmc_c5g/summary.txt $O/r06_c5_general_pmc_summary.txt; rm -rf $O/pmc_c5g)
unset DSABF_PAIRED
cd $R
# the DM stage of the loop by itself: per production block, then its HBM traffic and kernel / copy trace
for a in "64" "64 0 8" "64 0 1 4"; do python tools/dm_stage.py $a 2>/dev/null | tail -1; done > $O/r06_dm_stage.txt
tools/pmc_hbm.sh r06p/dm_stage_pmc "dedisperse|copy|Copy" python3 $R/tools/dm_stage.py 64 > /dev/null 2>&1; cp $O/dm_stage_pmc/summary.txt $O/r06_dm_stage_pmc_summary.txt; rm -rf $O/dm_stage_pmc
# the N = 8 line of BASELINE configs[4] in shape (eight rank processes on ONE GPU over the loopback stand-in: the numbers mean nothing)
DSABF_BENCH_ONE_GPU=1 DSABF_RCCL_LIB=$R/tests/support/libfakerccl.so FAKERCCL_MAILBOX_MB=8 python bench.py --gpus 8 --workload c5 --units 16 --steps 3 --warmup 1 \
  --dist-backend gloo --min-warm-seconds 0.1 --cpu-seconds 2 > $O/r06_bench_c5_n8_loopback.json 2> $O/r06_bench_c5_n8_loopback.err
# parity at large: the census' own report, the fuzzers on the final build
cp gpurun_out/census_gpu.txt $O/r06_census_gpu.txt 2>/dev/null
SEED=61 CASES=300 python tools/fuzz_dm_stream.py > $O/r06_fuzz_dm_stream.txt 2>&1
(SEED=2600 CASES=800 python tools/fuzz_calls.py; FUZZ=loops SEED=3600 CASES=150 python tools/fuzz_calls.py; FUZZ=debug SEED=4600 CASES=300 python tools/fuzz_calls.py; FUZZ=dmloops SEED=6600 CASES=200 python tools/fuzz_calls.py) 2>&1 | grep -v "obs Complete\|amdgpu.ids" > $O/r06_fuzz_calls.txt
(FUZZ_GENERIC=1 SEED=601 CASES=400 python tools/fuzz_long.py; SEED=603 CASES=600 python tools/fuzz_long.py; FUZZ_WIDE=1 SEED=604 CASES=300 python tools/fuzz_long.py) 2>&1 | grep -v amdgpu.ids > $O/r06_fuzz_geometry.txt
# the GPU suite: the budgeted default run with its durations, then every case
python -m pytest tests -m gpu -q --durations=25 -p no:cacheprovider > $O/r06_gputest_durations.txt 2>&1
cp gpurun_out/census_gpu.txt $O/r06_census_gpu.txt
DSABF_LONG_TESTS=1 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -4 > $O/r06_gputest_long_tail.txt
rm -f $O/*.log; ls -la $O
