#!/bin/bash
# GPU box, repo root: regenerate the profiles/r05_* artifacts into gpurun_out/r05p/ (round 5).
# New this round: the budgeted GPU suite's durations + the full suite (DSABF_LONG_TESTS=1) tail, the bench line with
# bound_measured / the staged transport's re-layout pass / the DM stage in the streaming record, the K' = 200 micro-benchmark.
# Kernel stats and PMC passes of the C3 pair kernel (headline) and the C5 general kernel are refreshed;
# the other kernels did not change since round 4 (profiles/r04_*).
set -x
R=$PWD; O=$R/gpurun_out/r05p; mkdir -p $O
python bench.py > $O/r05_c3_bench.json 2> $O/r05_c3_bench.err
python bench.py --workload c5 --units 16 --no-cpu-baseline > $O/r05_c5_bench.json 2> $O/r05_c5_bench.err
python bench.py --workload c2 --no-cpu-baseline > $O/r05_c2_bench.json 2> $O/r05_c2_bench.err
tools/ubench_k200 > $O/r05_ubench_k200.txt 2>&1
cd /tmp; export TMPDIR=/tmp
for V in paired general; do
  if [ $V = general ]; then export DSABF_PAIRED=0; else unset DSABF_PAIRED; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$V -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extras > $O/prof_$V.log 2>&1
  find $O/prof_$V -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c "grep -E 'Name|fused16|relayout|pair_check' {} > $O/r05_c3_${V}_kernel_stats.csv"
  rm -rf $O/prof_$V
  (cd $R && bash tools/pmc.sh r05p/pmc_$V > /dev/null 2>&1 && cp $O/pmc_$V/summary.txt $O/r05_c3_${V}_pmc_summary.txt; rm -rf $O/pmc_$V)
done
unset DSABF_PAIRED
for WL in c5 c2; do
  U=$([ $WL = c5 ] && echo 16 || echo 128)
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$WL -- python3 $R/bench.py --workload $WL --units $U --steps 50 --warmup 10 --no-cpu-baseline --no-extras > $O/prof_$WL.log 2>&1
  find $O/prof_$WL -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c "grep -E 'Name|fused16' {} > $O/r05_${WL}_kernel_stats.csv"
  rm -rf $O/prof_$WL
  (cd $R && bash tools/pmc.sh r05p/pmc_$WL --workload $WL --units $U > /dev/null 2>&1 && cp $O/pmc_$WL/summary.txt $O/r05_${WL}_pmc_summary.txt; rm -rf $O/pmc_$WL)
done
export DSABF_PAIRED=0   # C5 general kernel (a calibrated DSA100): stats + counters
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c5g -- python3 $R/bench.py --workload c5 --units 16 --steps 50 --warmup 10 --no-cpu-baseline --no-extras > $O/prof_c5g.log 2>&1
find $O/prof_c5g -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c "grep -E 'Name|fused16' {} > $O/r05_c5_general_kernel_stats.csv"
rm -rf $O/prof_c5g
(cd $R && bash tools/pmc.sh r05p/pmc_c5g --workload c5 --units 16 > /dev/null 2>&1 && cp $O/pmc_c5g/summary.txt $O/r05_c5_general_pmc_summary.txt; rm -rf $O/pmc_c5g)
unset DSABF_PAIRED
# the auxiliary kernels of the round: the staged transport's re-layout pass (and the DM kernels beside it), from a bench run WITH its
# supplementary records: kernel trace + HBM traffic counters
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_aux -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/prof_aux.log 2>&1
find $O/prof_aux -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c "grep -E 'Name|gather_relayout|dedisperse' {} > $O/r05_aux_kernel_stats.csv"
rm -rf $O/prof_aux
(cd $R && PMC_FILTER="gather_relayout" PMC_BENCH_ARGS="--steps 3 --warmup 1 --no-cpu-baseline" bash tools/pmc.sh r05p/pmc_aux > /dev/null 2>&1 && cp $O/pmc_aux/summary.txt $O/r05_relayout_pmc_summary.txt; rm -rf $O/pmc_aux)
cd $R
SEED=51 CASES=300 python tools/fuzz_dm_stream.py > $O/r05_fuzz_dm_stream.txt 2>&1
(SEED=2000 CASES=1200 python tools/fuzz_calls.py; SEED=10000 CASES=300 python tools/fuzz_calls.py; FUZZ=loops SEED=3000 CASES=150 python tools/fuzz_calls.py; FUZZ=debug SEED=4000 CASES=400 python tools/fuzz_calls.py; FUZZ=dmloops SEED=6000 CASES=200 python tools/fuzz_calls.py) 2>&1 | grep -v "obs Complete\|amdgpu.ids" > $O/r05_fuzz_calls.txt
# the in-contract geometries outside the BASELINE configs: run-time windows, dword-aligned rows beyond 128 antennas
python tools/rtw_perf.py 2>&1 | grep -v amdgpu.ids > $O/r05_rtw_perf.txt
python tools/deep_p4_perf.py 2>&1 | grep -v amdgpu.ids > $O/r05_deep_p4_perf.txt
python tools/generic_perf.py 2>&1 | grep -v amdgpu.ids > $O/r05_generic_perf.txt
(FUZZ_GENERIC=1 SEED=501 CASES=600 python tools/fuzz_long.py; FUZZ_GENERIC=1 SEED=502 CASES=600 python tools/fuzz_long.py; SEED=503 CASES=300 python tools/fuzz_long.py) 2>&1 | grep -v amdgpu.ids > $O/r05_fuzz_geometry.txt
# the GPU suite: the budgeted default run with its durations, then every case
python -m pytest tests -m gpu -q --durations=25 -p no:cacheprovider > $O/r05_gputest_durations.txt 2>&1
DSABF_LONG_TESTS=1 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -4 > $O/r05_gputest_long_tail.txt
rm -f $O/*.log; ls -la $O
