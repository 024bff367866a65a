#!/bin/bash
# Perf experiment (GPU box): rebuild with each flag set and time the bench kernel, interleaved rounds.
# The in-tree product library is rebuilt with the DEFAULT flags when the script ends, however it ends (trap), and
# dsabeamformer_amd/build.py also refuses to reuse a library whose recorded flag set differs (build/flags.stamp).
mkdir -p gpurun_out
restore() { DSABF_EXTRA_FLAGS="" python -m dsabeamformer_amd.build --force > /dev/null 2>&1 || echo "RESTORE BUILD FAILED"; }
trap restore EXIT
ROUNDS=${ROUNDS:-2}
for r in $(seq $ROUNDS); do
for fl in "$@"; do
  DSABF_EXTRA_FLAGS="$fl" python -m dsabeamformer_amd.build --force > /dev/null 2>&1 || echo "BUILD FAILED $fl"
  DSABF_EXTRA_FLAGS="$fl" python bench.py --steps 80 --warmup 20 --no-cpu-baseline --no-extras $BENCH_ARGS 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('%-44s kernel_ms avg %.4f med %.4f min %.4f  frac %.3f' % ('$fl', r['kernel_ms_avg'], r['kernel_ms_median'], r['kernel_ms_min'], r['frac']))"
done; done 2>&1 | tee -a gpurun_out/variants.txt
