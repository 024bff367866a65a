#!/bin/bash
# GPU box: scheduler strategies for the fused kernel (rebuild + bench each, 2 interleaved rounds)
restore() { DSABF_SCHED=max-ilp DSABF_EXTRA_FLAGS="" python -m dsabeamformer_amd.build --force > /dev/null 2>&1; }
trap restore EXIT
for r in 1 2; do
for cfg in "max-ilp|" "iterative-ilp|" "max-memory-clause|" "max-ilp|-mllvm -amdgpu-use-amdgpu-trackers=1" "max-ilp|-mllvm -misched-postra-direction=bottomup" "max-ilp|-mllvm -amdgpu-schedule-metric-bias=0"; do
  S=${cfg%%|*}; F=${cfg#*|}
  DSABF_SCHED=$S DSABF_EXTRA_FLAGS="$F" python -m dsabeamformer_amd.build --force > /dev/null 2>&1 || { echo "BUILD FAILED $cfg"; continue; }
  for P in 1 0; do
  DSABF_PAIRED=$P DSABF_SCHED=$S DSABF_EXTRA_FLAGS="$F" python bench.py --steps 80 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('%-70s paired=$P kernel_ms avg %.4f  frac %.3f' % ('$cfg', r['kernel_ms_avg'], r['frac']))"
  done
done; done
