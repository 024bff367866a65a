#!/bin/bash
# Perf experiment (GPU box): rebuild libdsabf.so with -DDSABF_ABLATE=n and time the bench kernel.  Results of
# n != 0 are WRONG by construction; only the timings are used ("where the cycles go").
mkdir -p gpurun_out
for n in "$@"; do
  DSABF_EXTRA_FLAGS="-DDSABF_ABLATE=$n" python -m dsabeamformer_amd.build --force > /dev/null 2>&1
  python bench.py --steps 60 --warmup 15 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('ablate=$n  kernel_ms avg %.4f min %.4f  frac %.3f' % (r['kernel_ms_avg'], r['kernel_ms_min'], r['frac']))"
done 2>&1 | tee -a gpurun_out/ablate.txt
