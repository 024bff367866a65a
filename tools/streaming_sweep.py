#!/usr/bin/env python3
"""The production observation loop end to end (junk source -> H2D -> fused kernel -> D2H, PCIe included) for the launch
granularities run_observation offers, interleaved rounds, best and median per configuration.
GPU box, repo root:  python tools/streaming_sweep.py [rounds] [blocks] > gpurun_out/r02_streaming.txt"""
import os

os.environ.setdefault("DSABF_LAB", "1")   # a measurement tool: the library reads its A/B switches from the environment only in lab mode
import sys

sys.path.insert(0, ".")
import dsabeamformer_amd as bfm  # noqa: E402
from dsabeamformer_amd import host  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n_blk = int(sys.argv[2]) if len(sys.argv) > 2 else 64
pc = bfm.production_config()
configs = [("reference: 1 gemm-unit per launch, 8 queues round-robin", {"DSABF_UNIT_LAUNCH": "1"}),
           ("2 gemm-units per launch", {"DSABF_UNITS_PER_LAUNCH": "2"}),
           ("4 gemm-units per launch", {"DSABF_UNITS_PER_LAUNCH": "4"}),
           ("8 gemm-units per launch", {"DSABF_UNITS_PER_LAUNCH": "8"}),
           ("16 gemm-units per launch", {"DSABF_UNITS_PER_LAUNCH": "16"}),
           ("whole block (32) per launch", {"DSABF_UNITS_PER_LAUNCH": "0"})]
host.run_observation_junk(pc, 8, ring_blocks=4, burn_in=2)   # one-off costs (pinning, first launches) outside the sweep
res = {name: [] for name, _ in configs}
for r in range(rounds):
    for name, env in configs:
        os.environ.update(env)
        try:
            out = host.run_observation_junk(pc, n_blk, ring_blocks=4, burn_in=4)
        finally:
            for k in env:
                os.environ.pop(k, None)
        res[name].append(out["ms"])
chunks = n_blk * pc.n_gemms_per_block * pc.n_out_per_gemm
in_b = n_blk * pc.n_gemms_per_block * pc.n_ant * pc.n_freq * pc.n_out_per_gemm * pc.n_pol * pc.n_avg
print("production geometry, %d blocks of 128 MiB, %d interleaved rounds; per beam-block (256 beams x 256 freq), PCIe included" % (n_blk, rounds))
for name, _ in configs:
    v = sorted(res[name])
    med = v[len(v) // 2]
    print("  %-58s median %6.2f us (best %6.2f, worst %6.2f)   input %5.1f GB/s, output %5.1f GB/s"
          % (name, med / chunks * 1e3, v[0] / chunks * 1e3, v[-1] / chunks * 1e3, in_b / (med * 1e-3) / 1e9, in_b / 2 / (med * 1e-3) / 1e9))
