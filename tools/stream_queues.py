#!/usr/bin/env python3
"""run_observation end to end (PCIe included) for 8 / 4 / 2 / 1 compute queues, block launches against the coalesced per-unit loop;
GPU box, repo root: python tools/stream_queues.py"""
import os, sys

os.environ.setdefault("DSABF_LAB", "1")   # a measurement tool: the library reads its A/B switches from the environment only in lab mode
sys.path.insert(0, "/root/repo")
import dsabeamformer_amd as bfm
from dsabeamformer_amd import host
pc = bfm.production_config()
host.run_observation_junk(pc, 8, ring_blocks=4, burn_in=2)
n_blk = 64
chunks = n_blk * pc.n_gemms_per_block * pc.n_out_per_gemm
for r in range(2):
    for ns in (8, 4, 2, 1):
        for mode, env in (("block", {"DSABF_UNITS_PER_LAUNCH": "0"}), ("coalesced-units", {"DSABF_UNIT_LAUNCH": "1"})):
            pc2 = bfm.production_config(); pc2.n_streams = ns
            os.environ.update(env)
            try:
                out = host.run_observation_junk(pc2, n_blk, ring_blocks=4, burn_in=4)
            finally:
                for k in env: os.environ.pop(k, None)
            print("n_streams %d %-16s %.2f us per beam-block" % (ns, mode, out["ms"] / chunks * 1e3), flush=True)
