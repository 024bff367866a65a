#!/usr/bin/env python3
"""Lifetimes of the DM stage's twice-mapped ring (round 6): create / push / destroy many bf_dm_streams -- small ones, then
production-sized ones until the address arenas roll over (a ring's addresses are never reused: 2 x its bytes per stage, taken from
256-GiB reservations) -- watching device memory (the physical side IS released) and checking one stage's chunks against the oracle
every so often.  GPU box, repo root: python tools/soak_dm_ring.py [small_cycles] [big_cycles]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import dsabeamformer_amd as bfm
import oracle as orc
from dsabeamformer_amd import api

small, big = (int(sys.argv[1]) if len(sys.argv) > 1 else 1500), (int(sys.argv[2]) if len(sys.argv) > 2 else 700)
rng = np.random.default_rng(66)
torch.zeros(1, device="cuda")
free0 = torch.cuda.mem_get_info()[0]
bf = bfm.Beamformer(bfm.debug_config(n_beams=64, n_freq=16, n_gemms_per_block=1, n_blocks_on_gpu=1, n_streams=1))
s = torch.cuda.current_stream().cuda_stream
bad = 0
for c in range(small):
    n_dm, rows = int(rng.integers(1, 9)), int(rng.choice([1, 4, 8, 16]))
    delays = np.ascontiguousarray(np.sort(rng.integers(0, 20, size=(n_dm, 16)), axis=1)[:, ::-1].astype(np.int32))
    dm = api.DmStream(bf, delays, 16, rows)
    assert bf.counter("dm_ring_stages") == 1
    if c % 50 == 0:
        D, n_t = int(delays.max()), 8 * rows + 24
        series = rng.random((n_t, 16, 64), dtype=np.float32)
        d_series = torch.from_numpy(series).cuda()
        host = torch.empty(n_dm * rows * 64, dtype=torch.float32).pin_memory()
        parts, pushed = [], 0
        while pushed < n_t:
            n = min(rows, n_t - pushed)
            first, n_out = dm.push(d_series[pushed:pushed + n], n, host, s)
            torch.cuda.synchronize()
            if n_out:
                parts.append(host[:n_dm * n_out * 64].numpy().reshape(n_dm, n_out, 64).copy())
            pushed += n
        got = np.concatenate(parts, axis=1) if parts else np.zeros((n_dm, 0, 64), np.float32)
        bad += not np.array_equal(got, orc.dedisperse_dm(series, delays, max(n_t - D, 0))[:, :got.shape[1]])
    dm.close()
bf.close()
torch.cuda.synchronize()
free1 = torch.cuda.mem_get_info()[0]
print("small stages: %d lifetimes, %d checked against the oracle, %d mismatches, device memory %+.1f MiB" % (small, (small + 49) // 50, bad, (free0 - free1) / 2 ** 20))
# production-sized stages: 891 rows x 256 KiB of physical memory each (released at destroy), 445 MiB of addresses each (never reused)
pc = bfm.production_config()
bf = bfm.Beamformer(pc)
delays = np.zeros((64, pc.n_freq), np.int32)
delays[:, 0] = 123
rows = pc.n_gemms_per_block * pc.n_out_per_gemm
torch.cuda.synchronize()
free_h = torch.cuda.mem_get_info()[0]            # with the production handle (its 1-GiB voltage ring) in place
worst = 0
for c in range(big):
    dm = api.DmStream(bf, delays, pc.n_freq, rows)
    assert bf.counter("dm_ring_stages") == 1, "cycle %d: the stage fell back to the linear buffer" % c
    dst = dm.reserve(rows, s)
    assert dst
    dm.push(dst, rows, None, s)
    torch.cuda.synchronize()
    dm.close()
    if c % 100 == 99:
        worst = max(worst, free_h - torch.cuda.mem_get_info()[0])
bf.close()
torch.cuda.synchronize()
free2 = torch.cuda.mem_get_info()[0]
print("production-sized stages: %d lifetimes = %.0f GiB of ring addresses taken (arenas of 256 GiB: %d rolled over), every one a ring, device memory %+.1f MiB at the end (between two lifetimes, against the handle alone: at most %+.1f MiB)"
      % (big, big * 2 * (123 + 3 * rows) * pc.n_freq * pc.n_beams * 4 / 2 ** 30, int(big * 2 * (123 + 3 * rows) * pc.n_freq * pc.n_beams * 4 // (256 << 30)),
         (free1 - free2) / 2 ** 20, worst / 2 ** 20))
sys.exit(1 if bad else 0)
