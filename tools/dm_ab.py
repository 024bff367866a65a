#!/usr/bin/env python3
"""DM-trial dedispersion: the round-3 shared-window kernel (bf_dm_wide.hip) against the per-thread-window kernel alone
(bf_set_switch dm_wide 0), same inputs, interleaved; fine and coarse ladders.  Also checks that both give the same bits.
GPU box, repo root:  python tools/dm_ab.py > gpurun_out/r03/dm_ab.txt"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import dsabeamformer_amd as bfm  # noqa: E402
from dsabeamformer_amd import host  # noqa: E402

bf = bfm.Beamformer(bfm.production_config())
s = torch.cuda.current_stream().cuda_stream
freq = [host.channel_frequency(0, c) for c in range(256)]


def timed(n=20):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record()
        bf.dedisperse_dm(d_series, n_t, d_delays, len(dms), n_t_out, d_dd, s)
        b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    return ms[len(ms) // 2]


for label, dm_max, n_t, n_dm in (("DM<=250 x64", 250.0, 1024, 64), ("DM<=250 x256", 250.0, 1024, 256), ("DM<=1000 x64", 1000.0, 2048, 64),
                                 ("DM<=2000 x64", 2000.0, 2048, 64), ("DM<=2000 all", 2000.0, 4096, 0)):
    ladder = host.dm_trials(dm_max=dm_max)
    dms = ladder if n_dm == 0 else ladder[:: max(1, len(ladder) // n_dm)][:n_dm]
    delays = host.dm_delays(dms, freq, freq[0], 0.131)
    n_t_out = n_t - int(delays.max())
    if n_t_out < 64:
        continue
    d_series = torch.rand(n_t * 256 * 256, device="cuda")
    d_delays = torch.from_numpy(delays).cuda()
    d_dd = torch.zeros(len(dms) * n_t_out * 256, device="cuda")
    res, outs = {}, {}
    for rnd in range(3):
        for mode in ("wide", "thread"):
            bf.set_switch("dm_wide", 0 if mode == "thread" else 1)
            for _ in range(3):
                bf.dedisperse_dm(d_series, n_t, d_delays, len(dms), n_t_out, d_dd, s)
            res.setdefault(mode, []).append(timed())
            if rnd == 0:
                outs[mode] = d_dd.clone()
    same = bool(torch.equal(outs["wide"], outs["thread"]))
    spread32 = max(int((delays[g * 32:(g + 1) * 32].max(0) - delays[g * 32:(g + 1) * 32].min(0)).max()) for g in range((len(dms) + 31) // 32))
    alg = 4 * (n_t * 256 * 256 + len(dms) * n_t_out * 256)
    w, t = sorted(res["wide"])[1], sorted(res["thread"])[1]
    print("%-13s trials %4d max delay %4d widest 32-trial spread %3d n_t_out %4d | shared-window %.3f ms (%.0f GB/s algorithmic = %.3f of 8 TB/s)"
          " | per-thread-window alone %.3f ms | x%.2f | same bits: %s"
          % (label, len(dms), int(delays.max()), spread32, n_t_out, w, alg / (w * 1e-3) / 1e9, alg / (w * 1e-3) / 8e12, t, t / w, same))
bf.close()
