#!/usr/bin/env python3
"""Stand-alone stage kernels against their rooflines (not the headline; the product path is the fused kernel):
   a1 bf_expand_device  -- HBM bound: 1 B read + 2 B written per packed byte
   a8 bf_dedisperse_device -- latency bound (256 KiB per unit)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import dsabeamformer_amd as bfm

bf = bfm.Beamformer(bfm.debug_config())
s = torch.cuda.current_stream().cuda_stream
res = {}
n = 1 << 30
d_in = torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda")
d_out = torch.empty(2 * n, dtype=torch.int8, device="cuda")
for _ in range(5):
    bf.expand(d_in, n, d_out, s)
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
for a, b in ev:
    a.record(); bf.expand(d_in, n, d_out, s); b.record()
torch.cuda.synchronize()
ms = sorted(a.elapsed_time(b) for a, b in ev)
res["expand"] = {"bytes_per_launch": 3 * n, "ms_median": ms[len(ms) // 2], "GBps": 3 * n / (ms[len(ms) // 2] * 1e-3) / 1e9,
                 "frac_of_8TBps": 3 * n / (ms[len(ms) // 2] * 1e-3) / 8e12}
d_unit = torch.rand(8 * 256 * 256, device="cuda")
d_ded = torch.empty(256, device="cuda")
for _ in range(5):
    bf.dedisperse(d_unit, d_ded, s)
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(50)]
for a, b in ev:
    a.record(); bf.dedisperse(d_unit, d_ded, s); b.record()
torch.cuda.synchronize()
ms = sorted(a.elapsed_time(b) for a, b in ev)
res["dedisperse"] = {"us_median": ms[len(ms) // 2] * 1e3}
# 8f-4 bf_dedisperse_dm_device: 1024 beam-blocks (0.134 s of sky at 131 us) x 64 DM trials of the notebook ladder spread
# to DM ~ 250; algorithmic bytes = series once + output once; the kernel re-reads the series once per trial from L2/MALL
from dsabeamformer_amd import host  # noqa: E402

n_t, n_dm = 1024, 64
freq = [host.channel_frequency(0, c) for c in range(256)]
ladder = host.dm_trials(dm_max=250.0)
dms = ladder[:: max(1, len(ladder) // n_dm)][:n_dm]
delays = host.dm_delays(dms, freq, freq[0], 0.131)
n_t_out = n_t - int(delays.max())
d_series = torch.rand(n_t * 256 * 256, device="cuda")
d_delays = torch.from_numpy(delays).cuda()
d_dd = torch.empty(len(dms) * n_t_out * 256, device="cuda")
for _ in range(3):
    bf.dedisperse_dm(d_series, n_t, d_delays, len(dms), n_t_out, d_dd, s)
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
for a, b in ev:
    a.record(); bf.dedisperse_dm(d_series, n_t, d_delays, len(dms), n_t_out, d_dd, s); b.record()
torch.cuda.synchronize()
ms = sorted(a.elapsed_time(b) for a, b in ev)
alg = 4 * (n_t * 256 * 256 + len(dms) * n_t_out * 256)
loads = 4 * len(dms) * n_t_out * 256 * 256   # one value per (trial, time, freq, beam): what a kernel without reuse loads
res["dedisperse_dm"] = {"n_t": n_t, "n_dm": len(dms), "max_delay": int(delays.max()), "n_t_out": n_t_out,
                        "ms_median": ms[len(ms) // 2], "algorithmic_bytes": alg,
                        "algorithmic_GBps": alg / (ms[len(ms) // 2] * 1e-3) / 1e9,
                        "bytes_summed": loads, "summed_GBps": loads / (ms[len(ms) // 2] * 1e-3) / 1e9}
print(json.dumps(res))
