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
print(json.dumps(res))
