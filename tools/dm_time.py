#!/usr/bin/env python3
"""Median time of the DM-trial dedispersion on the bench's shape (64 trials x 901 x 256 x 256) for the library DSABF_LIB_PATH names
(variant builds of tools/build_variant.py); one line.  usage: DSABF_LIB_PATH=variants/x/libdsabf.so python tools/dm_time.py [label]"""
import sys

import torch

sys.path.insert(0, "/root/repo")
import dsabeamformer_amd as bfm  # noqa: E402
from dsabeamformer_amd import host  # noqa: E402

bf = bfm.Beamformer(bfm.production_config())
s = torch.cuda.current_stream().cuda_stream
freq = [host.channel_frequency(0, c) for c in range(256)]
ladder = host.dm_trials(dm_max=250.0)
dms = ladder[:: max(1, len(ladder) // 64)][:64]
delays = host.dm_delays(dms, freq, freq[0], 0.131)
n_t = 1024
n_t_out = n_t - int(delays.max())
d_series = torch.rand(n_t * 256 * 256, device="cuda")
d_delays = torch.from_numpy(delays).cuda()
d_dd = torch.zeros(len(dms) * n_t_out * 256, device="cuda")
for _ in range(10):
    bf.dedisperse_dm(d_series, n_t, d_delays, len(dms), n_t_out, d_dd, s)
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(60)]
for a, b in ev:
    a.record()
    bf.dedisperse_dm(d_series, n_t, d_delays, len(dms), n_t_out, d_dd, s)
    b.record()
torch.cuda.synchronize()
ms = sorted(a.elapsed_time(b) for a, b in ev)
print("%-12s median %.4f ms  min %.4f  checksum %.6e" % (sys.argv[1] if len(sys.argv) > 1 else "", ms[len(ms) // 2], ms[0], float(d_dd.double().sum())))
