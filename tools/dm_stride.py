#!/usr/bin/env python3
"""Does the DM-trial dedispersion suffer from power-of-two row strides (L2 / HBM channel camping)?  Same kernel, same
trials and delays per channel, series rows of 256 / 255 / 250 / 192 channels x 256 beams: time per channel."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import dsabeamformer_amd as bfm  # noqa: E402
from dsabeamformer_amd import host  # noqa: E402

s = torch.cuda.current_stream().cuda_stream
freq_all = [host.channel_frequency(0, c) for c in range(256)]
ladder = host.dm_trials(dm_max=250.0)
dms = ladder[:: max(1, len(ladder) // 64)][:64]
n_t = 1024
for n_f in (256, 255, 250, 192, 256):
    bf = bfm.Beamformer(bfm.production_config(n_freq=n_f))
    delays = np.ascontiguousarray(host.dm_delays(dms, freq_all, freq_all[0], 0.131)[:, :n_f])
    n_t_out = n_t - 123
    d_series = torch.rand(n_t * n_f * 256, device="cuda")
    d_delays = torch.from_numpy(delays).cuda()
    d_dd = torch.zeros(len(dms) * n_t_out * 256, device="cuda")
    for _ in range(3):
        bf.dedisperse_dm(d_series, n_t, d_delays, len(dms), n_t_out, d_dd, s)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for a, b in ev:
        a.record(); bf.dedisperse_dm(d_series, n_t, d_delays, len(dms), n_t_out, d_dd, s); b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    print("n_freq %3d  row stride %7d B  median %.3f ms  = %.3f us per channel" % (n_f, n_f * 1024, ms[5], ms[5] * 1e3 / n_f))
    bf.close()
