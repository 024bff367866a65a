#!/usr/bin/env python3
"""A few launches of the DM-trial dedispersion on the stage bench's shape, for rocprofv3 (tools/dm_pmc.sh).
(The experiment switches that selected other kernel versions are gone; the summaries of those runs are profiles/r02_dm_*.)"""
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
for _ in range(4):
    bf.dedisperse_dm(d_series, n_t, d_delays, len(dms), n_t_out, d_dd, s)
torch.cuda.synchronize()
