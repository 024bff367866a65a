#!/usr/bin/env python3
"""Where a channel's cycles go in dedisperse_dm_wide_kernel: run against a -DDSABF_DW_PROBE=1 build
(DSABF_LIB_PATH=variants/dwprobe/libdsabf.so python tools/dm_probe.py); prints, per wave of tile 0, the average s_memtime
cycles per channel spent in the body (reads + adds + DMA issue), waiting for its own DMA, and at the barrier."""
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
v = d_dd[:64].cpu().numpy().reshape(16, 4)
print("wave   body   own-DMA wait   barrier   total   (s_memtime cycles per channel, tile 0)")
for w in range(16):
    print("%4d %7.0f %10.0f %12.0f %8.0f" % (w, v[w, 0], v[w, 1], v[w, 2], v[w, 3]))
