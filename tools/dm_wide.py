import sys
import numpy as np, torch
sys.path.insert(0, ".")
import dsabeamformer_amd as bfm
from dsabeamformer_amd import host
bf = bfm.Beamformer(bfm.production_config())
s = torch.cuda.current_stream().cuda_stream
freq = [host.channel_frequency(0, c) for c in range(256)]
for label, dm_max, n_t in (("DM<=250", 250.0, 1024), ("DM<=1000", 1000.0, 2048), ("DM<=2000", 2000.0, 2048)):
    ladder = host.dm_trials(dm_max=dm_max)
    dms = ladder[:: max(1, len(ladder) // 64)][:64]
    delays = host.dm_delays(dms, freq, freq[0], 0.131)
    n_t_out = n_t - int(delays.max())
    span4 = max(int((delays[b * 4 + 3] - delays[b * 4]).max()) for b in range(16))
    d_series = torch.rand(n_t * 256 * 256, device="cuda")
    d_delays = torch.from_numpy(delays).cuda()
    d_dd = torch.zeros(len(dms) * n_t_out * 256, device="cuda")
    for _ in range(3):
        bf.dedisperse_dm(d_series, n_t, d_delays, len(dms), n_t_out, d_dd, s)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for a, b in ev:
        a.record(); bf.dedisperse_dm(d_series, n_t, d_delays, len(dms), n_t_out, d_dd, s); b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    print("%-9s max delay %4d  widest 4-trial span %3d  n_t_out %4d  median %.3f ms = %.3f ns per (trial, time, beam)" % (
        label, int(delays.max()), span4, n_t_out, ms[5], ms[5] * 1e6 / (64 * n_t_out * 256)))
