#!/usr/bin/env python3
"""Two-k-step geometries at the C5 shape (512 beams, 1024 freq, n_ipo 32, 16 gemm-units per launch): kernel time for 96, 100,
112 and 128 antennas.  The MFMA and detect work is identical for all of them (two k-steps of 64, zero weights behind the
last antenna); what differs is the staging: 16-byte pieces (96, 112, 128: no LDS bank conflicts) vs 4-byte pieces (100:
2-way ds_write_b32 conflicts on 22 % of the LDS cycles).  GPU box, repo root: python tools/ant_sweep.py"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
import dsabeamformer_amd as bfm  # noqa: E402

units, n_avg, n_out = 16, 16, 8
pos100, dirs = bench.grid_100()
stream = torch.cuda.current_stream()
rows = {}
handles = {}
for n_ant in (68, 96, 100, 108, 112, 128):
    cfg = bfm.production_config(n_avg=n_avg, n_out_per_gemm=n_out, n_freq=1024, n_beams=512, n_ant=n_ant)
    from dsabeamformer_amd import host

    pos = np.zeros((n_ant, 3), np.float32)
    pos[:min(n_ant, 100)] = pos100[:min(n_ant, 100)]
    pos[100:, 0] = np.linspace(-240, 240, max(0, n_ant - 100))
    bf = bfm.Beamformer(cfg)
    bf.set_weights(host.make_weights(pos, dirs, cfg.n_freq, chan0=0, gpu=0))
    handles[n_ant] = (bf, cfg)
n_time = n_out * 2 * n_avg
d_out = torch.empty(units * n_out * 1024 * 512, dtype=torch.float32, device="cuda")
res = {a: [] for a in handles}
for rnd in range(4):
    for n_ant, (bf, cfg) in handles.items():
        d_in = [torch.randint(0, 256, (units * 1024 * n_time * n_ant,), dtype=torch.uint8, device="cuda") for _ in range(2)]
        fn = lambda i: bf.beamform(d_in[i & 1], units, d_out, stream.cuda_stream)  # noqa: E731
        for i in range(10):
            fn(i)
        res[n_ant].append(bench.time_launches(torch, fn, 60, stream)[0])
        del d_in
print("C5 shape, %d gemm-units per launch, 4 interleaved rounds, kernel ms (median of rounds)" % units)
for n_ant, (bf, cfg) in handles.items():
    v = sorted(res[n_ant])
    print("  %3d antennas  %-70s  %.4f ms   (%.4f .. %.4f)" % (n_ant, bf.kernel_info(units)["kernel"], v[len(v) // 2], v[0], v[-1]))
