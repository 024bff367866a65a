#!/usr/bin/env python3
"""129 ... 256 antennas in rows that are only dword-aligned (n_ant % 16 != 0): fused16_kernel's deep classes with 4-byte staging
pieces (round 5) against fusedg_kernel (DSABF_DEEP=0), which took them until round 4.  GPU box, repo root: python tools/deep_p4_perf.py"""
import os

os.environ.setdefault("DSABF_LAB", "1")   # a measurement tool: the library reads its A/B switches from the environment only in lab mode
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
import dsabeamformer_amd as bfm

stream = torch.cuda.current_stream()
rng = np.random.default_rng(3)


def run(n_ant, n_beams, n_avg, units, paired, deep):
    os.environ["DSABF_DEEP"] = "1" if deep else "0"
    cfg = bfm.production_config(n_avg=n_avg, n_out_per_gemm=16, n_freq=256)
    cfg.n_ant, cfg.n_beams = n_ant, n_beams
    bf = bfm.Beamformer(cfg)
    os.environ.pop("DSABF_DEEP", None)
    w = rng.integers(-127, 128, size=(256, n_ant, n_beams, 2), dtype=np.int8)
    if paired:
        h = n_beams // 2
        w[:, :, h:, 0] = w[:, :, :h, 0][:, :, ::-1]
        w[:, :, h:, 1] = -w[:, :, :h, 1][:, :, ::-1]
    bf.set_weights(w)
    n_time = 16 * 2 * n_avg
    d_in = [torch.randint(0, 256, (units * 256 * n_time * n_ant,), dtype=torch.uint8, device="cuda") for _ in range(2)]
    d_out = torch.empty(units * 16 * 256 * n_beams, dtype=torch.float32, device="cuda")
    fn = lambda i: bf.beamform(d_in[i & 1], units, d_out, stream.cuda_stream)  # noqa: E731
    for i in range(12):
        fn(i)
    torch.cuda.synchronize()
    avg, med, mn = bench.time_launches(torch, fn, 30, stream)
    info = bf.kernel_info(units)
    bf.close()
    ops = 8.0 * n_beams * n_ant * n_time * 256 * units
    return avg, ops / avg / 1e9 / 5000, info


for n_ant, n_beams, n_avg, paired in ((132, 256, 16, False), (132, 256, 16, True), (180, 256, 16, False), (180, 512, 16, True), (196, 256, 16, False),
                                      (252, 256, 16, False), (252, 256, 16, True), (252, 512, 16, True), (252, 256, 8, False), (252, 512, 8, True),
                                      (140, 512, 8, True), (228, 512, 16, False)):
    units = max(1, (4096 // n_ant) * 16 // n_avg * 256 // n_beams)
    a = run(n_ant, n_beams, n_avg, units, paired, True)
    b = run(n_ant, n_beams, n_avg, units, paired, False)
    print("ant %3d beams %3d n_ipo %2d units %2d %-7s | deep class: %.3f ms %.3f (vgprs %3d) %-60s | fusedg: %.3f ms %.3f | %+.1f %%"
          % (n_ant, n_beams, 2 * n_avg, units, "pair" if paired else "general", a[0], a[1], a[2]["vgprs"], a[2]["kernel"][7:67], b[0], b[1],
             (b[0] / a[0] - 1) * 100), flush=True)
