#!/usr/bin/env python3
"""Run-time accumulation windows (n_pol * n_avg without a compile-time instantiation): kernel time of fused16_kernel<NIPO = 0> with
the stream length the library picks, next to the one-chunk streams of round 4 (bf_set_switch "rtw_kout"), interleaved on one
handle.  GPU box, repo root:  python tools/rtw_perf.py > gpurun_out/rtw_perf.txt
Rates are ALGORITHMIC int8 ops / kernel time against the nominal 5.0 POP/s."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
import dsabeamformer_amd as bfm
from dsabeamformer_amd._lib import load

stream = torch.cuda.current_stream()
rng = np.random.default_rng(3)


def run(n_ant, n_beams, n_freq, n_avg, n_out, units, paired=False):
    cfg = bfm.production_config(n_avg=n_avg, n_out_per_gemm=n_out, n_freq=n_freq)
    cfg.n_ant, cfg.n_beams = n_ant, n_beams
    bf = bfm.Beamformer(cfg)
    w = rng.integers(-127, 128, size=(n_freq, n_ant, n_beams, 2), dtype=np.int8)
    if paired:
        h = n_beams // 2
        w[:, :, h:, 0] = w[:, :, :h, 0][:, :, ::-1]
        w[:, :, h:, 1] = -w[:, :, :h, 1][:, :, ::-1]
    bf.set_weights(w)
    L = 2 * n_avg
    n_time = n_out * L
    d_in = [torch.randint(0, 256, (units * n_freq * n_time * n_ant,), dtype=torch.uint8, device="cuda") for _ in range(2)]
    d_out = torch.empty(units * n_out * n_freq * n_beams, dtype=torch.float32, device="cuda")
    fn = lambda i: bf.beamform(d_in[i & 1], units, d_out, stream.cuda_stream)  # noqa: E731
    k, ch = C.c_int(), C.c_int()
    load().bf_rtw_plan(C.byref(cfg), units, 256, C.byref(k), C.byref(ch))
    old = 32 // L if L <= 32 else 1
    ops = 8.0 * n_beams * n_ant * n_time * n_freq * units
    res = {}
    for rep in range(3):
        for kout in (old, 0):
            bf.set_switch("rtw_kout", kout)
            for i in range(12):
                fn(i)
            torch.cuda.synchronize()
            avg, med, mn = bench.time_launches(torch, fn, 30, stream)
            res.setdefault(kout, []).append(avg)
    a, b = min(res[old]), min(res[0])
    print("ant %3d beams %3d n_ipo %3d units %3d %-7s | one-chunk streams (kout %2d): %.3f ms %.3f | library (kout %2d, %5d chunks): %.3f ms %.3f | %+.1f %%"
          % (n_ant, n_beams, L, units, "pair" if paired else "general", old, a, ops / a / 1e9 / 5000, k.value, ch.value, b, ops / b / 1e9 / 5000,
             (a / b - 1) * 100), flush=True)
    bf.close()


for n_avg in (3, 5, 6, 9, 10, 12, 14, 20, 24, 48):
    run(64, 256, 256, n_avg, 16, 32)
run(64, 256, 256, 12, 16, 32, paired=True)
run(64, 256, 256, 20, 16, 32, paired=True)
run(100, 512, 256, 12, 16, 16)
run(100, 512, 256, 20, 16, 16)
run(64, 256, 256, 12, 8, 1)       # a one-unit launch
run(64, 256, 256, 12, 8, 4)
run(192, 256, 256, 12, 16, 10)    # beyond 128 antennas: fusedg_kernel, the same stream choice
run(256, 256, 256, 20, 16, 8)
