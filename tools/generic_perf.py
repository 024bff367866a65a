#!/usr/bin/env python3
"""Kernel time of fusedg_kernel (csrc/bf_fusedg.hip) over antenna counts and accumulation windows, next to the specialised
fused16_kernel where both cover the geometry.  GPU box, repo root:  python tools/generic_perf.py > gpurun_out/generic_perf.txt
Rates are ALGORITHMIC int8 ops (8 * beams * antennas * samples * frequencies) / kernel time, against the nominal 5.0 POP/s."""
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


def run(n_ant, n_beams, n_freq, n_avg, n_out, units, generic=None, label="", paired=False):
    if generic is None:
        os.environ.pop("DSABF_GENERIC", None)
    else:
        os.environ["DSABF_GENERIC"] = "1" if generic else "0"
    cfg = bfm.production_config(n_avg=n_avg, n_out_per_gemm=n_out, n_freq=n_freq)
    cfg.n_ant, cfg.n_beams = n_ant, n_beams
    bf = bfm.Beamformer(cfg)
    os.environ.pop("DSABF_GENERIC", None)
    w = rng.integers(-127, 128, size=(n_freq, n_ant, n_beams, 2), dtype=np.int8)     # general weights (no conjugate symmetry)
    if paired:
        h = n_beams // 2
        w[:, :, h:, 0] = w[:, :, :h, 0][:, :, ::-1]
        w[:, :, h:, 1] = -w[:, :, :h, 1][:, :, ::-1]
    bf.set_weights(w)
    n_ipo = 2 * n_avg
    n_time = n_out * n_ipo
    d_in = [torch.randint(0, 256, (units * n_freq * n_time * n_ant,), dtype=torch.uint8, device="cuda") for _ in range(2)]
    d_out = torch.empty(units * n_out * n_freq * n_beams, dtype=torch.float32, device="cuda")
    fn = lambda i: bf.beamform(d_in[i & 1], units, d_out, stream.cuda_stream)  # noqa: E731
    import time
    t0, i = time.perf_counter(), 0
    while i < 10 or time.perf_counter() - t0 < 0.3:
        fn(i)
        i += 1
        if i % 16 == 0:
            torch.cuda.synchronize()
    avg, med, mn = bench.time_launches(torch, fn, 40, stream)
    info = bf.kernel_info(units)
    ops = 8.0 * n_beams * n_ant * n_time * n_freq * units
    print("%-34s ant %4d beams %4d freq %4d n_ipo %3d units %3d | %-78s grid %5d vgprs %3d | %.3f ms  %.0f TOP/s  %.3f of 5.0 POP/s"
          % (label, n_ant, n_beams, n_freq, n_ipo, units, info["kernel"][7:85], info["grid"], info["vgprs"], avg, ops / avg / 1e9,
             ops / avg / 1e9 / 5000.0), flush=True)
    bf.close()


# C3 geometry: specialised general kernel vs the generic one (what the accumulator-stationary structure costs at one k-step)
run(64, 256, 256, 16, 16, 32, None, "C3 shape, fused16 general")
run(64, 256, 256, 16, 16, 32, True, "C3 shape, fusedg")
run(128, 256, 256, 16, 16, 16, None, "128 ant, fused16 general")
run(128, 256, 256, 16, 16, 16, True, "128 ant, fusedg")
# three / four k-steps: the deep classes of fused16_kernel (general and conjugate-pair) next to fusedg_kernel
for n_ant in (144, 192, 224, 256):
    run(n_ant, 256, 256, 16, 16, 8192 // n_ant, None, "%d ant, fused16 deep general" % n_ant)
    run(n_ant, 256, 256, 16, 16, 8192 // n_ant, True, "%d ant, fusedg" % n_ant)
run(256, 512, 256, 16, 16, 16, None, "256 ant x 512 beams, deep general")
run(256, 512, 256, 16, 16, 16, None, "256 ant x 512 beams, deep pair", paired=True)
run(192, 512, 256, 16, 16, 16, None, "192 ant x 512 beams, deep pair", paired=True)
run(256, 256, 256, 16, 16, 32, None, "256 ant x 256 beams, deep pair (1 tile/wave)", paired=True)
run(192, 256, 256, 16, 16, 42, None, "192 ant x 256 beams, deep pair (1 tile/wave)", paired=True)
# beyond that: only the generic kernel
for n_ant in (132, 192, 256, 320, 512, 1024):
    run(n_ant, 256, 256, 16, 16, max(2, 2048 // n_ant), None, "%d antennas" % n_ant)
for n_ant in (192, 256, 512):       # the same with launches of 1 GiB of voltages (the bench's step size)
    run(n_ant, 256, 256, 16, 16, 8192 // n_ant, None, "%d antennas, 1 GiB launch" % n_ant)
run(256, 512, 1024, 16, 8, 4, None, "256 ant, C5-like band")
# accumulation windows that are not a power of two (64 antennas)
for n_avg in (3, 5, 12, 20, 48):
    run(64, 256, 256, n_avg, max(1, 256 // n_avg), 32, None, "n_avg %d, fused16 run-time window" % n_avg)
    run(64, 256, 256, n_avg, max(1, 256 // n_avg), 32, True, "n_avg %d, fusedg" % n_avg)
run(100, 512, 256, 12, 10, 16, None, "100 ant n_avg 12, fused16 run-time window")
