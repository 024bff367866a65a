#!/usr/bin/env python3
"""Round 6: is a compile-time antenna class of fused16_kernel (64 / 100 / 128 / 192 / 256 antennas) faster than
the run-time class that covers the same count (k1p16 / k2p4 / k2p16 / k3p16 / k4p16)?  Interleaved A/B on one box, two handles
per case (the run-time one created with DSABF_LAB=1 DSABF_RUNTIME_ANT=1), general and conjugate-pair kernel, at the bench's
launch sizes.  A class whose gain is inside the box noise is folded into the run-time class.
GPU box, repo root: python tools/class_fold_ab.py > gpurun_out/class_fold_ab.txt"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
import dsabeamformer_amd as bfm

stream = torch.cuda.current_stream()
rng = np.random.default_rng(6)
ROUNDS = int(os.environ.get("ROUNDS", "5"))


def make(n_ant, n_beams, n_avg, n_freq, paired, runtime):
    os.environ["DSABF_LAB"] = "1"
    os.environ["DSABF_RUNTIME_ANT"] = "1" if runtime else "0"
    cfg = bfm.production_config(n_avg=n_avg, n_out_per_gemm=16, n_freq=n_freq)
    cfg.n_ant, cfg.n_beams = n_ant, n_beams
    bf = bfm.Beamformer(cfg)
    os.environ.pop("DSABF_RUNTIME_ANT", None)
    w = rng.integers(-127, 128, size=(n_freq, n_ant, n_beams, 2), dtype=np.int8)
    if paired:
        h = n_beams // 2
        w[:, :, h:, 0] = w[:, :, :h, 0][:, :, ::-1]
        w[:, :, h:, 1] = -w[:, :, :h, 1][:, :, ::-1]
    bf.set_weights(w)
    return bf


def case(n_ant, n_beams, n_avg, n_freq, units, paired):
    n_time = 16 * 2 * n_avg
    d_in = [torch.randint(0, 256, (units * n_freq * n_time * n_ant,), dtype=torch.uint8, device="cuda") for _ in range(2)]
    d_out = torch.empty(units * 16 * n_freq * n_beams, dtype=torch.float32, device="cuda")
    hs = [make(n_ant, n_beams, n_avg, n_freq, paired, rt) for rt in (False, True)]
    names = [h.kernel_info(units) for h in hs]
    fns = [(lambda i, h=h: h.beamform(d_in[i & 1], units, d_out, stream.cuda_stream)) for h in hs]
    for fn in fns:
        for i in range(8):
            fn(i)
    torch.cuda.synchronize()
    t = [[], []]
    for _ in range(ROUNDS):
        for k in (0, 1):
            t[k].append(bench.time_launches(torch, fns[k], 20, stream)[1])
    for h in hs:
        h.close()
    a, b = float(np.median(t[0])), float(np.median(t[1]))
    spread = max(max(x) / min(x) - 1 for x in t) * 100
    ops = 8.0 * n_beams * n_ant * n_time * n_freq * units
    print("ant %3d beams %3d n_ipo %2d freq %4d units %3d %-7s | compile-time %.4f ms (%.3f, %3d vgprs) | run-time %.4f ms (%.3f, %3d vgprs) | run-time %+.2f %% "
          "(spread of a class's own rounds: %.2f %%)" % (n_ant, n_beams, 2 * n_avg, n_freq, units, "pair" if paired else "general", a,
                                                        ops / a / 1e9 / 5000, names[0]["vgprs"], b, ops / b / 1e9 / 5000, names[1]["vgprs"],
                                                        (b / a - 1) * 100, spread), flush=True)


if __name__ == "__main__":
    print("device:", torch.cuda.get_device_name(0), flush=True)
    cases = [
        (64, 256, 16, 256, 128, True), (64, 256, 16, 256, 128, False),     # C3: the bench line's workload
        (64, 256, 1, 256, 32, False), (64, 256, 1, 256, 32, True),         # C2 geometry (n_ipo 2), a block of 32 units
        (64, 256, 4, 256, 64, False),                                      # n_ipo 8
        (128, 256, 16, 256, 64, True), (128, 256, 16, 256, 64, False),     # plain two-k-step launch (one beam group)
        (128, 512, 16, 256, 32, True), (128, 512, 16, 256, 32, False),     # 8-slot pair kernel / 8-wave general kernel
        (128, 384, 16, 256, 32, True),                                     # 8-wave pair kernel
        (100, 512, 16, 1024, 16, True), (100, 512, 16, 1024, 16, False),   # C5 (k2p4 has no 8-slot kernel: 8-wave pair)
        (100, 256, 16, 256, 64, True),
        (192, 256, 16, 256, 32, False), (192, 512, 16, 256, 16, True),
        (256, 256, 16, 256, 32, False), (256, 512, 16, 256, 16, True), (256, 256, 16, 256, 32, True),
    ]
    for c in cases:
        case(*c)
