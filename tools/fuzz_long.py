#!/usr/bin/env python3
"""Long randomized parity run (GPU box): CASES random geometries / launch shapes / paired or general weights through
bf_beamform_device, each compared bit for bit with the CPU oracle.  Not part of the test suite (the suite runs a 14-case
seeded subset); round 1: 3 seeds x 150 cases, 0 mismatches.   usage: SEED=1 CASES=150 python tools/fuzz_long.py"""
import os, sys, numpy as np, torch
sys.path.insert(0, ".")
import dsabeamformer_amd as bfm, oracle as orc
rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
combos = [(64, a) for a in (1, 2, 4, 8, 16, 32)] + [(16, 1), (16, 16), (32, 1), (32, 16), (100, 1), (100, 16), (128, 1), (128, 16)]
bad = 0
N = int(os.environ.get("CASES", "150"))
for case in range(N):
    n_ant, n_avg = combos[int(rng.integers(len(combos)))]
    n_ipo = 2 * n_avg
    n_out = int(rng.integers(1, 7)) * max(1, 16 // n_ipo)
    g = orc.Geom(n_beams=32 * int(rng.integers(1, 13)), n_ant=n_ant, n_freq=int(rng.integers(1, 18)), n_avg=n_avg, n_out_per_gemm=n_out)
    n_units = int(rng.integers(1, 1 + max(1, 900 // g.n_time)))
    w = rng.integers(-127, 128, size=(g.n_freq, g.n_ant, g.n_beams, 2), dtype=np.int8)
    paired = bool(rng.integers(2))
    if paired:
        B = g.n_beams
        w[:, :, B // 2:, 0] = w[:, :, :B // 2, 0][:, :, ::-1]
        w[:, :, B // 2:, 1] = -w[:, :, :B // 2, 1][:, :, ::-1]
    packed = rng.integers(0, 256, size=(n_units, g.n_freq, g.n_time, g.n_ant), dtype=np.uint8)
    os.environ["DSABF_TSPLIT"] = str(int(rng.integers(1, 5)))
    bf = bfm.Beamformer(bfm.debug_config(n_beams=g.n_beams, n_ant=g.n_ant, n_freq=g.n_freq, n_avg=g.n_avg, n_out_per_gemm=g.n_out_per_gemm))
    bf.set_weights(w)
    d_in = torch.from_numpy(packed).cuda()
    want = orc.beamform(g, w, packed)
    d_out = torch.full((want.size,), float("nan"), dtype=torch.float32, device="cuda")
    bf.beamform(d_in, n_units, d_out, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    ok = np.array_equal(d_out.cpu().numpy().reshape(want.shape), want)
    if not ok:
        bad += 1
        print("MISMATCH", case, g, n_units, paired, os.environ["DSABF_TSPLIT"])
    bf.close()
print("cases", N, "mismatches", bad)
