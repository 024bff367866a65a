#!/usr/bin/env python3
"""Randomised bit-exact parity run of the DM-trial dedispersion (both kernels: the shared-window kernel where trial groups
fit, the per-thread-window kernel elsewhere and alone) against the oracle.  Random series length, channels, beams (any
multiple of 4), trial counts and delay tables: fine ladders, coarse ladders, mixed groups, negative-going delays, random noise
on top.  GPU box, repo root:  SEED=1 CASES=300 python tools/fuzz_dm.py > gpurun_out/r03/fuzz_dm.txt"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import dsabeamformer_amd as bfm  # noqa: E402
import oracle as orc  # noqa: E402

seed, cases = int(os.environ.get("SEED", "1")), int(os.environ.get("CASES", "200"))
rng = np.random.default_rng(seed)
s = torch.cuda.current_stream().cuda_stream
bad, kinds = 0, {}
for case in range(cases):
    n_f = int(rng.choice([1, 3, 8, 17, 32, 64, 100, 128]))
    n_b = 4 * int(rng.integers(1, 80))
    n_dm = int(rng.integers(1, 100))
    n_t = int(rng.integers(20, 260))
    kind = str(rng.choice(["fine", "coarse", "mixed", "negative", "random", "constant"]))
    slope = np.linspace(1.0, 0.0, n_f) ** 2 if n_f > 1 else np.ones(1)
    if kind == "fine":
        d = np.arange(n_dm)[:, None] * rng.uniform(0.2, 2.0) * slope[None, :]
    elif kind == "coarse":
        d = np.arange(n_dm)[:, None] * rng.uniform(5.0, 20.0) * slope[None, :]
    elif kind == "mixed":
        step = np.where(rng.random(n_dm) < 0.1, rng.uniform(5, 30), rng.uniform(0.2, 1.5))
        d = np.cumsum(step)[:, None] * slope[None, :]
    elif kind == "negative":
        d = np.arange(n_dm)[:, None] * rng.uniform(0.3, 2.0) * (slope[None, :] - 0.4) - rng.integers(0, 10)
    elif kind == "random":
        d = rng.integers(-5, 40, size=(n_dm, n_f)).astype(float)
    else:
        d = np.full((n_dm, n_f), float(rng.integers(0, 5)))
    delays = np.ascontiguousarray((d + (rng.integers(0, 2, size=d.shape) if rng.random() < 0.3 else 0)).astype(np.int32))
    series = (rng.random((n_t, n_f, n_b), dtype=np.float32) * 1e3).astype(np.float32)
    n_t_out = int(rng.choice([n_t, max(1, n_t - max(0, int(delays.max()))), max(1, n_t // 2)]))
    bf = bfm.Beamformer(bfm.debug_config(n_beams=n_b, n_freq=n_f))
    d_series, d_delays = torch.from_numpy(series).cuda(), torch.from_numpy(delays).cuda()
    want = orc.dedisperse_dm(series, delays, n_t_out)
    for mode in ("shared", "thread"):
        bf.set_switch("dm_wide", 0 if mode == "thread" else 1)
        d_out = torch.full((n_dm, n_t_out, n_b), float("nan"), dtype=torch.float32, device="cuda")
        bf.dedisperse_dm(d_series, n_t, d_delays, n_dm, n_t_out, d_out, s)
        torch.cuda.synchronize()
        if not np.array_equal(d_out.cpu().numpy(), want):
            bad += 1
            print("MISMATCH case %d mode %s kind %s n_t %d n_f %d n_b %d n_dm %d n_t_out %d" % (case, mode, kind, n_t, n_f, n_b, n_dm, n_t_out))
    kinds[kind] = kinds.get(kind, 0) + 1
    bf.close()
print("seed %d cases %d (x 2 kernel selections) mismatches %d kinds %s" % (seed, cases, bad, kinds))
sys.exit(1 if bad else 0)
