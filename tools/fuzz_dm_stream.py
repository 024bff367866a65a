#!/usr/bin/env python3
"""Randomised bit-exact parity run of the DM stage of the loop (bf_dm_stream, include/dsabf.h): a detected series pushed in
random pieces on alternating HIP streams, the delay window carried over on the device, against ONE oracle call over the whole
series.  Random series length, channels, beams (any multiple of 4), trial counts, delay tables (fine / coarse / mixed / random /
constant: the wide kernel, the per-thread kernel and both in one call), push sizes from 1 row to the whole series; both kernel
selections.  Round 6: every piece is fed either by a copying push or through bf_dm_stream_reserve (the producer -- a device copy
here -- writes into the stage's own buffer), the buffer is the twice-mapped ring or the linear one, and in half of the problems
NOTHING is synchronised between pushes (the chunks go to their own host buffers): the stage's own ordering -- a push behind the
previous one, a producer behind the push three back -- is all that keeps the rows apart.
GPU box, repo root:  SEED=1 CASES=300 python tools/fuzz_dm_stream.py > gpurun_out/r06p/r06_fuzz_dm_stream.txt"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import dsabeamformer_amd as bfm  # noqa: E402
import oracle as orc  # noqa: E402
from dsabeamformer_amd import _lib, api  # noqa: E402

hip = _lib._preload_hip_runtime()

seed, cases = int(os.environ.get("SEED", "1")), int(os.environ.get("CASES", "200"))
rng = np.random.default_rng(seed)
streams = [torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.current_stream()]
bad, kinds, pushes_total = 0, {}, 0
for case in range(cases):
    n_f = int(rng.choice([1, 3, 8, 17, 32, 64, 100, 128]))
    n_b = 4 * int(rng.integers(1, 80))
    n_dm = int(rng.integers(1, 100))
    n_t = int(rng.integers(30, 400))
    kind = str(rng.choice(["fine", "coarse", "mixed", "random", "constant"]))
    slope = np.linspace(1.0, 0.0, n_f) ** 2 if n_f > 1 else np.ones(1)
    if kind == "fine":
        d = np.arange(n_dm)[:, None] * rng.uniform(0.2, 2.0) * slope[None, :]
    elif kind == "coarse":
        d = np.arange(n_dm)[:, None] * rng.uniform(3.0, 8.0) * slope[None, :]
    elif kind == "mixed":
        step = np.where(rng.random(n_dm) < 0.1, rng.uniform(5, 20), rng.uniform(0.2, 1.5))
        d = np.cumsum(step)[:, None] * slope[None, :]
    elif kind == "random":
        d = rng.integers(0, 40, size=(n_dm, n_f)).astype(float)
    else:
        d = np.full((n_dm, n_f), float(rng.integers(0, 5)))
    delays = np.ascontiguousarray(np.minimum(d, n_t - 2).astype(np.int32))     # (a streamed dedispersion needs delays >= 0)
    D = int(delays.max())
    max_rows = int(rng.choice([1, 7, 16, 32, 64, n_t]))
    series = (rng.random((n_t, n_f, n_b), dtype=np.float32) * 1e3).astype(np.float32)
    want = orc.dedisperse_dm(series, delays, n_t - D)
    bf = bfm.Beamformer(bfm.debug_config(n_beams=n_b, n_freq=n_f))
    d_series = torch.from_numpy(series).cuda()
    row_bytes = n_f * n_b * 4
    for mode in ("shared", "thread"):
        bf.set_switch("dm_wide", 0 if mode == "thread" else 1)
        ring = int(rng.integers(0, 4) != 0)
        bf.set_switch("dm_ring", ring)
        dm = api.DmStream(bf, delays, n_f, max_rows)
        lazy = bool(rng.integers(0, 2))            # no synchronisation between pushes: every chunk to a host buffer of its own
        plan, pushed = [], 0
        while pushed < n_t:
            n = min(int(rng.integers(1, max_rows + 1)), n_t - pushed)
            plan.append((pushed, n))
            pushed += n
        hosts = [torch.full((n_dm * n * n_b,), float("nan"), dtype=torch.float32).pin_memory() for _, n in plan] if lazy else \
            [torch.full((n_dm * max_rows * n_b,), float("nan"), dtype=torch.float32).pin_memory()]
        parts, at, k, ok, outs = [], 0, 0, True, []
        for pushed, n in plan:
            st = streams[int(rng.integers(0, len(streams)))]
            host = hosts[k if lazy else 0]
            src = d_series.data_ptr() + pushed * row_bytes
            if rng.integers(0, 3):                 # the zero-copy feed: the producer writes where the stage says
                dst = dm.reserve(n, st.cuda_stream)
                ok &= hip.hipMemcpyAsync(C.c_void_p(dst), C.c_void_p(src), C.c_size_t(n * row_bytes), 3, C.c_void_p(st.cuda_stream)) == 0
                src = dst
            first, n_out = dm.push(src, n, host, st.cuda_stream)
            ok &= first == at and n_out == max(0, pushed + n - D) - max(0, pushed - D)
            if not lazy:
                st.synchronize()
                if n_out:
                    parts.append(host[:n_dm * n_out * n_b].numpy().reshape(n_dm, n_out, n_b).copy())
            outs.append(n_out)
            at += n_out
            k += 1
        if lazy:
            torch.cuda.synchronize()
            parts = [hosts[i][:n_dm * o * n_b].numpy().reshape(n_dm, o, n_b).copy() for i, o in enumerate(outs) if o]
        pushes_total += k
        got = np.concatenate(parts, axis=1) if parts else np.zeros((n_dm, 0, n_b), np.float32)
        if not ok or got.shape != want.shape or not np.array_equal(got, want):
            bad += 1
            print("MISMATCH case %d mode %s kind %s n_t %d n_f %d n_b %d n_dm %d D %d max_rows %d ring %d lazy %d" % (case, mode, kind, n_t, n_f, n_b, n_dm, D, max_rows, ring, lazy))
        dm.close()
    kinds[kind] = kinds.get(kind, 0) + 1
    bf.close()
print("seed %d cases %d (x 2 kernel selections, %d pushes) mismatches %d kinds %s" % (seed, cases, pushes_total, bad, kinds))
sys.exit(1 if bad else 0)
