#!/usr/bin/env python3
"""Kernel time per beam-block of one launch over 1 / 2 / 8 / 32 / 128 gemm-units (input resident in HBM), for the time
splits bf_set_switch("tsplit") can force -- how fused_launch_shape's choice compares with the alternatives.  GPU box, repo root:
python tools/launch_size.py [workload] > gpurun_out/r02_launch_size.txt"""
import os
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
import dsabeamformer_amd as bfm  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
shard = wl == "c5shard"          # one rank's share of BASELINE configs[4]: 128 of 1024 freq x 512 beams x 100 ant
if shard:
    wl = "c5"
shard4 = wl.startswith("c4shard")   # one rank's share of BASELINE configs[3]: 256/N freq (c4shard8: 32, c4shard2: 128 ...)
n_ranks = int(wl[7:] or 8) if shard4 else 1
if shard4:
    wl = "c3"
n_avg, n_out = bench.geometry(wl)
cfg = bfm.production_config(n_avg=n_avg, n_out_per_gemm=n_out)
if wl == "c5":
    cfg.n_ant, cfg.n_beams, cfg.n_freq = 100, 512, 128 if shard else 1024
if shard4:
    cfg.n_freq = 256 // n_ranks
bf = bfm.Beamformer(cfg)
bf.set_weights(bench.product_weights(cfg, 0))
n_time = n_out * cfg.n_pol * cfg.n_avg
stream = torch.cuda.current_stream()
print("workload %s%s: kernel %s" % (wl, " (%d of 256 freq: one of %d ranks)" % (cfg.n_freq, n_ranks) if shard4 else "", bf.kernel_info(1)["kernel"]))
for units in ((1, 4, 16, 64) if wl == "c5" else (32, 128) if shard4 else (1, 2, 4, 8, 32, 128)):
    d_in = [torch.randint(0, 256, (units * cfg.n_freq * n_time * cfg.n_ant,), dtype=torch.uint8, device="cuda") for _ in range(4)]
    d_out = torch.empty(units * n_out * cfg.n_freq * cfg.n_beams, dtype=torch.float32, device="cuda")
    chunks = units * n_time // 128
    rows = []
    for ts in [None] + [t for t in (1, 2, 4, 8, 16, 32, 64) if t <= chunks]:
        bf.set_switch("tsplit", ts or 0)     # per handle (bf_set_switch); 0: the library's own choice
        info = bf.kernel_info(units)
        fn = lambda i: bf.beamform(d_in[i % 4], units, d_out, stream.cuda_stream)  # noqa: E731
        import time as _t
        t_w, i = _t.perf_counter(), 0
        while i < 10 or _t.perf_counter() - t_w < 0.2:   # warm the clock back up after the allocations above
            fn(i)
            i += 1
            if i % 32 == 0:
                torch.cuda.synchronize()
        avg, med, mn = bench.time_launches(torch, fn, 200 if units <= 8 else 60, stream)
        rows.append((ts, info["grid"], avg, mn))
    bf.set_switch("tsplit", 0)
    print("units %3d (%4d chunks per frequency)" % (units, chunks))
    for ts, grid, avg, mn in rows:
        print("   tsplit %-7s grid %5d  kernel avg %8.2f us  min %8.2f us  = %6.3f us per beam-block%s"
              % ("default" if ts is None else ts, grid, avg * 1e3, mn * 1e3, avg * 1e3 / (units * n_out),
                 "   <- fused_launch_shape" if ts is None else ""))
bf.close()
