#!/usr/bin/env python3
"""Soak run (GPU box): handle create / set_weights / beamform / destroy in a loop over changing geometries, watching device
memory, then a long observation (`beam -j N`).  Not part of the test suite.  usage: python tools/soak.py [cycles] [blocks]"""
import subprocess
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import dsabeamformer_amd as bfm  # noqa: E402

cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 300
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 600
rng = np.random.default_rng(5)
s = torch.cuda.current_stream().cuda_stream
free0 = torch.cuda.mem_get_info()[0]
geoms = [dict(), dict(n_ant=100, n_beams=512, n_freq=64), dict(n_ant=108, n_beams=512, n_freq=16), dict(n_avg=1, n_out_per_gemm=8),
         dict(n_ant=32, n_beams=96, n_freq=7, n_avg=4, n_out_per_gemm=4)]
t0 = time.time()
free_mid = None
for c in range(cycles):
    if c == cycles // 3:
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        free_mid = torch.cuda.mem_get_info()[0]
    kw = geoms[c % len(geoms)]
    cfg = bfm.production_config(**kw)
    bf = bfm.Beamformer(cfg)
    w = rng.integers(-127, 128, size=(cfg.n_freq, cfg.n_ant, cfg.n_beams, 2), dtype=np.int8)
    bf.set_weights(w)
    n_time = cfg.n_out_per_gemm * cfg.n_pol * cfg.n_avg
    d_in = torch.randint(0, 256, (2 * cfg.n_freq * n_time * cfg.n_ant,), dtype=torch.uint8, device="cuda")
    d_out = torch.empty(2 * cfg.n_out_per_gemm * cfg.n_freq * cfg.n_beams, dtype=torch.float32, device="cuda")
    for _ in range(3):
        bf.beamform(d_in, 2, d_out, s)
    torch.cuda.synchronize()
    assert torch.isfinite(d_out).all()
    bf.close()
    del d_in, d_out
torch.cuda.empty_cache()
free1 = torch.cuda.mem_get_info()[0]
print("create/destroy cycles %d in %.1f s; device memory free before %.1f MiB, after a third of the cycles %.1f MiB, at the end %.1f MiB: "
      "one-time (runtime, code objects, allocator) %.1f MiB, growth over the last two thirds %.1f MiB"
      % (cycles, time.time() - t0, free0 / 2**20, free_mid / 2**20, free1 / 2**20, (free0 - free_mid) / 2**20, (free_mid - free1) / 2**20))
r = subprocess.run(["dsabeamformer_amd/beam", "-j", str(blocks)], capture_output=True, text=True, timeout=900)
print("beam -j %d: rc %d" % (blocks, r.returncode))
print("\n".join(l for l in (r.stdout + r.stderr).splitlines() if "amdgpu.ids" not in l)[-600:])
