#!/usr/bin/env python3
"""A few launches of fusedg_kernel on one geometry, for rocprofv3 (tools/generic_pmc.sh):  generic_one.py n_ant n_avg units"""
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/repo")
import dsabeamformer_amd as bfm  # noqa: E402

n_ant, n_avg, units = (int(x) for x in sys.argv[1:4])
cfg = bfm.production_config(n_avg=n_avg, n_out_per_gemm=16, n_freq=256)
cfg.n_ant = n_ant
bf = bfm.Beamformer(cfg)
rng = np.random.default_rng(3)
bf.set_weights(rng.integers(-127, 128, size=(256, n_ant, 256, 2), dtype=np.int8))
n_time = 16 * 2 * n_avg
d_in = torch.randint(0, 256, (units * 256 * n_time * n_ant,), dtype=torch.uint8, device="cuda")
d_out = torch.empty(units * 16 * 256 * 256, dtype=torch.float32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for _ in range(6):
    bf.beamform(d_in, units, d_out, s)
torch.cuda.synchronize()
print(bf.kernel_info(units))
