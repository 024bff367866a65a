#!/usr/bin/env python3
"""Interleaved A/B timing of the fused kernel's variants on the bench workload (C3, 128 gemm-units per launch): detect mode
(canonical / contracted / fast) x weight handling (conjugate-pair / general).  Several rounds, every variant measured in
every round right after the others, so box-to-box and minute-to-minute clock drift cancels; medians over the rounds.
GPU box, repo root:  python tools/ab_modes.py [rounds] > gpurun_out/r02_ab_modes.txt"""
import os
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
import dsabeamformer_amd as bfm  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
units, (n_avg, n_out) = 128, bench.geometry("c3")
variants = []
for mode_name, mode in (("canonical", 0), ("contracted", 2), ("fast", 1)):
    for pair_name, env in (("paired", None), ("general", "0")):
        if env is None:
            os.environ.pop("DSABF_PAIRED", None)
        else:
            os.environ["DSABF_PAIRED"] = env
        cfg = bfm.production_config(n_avg=n_avg, n_out_per_gemm=n_out, detect_mode=mode)
        bf = bfm.Beamformer(cfg)
        bf.set_weights(bench.product_weights(cfg, 0))
        variants.append(("%s/%s" % (mode_name, pair_name), bf))
os.environ.pop("DSABF_PAIRED", None)
cfg = variants[0][1].cfg
n_time = n_out * cfg.n_pol * cfg.n_avg
gen = torch.Generator(device="cuda").manual_seed(0xD5A)
d_in = [torch.randint(0, 256, (units * cfg.n_freq * n_time * cfg.n_ant,), dtype=torch.uint8, device="cuda", generator=gen) for _ in range(2)]
d_out = [torch.empty(units * n_out * cfg.n_freq * cfg.n_beams, dtype=torch.float32, device="cuda") for _ in range(2)]
stream = torch.cuda.current_stream()
ops = 8 * cfg.n_beams * cfg.n_ant * cfg.n_pol * cfg.n_avg * cfg.n_freq * units * n_out
res = {name: [] for name, _ in variants}
for name, bf in variants:       # warm every variant (and the clock) once
    for i in range(200):
        bf.beamform(d_in[i & 1], units, d_out[i & 1], stream.cuda_stream)
torch.cuda.synchronize()
for r in range(rounds):
    for name, bf in variants:
        fn = lambda i: bf.beamform(d_in[i & 1], units, d_out[i & 1], stream.cuda_stream)  # noqa: E731
        for i in range(20):
            fn(i)
        avg, med, mn = bench.time_launches(torch, fn, 150, stream)
        res[name].append(avg)
print("C3, %d gemm-units per launch, %d interleaved rounds of 150 launches; kernel ms (HIP events), median over rounds" % (units, rounds))
base = sorted(res["canonical/paired"])[rounds // 2]
for name, bf in variants:
    v = sorted(res[name])
    med = v[len(v) // 2]
    print("  %-22s %-78s ms %.4f (min %.4f max %.4f)  frac %.3f  vs canonical/paired %+.1f %%"
          % (name, bf.kernel_info(units)["kernel"], med, v[0], v[-1], ops / (med * 1e-3) / 1e12 / 5000.0, (med / base - 1) * 100))
for _, bf in variants:
    bf.close()
