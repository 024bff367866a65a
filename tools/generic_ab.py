#!/usr/bin/env python3
"""Interleaved timing of fusedg_kernel across measurement builds (tools/build_variant.py), one subprocess per measurement:
   python tools/generic_ab.py n_ant n_avg units NAME=LIB ...     (LIB = product | variants/NAME/libdsabf.so)"""
import json
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")

if sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch

    import bench
    import dsabeamformer_amd as bfm

    n_ant, n_avg, units = (int(x) for x in sys.argv[2:5])
    n_beams = int(os.environ.get("G_BEAMS", "256"))          # G_BEAMS / G_PAIRED: beam count, conjugate-symmetric weights
    cfg = bfm.production_config(n_avg=n_avg, n_out_per_gemm=16, n_freq=256)
    cfg.n_ant, cfg.n_beams = n_ant, n_beams
    bf = bfm.Beamformer(cfg)
    w = np.random.default_rng(3).integers(-127, 128, size=(256, n_ant, n_beams, 2), dtype=np.int8)
    if os.environ.get("G_PAIRED") == "1":
        h = n_beams // 2
        w[:, :, h:, 0] = w[:, :, :h, 0][:, :, ::-1]
        w[:, :, h:, 1] = -w[:, :, :h, 1][:, :, ::-1]
    bf.set_weights(w)
    n_time = 16 * 2 * n_avg
    d_in = [torch.randint(0, 256, (units * 256 * n_time * n_ant,), dtype=torch.uint8, device="cuda") for _ in range(2)]
    d_out = torch.empty(units * 16 * 256 * n_beams, dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream()
    fn = lambda i: bf.beamform(d_in[i & 1], units, d_out, st.cuda_stream)  # noqa: E731
    for i in range(10):
        fn(i)
    torch.cuda.synchronize()
    avg, med, mn = bench.time_launches(torch, fn, 30, st)
    print(json.dumps({"avg": avg, "vgprs": bf.kernel_info(units)["vgprs"], "grid": bf.kernel_info(units)["grid"]}))
    sys.exit(0)

n_ant, n_avg, units = sys.argv[1:4]
variants = [v.split("=", 1) for v in sys.argv[4:]]
res = {n: [] for n, _ in variants}
info = {}
for rnd in range(3):
    for name, lib in variants:
        env = dict(os.environ)
        if lib != "product":
            env["DSABF_LIB_PATH"] = os.path.join(ROOT, lib)
        r = subprocess.run([sys.executable, __file__, "--child", n_ant, n_avg, units], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(name, "FAILED", r.stderr[-500:])
            continue
        d = json.loads(line[-1])
        res[name].append(d["avg"])
        info[name] = d
ops = 8.0 * int(os.environ.get("G_BEAMS", "256")) * int(n_ant) * 16 * 2 * int(n_avg) * 256 * int(units)
for name, _ in variants:
    if res[name]:
        t = sorted(res[name])[len(res[name]) // 2]
        print("%-28s %.3f ms  %.3f of 5.0 POP/s   vgprs %d grid %d" % (name, t, ops / t / 1e9 / 5000, info[name]["vgprs"], info[name]["grid"]))
