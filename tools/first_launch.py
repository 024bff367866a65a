#!/usr/bin/env python3
"""What a fresh process pays before its first beam-block: dlopen of libdsabf.so (its code objects: 549 kernels / 16.9 MB in round 5,
382 / 13.1 MB after the census of round 6), bf_create, bf_set_weights, the first fused launch (the runtime loads the code object of
the translation unit then) and the second one.  One subprocess per library, interleaved:
  python tools/first_launch.py [NAME=LIB ...]        (default: r05=variants/r05/libdsabf.so product)"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child():
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch

    torch.cuda.init()
    torch.zeros(1, device="cuda")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    import dsabeamformer_amd as bfm
    lib = bfm.load()
    t1 = time.perf_counter()
    cfg = bfm.production_config(n_out_per_gemm=16)
    bf = bfm.Beamformer(cfg)
    t2 = time.perf_counter()
    import bench
    w = bench.product_weights(cfg, 0)
    t3 = time.perf_counter()
    bf.set_weights(w)
    t4 = time.perf_counter()
    units = 8
    d_in = torch.randint(0, 256, (units * 256 * 512 * 64,), dtype=torch.uint8, device="cuda")
    d_out = torch.empty(units * 16 * 256 * 256, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    s = torch.cuda.current_stream().cuda_stream
    t5 = time.perf_counter()
    bf.beamform(d_in, units, d_out, s)
    torch.cuda.synchronize()
    t6 = time.perf_counter()
    bf.beamform(d_in, units, d_out, s)
    torch.cuda.synchronize()
    t7 = time.perf_counter()
    print(json.dumps({"dlopen_ms": (t1 - t0) * 1e3, "bf_create_ms": (t2 - t1) * 1e3, "bf_set_weights_ms": (t4 - t3) * 1e3,
                      "first_launch_ms": (t6 - t5) * 1e3, "second_launch_ms": (t7 - t6) * 1e3, "lib_bytes": os.path.getsize(lib._name),
                      "version": lib.bf_version().decode()}))


def main():
    if "--child" in sys.argv:
        return child()
    variants = [a.split("=", 1) for a in sys.argv[1:]] or [["r05", "variants/r05/libdsabf.so"], ["r06", "product"]]
    res = {n: [] for n, _ in variants}
    for rnd in range(5):
        for name, lib in variants:
            env = dict(os.environ)
            if lib != "product":
                env["DSABF_LIB_PATH"] = os.path.abspath(lib)
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True, timeout=600)
            try:
                res[name].append(json.loads(p.stdout.strip().splitlines()[-1]))
            except Exception:
                print("FAILED", name, (p.stdout + p.stderr)[-500:])
    for name, rows in res.items():
        if not rows:
            continue
        med = {k: sorted(r[k] for r in rows)[len(rows) // 2] for k in rows[0] if k.endswith("_ms")}
        print("%-6s %9d bytes  dlopen %7.1f ms  bf_create %6.1f ms  bf_set_weights %6.1f ms  first launch %7.2f ms  second %6.2f ms   (%s; median of %d fresh processes)"
              % (name, rows[0]["lib_bytes"], med["dlopen_ms"], med["bf_create_ms"], med["bf_set_weights_ms"], med["first_launch_ms"], med["second_launch_ms"],
                 rows[0]["version"], len(rows)))


if __name__ == "__main__":
    main()
