#!/usr/bin/env python3
"""Interleaved A/B timing of the fused kernel across BUILDS and environment switches, one subprocess per measurement.

  python tools/ab_libs.py [--workload c3|c5|c2] [--paired 0|1] [--detect canonical|contracted|fast] [--weights fan|calibrated]
                          [--units N] [--n-freq F] [--n-beams B] [--rounds R] [--launches L]  NAME=LIB[,ENV=VAL...] ...

LIB is a library made by tools/build_variant.py (or `product` for the in-tree one); ENV=VAL pairs are set for that variant's
process (e.g. DSABF_TSPLIT=24).  Every round measures every variant once, back to back, so clock drift cancels; the table
gives the median over rounds of the per-round average kernel time (HIP events around every launch)."""
import argparse
import json
import os

os.environ.setdefault("DSABF_LAB", "1")   # a measurement tool: the library reads its A/B switches from the environment only in lab mode
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def child(a):
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch

    import bench
    import dsabeamformer_amd as bfm

    n_avg, n_out = bench.geometry(a.workload)
    cfg = bfm.production_config(n_avg=n_avg, n_out_per_gemm=n_out, n_freq=1024 if a.workload == "c5" else 256,
                                detect_mode=bench.DETECT[a.detect])
    if a.workload == "c5":
        cfg.n_ant, cfg.n_beams = 100, 512
    if a.n_freq:
        cfg.n_freq = a.n_freq          # e.g. 128: one rank's shard of BASELINE configs[4]
    if a.n_beams:
        cfg.n_beams = a.n_beams
    if a.n_ant:
        cfg.n_ant = a.n_ant            # e.g. 128 with --workload c5: the full two-k-step class
    if not a.paired:
        os.environ["DSABF_PAIRED"] = "0"
    bf = bfm.Beamformer(cfg)
    w = bench.product_weights(cfg, 0)
    if a.weights == "calibrated":
        w = bench.calibrated_weights(w, seed=7)
    bf.set_weights(w)
    units = a.units or (16 if a.workload == "c5" else 128)
    n_time = n_out * cfg.n_pol * cfg.n_avg
    gen = torch.Generator(device="cuda").manual_seed(0xD5A)
    d_in = [torch.randint(0, 256, (units * cfg.n_freq * n_time * cfg.n_ant,), dtype=torch.uint8, device="cuda", generator=gen)
            for _ in range(2)]
    d_out = [torch.empty(units * n_out * cfg.n_freq * cfg.n_beams, dtype=torch.float32, device="cuda") for _ in range(2)]
    stream = torch.cuda.current_stream()
    fn = lambda i: bf.beamform(d_in[i & 1], units, d_out[i & 1], stream.cuda_stream)  # noqa: E731
    import time
    t0 = time.time()
    i = 0
    while time.time() - t0 < a.warm:
        for _ in range(50):
            fn(i)
            i += 1
        torch.cuda.synchronize()
    avg, med, mn = bench.time_launches(torch, fn, a.launches, stream)
    info = bf.kernel_info(units)
    ops = 8 * cfg.n_beams * cfg.n_ant * cfg.n_pol * cfg.n_avg * cfg.n_freq * units * n_out
    print(json.dumps({"avg": avg, "med": med, "min": mn, "kernel": info["kernel"], "vgprs": info.get("vgprs"),
                      "grid": info.get("grid"), "frac": ops / (avg * 1e-3) / 5e15,
                      "sum": float(d_out[0][:4096].double().sum().item())}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--paired", type=int, default=0)
    ap.add_argument("--detect", default="canonical")
    ap.add_argument("--weights", default="fan")
    ap.add_argument("--units", type=int, default=0)
    ap.add_argument("--n-freq", type=int, default=0)
    ap.add_argument("--n-beams", type=int, default=0)
    ap.add_argument("--n-ant", type=int, default=0)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--launches", type=int, default=150)
    ap.add_argument("--warm", type=float, default=1.0)
    ap.add_argument("--child", action="store_true")
    ap.add_argument("variants", nargs="*")
    a = ap.parse_args()
    if a.child:
        return child(a)
    variants = []
    for v in a.variants:
        name, rest = v.split("=", 1)
        parts = rest.split(",")
        env = dict(p.split("=", 1) for p in parts[1:])
        if parts[0] != "product":
            env["DSABF_LIB_PATH"] = os.path.abspath(parts[0])
        variants.append((name, env))
    res = {n: [] for n, _ in variants}
    last = {}
    for r in range(a.rounds):
        for name, env in variants:
            cmd = [sys.executable, os.path.abspath(__file__), "--child", "--workload", a.workload, "--paired", str(a.paired),
                   "--detect", a.detect, "--weights", a.weights, "--units", str(a.units), "--n-freq", str(a.n_freq), "--n-beams", str(a.n_beams), "--n-ant", str(a.n_ant), "--launches", str(a.launches),
                   "--warm", str(a.warm)]
            p = subprocess.run(cmd, env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
            try:
                d = json.loads(p.stdout.strip().splitlines()[-1])
            except Exception:
                print("FAILED %s: %s" % (name, (p.stdout + p.stderr)[-600:]))
                continue
            res[name].append(d["avg"])
            last[name] = d
    print("workload %s paired %d detect %s weights %s; %d rounds x %d launches; kernel ms = median over rounds of the round average"
          % (a.workload, a.paired, a.detect, a.weights, a.rounds, a.launches))
    base = None
    for name, _ in variants:
        v = sorted(res[name])
        if not v:
            continue
        med = v[len(v) // 2]
        base = base or med
        d = last[name]
        print("  %-28s ms %.4f (min %.4f max %.4f) frac %.3f  vs first %+.1f %%  vgprs %s grid %s  %s  chk %.6g"
              % (name, med, v[0], v[-1], d["frac"] * d["avg"] / med, (med / base - 1) * 100, d["vgprs"], d["grid"], d["kernel"][:60], d["sum"]))


if __name__ == "__main__":
    main()
