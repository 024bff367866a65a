#!/usr/bin/env python3
"""bench.py -- throughput of the beamformer hot path (fused 4-bit expand -> int8 MFMA GEMM -> power detect).

Contract (driver):  python bench.py --gpus N --steps K --warmup W   (N > 1: one rank per GPU under
torch.distributed.run; RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment).  Rank 0 prints ONE JSON line.

Workload (BASELINE.json configs[2] geometry, "C3" in SURVEY.md section 8d): 64 antennas x 2 pol, 256 frequencies,
256 beams, N_TIME = 512 voltage columns per gemm-unit (16 detected outputs x n_ipo 32), input = uniform random
nibbles (all 16 codes) already resident in HBM.  One *step* = one launch over `--units` gemm-units (default 128 =
the MAX_TOTAL_SEP = 4 PSRDADA blocks of 32 gemm-units that the reference's scheduler keeps in flight,
src/beamformer.hh:85,114) = units*16 beam-blocks.  The metric unit is the
beam-block: one detected [256 freq][256 beams] float32 output.

N > 1 (strong scaling, BASELINE.json configs[3]): rank r owns frequencies [r*256/N, (r+1)*256/N) of every
gemm-unit; the only collective is the detected-power gather (RCCL all_to_all_single: every rank becomes the owner of
the full band for 1/N of the outputs; `--gather root` gathers everything on rank 0 instead, `--gather none` skips it).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

INT8_DENSE_PEAK_TOPS = 5000.0  # MI355X_MICROARCH.md: I8 MFMA = 2x the BF16 rate (~2.5 PF dense) -> ~5 POP/s
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=25)  # mirrors BURNIN 25, src/beamformer.hh:45
    ap.add_argument("--units", type=int, default=128,
                    help="gemm-units per step = one launch; default 128 = MAX_TOTAL_SEP (4) PSRDADA blocks of "
                         "N_GEMMS_PER_BLOCK (32) gemm-units, what the reference's scheduler keeps in flight "
                         "(src/beamformer.hh:85,114)")
    ap.add_argument("--nbuf", type=int, default=2, help="distinct input step-buffers cycled (defeats L2/MALL reuse)")
    ap.add_argument("--workload", default="c3", choices=["c3", "prod", "c2", "c5"],
                    help="c3: N_TIME 512 (16 outputs x n_ipo 32); prod: reference production N_TIME 256; "
                         "c2: DEBUG geometry N_TIME 16 (n_ipo 2, parity config; HBM-write bound); "
                         "c5: DSA100 scale-up, 100 ant x 512 beams x 1024 freq, N_TIME 256 (use --units 4)")
    ap.add_argument("--gather", default="alltoall", choices=["alltoall", "root", "none"])
    ap.add_argument("--detect", default="canonical", choices=["canonical", "fast"],
                    help="canonical (default): bit-exact detect; fast: opt-in fma-contracted detect (tolerance mode)")
    ap.add_argument("--input", default="random", choices=["random", "zeros", "const"],
                    help="experiment only: voltage bit patterns (MFMA power depends on operand toggling)")
    ap.add_argument("--force-dist", action="store_true",
                    help="single process: still create the RCCL process group (world size 1) and run the gather path "
                         "(plumbing check on a 1-GPU box; the driver's multi-GPU runs use torch.distributed.run)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the supplementary runs (fast detect, general kernel): profiling passes see one kernel")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def geometry(workload):
    # (n_avg, n_out_per_gemm): n_ipo = 2*n_avg, N_TIME = n_out*n_ipo
    return {"c3": (16, 16), "prod": (16, 8), "c2": (1, 8), "c5": (16, 8)}[workload]


def product_weights(torch, cfg, f0):
    """Steering weights for the linear DSA geometry, computed by the product's own host code (a5)."""
    import ctypes as C

    import numpy as np

    from dsabeamformer_amd import host

    if cfg.n_ant == 100:  # c5: 10x10 grid, 32x16 beam grid (notebook formulas)
        ax = np.linspace(-250, 250, 10)
        pos = np.zeros((100, 3), np.float32)
        pos[:, 0], pos[:, 1] = [v.ravel() for v in np.meshgrid(ax, ax)]
        th, ph = np.meshgrid(np.linspace(-3.5, 3.5, 32) * np.pi / 180, np.linspace(-3.5, 3.5, 16) * np.pi / 180)
        dirs = np.stack([th.ravel(), ph.ravel()], 1).astype(np.float32)
        return host.make_weights(pos, dirs, cfg.n_freq, chan0=f0, gpu=0)
    w = host.make_weights_default(n_beams=cfg.n_beams, n_ant=cfg.n_ant, n_freq_total=256, gpu=0)
    return np.ascontiguousarray(w[f0:f0 + cfg.n_freq])


def pmc_traffic(args, world):
    """HBM bytes per launch of the fused kernel from the committed rocprofv3 PMC passes (tools/pmc.sh: FETCH_SIZE
    and WRITE_SIZE collected in separate passes, KiB units; FETCH_SIZE doubled per MI355X_MICROARCH.md section HBM --
    gfx950 tallies 128-B read requests at 64 B).  Only valid for the exact launch it was measured on."""
    # (workload, units) each committed summary was measured on
    measured = {("c3", 128): "r01_c3_pmc_summary.txt", ("c5", 16): "r01_c5_pmc_summary.txt",
                ("c2", 128): "r01_c2_pmc_summary.txt"}
    name = measured.get((args.workload, args.units))
    if name is None or world != 1:
        return None
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None
    vals = {}
    for line in open(path):
        parts = line.split()
        if len(parts) >= 3 and parts[1] == "mean":
            vals[parts[0]] = float(parts[2])
    if "FETCH_SIZE" not in vals or "WRITE_SIZE" not in vals:
        return None
    return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0


def cpu_baseline(args, n_avg, n_out, seconds):
    """Oracle (C restatement, OpenMP, all host cores) on a bounded sample of the SAME workload."""
    import numpy as np

    import oracle as orc

    g = orc.Geom(n_avg=n_avg, n_out_per_gemm=n_out)
    pos, dirs = orc.default_positions(g.n_ant), orc.default_directions(g.n_beams)
    w = orc.make_weights(g, pos, dirs, 0)
    rng = np.random.default_rng(1)
    unit = rng.integers(0, 256, size=(1, g.n_freq, g.n_time, g.n_ant), dtype=np.uint8)
    orc.beamform(g, w, unit)  # warm-up (thread pool, page faults)
    n, t0 = 0, time.perf_counter()
    while True:
        orc.beamform(g, w, unit)
        n += 1
        el = time.perf_counter() - t0
        if el >= seconds or n >= 4096:
            break
    return {"value": n * g.n_out_per_gemm / el, "unit": "beam-blocks/s", "cores": orc.get_threads(), "kind": "port",
            "sample": "%d gemm-unit(s) of the bench workload (%d beam-blocks), oracle/dsabf_oracle.c -O3 -mavx2 "
                      "-fopenmp, %.1f s" % (n, n * g.n_out_per_gemm, el)}


def main():
    args = parse()
    # the pool's host driver only supports dmabuf IPC; without this RCCL's cross-process handles fail
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d"
                     % (args.gpus, args.gpus))
        args.gpus = world
    torch.cuda.set_device(local)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist

        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29541")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    import dsabeamformer_amd as bfm

    n_avg, n_out = geometry(args.workload)
    n_freq_total = 1024 if args.workload == "c5" else 256
    assert n_freq_total % world == 0
    n_freq = n_freq_total // world
    cfg = bfm.production_config(n_avg=n_avg, n_out_per_gemm=n_out, n_freq=n_freq)
    if args.workload == "c5":
        cfg.n_ant, cfg.n_beams = 100, 512
    cfg.detect_mode = 1 if args.detect == "fast" else 0
    n_ipo, n_time = cfg.n_pol * cfg.n_avg, n_out * cfg.n_pol * cfg.n_avg
    bf = bfm.Beamformer(cfg, device=local)
    bf.set_weights(product_weights(torch, cfg, rank * n_freq))

    units = args.units
    blocks_per_step = units * n_out                      # beam-blocks per step (whole job)
    in_bytes = units * n_freq * n_time * cfg.n_ant       # this rank's packed bytes per step
    out_floats = units * n_out * n_freq * cfg.n_beams    # this rank's detected floats per step
    gen = torch.Generator(device="cuda").manual_seed(0xD5A + rank)
    d_in = [torch.randint(0, 256, (in_bytes,), dtype=torch.uint8, device="cuda", generator=gen)
            for _ in range(max(1, args.nbuf))]
    if args.input != "random":
        for t in d_in:
            t.fill_(0 if args.input == "zeros" else 0x31)
    d_out = [torch.empty(out_floats, dtype=torch.float32, device="cuda") for _ in range(2)]
    stream = torch.cuda.current_stream()
    sptr = stream.cuda_stream

    # ---- gather plumbing (N > 1): dsabeamformer_amd/shard.py, covered by tests/test_shard_gloo.py ------------
    og = units * n_out
    gather = None
    if dist is not None and args.gather != "none":
        from dsabeamformer_amd.shard import DetectedGather

        gather = DetectedGather(torch, dist, args.gather, og, n_freq, cfg.n_beams, torch.device("cuda", local))

    def step(i, ev_pair=None):
        slot = i & 1
        if gather is not None:
            gather.finish(slot)  # the output buffer about to be overwritten must have left (and been re-laid out)
        if ev_pair:
            ev_pair[0].record(stream)
        bf.beamform(d_in[i % len(d_in)], units, d_out[slot], sptr)
        if ev_pair:
            ev_pair[1].record(stream)
        if gather is not None:
            gather.start(slot, d_out[slot])  # RCCL runs on its own stream, overlapping the next step's kernel

    def drain():
        if gather is not None:
            gather.finish(0)
            gather.finish(1)
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    drain()

    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i, events[i])
    drain()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if args.force_dist and world == 1 and gather is not None:
        # one rank: the gathered tensor must be the kernel's output of that slot, re-laid out as [o][f][b] (= unchanged)
        for slot in (0, 1):
            got = gather.full[slot]
            if got is not None and not torch.equal(got.reshape(-1), d_out[slot]):
                sys.exit("gather plumbing check failed: slot %d differs from the kernel output" % slot)

    kern_ms = sorted(a.elapsed_time(b) for a, b in events)
    kern_avg_ms = sum(kern_ms) / len(kern_ms)

    if rank == 0:
        total_blocks = args.steps * blocks_per_step
        value = total_blocks / elapsed
        # algorithmic work per beam-block: ops = 8*B*A*n_ipo*F ; bytes = A*n_ipo*F + 4*B*F
        ops_per_block = 8 * cfg.n_beams * cfg.n_ant * n_ipo * n_freq_total
        bytes_per_block = cfg.n_ant * n_ipo * n_freq_total + 4 * cfg.n_beams * n_freq_total
        launch_ops = ops_per_block * blocks_per_step / world    # per launch (this rank's kernel)
        launch_bytes = bytes_per_block * blocks_per_step / world
        mfma_bound = args.workload != "c2"
        if mfma_bound:
            achieved = launch_ops / (kern_avg_ms * 1e-3) / 1e12
            roof = {"bound": "mfma", "achieved": achieved, "peak": INT8_DENSE_PEAK_TOPS, "unit": "TFLOP/s",
                    "frac": achieved / INT8_DENSE_PEAK_TOPS, "traffic": pmc_traffic(args, world)}
        else:
            achieved = launch_bytes / (kern_avg_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(args, world)}
        info = bf.kernel_info(units)
        roof.update({"kernel": info["kernel"],
                     "kernel_ms_avg": kern_avg_ms, "kernel_ms_median": kern_ms[len(kern_ms) // 2],
                     "kernel_ms_min": kern_ms[0], "algorithmic_ops_per_launch": launch_ops,
                     "algorithmic_bytes_per_launch": launch_bytes,
                     "note": "unit is int8 TOP/s (1 complex MAC = 8 ops); traffic = HBM bytes per launch from the "
                             "committed PMC passes (profiles/r01_c3_pmc_summary.txt, r01_c5_pmc_summary.txt), null if this launch differs"})
        out = {
            "metric": "beam-blocks/sec (%d beams x %d freq x N_TIME)" % (cfg.n_beams, n_freq_total), "value": value,
            "unit": "beam-blocks/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "ms_per_block": elapsed / total_blocks * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "int8 (int32 accumulate, fp32 detect)",
            "data": "synthetic: uniform random 4-bit complex voltages resident in HBM (%d step buffers), steering "
                    "weights of the linear 64-antenna / 256-beam DSA geometry" % len(d_in),
            "config": {"workload": {"c3": "C3: 64 ant x 2 pol, 256 freq, 256 beams, N_TIME=512 (16 outputs x n_ipo 32)",
                                    "prod": "reference production: N_TIME=256 (8 outputs x n_ipo 32)",
                                    "c2": "C2 DEBUG geometry: N_TIME=16 (8 outputs x n_ipo 2)",
                                    "c5": "C5 DSA100 scale-up: 100 ant x 2 pol, 1024 freq, 512 beams, N_TIME=256; the "
                                          "unit is a 512-beam x 1024-freq block"}[args.workload],
                       "gemm_units_per_step": units, "beam_blocks_per_step": blocks_per_step,
                       "freq_per_gpu": n_freq, "gather": args.gather if dist is not None else "n/a",
                       "detect_mode": args.detect,
                       "launch": info},
            "roofline": roof,
        }
        paired = "PAIRED" in info["kernel"]
        roof["executed_mfma_ops_per_launch"] = launch_ops / 2 if paired else launch_ops
        if paired:
            roof["note"] += ("; the beam set is symmetric about the boresight, so the conjugate-pair kernel executes half "
                             "the algorithmic int8 ops on the MFMA pipe (same bits) -- achieved/frac stay ALGORITHMIC "
                             "ops over time, the general kernel on the same input is reported under general_kernel")
        if world == 1 and args.detect == "canonical" and args.workload in ("c3", "prod") and not args.no_extras:
            def supplementary(detect_mode, env=None):
                old = os.environ.get("DSABF_PAIRED")
                if env is not None:
                    os.environ["DSABF_PAIRED"] = env
                try:
                    cfg2 = bfm.production_config(n_avg=n_avg, n_out_per_gemm=n_out, n_freq=n_freq, detect_mode=detect_mode)
                    bf2 = bfm.Beamformer(cfg2, device=local)
                    bf2.set_weights(product_weights(torch, cfg2, 0))
                finally:
                    if env is not None:
                        if old is None:
                            del os.environ["DSABF_PAIRED"]
                        else:
                            os.environ["DSABF_PAIRED"] = old
                n2 = max(10, args.steps // 4)
                for i in range(5):
                    bf2.beamform(d_in[i % len(d_in)], units, d_out[i & 1], sptr)
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                for i in range(n2):
                    bf2.beamform(d_in[i % len(d_in)], units, d_out[i & 1], sptr)
                torch.cuda.synchronize()
                el2 = time.perf_counter() - t2
                name = bf2.kernel_info(units)["kernel"]
                bf2.close()
                v2 = n2 * blocks_per_step / el2
                return {"value": v2, "unit": "beam-blocks/s", "frac": v2 * ops_per_block / 1e12 / INT8_DENSE_PEAK_TOPS,
                        "kernel": name}

            # supplementary, never the headline: the opt-in tolerance mode (BF_DETECT_FAST) on the same inputs ...
            out["fast_detect_mode"] = supplementary(1)
            out["fast_detect_mode"].update({"tolerance": "4*n_ipo*2^-24 relative to the canonical (bit-exact) result",
                                            "note": "opt-in bf_config.detect_mode = BF_DETECT_FAST; not the headline"})
            if paired:
                # ... and the general kernel (what weights without the conjugate symmetry run), same inputs, same bits
                out["general_kernel"] = supplementary(0, env="0")
                out["general_kernel"]["note"] = ("DSABF_PAIRED=0: every int8 op of the algorithmic count executes on "
                                                 "the MFMA pipe; not the headline")
        if world == 1 and args.workload == "c3" and not args.no_extras:
            # supplementary: BASELINE configs[1], the reference's DEBUG geometry (N_TIME 16, n_ipo 2) -- the parity
            # configuration; HBM-write-bound (4 B out per 0.125 B in per beam), so its roofline is the HBM one
            cfg3 = bfm.production_config(n_avg=1, n_out_per_gemm=8, n_freq=n_freq)
            bf3 = bfm.Beamformer(cfg3, device=local)
            bf3.set_weights(product_weights(torch, cfg3, 0))
            in3 = units * n_freq * 16 * cfg3.n_ant
            out3 = torch.empty(units * 8 * n_freq * cfg3.n_beams, dtype=torch.float32, device="cuda")
            for i in range(10):
                bf3.beamform(d_in[i % len(d_in)][:in3], units, out3, sptr)
            ev3 = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(100)]
            for i, (a, b) in enumerate(ev3):
                a.record(stream)
                bf3.beamform(d_in[i % len(d_in)][:in3], units, out3, sptr)
                b.record(stream)
            torch.cuda.synchronize()
            ms3 = sum(a.elapsed_time(b) for a, b in ev3) / len(ev3)
            bytes3 = (cfg3.n_ant * 2 * n_freq + 4 * cfg3.n_beams * n_freq) * units * 8
            out["debug_geometry"] = {"workload": "C2: BASELINE configs[1], N_TIME=16 (8 outputs x n_ipo 2), %d gemm-units per launch" % units,
                                     "value": units * 8 / (ms3 * 1e-3), "unit": "beam-blocks/s", "kernel_ms_avg": ms3,
                                     "kernel": bf3.kernel_info(units)["kernel"],
                                     "roofline": {"bound": "hbm", "achieved": bytes3 / (ms3 * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                                                  "unit": "GB/s", "frac": bytes3 / (ms3 * 1e-3) / 1e9 / HBM_PEAK_GBS},
                                     "note": "not the headline; bit-exact parity on this geometry is what tests/ check"}
            bf3.close()
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, n_avg, n_out, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    bf.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
