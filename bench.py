#!/usr/bin/env python3
"""bench.py -- throughput of the beamformer hot path (fused 4-bit expand -> int8 MFMA GEMM -> power detect).

Contract (driver):  python bench.py --gpus N --steps K --warmup W   (N > 1: one rank per GPU under
torch.distributed.run; RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment).  Rank 0 prints ONE JSON line.
Called plainly with --gpus N > 1 (no WORLD_SIZE in the environment) it starts that launcher itself as a CHILD process --
before anything here touches the GPU -- and passes on rank 0's line and the exit code (self_launch below).

Workload (BASELINE.json configs[2] geometry, "C3" in SURVEY.md section 8d): 64 antennas x 2 pol, 256 frequencies,
256 beams, N_TIME = 512 voltage columns per gemm-unit (16 detected outputs x n_ipo 32), input = uniform random
nibbles (all 16 codes) already resident in HBM.  One *step* = one launch over `--units` gemm-units (default 128 =
the MAX_TOTAL_SEP = 4 PSRDADA blocks of 32 gemm-units that the reference's scheduler keeps in flight,
src/beamformer.hh:85,114) = units*16 beam-blocks.  The metric unit is the beam-block: one detected [256 freq][256 beams]
float32 output.

Order of work (so that the GPU is busy for seconds around the timed region and nothing long runs before it):
  warm-up (W steps, then more of the same until >= 1 s has passed: DVFS settles, the smi sampler sees the burst)
  -> the timed K steps (barrier + synchronize on both sides, max over ranks) -> supplementary GPU records
  -> the CPU baselines, LAST (announced on stderr; ~20 s of host work).

N > 1 (strong scaling, BASELINE.json configs[3]): rank r owns frequencies [r*256/N, (r+1)*256/N) of every gemm-unit; the
only collective is the gather of the detected powers, through the C-ABI (bf_comm_create / bf_gather_detected: RCCL
point-to-point over xGMI, include/dsabf.h).  `--gather` picks the headline mode (default alltoall = distributed owners);
every mode -- none, root, alltoall, each in both layouts and transports -- is then timed side by side under "gather_modes",
so a scaling run separates kernel scaling from xGMI.

N > 1 cannot fail softly:
  * a process-level deadline (class Deadline) is armed at the top of main(), before torch is imported, before the process
    group and before any RCCL call; every stage that can block names itself and has its own time limit.  When one expires
    rank 0 prints ONE JSON line with what exists -- the stage, the rank, the kernel-only record if it was taken -- and the
    process ends with a NON-ZERO status (a fresh exit, never a re-exec);
  * the control plane (barrier, max-over-ranks, unique-id broadcast, checksum exchange) runs on gloo over 127.0.0.1: the
    GPU fabric carries the data path only, so a fabric problem cannot take the measurement's own bookkeeping with it;
  * kernel-only strong scaling (gather mode "none": no RCCL communicator exists yet) is measured FIRST and stored; only then
    is the communicator created and the headline gather warmed up and timed.  If that hangs or fails, the line still carries
    gather_modes.none, "gather_error" names the stage, value is null and the exit status is non-zero;
  * every gather mode is VERIFIED after it is timed: each rank publishes position-weighted 64-bit checksums of its local
    rows (a small all-gather on the control plane), each receiver recomputes them on what it holds:
    gather_modes[k].verified.  A headline mode that delivered anything else ends the run non-zero.
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import signal
import sys
import threading
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
    ap.add_argument("--gather", default="alltoall", choices=["alltoall", "root", "none"],
                    help="N > 1: the headline's gather mode (all modes are also timed side by side)")
    ap.add_argument("--layout", default="rank", choices=["rank", "freq"],
                    help="gathered layout: rank = sub-band-major [rank][row][f_local][b], one message per sender; "
                         "freq = the reference's [row][f over the band][b], one message per (row, sender)")
    ap.add_argument("--detect", default="canonical", choices=["canonical", "contracted", "fast"],
                    help="canonical (default): x*x + y*y as written; contracted: nvcc's fma(x,x,y*y); fast: tolerance mode")
    ap.add_argument("--input", default="random", choices=["random", "zeros", "const"],
                    help="experiment only: voltage bit patterns (MFMA power depends on operand toggling)")
    ap.add_argument("--force-dist", action="store_true",
                    help="single process: still create the process group and a one-rank RCCL communicator and run the "
                         "gather path (plumbing check on a 1-GPU box)")
    ap.add_argument("--dist-backend", default="gloo", choices=["gloo", "nccl"],
                    help="torch.distributed backend of the CONTROL PLANE only (barrier, max-over-ranks, id broadcast, checksum "
                         "exchange).  gloo (default, loopback TCP): no RCCL traffic before the kernel-only record exists, and a "
                         "fabric problem cannot hang the bookkeeping; nccl: torch's own RCCL communicator.  The data path -- "
                         "the gather -- is RCCL point-to-point behind the C-ABI either way")
    ap.add_argument("--deadline-seconds", type=float, default=1500.0,
                    help="process-level limit from the top of main(): when it expires before the headline exists, rank 0 prints "
                         "a diagnostic line (stage, rank) and every rank exits non-zero")
    ap.add_argument("--stage-seconds", type=float, default=300.0,
                    help="limit of each blocking stage before the headline (imports, process group, handle, kernel-only region)")
    ap.add_argument("--gather-timeout", type=float, default=180.0,
                    help="limit of each stage that runs RCCL: communicator creation, and the warm-up / timed region / "
                         "verification of each gather mode")
    ap.add_argument("--no-verify", action="store_true", help="skip the checksum verification of the gather modes")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--budget-seconds", type=float, default=900.0,
                    help="wall-clock limit for everything AFTER the headline measurement (supplementary records, gather modes, "
                         "CPU baselines): when it expires the line is printed with what is there and the process exits 0")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the supplementary records: profiling passes see one kernel")
    ap.add_argument("--min-warm-seconds", type=float, default=1.0)
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    return ap.parse_args()


DETECT = {"canonical": 0, "fast": 1, "contracted": 2}


def geometry(workload):
    # (n_avg, n_out_per_gemm): n_ipo = 2*n_avg, N_TIME = n_out*n_ipo
    return {"c3": (16, 16), "prod": (16, 8), "c2": (1, 8), "c5": (16, 8)}[workload]


def grid_100():
    """c5: 10x10 antenna grid, 32x16 beam grid (the notebook's formulas)."""
    import numpy as np

    ax = np.linspace(-250, 250, 10)
    pos = np.zeros((100, 3), np.float32)
    pos[:, 0], pos[:, 1] = [v.ravel() for v in np.meshgrid(ax, ax)]
    th, ph = np.meshgrid(np.linspace(-3.5, 3.5, 32) * np.pi / 180, np.linspace(-3.5, 3.5, 16) * np.pi / 180)
    return pos, np.stack([th.ravel(), ph.ravel()], 1).astype(np.float32)


def product_weights(cfg, f0):
    """Steering weights computed by the product's own host code (a5)."""
    import numpy as np

    from dsabeamformer_amd import host

    if cfg.n_ant == 100:
        pos, dirs = grid_100()
        return host.make_weights(pos, dirs, cfg.n_freq, chan0=f0, gpu=0)
    w = host.make_weights_default(n_beams=cfg.n_beams, n_ant=cfg.n_ant, n_freq_total=256, gpu=0)
    return np.ascontiguousarray(w[f0:f0 + cfg.n_freq])


def calibrated_weights(w, seed=7):
    """What a real array uploads: the steering fan times a per-(frequency, antenna) complex calibration gain.  Here the
    gains are unit-modulus with a seeded random phase; the product is re-quantised to int8 (clipped to +-127).  Any such
    set breaks W[B-1-b] = conj(W[b]), so bf_set_weights selects the GENERAL kernel -- the production number."""
    import numpy as np

    rng = np.random.default_rng(seed)
    phi = rng.uniform(0.0, 2.0 * np.pi, size=w.shape[:2])                     # [f][a]
    g = np.exp(1j * phi)[:, :, None]
    c = (w[..., 0].astype(np.float64) + 1j * w[..., 1].astype(np.float64)) * g
    out = np.empty(w.shape, np.int8)
    out[..., 0] = np.clip(np.rint(c.real), -127, 127)
    out[..., 1] = np.clip(np.rint(c.imag), -127, 127)
    return out


def pmc_key(kernels, variant, workload, units, paired, detect):
    """What identifies a counter pass: the device code (build.kernel_build_id(), also the tail of bf_version()), the kernel
    instantiation and the launch.  bench.py prints it on stderr at start (tools/pmc.sh copies that line into the summary's first
    line); pmc_for_launch() accepts a committed summary only if its key is THIS run's."""
    return "kernels=%s variant=%s workload=%s units=%d paired=%d detect=%s" % (kernels, variant.replace(" ", ""), workload, units, int(bool(paired)), detect)


def pmc_for_launch(key, profiles_dir=None):
    """(counter means, file name, stale) of the committed rocprofv3 PMC summary (profiles/*pmc_summary*.txt, tools/pmc.sh) whose first
    line carries exactly `key` -- the counters of THIS kernel build and launch (a kernel change without a profile
    refresh must not pair new timings with old counters).  No such file: ({}, the newest summary of the same workload and launch
    if there is one, True) -- the bench line then says traffic: null, pmc_stale: true and names what it did not use."""
    profiles_dir = profiles_dir or os.path.join(ROOT, "profiles")
    launch = key.split(" ", 2)[2]                      # workload= units= paired= detect=
    near = None
    for path in sorted(glob.glob(os.path.join(profiles_dir, "*pmc_summary*.txt")), reverse=True):
        try:
            first = open(path).readline().strip()
        except OSError:
            continue
        if not first.startswith("# pmc_key "):
            continue
        have = first[len("# pmc_key "):]
        if have == key:
            return pmc_summary(os.path.basename(path), profiles_dir), os.path.basename(path), False
        if near is None and have.split(" ", 2)[2:] == [launch]:
            near = os.path.basename(path)
    return {}, near, True


def pmc_summary(name, profiles_dir=None):
    """Counter means of one committed rocprofv3 PMC summary (tools/pmc.sh), {} if absent."""
    path = os.path.join(profiles_dir or os.path.join(ROOT, "profiles"), name)
    vals = {}
    if os.path.exists(path):
        fused = True        # sections are headed by a kernel name; only the fused kernel's counters are wanted (the streaming
        for line in open(path):   # micro-benchmark of an HBM-bound line launches expand_kernel in the same process)
            parts = line.split()
            if len(parts) >= 3 and parts[1] == "mean":
                if fused:
                    vals[parts[0]] = float(parts[2])
            elif line.strip():
                fused = "fused16_kernel" in line
    return vals


def pmc_traffic(vals):
    """HBM bytes per launch: FETCH_SIZE and WRITE_SIZE (KiB) come from separate passes; FETCH_SIZE doubled per
    MI355X_MICROARCH.md section HBM (gfx950 tallies 128-B read requests at 64 B)."""
    if "FETCH_SIZE" not in vals or "WRITE_SIZE" not in vals:
        return None
    return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0


def pmc_mfma_busy(vals):
    """Fraction of SIMD cycles the matrix pipe is busy: SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)."""
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in vals or "GRBM_GUI_ACTIVE" not in vals:
        return None
    return vals["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * vals["GRBM_GUI_ACTIVE"] / 8.0)


def issue_model(vals, kernel_ms):
    """What binds the fused kernel, from the committed PMC pass of this launch (docs/PERF_MODEL.md section 2):
    a wave that issues MFMAs and the detect's VALU ops needs ~13 + 2.45 K cycles of its SIMD's issue per MFMA, K = VALU ops
    per MFMA.  issue_occupancy = (MFMAs per SIMD x that) / the cycles the launch took: near 1 means the SIMDs' instruction
    issue is the bound, not the matrix pipe (whose own busy fraction is mfma_busy_frac).  {} if the pass is not committed."""
    need = ("SQ_INSTS_VALU", "SQ_INSTS_VALU_MFMA_I8", "GRBM_GUI_ACTIVE")
    if not all(k in vals for k in need) or not vals["SQ_INSTS_VALU_MFMA_I8"]:
        return {"bound_measured": None}
    mfma = vals["SQ_INSTS_VALU_MFMA_I8"]
    k = (vals["SQ_INSTS_VALU"] - mfma) / mfma
    cyc_per_mfma = 13.0 + 2.45 * k
    cycles = vals["GRBM_GUI_ACTIVE"] / 8.0                       # the counter sums the 8 XCDs
    occupancy = (mfma / 1024.0) * cyc_per_mfma / cycles          # 1024 SIMDs
    busy = pmc_mfma_busy(vals)
    rec = {"valu_per_mfma": k, "issue_model_cycles_per_mfma": cyc_per_mfma, "issue_occupancy": occupancy,
           "bound_measured": "simd-issue" if (occupancy > 0.8 and (busy is None or busy < 0.7)) else "mfma-pipe" if (busy or 0) >= 0.7 else "unclear",
           "issue_model": "13 + 2.45 * valu_per_mfma cycles of a SIMD's issue per MFMA (docs/PERF_MODEL.md section 2); "
                          "issue_occupancy = MFMAs per SIMD x that / launch cycles; bound = the algorithmic roof the fraction "
                          "is quoted against, bound_measured = what the counters say limits the kernel"}
    # the clock the chip held in the profiled pass (profiled passes clock a little lower than timed ones)
    prof_ms = vals["_KERNEL_NS_GRBM_PASS"] * 1e-6 if vals.get("_KERNEL_NS_GRBM_PASS") else None
    rec["clock_ghz_under_load"] = cycles / ((prof_ms or kernel_ms) * 1e-3) / 1e9
    rec["clock_note"] = ("GRBM_GUI_ACTIVE / 8 XCDs / %s" % ("the kernel's average duration in the same rocprofv3 pass" if prof_ms else
                                                            "this run's HIP-event kernel time (the counters are from the committed pass)"))
    return rec


def time_launches(torch, fn, n, stream):
    """Average duration (ms) of n launches of fn(i), HIP events on the launch stream around each."""
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for i, (a, b) in enumerate(ev):
        a.record(stream)
        fn(i)
        b.record(stream)
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    return sum(ms) / len(ms), ms[len(ms) // 2], ms[0]


def dm_stage_record(torch, bfm, device, n_dm=64, trial_share=(0, 1), pushes=24, warm=6, blocks_per_push=1, queues=1):
    """The DM-trial stage of the observation loop by itself: kernel-only time per PRODUCTION
    block -- 32 gemm-units x 8 outputs = 256 beam-blocks of 256 x 256 floats, 64 MiB -- for `n_dm` trials of the notebook's ladder to
    DM 250, the delay window carried over on the device.  Zero-copy feed: bf_dm_stream_reserve hands out the rows' place inside the
    stage's buffer (in the loop the beamformer or the gather writes them there; here they hold what a first fill left), the push only
    launches -- so what is timed is the stage's own work: the dedispersion kernels over [carry | new rows] (the buffer is a ring
    mapped twice back to back: nothing slides).  trial_share = (r, R): rank r's share of the ladder when a sharded run splits
    the trials (`beam -X`; R = 1: the whole ladder on one GPU).  blocks_per_push > 1: the stage fed every few blocks -- a block's
    64 trials x 256 times x 256 beams are 64 workgroups of the shared-window kernel, a quarter of the chip.  queues > 1: consecutive
    blocks pushed on alternating HIP streams, as run_observation's two compute queues do: up to three pushes are in flight and
    their tiles run side by side (us_per_block is then the host-timed rate over all pushes, not one push's HIP-event time)."""
    import ctypes as C

    import numpy as np

    from dsabeamformer_amd import _lib, api, host

    hip = _lib._preload_hip_runtime()
    pc = bfm.production_config()
    rows = pc.n_gemms_per_block * pc.n_out_per_gemm * blocks_per_push
    freq = [host.channel_frequency(0, c) for c in range(pc.n_freq)]
    ladder = host.dm_trials(dm_max=250.0)
    dms = ladder[:: max(1, len(ladder) // n_dm)][:n_dm]
    delays = host.dm_delays(dms, freq, freq[0], 0.131)
    first, count = host.dm_trial_share(len(dms), trial_share[1], trial_share[0]) if trial_share[1] > 1 else (0, len(dms))
    delays = np.ascontiguousarray(delays[first:first + count])
    D = int(delays.max())
    stream = torch.cuda.current_stream()
    sptr = stream.cuda_stream
    qs = [sptr] if queues <= 1 else [torch.cuda.Stream() for _ in range(queues)]
    qptr = [q if isinstance(q, int) else q.cuda_stream for q in qs]
    b = bfm.Beamformer(pc, device=device)
    dm = api.DmStream(b, delays, pc.n_freq, rows)
    ring = bool(b.counter("dm_ring_stages"))     # the stage's buffer: the twice-mapped ring (nothing slides), or the linear fallback
    row_floats = pc.n_freq * pc.n_beams
    fill = torch.rand(rows * row_floats, device="cuda")

    def one(i):
        q = qptr[i % len(qptr)]
        dst = dm.reserve(rows, q)
        if i < n_fill:       # (first pass over the buffer: real values in every row the kernels will read)
            hip.hipMemcpyAsync(C.c_void_p(dst), C.c_void_p(fill.data_ptr()), C.c_size_t(rows * row_floats * 4), 3, C.c_void_p(q))
        dm.push(dst, rows, None, q)

    n_fill = 2 * ((D + rows) // rows + 2)
    for i in range(n_fill + warm):
        one(i)
    torch.cuda.synchronize()
    if len(qptr) == 1:
        avg, med, mn = time_launches(torch, lambda i: one(n_fill + warm + i), pushes, stream)
    else:
        reps = []
        for r in range(5):
            t0 = time.perf_counter()
            for i in range(pushes):
                one(n_fill + warm + r * pushes + i)
            torch.cuda.synchronize()
            reps.append((time.perf_counter() - t0) * 1e3 / pushes)
        reps.sort()
        avg, med, mn = sum(reps) / len(reps), reps[len(reps) // 2], reps[0]
    dm.close()
    b.close()
    window = D + rows                                           # rows the kernels read per push: [carry | new]
    alg = 4 * (window * row_floats + count * rows * pc.n_beams)   # read once + the chunk [dm][t][b] written once
    adds = float(count) * rows * pc.n_freq * pc.n_beams
    return {"us_per_block": avg * 1e3 / blocks_per_push, "us_per_push": avg * 1e3, "us_per_push_median": med * 1e3, "us_per_beam_block": avg * 1e3 / rows,
            "dm_trials": count, "rows_per_push": rows, "blocks_per_push": blocks_per_push, "queues": len(qptr), "max_delay_rows": D, "ring_buffer": ring, "algorithmic_bytes_per_push": alg,
            "hbm_gbs_algorithmic": alg / (avg * 1e-3) / 1e9, "gadds_per_s": adds / (avg * 1e-3) / 1e9}


def cpu_baselines(n_avg, n_out, seconds, full=True):
    """Host-side baselines, on a bounded sample (full=False, the N > 1 line: record (1) alone, on all host cores): (1) the oracle's beamform port (expand + GEMM + detect, OpenMP, all cores)
    on gemm-units of the bench workload; (2) the reference's only CPU code on the path, generate_test_data
    (src/test_data_generator.hh:63-95): the PRODUCT's generator for one 1024-unit DEBUG batch on 1 core and on all cores,
    and the oracle's literal restatement of the reference loop (trig for every time column, as `make fast_debug` runs it)."""
    import numpy as np

    import oracle as orc
    from dsabeamformer_amd import debug_config, host

    g = orc.Geom(n_avg=n_avg, n_out_per_gemm=n_out)
    pos, dirs = orc.default_positions(g.n_ant), orc.default_directions(g.n_beams)
    w = orc.make_weights(g, pos, dirs, 0)
    if os.environ.get("OMP_NUM_THREADS") == "1" and "WORLD_SIZE" in os.environ:
        orc.set_threads(os.cpu_count() or 1)     # a rank under torchrun (which pins OMP_NUM_THREADS=1): still all the cores
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
    out = {"value": n * g.n_out_per_gemm / el, "unit": "beam-blocks/s", "cores": orc.get_threads(), "kind": "port",
           "sample": "%d gemm-unit(s) of the bench workload (%d beam-blocks), oracle/dsabf_oracle.c -O3 -mavx2 "
                     "-fopenmp, %.1f s" % (n, n * g.n_out_per_gemm, el)}
    out["ms_per_beam_block"] = 1e3 * el / (n * g.n_out_per_gemm)
    try:
        out["cpu_model"] = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
        out["nproc"] = os.cpu_count()
    except Exception:   # pragma: no cover
        pass
    if not full:
        return out
    # the same port at the parity geometry (C1/C2: N_IPO 2, N_TIME 16), BASELINE.md section 4 item 2: a 3 s sample
    g1 = orc.DEBUG_GEOM
    w1 = orc.make_weights(g1, orc.default_positions(g1.n_ant), orc.default_directions(g1.n_beams), 0)
    units1 = rng.integers(0, 256, size=(8, g1.n_freq, g1.n_time, g1.n_ant), dtype=np.uint8)
    orc.beamform(g1, w1, units1)
    n1, t1 = 0, time.perf_counter()
    while True:
        orc.beamform(g1, w1, units1)
        n1 += units1.shape[0]
        el1 = time.perf_counter() - t1
        if el1 >= 3.0 or n1 >= 1 << 16:
            break
    out["debug_geometry"] = {"value": n1 * g1.n_out_per_gemm / el1, "unit": "beam-blocks/s", "cores": orc.get_threads(),
                             "ms_per_beam_block": 1e3 * el1 / (n1 * g1.n_out_per_gemm),
                             "sample": "%d gemm-units of the DEBUG geometry (N_IPO 2), %.1f s" % (n1, el1)}
    # (2) the generator
    dbg = debug_config()
    src = host.read_directions(os.path.join(ROOT, "tests", "golden", "config", "linear_source_directions_1024.txt"))
    gen_rec = {"unit": "s per 1024-gemm-unit DEBUG batch (256 MiB)", "reference": "src/test_data_generator.hh:63-95"}
    ncores = os.cpu_count() or 1
    for label, nt in (("product_1_core", 1), ("product_all_cores", ncores)):
        os.environ["DSABF_THREADS"] = str(nt)
        gen = host.TestDataGenerator(dbg, 1024, pin=False)
        gen.set_source_directions(src)
        ppos = host.default_positions(64)
        gen.data()[:] = 0                        # first touch of the 256 MiB buffer outside the timed call
        t0 = time.perf_counter()
        gen.generate_test_data(ppos, 0)          # batch 0 = sources 0..1023 of the catalogue
        gen_rec[label] = {"value": time.perf_counter() - t0, "cores": nt}
        gen.close()
    os.environ.pop("DSABF_THREADS", None)
    gen_rec["product_note"] = ("dsabf::test_data_generator evaluates the trig once per (source, frequency, antenna) and "
                               "replicates it over the time columns (the reference's expression has no column index)")
    # the reference's loop as written (trig for every column), OpenMP over sources like `make fast_debug` (makefile:16):
    # a bounded 128-unit sample, scaled to the 1024-unit batch
    gd = orc.DEBUG_GEOM
    opos = orc.default_positions(gd.n_ant)
    t0 = time.perf_counter()
    orc.generate_test_data(gd, opos, src, 0, 0, 128, literal=True)
    el = time.perf_counter() - t0
    gen_rec["reference_loop_all_cores"] = {"value": el * 8.0, "cores": orc.get_threads(), "kind": "port",
                                           "sample": "128 of 1024 gemm-units, literal restatement, x8"}
    out["generator"] = gen_rec
    return out


def launcher_command(n, argv, port):
    """The command the driver itself uses for N > 1: one rank per GPU of ONE node, rendezvous on 127.0.0.1."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
            "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


METRIC_NAME = "beam-blocks/sec (256 beams x 256 freq x N_TIME)"


def count_gpus():
    """GPUs this process could open, WITHOUT loading the HIP runtime (torch.cuda.device_count() loads it; the parent of the
    ranks must stay a process that never touched the GPU): KFD topology nodes with SIMDs whose render node is present and
    accessible (a container that was handed one GPU of an 8-GPU host sees all eight topology nodes but one /dev/dri/renderD*),
    cut to HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES when those are set.  DSABF_KFD_TOPOLOGY / DSABF_DRI_DIR: tests."""
    base = os.environ.get("DSABF_KFD_TOPOLOGY", "/sys/class/kfd/kfd/topology/nodes")
    dri = os.environ.get("DSABF_DRI_DIR", "/dev/dri")
    n = 0
    for path in sorted(glob.glob(os.path.join(base, "*", "properties"))):
        props = {}
        try:
            for line in open(path):
                k, _, v = line.strip().partition(" ")
                props[k] = v
        except OSError:
            continue
        try:
            if int(props.get("simd_count", "0")) <= 0:
                continue                                   # a CPU node
            minor = int(props.get("drm_render_minor", "-1"))
        except ValueError:
            continue
        if minor >= 0 and not os.access(os.path.join(dri, "renderD%d" % minor), os.R_OK | os.W_OK):
            continue                                       # not handed to this container
        n += 1
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip()]))
    return n


def count_gpus_in_child(timeout_s=120.0):
    """What a short-lived CHILD process says torch.cuda.device_count() is (counting devices does not initialise the GPU on this image;
    the parent still never loads HIP).  The second opinion when the sysfs walk of count_gpus() comes up short: a container that exposes
    /dev/kfd without that sysfs tree, or render nodes under another permission model, must not lose the whole run to it.  None: the
    child failed or did not answer in time (first `import torch` on a fresh box can take a minute or two).  DSABF_BENCH_CHILD_COUNT:
    tests (the child's answer, without a child)."""
    import subprocess

    fake = os.environ.get("DSABF_BENCH_CHILD_COUNT")
    if fake is not None:
        return int(fake) if fake.strip().lstrip("-").isdigit() else None
    try:
        r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True,
                           timeout=timeout_s, start_new_session=True)
        return int(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 and r.stdout.strip() else None
    except Exception:
        return None


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: run torch.distributed.run as a CHILD process and exit
    with its status.  This process never initialises the GPU -- the device count comes from sysfs (count_gpus), torch is not
    even imported -- and the ranks are children, never an exec of this process.  stdout / stderr are inherited, so rank 0's
    one JSON line is this command's one JSON line.  The child runs in its own session under this process's deadline: if the
    launcher itself does not come back (a rendezvous that never completes), its whole process group is killed, a diagnostic
    line is printed and the status is non-zero."""
    import socket
    import subprocess

    one_gpu = os.environ.get("DSABF_BENCH_ONE_GPU") == "1"     # test mode: the ranks time-share GPU 0
    if not one_gpu:
        have = count_gpus()
        have_child = None
        if have < args.gpus:
            # the sysfs walk can be wrong (no topology tree in this container, another permission model on the render nodes): ask a
            # child before refusing
            have_child = count_gpus_in_child()
        if have < args.gpus and (have_child is None or have_child < args.gpus):
            # refused -- as ONE JSON line on stdout like every other outcome of this command (value null + error), both counts in it
            msg = ("bench.py --gpus %d: this node shows %d GPU(s) in sysfs (%s) and %s to a child process's torch.cuda.device_count()"
                   % (args.gpus, have, os.environ.get("DSABF_KFD_TOPOLOGY", "/sys/class/kfd/kfd/topology/nodes"),
                      "no answer" if have_child is None else "%d" % have_child))
            print(json.dumps({"metric": METRIC_NAME, "value": None, "unit": "beam-blocks/s", "n_gpus": args.gpus, "error": msg,
                              "stage": "counting the GPUs (no rank was started)", "rank": None,
                              "gpus_seen": {"sysfs": have, "child_device_count": have_child}}), flush=True)
            print(msg, file=sys.stderr, flush=True)
            sys.exit(6)
        if have < args.gpus:
            print("bench.py: sysfs shows %d GPU(s), a child process's torch.cuda.device_count() %d: going by the child"
                  % (have, have_child), file=sys.stderr, flush=True)
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as s:                              # a free port of the loopback interface
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")           # dmabuf IPC: what RCCL needs on this pool
    env.setdefault("OMP_NUM_THREADS", "1")                      # (what torchrun would set, without its warning)
    cmd = launcher_command(args.gpus, sys.argv[1:], port)
    print("bench.py: --gpus %d without a launcher: starting %s" % (args.gpus, " ".join(cmd[1:9])), file=sys.stderr, flush=True)
    limit = args.deadline_seconds + float(os.environ.get("DSABF_LAUNCH_GRACE", "90"))   # the ranks' own deadlines fire first and say more
    child = subprocess.Popen(cmd, env=env, start_new_session=True)

    def pass_on(signum, _frame):
        """The ranks live in a session of their own: a signal that ends THIS process must end them too (first SIGTERM -- their
        sigwait threads print what they have -- then, if the launcher lingers, SIGKILL), or they would keep the GPUs."""
        for sig, wait_s in ((signal.SIGTERM, 15.0), (signal.SIGKILL, 5.0)):
            try:
                os.killpg(child.pid, sig)
            except OSError:
                break
            try:
                child.wait(timeout=wait_s)
                break
            except subprocess.TimeoutExpired:
                continue
        os._exit(128 + signum)

    for sig in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
        signal.signal(sig, pass_on)
    try:
        sys.exit(child.wait(timeout=limit))
    except subprocess.TimeoutExpired:
        try:
            os.killpg(child.pid, signal.SIGKILL)                # the launcher and every rank it started (its own session)
        except OSError:
            pass
        print(json.dumps({"metric": METRIC_NAME, "value": None, "unit": "beam-blocks/s", "n_gpus": args.gpus,
                          "error": "the launcher did not return within %.0f s; its process group was killed" % limit,
                          "stage": "torch.distributed.run (child)", "rank": None}), flush=True)
        sys.exit(5)


LINE_OUT = None     # where the ONE JSON line goes once main() has claimed stdout for it


def claim_stdout():
    """Only the JSON line may reach stdout.  Native libraries print there too -- gloo writes "[Gloo] Rank 0 is connected to 7 peer
    ranks" once per rank at N > 1 -- so fd 1 is pointed at stderr for everything else (python's prints included) and the line is
    written to a duplicate of the real stdout."""
    global LINE_OUT
    if LINE_OUT is None:
        try:
            sys.stdout.flush()
            keep = os.dup(1)
            os.dup2(2, 1)
            LINE_OUT = os.fdopen(keep, "w")
        except OSError:          # no stderr (or no stdout) to work with: leave the descriptors as they are
            LINE_OUT = None


def print_line(text):
    out = LINE_OUT or sys.stdout
    out.write(text + "\n")
    out.flush()


class Deadline:
    """Guarantees that the process ENDS, and that rank 0 prints exactly one JSON line, whatever blocks.

    Before the headline exists (complete == False) an expiry is a FAILURE: rank 0 prints a diagnostic line -- the stage that
    did not finish, the rank, the seconds, and `out` as far as it was recorded (the kernel-only record of an N > 1 run) with
    value null -- and the process ends with fail_code (non-zero) through os._exit: a collective that never returns cannot
    be caught as an exception, and atexit handlers of a hung runtime must not get a second chance to hang.
    After the headline (complete == True) an expiry only cuts supplementary records short: the line is printed with what is
    there plus "truncated" and the process ends with `exit_code` (0 unless a verification failed).
    Two timers: the process-level one (total_seconds, armed in __init__) and the current stage's (stage())."""

    def __init__(self, rank=0, n_gpus=1, total_seconds=None, out=None, complete=False, fail_code=3):
        self.rank, self.n_gpus, self.out, self.complete, self.fail_code = rank, n_gpus, out, complete, fail_code
        self.exit_code = 0
        self.t0 = time.time()
        self.stage_name, self.stage_timer, self.stage_limit = "start", None, None
        self.lock, self.printed = threading.Lock(), False
        self.error_key = "error"            # "gather_error" once the kernel-only record exists
        self.total = None
        if total_seconds:
            self.total = threading.Timer(total_seconds, self._expired, args=("the process-level deadline (%.0f s)" % total_seconds,))
            self.total.daemon = True
            self.total.start()

    # -- stages ------------------------------------------------------------------------------------------------
    def stage(self, name, seconds=None):
        """Names what the process is about to do; `seconds` (optional) is that stage's own limit."""
        with self.lock:
            if self.stage_timer is not None:
                self.stage_timer.cancel()
                self.stage_timer = None
            self.stage_name, self.stage_limit = name, seconds
            if seconds:
                self.stage_timer = threading.Timer(seconds, self._expired, args=("stage '%s' (%.0f s)" % (name, seconds),))
                self.stage_timer.daemon = True
                self.stage_timer.start()
        if os.environ.get("DSABF_BENCH_TRACE") == "1":
            print("bench.py[rank %d] +%.1f s: %s" % (self.rank, time.time() - self.t0, name), file=sys.stderr, flush=True)

    def set_rank(self, rank, n_gpus):
        self.rank, self.n_gpus = rank, n_gpus

    def adopt(self, out, error_key="gather_error"):
        """rank 0: the record so far (printed with value null if a later stage never finishes)."""
        self.out, self.error_key = out, error_key

    def headline_complete(self, budget_seconds=None):
        """From here on an expiry truncates instead of failing; the process-level timer is re-armed with the budget."""
        with self.lock:
            self.complete = True
            if self.total is not None:
                self.total.cancel()
                self.total = None
            if budget_seconds:
                self.total = threading.Timer(budget_seconds, self._expired, args=("the budget for supplementary records "
                                                                                  "(--budget-seconds)",))
                self.total.daemon = True
                self.total.start()

    # -- the one line -------------------------------------------------------------------------------------------
    def _line(self, out):
        try:
            return json.dumps(out)
        except Exception:   # a record half-written by the main thread
            return json.dumps({k: v for k, v in out.items() if k in (
                "metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "truncated", "error", "gather_error", "stage", "rank")})

    def _emit(self, note=None, failure=None):
        with self.lock:
            if self.printed:
                return
            self.printed = True
            if self.rank != 0:
                if failure:
                    print("bench.py[rank %d]: %s" % (self.rank, failure), file=sys.stderr, flush=True)
                return
            out = self.out
            if failure:
                if out is None:
                    out = {"metric": METRIC_NAME, "value": None, "unit": "beam-blocks/s", "n_gpus": self.n_gpus}
                out = dict(out)
                out["value"] = None
                out[self.error_key] = failure
                out["stage"], out["rank"], out["elapsed_s"] = self.stage_name, self.rank, round(time.time() - self.t0, 1)
            elif out is not None and note:
                out["truncated"] = note
            if out is not None:
                print_line(self._line(out))

    def _expired(self, what):
        if self.complete:
            self._emit("%s expired in stage '%s'; the headline and roofline are complete" % (what, self.stage_name))
            os._exit(self.exit_code)
        self._emit(failure="%s expired: '%s' did not finish (+%.0f s since start)" % (what, self.stage_name, time.time() - self.t0))
        os._exit(self.fail_code)

    def fail(self, message, code=None):
        """A failure the main thread noticed itself (an exception in a gather stage, a verification that failed before the
        headline): the same line, the same non-zero end."""
        self._emit(failure=message)
        sys.stdout.flush()
        os._exit(code or self.fail_code)

    def terminated(self, signum):
        """SIGTERM from the launcher (another rank failed): say what was there, end non-zero unless everything was done."""
        if self.complete:
            self._emit("terminated by signal %d in stage '%s'" % (signum, self.stage_name))
            os._exit(self.exit_code if self.exit_code else 143)
        self._emit(failure="terminated by signal %d (the launcher stops the other ranks when one fails) in stage '%s'"
                           % (signum, self.stage_name))
        os._exit(143)

    def finish(self):
        with self.lock:
            for t in (self.total, self.stage_timer):
                if t is not None:
                    t.cancel()
        self._emit()

    def cancel(self):
        with self.lock:
            for t in (self.total, self.stage_timer):
                if t is not None:
                    t.cancel()


class Watchdog(Deadline):
    """The post-headline form by itself: Watchdog(out, seconds) prints `out` exactly once -- through finish(), or, if the
    budget expires first, from the timer thread with "truncated" -- and then ends the process with status 0."""

    def __init__(self, out, seconds):
        super().__init__(rank=0, total_seconds=seconds, out=out, complete=True)


def watch_sigterm(deadline):
    """SIGTERM is blocked in every thread and picked up by a thread of its own (sigwait): a Python-level handler would never
    run while the main thread sits inside a C call that does not return -- which is exactly when the launcher sends it."""
    try:
        signal.pthread_sigmask(signal.SIG_BLOCK, {signal.SIGTERM})
    except (AttributeError, ValueError, OSError):   # pragma: no cover
        return

    def wait():
        signum = signal.sigwait({signal.SIGTERM})
        deadline.terminated(signum)

    t = threading.Thread(target=wait, daemon=True)
    t.start()


def main():
    args = parse()
    # the pool's host driver only supports dmabuf IPC; without this RCCL's cross-process handles fail
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)          # never returns
    claim_stdout()                 # (every rank: what gloo, RCCL or anybody else prints goes to stderr from here on)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # ---- the deadline is armed HERE: before torch is imported, before the process group, before any RCCL call.  Ranks other
    # than 0 give rank 0 a few seconds to print its line first (the launcher stops everybody when the first rank exits).
    lag = 0.0 if rank == 0 else 5.0
    try:
        signal.pthread_sigmask(signal.SIG_BLOCK, {signal.SIGTERM})   # before any thread exists: see watch_sigterm
    except (AttributeError, ValueError, OSError):   # pragma: no cover
        pass
    deadline = Deadline(rank, max(world, args.gpus), args.deadline_seconds + lag)
    watch_sigterm(deadline)
    stage_s, gather_s = args.stage_seconds + lag, args.gather_timeout + lag
    deadline.stage("import torch", stage_s)
    import torch

    if args.gpus != world:
        if world == 1 and args.gpus > 1:   # WORLD_SIZE=1 in the environment next to --gpus N: a launcher started ONE rank
            sys.exit("bench.py --gpus %d under a launcher with WORLD_SIZE=1: start it with --nproc-per-node %d, or call "
                     "bench.py plainly and it launches its ranks itself" % (args.gpus, args.gpus))
        args.gpus = world
    if os.environ.get("DSABF_BENCH_ONE_GPU") == "1":
        local = 0          # test mode: every rank time-shares GPU 0
    deadline.stage("torch.cuda.set_device(%d)" % local, stage_s)
    torch.cuda.set_device(local)
    backend = args.dist_backend
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist

        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29541")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        deadline.stage("control plane: init_process_group(%s)" % backend, stage_s)
        if backend == "gloo":
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")    # the container's hostname may not resolve; loopback always does
            try:
                dist.init_process_group("gloo")
            except Exception as e:   # pragma: no cover  (a gloo that cannot bind: fall back to torch's RCCL communicator)
                print("bench.py[rank %d]: gloo control plane failed (%s); using nccl" % (rank, str(e)[:200]), file=sys.stderr, flush=True)
                backend = "nccl"
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    dist_dev = "cuda" if backend == "nccl" else "cpu"

    deadline.stage("import dsabeamformer_amd (libdsabf.so)", stage_s)
    import dsabeamformer_amd as bfm
    from dsabeamformer_amd import api

    n_avg, n_out = geometry(args.workload)
    n_freq_total = 1024 if args.workload == "c5" else 256
    assert n_freq_total % world == 0
    n_freq = n_freq_total // world
    cfg = bfm.production_config(n_avg=n_avg, n_out_per_gemm=n_out, n_freq=n_freq)
    if args.workload == "c5":
        cfg.n_ant, cfg.n_beams = 100, 512
    cfg.detect_mode = DETECT[args.detect]
    n_ipo, n_time = cfg.n_pol * cfg.n_avg, n_out * cfg.n_pol * cfg.n_avg
    deadline.stage("bf_create + bf_set_weights", stage_s)
    bf = bfm.Beamformer(cfg, device=local)
    bf.set_weights(product_weights(cfg, rank * n_freq))

    units = args.units
    blocks_per_step = units * n_out                      # beam-blocks per step (whole job)
    in_bytes = units * n_freq * n_time * cfg.n_ant       # this rank's packed bytes per step
    out_floats = units * n_out * n_freq * cfg.n_beams    # this rank's detected floats per step
    deadline.stage("synthetic input in HBM", stage_s)
    gen = torch.Generator(device="cuda").manual_seed(0xD5A + rank)
    d_in = [torch.randint(0, 256, (in_bytes,), dtype=torch.uint8, device="cuda", generator=gen)
            for _ in range(max(1, args.nbuf))]
    if args.input != "random":
        for t in d_in:
            t.fill_(0 if args.input == "zeros" else 0x31)
    d_out = [torch.empty(out_floats, dtype=torch.float32, device="cuda") for _ in range(2)]
    stream = torch.cuda.current_stream()
    sptr = stream.cuda_stream

    # ---- the gather (N > 1 or --force-dist): bf_comm / bf_gather_detected behind the C-ABI -------------------------------
    n_rows, row_floats = units * n_out, n_freq * cfg.n_beams
    # There is ONE gather path.  If the communicator cannot be created the run ends non-zero: a scaling line must never
    # come from a second code path.  The communicator is created AFTER the kernel-only region (see below).
    comm_box = {"comm": None}
    rccl_info = None
    side = torch.cuda.Stream() if dist is not None else None

    class GatherMode:
        """One way of bringing the shards together, double-buffered: the collective of step i runs on a side stream and
        overlaps the kernel of step i+1; the compute stream waits for it only before it overwrites that slot again."""

        def __init__(self, mode, layout, transport="inplace"):
            self.mode, self.layout, self.transport = mode, layout, transport
            self.root = {"root": 0, "alltoall": api.GATHER_ROOT_DISTRIBUTED}.get(mode)
            self.lay = api.GATHER_RANK_MAJOR if layout == "rank" else api.GATHER_FREQ_MAJOR
            self.kernel_done = [torch.cuda.Event() for _ in range(2)]
            self.gather_ev = [torch.cuda.Event() for _ in range(2)]     # created once: the step loop creates nothing
            self.gather_done = [None, None]
            self.full = [None, None]
            self.stage = [None, None]
            self.held = 0
            if mode == "none":
                return
            comm = comm_box["comm"]
            self.held = comm.rows_held(n_rows, self.root)
            if self.held:
                self.full = [torch.empty(self.held * world * row_floats, dtype=torch.float32, device="cuda") for _ in range(2)]
                if transport == "staged":
                    self.stage = [torch.empty(self.held * world * row_floats, dtype=torch.float32, device="cuda") for _ in range(2)]

        @property
        def key(self):
            if self.mode == "none":
                return "none"
            return "%s_%s_major" % (self.mode, self.layout) + ("_staged" if self.transport == "staged" else "")

        def before_kernel(self, slot):
            if self.gather_done[slot] is not None:
                stream.wait_event(self.gather_done[slot])
                self.gather_done[slot] = None

        def after_kernel(self, slot):
            if self.mode == "none":
                return
            self.kernel_done[slot].record(stream)
            side.wait_event(self.kernel_done[slot])
            comm = comm_box["comm"]
            if self.transport == "staged":
                comm.gather_staged(d_out[slot], n_rows, row_floats, self.root, self.full[slot], self.stage[slot], side.cuda_stream)
            else:
                comm.gather(d_out[slot], n_rows, row_floats, self.root, self.lay, self.full[slot], side.cuda_stream)
            self.gather_ev[slot].record(side)
            self.gather_done[slot] = self.gather_ev[slot]

        def drain(self):
            for slot in (0, 1):
                self.before_kernel(slot)
            torch.cuda.synchronize()

    def step(i, gm, ev_pair=None):
        slot = i & 1
        gm.before_kernel(slot)  # the output buffer about to be overwritten must have left
        if ev_pair:
            ev_pair[0].record(stream)
        bf.beamform(d_in[i % len(d_in)], units, d_out[slot], sptr)
        if ev_pair:
            ev_pair[1].record(stream)
        gm.after_kernel(slot)

    closing = {}      # the last timed region's cost of the closing barrier (reported, not part of the step time)

    def timed_region(gm, n_steps, with_events):
        """K steps between (barrier + synchronize) and (synchronize + barrier).  Every rank reads its clock when ITS K steps are
        complete on the device (drain = synchronize: with a gather that includes what it had to receive); the job's time is the MAX
        over ranks.  The closing barrier -- the MAX all-reduce itself, on the control plane -- is bookkeeping: its own latency
        (a loopback TCP round of gloo, of the order of a whole step at N = 8) is measured beside it, not charged to the steps."""
        events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_steps)] if with_events else None
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n_steps):
            step(i, gm, events[i] if events else None)
        closing["enqueue_ms"] = (time.perf_counter() - t0) * 1e3 / n_steps     # host time to SUBMIT a step (kernel + gather calls)
        gm.drain()                                   # torch.cuda.synchronize(): this rank's K steps are complete
        elapsed = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dist_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)  # the closing barrier: nobody leaves before everybody has finished
            torch.cuda.synchronize()
            closing["ms"] = (time.perf_counter() - t0 - elapsed) * 1e3
            elapsed = float(t.item())
        return elapsed, events

    def warm_up(gm, n_steps, min_seconds):
        """W steps, then more of the same -- the same number on every rank -- until the chip has been busy for min_seconds."""
        t_w = time.perf_counter()
        for i in range(n_steps):
            step(i, gm)
        extra = 0
        while True:
            gm.drain()
            go = torch.tensor([1.0 if time.perf_counter() - t_w < min_seconds else 0.0], device=dist_dev)
            if dist is not None:
                dist.all_reduce(go, op=dist.ReduceOp.MAX)   # every rank takes the same number of extra steps
            if go.item() == 0.0:
                break
            for i in range(32):
                step(i, gm)
            extra += 32
        return extra

    CS_MOD = 4093

    def row_checksums(flat, rows, floats):
        """Position-weighted 64-bit checksum of every row of a float32 array, on the device, exact integer arithmetic (the bit
        patterns as int32 times weights 1 .. 4093: |sum| < 2^62 for rows of up to 2^19 floats)."""
        x = flat.view(torch.int32).view(rows, floats)
        wgt = (torch.arange(floats, device=flat.device, dtype=torch.int64) % CS_MOD) + 1
        cs = torch.empty(rows, dtype=torch.int64, device=flat.device)
        chunk = max(1, (32 << 20) // floats)
        for r0 in range(0, rows, chunk):
            cs[r0:r0 + chunk] = (x[r0:r0 + chunk].to(torch.int64) * wgt).sum(dim=1)
        return cs

    def verify(gm):
        """One more step of this mode, then: every rank publishes the checksums of its local rows (all-gather on the control
        plane), every receiver recomputes them on what the gather delivered.  True on every rank iff every receiver agrees."""
        step(0, gm)
        gm.drain()
        mine = row_checksums(d_out[0], n_rows, row_floats).to(dist_dev)
        everyone = [torch.empty_like(mine) for _ in range(world)]
        if world > 1:
            dist.all_gather(everyone, mine)
        else:
            everyone = [mine]
        sent = torch.stack(everyone).cpu()                       # [rank][row]
        ok = 1
        if gm.held:
            first = comm_box["comm"].rank * gm.held if gm.root == api.GATHER_ROOT_DISTRIBUTED else 0
            got = row_checksums(gm.full[0], gm.held * world, row_floats).cpu()
            if gm.lay == api.GATHER_RANK_MAJOR:
                got = got.view(world, gm.held)                   # [rank][row held]
            else:
                got = got.view(gm.held, world).t()               # [row held][rank] -> [rank][row held]
            ok = int(torch.equal(got, sent[:, first:first + gm.held]))
        flag = torch.tensor([ok], dtype=torch.int32, device=dist_dev)
        if world > 1:
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(flag.item())

    info = bf.kernel_info(units)
    paired = "PAIRED" in info["kernel"]
    # the key a committed counter pass must carry to be quoted beside this run's timings: the device code the LIBRARY says it holds
    # (bf_version() ends in "kernels <id>"), the instantiation this handle launches, the launch
    kernels_id = bfm.load().bf_version().decode().rsplit("kernels ", 1)[-1].rstrip(")")
    this_pmc_key = pmc_key(kernels_id, bf.variant_key(), args.workload, units, paired, args.detect)
    if rank == 0:
        print("bench.py: pmc_key %s" % this_pmc_key, file=sys.stderr, flush=True)
    peak_cache = {}        # the micro-benchmarks behind roofline.peak_measured run once per process

    def build_record(elapsed, events, extra_warm):
        """rank 0: the JSON record of one timed region (elapsed = max over ranks; events = this rank's HIP-event pairs)."""
        kern_ms = sorted(a.elapsed_time(b) for a, b in events)
        kern_avg_ms = sum(kern_ms) / len(kern_ms)
        if rank != 0:
            return None
        total_blocks = args.steps * blocks_per_step
        value = total_blocks / elapsed
        # algorithmic work per beam-block: ops = 8*B*A*n_ipo*F ; bytes = A*n_ipo*F + 4*B*F
        ops_per_block = 8 * cfg.n_beams * cfg.n_ant * n_ipo * n_freq_total
        bytes_per_block = cfg.n_ant * n_ipo * n_freq_total + 4 * cfg.n_beams * n_freq_total
        launch_ops = ops_per_block * blocks_per_step / world    # per launch (this rank's kernel)
        launch_bytes = bytes_per_block * blocks_per_step / world
        mfma_bound = args.workload != "c2"
        # the committed rocprofv3 PMC summary of exactly this kernel build and launch (tools/pmc.sh), if there is one: resolved by the
        # key stored IN the summary, never by a file name
        pmc, pmc_name, pmc_stale = pmc_for_launch(this_pmc_key) if world == 1 else ({}, None, False)
        if mfma_bound:
            achieved = launch_ops / (kern_avg_ms * 1e-3) / 1e12
            roof = {"bound": "mfma", "achieved": achieved, "peak": INT8_DENSE_PEAK_TOPS, "unit": "TOP/s",
                    "frac": achieved / INT8_DENSE_PEAK_TOPS, "traffic": pmc_traffic(pmc)}
            # "report utilisation against both nominal and measured peak" -- the matrix pipe by itself, live on
            # this box: back-to-back v_mfma_i32_16x16x64_i8 on the workload's own bytes (bf_mfma_peak_device), after the timed
            # region, same HIP-event timing.  The chip does not hold 2.4 GHz under this load (DVFS), so this is what "100 %"
            # means here for a kernel that did nothing but MFMAs.
            try:
                if "mfma" in peak_cache:
                    raise StopIteration
                d_sink = torch.empty(4 << 20, dtype=torch.uint8, device="cuda")
                peak_ops = [0.0]

                def peak_fn(i):
                    peak_ops[0] = bf.mfma_peak(d_in[i % len(d_in)], in_bytes, d_sink, 4 << 20, 2000, sptr)
                for i in range(20):
                    peak_fn(i)
                torch.cuda.synchronize()
                p_avg, _, p_min = time_launches(torch, peak_fn, 50, stream)
                peak_cache["mfma"] = (peak_ops[0] / (p_avg * 1e-3) / 1e12,
                                      "back-to-back v_mfma_i32_16x16x64_i8, 2 accumulator chains per wave issued chain by chain, 4 waves "
                                      "per SIMD, operands = this workload's bytes (A = 16 * nibble, B = random "
                                      "int8), %.3f ms per launch of %.3g ops, measured after the first timed region"
                                      % (p_avg, peak_ops[0]))
            except StopIteration:
                pass
            except Exception as e:   # the headline never depends on the micro-benchmark
                peak_cache["mfma"] = (None, "micro-benchmark failed: %s" % e)
            peak_measured, peak_note = peak_cache["mfma"]
            roof.update({"peak_measured": peak_measured, "peak_measured_note": peak_note})
            if peak_measured:
                roof["frac_of_measured_peak"] = achieved / peak_measured
        else:
            achieved = launch_bytes / (kern_avg_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(pmc)}
            # the measured counterpart of the nominal 8 TB/s: a pure streaming kernel on this box, after the timed region --
            # bf_expand_device (a1 alone: 1 byte in, 2 out per packed byte, whole 128-byte lines, nontemporal both ways)
            try:
                if "hbm" in peak_cache:
                    raise StopIteration
                n_exp = 256 << 20      # 256 MiB in, 512 MiB out: well past the 256 MiB Infinity Cache
                d_exp_in = torch.randint(0, 256, (n_exp,), dtype=torch.uint8, device="cuda")
                d_exp = torch.empty(2 * n_exp, dtype=torch.uint8, device="cuda")

                def exp_fn(i):
                    bf.expand(d_exp_in, n_exp, d_exp, sptr)
                for i in range(10):
                    exp_fn(i)
                torch.cuda.synchronize()
                e_avg, _, _ = time_launches(torch, exp_fn, 30, stream)
                peak_cache["hbm"] = (3 * n_exp / (e_avg * 1e-3) / 1e9,
                                     "bf_expand_device streaming %d MiB in / %d MiB out: %.3f ms per launch, "
                                     "measured after the first timed region" % (n_exp >> 20, n_exp >> 19, e_avg))
                del d_exp, d_exp_in
            except StopIteration:
                pass
            except Exception as e:
                peak_cache["hbm"] = (None, "streaming micro-benchmark failed: %s" % e)
            peak_measured, peak_note = peak_cache["hbm"]
            roof.update({"peak_measured": peak_measured, "peak_measured_note": peak_note})
            if peak_measured:
                roof["frac_of_measured_peak"] = achieved / peak_measured
        executed = launch_ops / 2 if paired else launch_ops
        roof.update({"kernel": info["kernel"],
                     "kernel_ms_avg": kern_avg_ms, "kernel_ms_median": kern_ms[len(kern_ms) // 2],
                     "kernel_ms_min": kern_ms[0], "algorithmic_ops_per_launch": launch_ops,
                     "algorithmic_bytes_per_launch": launch_bytes,
                     "executed_mfma_ops_per_launch": executed,
                     "executed_frac": (executed / (kern_avg_ms * 1e-3) / 1e12 / INT8_DENSE_PEAK_TOPS) if mfma_bound else None,
                     "mfma_busy_frac": pmc_mfma_busy(pmc),
                     "pmc_source": ("profiles/" + pmc_name) if pmc else None,
                     # true: no committed counter pass carries this run's key (kernel build id + instantiation + launch) -- the
                     # counters of an OLDER build (pmc_not_used) are not paired with these timings; traffic etc. stay null
                     "pmc_stale": bool(pmc_stale and world == 1), "pmc_key": this_pmc_key,
                     "pmc_not_used": ("profiles/" + pmc_name) if (pmc_stale and pmc_name) else None,
                     # traffic and mfma_busy_frac are read from the committed rocprofv3 summary named above -- measured
                     # on an earlier box with the same kernel, NOT observed in this run (everything else in this object is)
                     "from_committed_profile": (["traffic", "mfma_busy_frac"] + (["valu_per_mfma", "issue_occupancy", "bound_measured",
                                                                                   "clock_ghz_under_load"] if mfma_bound else [])) if pmc else [],
                     "note": "int8 ops: 1 complex MAC = 8 ops.  frac = ALGORITHMIC ops / kernel time / peak; executed_frac = "
                             "the int8 ops the MFMA pipe really executes / time / peak (the conjugate-pair kernel forms two "
                             "beams from shared products: half the algorithmic ops, same bits); mfma_busy_frac = "
                             "SQ_VALU_MFMA_BUSY_CYCLES / SIMD cycles from the committed rocprofv3 PMC pass of this launch "
                             "(profiled passes clock lower); traffic = HBM bytes per launch from the same passes (FETCH_SIZE "
                             "doubled per the guide), null if this launch has no committed pass"})
        roof.update(issue_model(pmc, kern_avg_ms) if mfma_bound else {})
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
                       "freq_per_gpu": n_freq,
                       "gather": ("%s, %s-major layout, bf_gather_detected (RCCL p2p behind the C-ABI)"
                                  % (args.gather, args.layout)) if dist is not None else "n/a",
                       "control_plane": ("torch.distributed %s over 127.0.0.1: barrier, max-over-ranks, unique-id broadcast, "
                                         "checksum exchange -- no detected power travels on it" % backend) if dist is not None else "n/a",
                       "detect_mode": args.detect,
                       "detect_reading": {"canonical": "x*x + y*y as two multiplies and an add: what g++ makes of "
                                                       "src/beamformer.cuh:151 and what the CPU oracle computes (bit-exact "
                                                       "parity); the reference's production build is nvcc -O3 "
                                                       "-use_fast_math (makefile:13-16), which most likely contracts it: "
                                                       "see contracted_detect_mode for that reading's rate",
                                          "contracted": "acc + fma(x, x, y*y): nvcc's default -fmad=true reading of "
                                                        "src/beamformer.cuh:151 (makefile:13-16), most likely what the "
                                                        "reference's GPU build computes; bit-exact vs the oracle's "
                                                        "ORC_CONTRACT_NVCC",
                                          "fast": "opt-in tolerance mode (include/dsabf.h)"}[args.detect],
                       "weights": "the reference's linear fan (conjugate-symmetric: the pair kernel); a calibrated "
                                  "instrument's weights run the general kernel: see calibrated_weights",
                       "extra_warmup_steps": extra_warm,
                       "launch": info},
            "roofline": roof,
        }
        if rccl_info is not None:
            out["rccl"] = rccl_info
        if "enqueue_ms" in closing:     # the host's time to submit one step of the region this record was built from
            out["host_submit_ms_per_step"] = closing["enqueue_ms"]
        return out


    # ---- kernel-only first: no RCCL communicator exists yet (at N = 1 this IS the headline) ------------------------------
    none_mode = GatherMode("none", "rank")
    deadline.stage("kernel-only warm-up", stage_s)
    extra_warm = warm_up(none_mode, args.warmup, args.min_warm_seconds)
    deadline.stage("kernel-only timed region", stage_s)
    elapsed, events = timed_region(none_mode, args.steps, True)
    deadline.stage("record of the kernel-only region", stage_s)
    out = build_record(elapsed, events, extra_warm)
    deadline.adopt(out, "error")          # rank 0: the record the one line is printed from (None on the other ranks)
    headline_mode = none_mode
    gather_modes = {}

    def mode_record(el, n_steps, ev=None, gm=None):
        rec = {"value": n_steps * blocks_per_step / el, "unit": "beam-blocks/s", "ms_per_step": el / n_steps * 1e3, "steps": n_steps}
        if gm is not None and gm.mode != "none" and world > 1:
            # what the fabric carries: a sender's bytes per step (root: everything it computed; alltoall: all but its own rows)
            sent = out_floats * 4 * (1.0 if gm.mode == "root" else (world - 1) / world)
            one_per_pair = gm.layout == "rank" or gm.transport == "staged"      # else: one message per (row, sender), in place
            if gm.mode == "root":
                msgs = 1 if one_per_pair else n_rows                            # a sender -> the root
            else:
                msgs = (world - 1) * (1 if one_per_pair else n_rows // world)   # a sender -> every other owner
            rec.update({"bytes_sent_per_rank_per_step": sent, "send_gbs_per_rank": sent / (el / n_steps) / 1e9,
                        "messages_sent_per_rank_per_step": msgs})
            # a model to read the figure with (docs/DESIGN_r03_sections.md 5): the busiest link of the step -- alltoall: every pair of
            # GPUs exchanges 1 / world^2 of the step's powers in each direction; root: the root's links take a whole shard each, IN --
            # at 50 and 75 GB/s per direction and link (what RCCL point-to-point holds of an xGMI link's 76.8).  The step cannot be
            # faster than max(kernel, this); send_gbs_per_rank / (world - 1) is the per-link rate it really ran at
            per_link = out_floats * 4 * (1.0 if gm.mode == "root" else 1.0 / world)
            rec["link_model"] = {"bytes_on_the_busiest_link_per_step": per_link, "ms_at_50_gbs": per_link / 50e9 * 1e3,
                                 "ms_at_75_gbs": per_link / 75e9 * 1e3,
                                 "measured_gbs_per_link": (sent / (world - 1) if gm.mode != "root" else per_link) / (el / n_steps) / 1e9}
        if "ms" in closing:
            rec["closing_barrier_ms"] = closing["ms"]      # rank 0's wait in the MAX all-reduce behind its own K steps
        if "enqueue_ms" in closing:
            # rank 0's host time to submit one step: if it approaches ms_per_step the region was bound by the launching thread
            # (python + ctypes + the RCCL group call), not by the GPUs or the fabric
            rec["host_submit_ms_per_step"] = closing["enqueue_ms"]
        if ev:
            ms = [a.elapsed_time(b) for a, b in ev]
            rec["kernel_ms_avg"] = sum(ms) / len(ms)
        return rec

    if dist is not None and args.gather != "none":
        # ---- N > 1: the record so far is the kernel-only strong-scaling figure; the headline needs the gather ------------------
        gather_modes["none"] = mode_record(elapsed, args.steps, events)
        gather_modes["none"]["note"] = ("kernel-only strong scaling: measured first, before any RCCL communicator existed; "
                                        "no collective at all")
        if rank == 0:
            out["gather_modes"] = gather_modes
            out["kernel_only"] = {k: out[k] for k in ("value", "ms_per_step", "ms_per_block")}
            out["kernel_only"]["roofline_frac"] = out["roofline"]["frac"]
            deadline.adopt(out)         # printed with value null + gather_error if anything below never comes back

        def gather_stage(name, fn):
            """A stage that runs RCCL: its own time limit; an exception ends the run like an expiry does (line + non-zero)."""
            deadline.stage(name, gather_s)
            try:
                return fn()
            except Exception as e:
                deadline.fail("%s failed: %s: %s" % (name, type(e).__name__, str(e)[:400]), code=4)

        if rank == 0:       # which library is about to be asked for a communicator: known to the line even if it never answers
            try:
                out["rccl"] = dict(api.comm_library_info(), ranks=None, note="bound, no communicator yet")
            except Exception as e:
                out["rccl"] = {"error": str(e)[:300]}

        def make_comm():
            idt = torch.zeros(128, dtype=torch.uint8, device=dist_dev)
            if rank == 0:
                idt.copy_(torch.frombuffer(bytearray(api.comm_unique_id()), dtype=torch.uint8))
            dist.broadcast(idt, 0)
            c = api.Comm(rank, world, bytes(idt.cpu().numpy().tobytes()), device=local)
            inf = c.info()      # ranks as the LIBRARY counts them, its version, the file it was loaded from
            if inf["ranks"] not in (world, -1):
                raise RuntimeError("bf_comm: the library reports %d ranks, the launcher %d" % (inf["ranks"], world))
            return c, inf

        deadline.fail_code = 4
        comm_box["comm"], rccl_info = gather_stage("RCCL communicator (bf_comm_unique_id, bf_comm_create)", make_comm)
        headline_mode = GatherMode(args.gather, args.layout)
        tag = "gather '%s'" % headline_mode.key
        gather_stage(tag + ": warm-up", lambda: warm_up(headline_mode, args.warmup, 0.0))
        elapsed, events = gather_stage(tag + ": timed region", lambda: timed_region(headline_mode, args.steps, True))
        verified = None if args.no_verify else gather_stage(tag + ": verification", lambda: verify(headline_mode))
        deadline.stage("record of the headline", stage_s)
        out = build_record(elapsed, events, extra_warm)
        gather_modes[headline_mode.key] = mode_record(elapsed, args.steps, events, headline_mode)
        gather_modes[headline_mode.key].update({"verified": verified, "headline": True})
        if rank == 0:
            out["gather_modes"] = gather_modes
            out["kernel_only"] = deadline.out["kernel_only"]
            deadline.adopt(out)
        if verified is False:
            # the line still says everything that was measured; the number it would have carried is not a result
            if rank == 0:
                out["unverified_value"] = out["value"]
            deadline.fail("the headline gather '%s' delivered other bits than the ranks sent (per-row checksums differ)"
                          % headline_mode.key, code=6)

    # ---- from here on nothing may cost the headline: the deadline now only truncates (prints the line with what is there) ----
    deadline.headline_complete(args.budget_seconds + lag)
    watchdog = deadline
    deadline.stage("supplementary records")

    # ---- every gather mode side by side (N > 1) ----------------------------------------------------------------------------
    if dist is not None and comm_box["comm"] is not None and not args.no_extras:
        n_side = max(5, min(args.steps, 50))
        for mode, layout, transport in (("root", "rank", "inplace"), ("root", "freq", "inplace"), ("root", "freq", "staged"),
                                        ("alltoall", "rank", "inplace"), ("alltoall", "freq", "inplace"),
                                        ("alltoall", "freq", "staged")):
            gm = GatherMode(mode, layout, transport)
            if gm.key in gather_modes:
                continue
            deadline.stage("gather '%s' (supplementary)" % gm.key, gather_s)
            try:   # a supplementary record must never cost the headline line
                for i in range(4):
                    step(i, gm)
                gm.drain()
                el, _ = timed_region(gm, n_side, False)
                gather_modes[gm.key] = mode_record(el, n_side, None, gm)
                if not args.no_verify:
                    gather_modes[gm.key]["verified"] = verify(gm)
                    if gather_modes[gm.key]["verified"] is False:
                        deadline.exit_code = 6
            except Exception as e:  # pragma: no cover
                gather_modes[gm.key] = {"error": str(e)[:300]}
                torch.cuda.synchronize()
            del gm
        deadline.stage("supplementary records")

    if rank == 0:
        if "gather_modes" in out:
            out["gather_modes"]["note"] = ("the same kernel and inputs, only the collective differs; rank-major = one message "
                                           "per sender, freq-major = the reference's [o][f][b], one message per (row, sender), "
                                           "received in place; freq-major staged = the same result by bf_gather_detected_staged "
                                           "(one message per sender + one device re-layout pass); root = everything to rank 0, "
                                           "alltoall = rank j owns rows j*n/N.. of the whole band; verified = the receivers' "
                                           "per-row checksums of what arrived equal the senders' (one extra step after the timing)")

        def variant(detect_mode, paired_env=None, wl=None, n_units=None, reps=None, calibrated=False, n_ant=None, n_freq_rank=None):
            """Kernel-time record of another variant / workload on this GPU (HIP events around every launch)."""
            old = os.environ.get("DSABF_PAIRED")
            if paired_env is not None:
                os.environ["DSABF_PAIRED"] = paired_env
            try:
                na, no = geometry(wl) if wl else (n_avg, n_out)
                c2 = bfm.production_config(n_avg=na, n_out_per_gemm=no, n_freq=n_freq, detect_mode=detect_mode)
                if wl == "c5":
                    c2.n_ant, c2.n_beams, c2.n_freq = 100, 512, 128
                if n_ant:
                    c2.n_ant = n_ant
                if n_freq_rank:
                    c2.n_freq = n_freq_rank
                b2 = bfm.Beamformer(c2, device=local)
                w2 = product_weights(c2, 0)
                b2.set_weights(calibrated_weights(w2) if calibrated else w2)
            finally:
                if paired_env is not None:
                    if old is None:
                        del os.environ["DSABF_PAIRED"]
                    else:
                        os.environ["DSABF_PAIRED"] = old
            nu = n_units or units
            nt2 = no * c2.n_pol * c2.n_avg
            in2 = nu * c2.n_freq * nt2 * c2.n_ant
            of2 = nu * no * c2.n_freq * c2.n_beams
            src = d_in if in2 <= in_bytes else [torch.randint(0, 256, (in2,), dtype=torch.uint8, device="cuda", generator=gen)
                                                for _ in range(2)]
            dst = d_out if of2 <= out_floats else [torch.empty(of2, dtype=torch.float32, device="cuda") for _ in range(2)]
            fn = lambda i: b2.beamform(src[i % len(src)][:in2], nu, dst[i & 1][:of2], sptr)  # noqa: E731
            t_w2 = time.perf_counter()     # the handle was just built (idle GPU): warm the clock back up before timing
            i = 0
            while i < 8 or time.perf_counter() - t_w2 < 0.15:
                fn(i)
                i += 1
                if i % 16 == 0:
                    torch.cuda.synchronize()
            avg, med, mn = time_launches(torch, fn, reps or max(40, args.steps // 2), stream)
            name = b2.kernel_info(nu)["kernel"]
            li = b2.kernel_info(nu)
            b2.close()
            n_ipo2 = c2.n_pol * c2.n_avg
            ops = 8 * c2.n_beams * c2.n_ant * n_ipo2 * c2.n_freq * nu * no
            byts = (c2.n_ant * n_ipo2 * c2.n_freq + 4 * c2.n_beams * c2.n_freq) * nu * no
            return {"kernel": name, "kernel_ms_avg": avg, "kernel_ms_median": med, "kernel_ms_min": mn,
                    "grid": li["grid"], "vgprs": li["vgprs"], "gemm_units_per_launch": nu,
                    "value": nu * no / (avg * 1e-3), "unit": "beam-blocks/s (of this record's geometry)",
                    "tops": ops / (avg * 1e-3) / 1e12, "frac": ops / (avg * 1e-3) / 1e12 / INT8_DENSE_PEAK_TOPS,
                    "gbs": byts / (avg * 1e-3) / 1e9}

        def guarded(name, fn):
            """A supplementary record must never cost the headline line: an exception becomes {"error": ...}."""
            try:
                fn()
            except Exception as e:  # pragma: no cover
                out[name] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
                try:
                    torch.cuda.synchronize()
                except Exception:
                    pass

        def extras_variants():
            # supplementary, never the headline, same inputs:
            if paired:
                g = variant(0, "0")
                gkey = pmc_key(kernels_id, bf.variant_key().replace("true, 4, 4>", "false, 4, 4>"), args.workload, units, False, "canonical")
                gp, gp_name, gp_stale = pmc_for_launch(gkey)
                g.update({"mfma_busy_frac": pmc_mfma_busy(gp), "note": "DSABF_PAIRED=0: the kernel any weight set without the "
                          "conjugate symmetry runs; every algorithmic int8 op executes on the MFMA pipe; same bits"})
                out["general_kernel"] = g
            c = variant(2)
            c["note"] = ("bf_config.detect_mode = BF_DETECT_CONTRACTED: acc + fma(x,x,y*y), nvcc's default reading of "
                         "src/beamformer.cuh:151; 5 instead of 6 VALU ops per sample; bit-identical to the oracle's "
                         "ORC_CONTRACT_NVCC; within (n_ipo+4)*2^-24 of exact like the canonical mode")
            out["contracted_detect_mode"] = c
            if paired:
                out["contracted_detect_mode_general_kernel"] = variant(2, "0")
            # what a real array uploads: the fan times per-(frequency, antenna) complex gains -> no conjugate symmetry ->
            # bf_set_weights selects the general kernel by itself (no environment switch): the PRODUCTION number
            for key, dm in (("calibrated_weights", 0), ("calibrated_weights_contracted", 2)):
                cw = variant(dm, calibrated=True)
                assert "PAIRED" not in cw["kernel"]
                cw["note"] = ("steering fan x seeded random unit-modulus per-(frequency, antenna) gains, re-quantised to int8 "
                              "(bench.calibrated_weights): the weight set of a calibrated instrument; bf_set_weights finds no "
                              "conjugate symmetry and runs the general kernel; bit-exact vs the oracle in "
                              "tests/test_gpu_round3.py")
                out[key] = cw
            f = variant(1)
            f["tolerance"] = "(n_ipo+1)*2^-23 relative to the exact value (include/dsabf.h); canonical: (n_ipo+4)*2^-24"
            f["note"] = "opt-in bf_config.detect_mode = BF_DETECT_FAST; not the headline"
            out["fast_detect_mode"] = f

        def extras_geometries():
            # BASELINE configs[1], the reference's DEBUG geometry (N_TIME 16, n_ipo 2): the parity configuration;
            # HBM-write-bound (4 B out per 0.125 B in per beam), so its roofline is the HBM one
            d = variant(0, wl="c2", reps=100)
            out["debug_geometry"] = {"workload": "C2: BASELINE configs[1], N_TIME=16 (8 outputs x n_ipo 2), %d gemm-units per launch" % units,
                                     "value": d["value"], "unit": "beam-blocks/s", "kernel_ms_avg": d["kernel_ms_avg"],
                                     "kernel": d["kernel"],
                                     "roofline": {"bound": "hbm", "achieved": d["gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                  "frac": d["gbs"] / HBM_PEAK_GBS},
                                     "note": "not the headline; bit-exact parity on this geometry is what tests/ check"}
            # one rank's share of BASELINE configs[3] (C4): 32 of the 256 channels, the headline step's 128 gemm-units -- what ONE
            # GPU of an 8-GPU run launches, alone on this chip.  8 x its rate over the whole band's = the kernel-only strong
            # scaling the partition itself allows at N = 8 (launch shape and tail effects; no fabric, no host): the figure the
            # driver's N = 8 run is to be read against while no 8-GPU node has run the bench
            if world == 1 and args.workload == "c3":
                s4 = variant(0, n_freq_rank=32, reps=60)
                s4["workload"] = ("C4 shard: one of 8 ranks of BASELINE configs[3] = 32 freq x 256 beams x 64 ant x 2 pol, N_TIME 512, "
                                  "%d gemm-units per launch; beam-blocks here are 256 beams x 32 freq" % units)
                s4["kernel_only_scaling_at_8_ranks_by_launch_shape"] = out["roofline"]["kernel_ms_avg"] / (8 * s4["kernel_ms_avg"])
                out["c4_shard"] = s4
            # one rank's share of BASELINE configs[4]: 128 of 1024 freq x 512 beams x 100 ant, n_ipo 32, 16 gemm-units
            s5 = variant(0, wl="c5", n_units=16, reps=30)
            s5["workload"] = ("C5 shard: one of 8 ranks of BASELINE configs[4] = 128 freq x 512 beams x 100 ant x 2 pol, "
                              "N_TIME 256, 16 gemm-units per launch; beam-blocks here are 512 beams x 128 freq")
            s5["pmc_source"] = "the committed C5 counter pass is the whole-band launch (16 gemm-units x 1024 freq), not this shard"
            out["c5_shard"] = s5
            # the same C3 shape with 256 antennas (four k-steps: the deep classes of fused16_kernel) -- the
            # detect is amortised over 4 x the MACs.  The linear fan is conjugate-symmetric (pair kernel); the calibrated set is not.
            a256 = {}
            for key, cal in (("fan_pair_kernel", False), ("calibrated_general_kernel", True)):
                r = variant(0, n_units=32, reps=40, calibrated=cal, n_ant=256)
                a256[key] = {k: r[k] for k in ("kernel", "kernel_ms_avg", "grid", "vgprs", "tops", "frac")}
            a256["workload"] = "256 ant x 2 pol, 256 freq, 256 beams, N_TIME 512, 32 gemm-units per launch (1 GiB of voltages)"
            out["antennas_256"] = a256
            # launch granularity: what one launch over 1 / 8 / 32 gemm-units costs per beam-block (input resident)
            ls = {}
            for nu in (1, 8, 32, 128):
                r = variant(0, n_units=nu, reps=60)
                ls["%d_units" % nu] = {"kernel_us_per_beam_block": r["kernel_ms_avg"] * 1e3 / (nu * n_out), "grid": r["grid"],
                                       "kernel_ms_avg": r["kernel_ms_avg"], "frac": r["frac"]}
            # what the reference's loop really issues: N_STREAMS = 8 one-unit launches in flight at once, one per queue
            qs = [torch.cuda.Stream() for _ in range(8)]
            per_in, per_out = n_freq * n_time * cfg.n_ant, n_out * n_freq * cfg.n_beams
            done = [torch.cuda.Event() for _ in qs]

            N_EACH = 16                                   # one-unit launches per queue between the fork and the join

            def eight(i):
                fork = torch.cuda.Event()
                fork.record(stream)
                for q, sq in enumerate(qs):
                    sq.wait_event(fork)
                for k in range(N_EACH):
                    for q, sq in enumerate(qs):
                        u = (q * N_EACH + k) % units
                        bf.beamform(d_in[i % len(d_in)][u * per_in:(u + 1) * per_in], 1,
                                    d_out[i & 1][u * per_out:(u + 1) * per_out], sq.cuda_stream)
                for q, sq in enumerate(qs):
                    done[q].record(sq)
                    stream.wait_event(done[q])

            for i in range(5):
                eight(i)
            torch.cuda.synchronize()
            avg8, _, _ = time_launches(torch, eight, 30, stream)
            ops1 = 8 * cfg.n_beams * cfg.n_ant * n_ipo * n_freq * n_out
            n_l = 8 * N_EACH
            ls["8x1_units_concurrent"] = {"kernel_us_per_beam_block": avg8 * 1e3 / (n_l * n_out), "ms_for_%d_units" % n_l: avg8,
                                          "frac": n_l * ops1 / (avg8 * 1e-3) / 1e12 / INT8_DENSE_PEAK_TOPS,
                                          "note": "%d one-unit launches, %d on each of 8 queues, between one fork and one "
                                                  "join event: the aggregate rate of the reference's launch pattern "
                                                  "(N_STREAMS = 8 queues of one-unit launches, src/beamformer.cu:454-519); "
                                                  "1_units above is ONE such launch alone on the chip" % (n_l, N_EACH)}
            ls["note"] = ("bf_enqueue_gemm_unit launches 1 unit (the reference's pattern, src/beamformer.cu:454-519), "
                          "bf_enqueue_block 32 (one PSRDADA block), the headline step 128")
            out["launch_size"] = ls

        def extras_streaming():
            # the production loop end to end, PCIe included: junk source -> H2D -> kernel -> D2H, as the reference's
            # "Time per data chunk" (src/beamformer.cu:539-546)
            from dsabeamformer_amd import host

            pc = bfm.production_config()
            n_blk = 64
            st = {}
            host.run_observation_junk(pc, 8, ring_blocks=4, device=local, burn_in=2)   # one-off costs outside the records
            # (DSABF_UNIT_LAUNCH is a measurement switch: the library reads it only beside DSABF_LAB=1, set for these two records alone)
            for label, env in (("block_launches", {}), ("reference_unit_launches", {"DSABF_LAB": "1", "DSABF_UNIT_LAUNCH": "1"}),
                               ("reference_unit_launches_literal", {"DSABF_LAB": "1", "DSABF_UNIT_LAUNCH": "1", "DSABF_COALESCE": "0"})):
                os.environ.update(env)
                try:
                    r = host.run_observation_junk(pc, n_blk, ring_blocks=4, device=local, burn_in=4)
                finally:
                    for k in env:
                        os.environ.pop(k, None)
                chunks = n_blk * pc.n_gemms_per_block * pc.n_out_per_gemm
                in_b = n_blk * pc.n_gemms_per_block * pc.n_ant * pc.n_freq * pc.n_out_per_gemm * pc.n_pol * pc.n_avg
                out_b = chunks * pc.n_freq * pc.n_beams * 4
                st[label] = {"ms_per_beam_block": r["ms"] / chunks, "input_gbs": in_b / (r["ms"] * 1e-3) / 1e9,
                             "output_gbs": out_b / (r["ms"] * 1e-3) / 1e9, "observation_ms": r["ms"], "blocks": n_blk}
            # ... and with the DM stage in the loop: 64 trials of the notebook's ladder to DM 250 on every
            # analysed block, the delay window carried over on the device, chunks [dm][t][b] copied to pinned host buffers
            freq = [host.channel_frequency(0, c) for c in range(pc.n_freq)]
            ladder = host.dm_trials(dm_max=250.0)
            dms = ladder[:: max(1, len(ladder) // 64)][:64]
            delays = host.dm_delays(dms, freq, freq[0], 0.131)
            host.run_observation_junk_dm(pc, 8, delays, None, ring_blocks=4, device=local, burn_in=2)
            r = host.run_observation_junk_dm(pc, n_blk, delays, None, ring_blocks=4, device=local, burn_in=4)
            chunks = n_blk * pc.n_gemms_per_block * pc.n_out_per_gemm
            st["block_launches_dm_stage"] = {"ms_per_beam_block": r["ms"] / chunks, "observation_ms": r["ms"], "blocks": n_blk,
                                             "dm_trials": len(dms), "max_delay_rows": int(delays.max()), "dm_output_times": r["dm_times"],
                                             "dm_output_gbs": r["dm_times"] * len(dms) * pc.n_beams * 4 / (r["ms"] * 1e-3) / 1e9}
            st["note"] = ("run_observation, production geometry (N_AVERAGING 16, 128 MiB blocks), in-memory junk source, "
                          "pinned host buffers: H2D of every block and D2H of every gemm-unit's detected powers included -- the "
                          "reference's 'Time per data chunk'; never the headline.  block_launches: the caller uses bf_enqueue_block; "
                          "reference_unit_launches: the caller keeps the reference's loop (one bf_enqueue_gemm_unit per gemm-unit "
                          "round-robin over 8 queues, src/beamformer.cu:454-519) and the library coalesces it into one launch per "
                          "block; ..._literal: the same loop with DSABF_COALESCE=0, one launch per call; block_launches_dm_stage: block_launches + the "
                          "DM-trial stage inside the loop (bf_dm_stream_push behind every block launch, 64 trials, window carried over on "
                          "the device; bit-exact vs the oracle over the whole series in tests/test_gpu_round5.py).  Real-time budget: 0.131 ms per beam-block.  "
                          "PCIe-bound either way; interleaved sweep over the launch granularities: profiles/r02_streaming.txt (round 2)")
            out["streaming"] = st

        def extras_dm():
            # DM-trial dedispersion of a detected series -- an HBM-roofline kernel of its own (algorithmic
            # bytes = the series once + the output once); 64 trials of the reference notebook's ladder to DM 250 over 1024
            # beam-blocks of 256 x 256, the shape of tools/bench_stages.py / profiles/r0N_stage_kernels.json
            from dsabeamformer_amd import host

            n_t, n_dm = 1024, 64
            freq = [host.channel_frequency(0, c) for c in range(256)]
            ladder = host.dm_trials(dm_max=250.0)
            dms = ladder[:: max(1, len(ladder) // n_dm)][:n_dm]
            delays = host.dm_delays(dms, freq, freq[0], 0.131)
            n_t_out = n_t - int(delays.max())
            b2 = bfm.Beamformer(bfm.production_config(), device=local)
            d_series = torch.rand(n_t * 256 * 256, device="cuda")
            d_delays = torch.from_numpy(delays).cuda()
            d_dd = torch.empty(len(dms) * n_t_out * 256, device="cuda")
            rec = {}
            for label, wide in (("shared_window", 1), ("per_thread_window_alone", 0)):
                b2.set_switch("dm_wide", wide)     # per handle (bf_set_switch); the environment is never changed mid-process
                fn = lambda i: b2.dedisperse_dm(d_series, n_t, d_delays, len(dms), n_t_out, d_dd, sptr)  # noqa: E731
                for i in range(5):
                    fn(i)
                avg, med, mn = time_launches(torch, fn, 30, stream)
                rec[label] = {"ms_avg": avg, "ms_median": med}
            # the same kernel on a launch that FILLS the chip: 64 trials x 901 times are 2 x 57 x 2 = 228 tiles of the shared-window
            # kernel for 256 CUs (one tile per CU: a single, incomplete round); 256 trials of the same ladder are 912
            b2.set_switch("dm_wide", 1)
            dms4 = ladder[:: max(1, len(ladder) // 256)][:256]
            delays4 = host.dm_delays(dms4, freq, freq[0], 0.131)
            n_t_out4 = n_t - int(delays4.max())
            d_delays4 = torch.from_numpy(delays4).cuda()
            d_dd4 = torch.empty(len(dms4) * n_t_out4 * 256, device="cuda")
            fn4 = lambda i: b2.dedisperse_dm(d_series, n_t, d_delays4, len(dms4), n_t_out4, d_dd4, sptr)  # noqa: E731
            for i in range(3):
                fn4(i)
            avg4, med4, _ = time_launches(torch, fn4, 15, stream)
            adds4 = float(len(dms4)) * n_t_out4 * 256 * 256
            rec["shared_window_256_trials"] = {"ms_avg": avg4, "ms_median": med4, "dm_trials": len(dms4), "tiles": ((len(dms4) + 31) // 32) * ((n_t_out4 + 15) // 16) * 2,
                                               "gadds_per_s": adds4 / (avg4 * 1e-3) / 1e9, "frac": adds4 / (avg4 * 1e-3) / 1e9 / (256 * 64 * 2.4e9 / 1e9),
                                               "note": "the same kernel and ladder, four times as dense: a launch of several rounds of tiles per CU instead of 228 tiles for 256 CUs"}
            del d_dd4
            b2.close()
            alg = 4 * (n_t * 256 * 256 + len(dms) * n_t_out * 256)
            ms = rec["shared_window"]["ms_avg"]
            # The bound: every output is a chain of n_freq fp32 adds, each with one 4-byte
            # operand that cannot come from a register (the trial's row offset is run-time data) -- so the floors are the VALU add
            # rate, 64 lanes per clock per CU (v_pk_add_f32 issues at half rate: the same 64), and the LDS operand rate, 256 B
            # per clock per CU with ds_read_b128 = the same 64 operands.  HBM is NOT the bound (the series is read about once: the
            # algorithmic bytes are reported as `traffic`-style figures beside it).
            adds = float(len(dms)) * n_t_out * 256 * 256
            peak_adds = 256 * 64 * 2.4e9 / 1e9           # G adds/s at the nominal clock the MFMA peak assumes
            rec.update({"workload": "%d DM trials (notebook ladder to DM 250) x %d output samples x 256 freq x 256 beams, series of %d "
                                    "beam-blocks" % (len(dms), n_t_out, n_t),
                        "roofline": {"bound": "lds/valu", "achieved": adds / (ms * 1e-3) / 1e9, "peak": peak_adds, "unit": "Gadd/s",
                                     "frac": adds / (ms * 1e-3) / 1e9 / peak_adds, "traffic": None,
                                     "adds_per_launch": adds,
                                     "algorithmic_bytes_per_launch": alg,
                                     "hbm_gbs_algorithmic": alg / (ms * 1e-3) / 1e9, "hbm_frac_algorithmic": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                     "kernel": "dsabf::dedisperse_dm_wide_kernel (+ dedisperse_dm_kernel for "
                                               "trial groups whose delays do not fit a window)",
                                     "note": "peak = 256 CUs x 64 fp32 adds (= 64 LDS operands of 4 bytes) per clock x 2.4 GHz; one "
                                             "add and one LDS operand per (trial, time, beam, channel).  What binds the kernel in practice is "
                                             "instruction issue (92 instructions per wave and channel in lock step between barriers: "
                                             "DESIGN.md 3.4), not this floor"},
                        "note": "not the headline; ascending-f fp32 sum per (trial, time, beam), bit-exact vs the oracle in both "
                                "kernels; the delay law and the ladder are pinned by executing the reference's notebook "
                                "(tests/golden/make_dispersion_golden.py)"})
            out["dedisperse_dm"] = rec

        def extras_dm_stage():
            # the stage as the loop runs it (bf_dm_stream, zero-copy feed): per production block, beside a rank's beamforming time
            rec = {"whole_band_one_gpu": dm_stage_record(torch, bfm, local, 64, (0, 1)),
                   "whole_band_one_gpu_two_queues": dm_stage_record(torch, bfm, local, 64, (0, 1), pushes=48, queues=2),
                   "trial_share_1_of_8_two_queues": dm_stage_record(torch, bfm, local, 64, (0, 8), pushes=48, queues=2),
                   "whole_band_one_gpu_4_blocks_per_push": dm_stage_record(torch, bfm, local, 64, (0, 1), pushes=12, blocks_per_push=4),
                   "trial_share_1_of_8": dm_stage_record(torch, bfm, local, 64, (0, 8)),
                   "note": "kernel-only, rows already in the stage's buffer (bf_dm_stream_reserve: the beamformer / the gather writes them "
                           "there, bf_dm_stream_push only launches -- round 5 copied every row device-to-device first); 64 trials of the "
                           "notebook's ladder to DM 250 on a production block (256 beam-blocks); trial_share_1_of_8 = one rank's 8 trials "
                           "over the whole band when a sharded run splits the ladder (`beam -X`).  A rank's beamforming at N = 8: c4_shard.  "
                           "A production block gives the shared-window kernel 64 workgroups (2 trial groups x 16 time tiles x 2 beam tiles) "
                           "for 256 CUs: ..._4_blocks_per_push shows the same stage fed every four blocks (real time: 33.5 ms per block)"}
            out["dm_stage"] = rec

        def extras_relayout():
            # the device pass of the staged gather transport by itself, at the shape one receiver of BASELINE configs[3] sees in
            # the distributed-owner gather: 8 ranks, 2048 / 8 = 256 rows held, 32 channels x 256 beams per row and sender
            w8, held, rf = 8, (units * n_out) // 8, (256 // 8) * cfg.n_beams
            NB = 4         # distinct buffer pairs cycled (4 x 117 MB: the 256 MiB Infinity Cache cannot hold the stream)
            d_sts = [torch.rand(w8 * held * rf, device="cuda") for _ in range(NB)]
            d_fus = [torch.empty(w8 * held * rf, device="cuda") for _ in range(NB)]
            d_st, d_fu = d_sts[0], d_fus[0]
            REP = 8        # launches between one pair of HIP events: a 20-us kernel would otherwise carry the events' own 2-3 us

            def fn(i):
                for k in range(REP):
                    bf.gather_relayout(d_sts[k % NB], d_fus[k % NB], held, w8, rf, 0, sptr)
            for i in range(5):
                fn(i)
            torch.cuda.synchronize()
            want = d_st.view(w8, held, rf).transpose(0, 1).reshape(-1)
            ok = bool(torch.equal(d_fu.view(held, w8, rf)[:, 1:], want.view(held, w8, rf)[:, 1:]))
            avg, med, mn = (v / REP for v in time_launches(torch, fn, 30, stream))
            moved = 2.0 * (w8 - 1) * held * rf * 4
            out["gather_relayout"] = {"kernel": "dsabf::gather_relayout_kernel", "ms_avg": avg, "ms_median": med, "bit_equal_to_a_transpose": ok,
                                      "shape": "stage [8 ranks][%d rows][%d floats] -> full [row][rank][floats], own rank skipped" % (held, rf),
                                      "roofline": {"bound": "hbm", "achieved": moved / (avg * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                   "frac": moved / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": moved},
                                      "note": "bf_gather_detected_staged = one message per sender + this pass; every float read once and "
                                              "written once, whole 128-byte lines, nontemporal; per step and receiver at N = 8; %d launches "
                                              "between each pair of HIP events" % REP}

        if world == 1 and args.detect == "canonical" and args.workload in ("c3", "prod") and not args.no_extras:
            guarded("general_kernel", extras_variants)
            guarded("gather_relayout", extras_relayout)
        def north_star_summary():
            """north_star: '>= 50 % int8 MFMA utilisation' -- which kernel / detect reading meets it in THIS run, one line each:
            algorithmic fraction of the nominal 5.0 POP/s from the records above (executed fraction = half for the pair kernel)."""
            rows = {"pair_kernel_canonical (headline)": out["roofline"]["frac"] if paired else None}
            for key, label in (("contracted_detect_mode", "pair_kernel_contracted" if paired else "general_kernel_contracted"),
                               ("fast_detect_mode", "pair_kernel_fast" if paired else "general_kernel_fast"),
                               ("general_kernel", "general_kernel_canonical"),
                               ("contracted_detect_mode_general_kernel", "general_kernel_contracted"),
                               ("calibrated_weights", "general_kernel_canonical_calibrated_weights"),
                               ("calibrated_weights_contracted", "general_kernel_contracted_calibrated_weights")):
                if isinstance(out.get(key), dict) and "frac" in out[key]:
                    rows[label] = out[key]["frac"]
            if not paired:
                rows["general_kernel_canonical (headline)"] = out["roofline"]["frac"]
            rows = {k: v for k, v in rows.items() if v is not None}
            out["north_star_check"] = {"target": 0.50, "frac_by_reading": rows, "meets": sorted(k for k, v in rows.items() if v >= 0.50),
                                       "misses": sorted(k for k, v in rows.items() if v < 0.50),
                                       "note": "algorithmic int8 ops / kernel time / 5.0 POP/s, this run, this box (boxes differ by ~2.5 %); "
                                               "canonical = the bit-exact reading of the CPU oracle (g++), contracted = nvcc's -fmad reading"}

        if world == 1 and args.detect == "canonical" and args.workload in ("c3", "prod") and not args.no_extras:
            guarded("north_star_check", north_star_summary)
        if world == 1 and args.workload == "c3" and not args.no_extras:
            guarded("debug_geometry", extras_geometries)
            guarded("streaming", extras_streaming)
            guarded("dedisperse_dm", extras_dm)
            guarded("dm_stage", extras_dm_stage)
        if not args.no_cpu_baseline:
            # N = 1: the full set.  N > 1: the same key, the beamform port alone on a shorter sample (the other ranks have
            # finished; the whole-band beam-block is the unit at every N, so the number is comparable across the lines)
            secs = args.cpu_seconds if world == 1 else min(args.cpu_seconds, 5.0)
            print("bench.py: GPU part done; timing the CPU baseline%s on the host cores (~%.0f s) ..."
                  % ("s" if world == 1 else "", secs + (15 if world == 1 else 2)), file=sys.stderr, flush=True)
            guarded("cpu_baseline", lambda: out.__setitem__("cpu_baseline", cpu_baselines(n_avg, n_out, secs, full=world == 1)))
        deadline.stage("printing the line")
        watchdog.finish()
    else:
        watchdog.cancel()
    # teardown can block too (a communicator whose peers are gone): under its own limit, after the line is out
    deadline = Deadline(rank, world, 60.0 + lag, out=None, complete=True)
    deadline.exit_code = watchdog.exit_code
    bf.close()
    if comm_box["comm"] is not None:
        comm_box["comm"].close()
    if dist is not None:
        dist.destroy_process_group()
    deadline.cancel()
    if watchdog.exit_code:
        sys.exit(watchdog.exit_code)


if __name__ == "__main__":
    main()
