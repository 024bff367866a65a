#!/usr/bin/env python3
"""Long randomized parity run (GPU box): CASES random geometries over the WHOLE contract of the reference (any n_ant % 4 up to
128, any n_beams % 4, n_ipo 2..64), random launch shapes, conjugate-pair or general weights, canonical or contracted
detect, through bf_beamform_device, each compared bit for bit with the CPU oracle in the same reading.  Not part of the
test suite (the suite runs seeded subsets).  Round 1: 3 seeds x 150 cases on the then whitelist, 0 mismatches.
Round 4: FUZZ_GENERIC=1 draws antenna counts up to 636 and accumulation windows 1 ... 138 (fusedg_kernel).
FUZZ_DEEP=1: 144 ... 256 antennas in 16-byte rows, windows 16 / 32 / 64 (the deep classes).
usage: SEED=1 CASES=150 [FUZZ_WIDE=1 | FUZZ_GENERIC=1 | FUZZ_DEEP=1] python tools/fuzz_long.py"""
import os

os.environ.setdefault("DSABF_LAB", "1")   # a measurement tool: the library reads its A/B switches from the environment only in lab mode
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import dsabeamformer_amd as bfm  # noqa: E402
import oracle as orc  # noqa: E402

rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
bad = 0
N = int(os.environ.get("CASES", "150"))
classes = {}
for case in range(N):
    n_ant = int(rng.choice([64, 100, 128])) if rng.integers(3) == 0 else 4 * int(rng.integers(1, 33))
    n_avg = int(rng.choice([1, 2, 4, 8, 16, 32]))
    n_ipo = 2 * n_avg
    n_out = int(rng.integers(1, 7)) * max(1, 16 // n_ipo)
    n_beams = 32 * int(rng.integers(1, 13)) if rng.integers(2) else 4 * int(rng.integers(1, 100))
    n_pol = 2
    if os.environ.get("FUZZ_GENERIC") == "1":   # round 4: the geometries fusedg_kernel takes -- any antenna count, any window
        n_ant = 4 * int(rng.integers(33, 160)) if rng.integers(3) else 4 * int(rng.integers(1, 33))
        n_pol = int(rng.choice([1, 2, 2, 2]))
        n_avg = int(rng.integers(1, 70)) if rng.integers(2) else int(rng.choice([3, 5, 6, 7, 12, 20, 24, 48]))
        n_ipo = n_pol * n_avg
        n_out = int(rng.integers(1, 9))
        n_beams = 32 * int(rng.integers(1, 13)) if rng.integers(2) else 4 * int(rng.integers(1, 100))
    if os.environ.get("FUZZ_DEEP") == "1":      # round 4: the three / four k-step classes of fused16_kernel (129 ... 256 antennas)
        n_ant = 4 * int(rng.integers(33, 65))        # (any multiple of 4: rows that are only dword-aligned since round 5)
        n_pol, n_avg = 2, int(rng.choice([8, 16, 32]))
        n_ipo = 2 * n_avg
        n_out = int(rng.integers(1, 6))
        n_beams = 512 * int(rng.integers(1, 3)) if rng.integers(3) == 0 else 32 * int(rng.integers(1, 20)) if rng.integers(2) else 4 * int(rng.integers(1, 150))
    if os.environ.get("FUZZ_WIDE") == "1":   # bias towards the 8-wave workgroups of the two-k-step classes (fused_wg_waves)
        n_ant = int(rng.choice([100, 128])) if rng.integers(3) == 0 else 4 * int(rng.integers(17, 33))
        n_avg = int(rng.choice([8, 16, 32]))
        n_ipo = 2 * n_avg
        n_out = int(rng.integers(1, 7))
        n_beams = 32 * int(rng.integers(9, 36)) if rng.integers(2) else 4 * int(rng.integers(65, 280))
        if rng.integers(3) == 0:
            n_beams = 512 * int(rng.integers(1, 3))
    g = orc.Geom(n_beams=n_beams, n_ant=n_ant, n_freq=int(rng.integers(1, 18 if n_beams <= 400 and n_ant <= 128 else 6)), n_pol=n_pol, n_avg=n_avg,
                 n_out_per_gemm=n_out)
    n_units = int(rng.integers(1, 1 + max(1, 900 // g.n_time)))
    w = rng.integers(-127, 128, size=(g.n_freq, g.n_ant, g.n_beams, 2), dtype=np.int8)
    paired = bool(rng.integers(2)) and n_beams % 32 == 0
    if paired:
        B = g.n_beams
        w[:, :, B // 2:, 0] = w[:, :, :B // 2, 0][:, :, ::-1]
        w[:, :, B // 2:, 1] = -w[:, :, :B // 2, 1][:, :, ::-1]
    mode = int(rng.choice([0, 2]))
    packed = rng.integers(0, 256, size=(n_units, g.n_freq, g.n_time, g.n_ant), dtype=np.uint8)
    os.environ["DSABF_TSPLIT"] = str(int(rng.integers(1, 5)))
    bf = bfm.Beamformer(bfm.debug_config(n_beams=g.n_beams, n_ant=g.n_ant, n_freq=g.n_freq, n_pol=g.n_pol, n_avg=g.n_avg,
                                          n_out_per_gemm=g.n_out_per_gemm, detect_mode=mode))
    bf.set_weights(w)
    kout = 0
    if os.environ.get("FUZZ_GENERIC") == "1":   # round 5: any stream length of the run-time-window launches (0: the library's choice)
        kout = int(rng.integers(0, 33))
        bf.set_switch("rtw_kout", kout)
    info = bf.kernel_info(n_units)
    name = info["kernel"]
    generic = "fusedg_kernel" in name
    assert generic or ("PAIRED" in name) == paired, (name, paired)
    key = (bf.variant_key(), n_ipo, paired, mode,      # the compiled instantiation the case ran (round 6: tools/census.py names all of them)
           "generic" if generic else "slots8" if "SLOTS=8" in name else "waves8" if "WAVES=8" in name else "plain")
    classes[key] = classes.get(key, 0) + 1
    d_in = torch.from_numpy(packed).cuda()
    with orc.detect_contract(orc.CONTRACT_NVCC if mode == 2 else orc.CONTRACT_NONE):
        want = orc.beamform(g, w, packed)
    d_out = torch.full((want.size,), float("nan"), dtype=torch.float32, device="cuda")
    bf.beamform(d_in, n_units, d_out, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    ok = np.array_equal(d_out.cpu().numpy().reshape(want.shape), want)
    if not ok:
        bad += 1
        print("MISMATCH", case, g, n_units, paired, mode, os.environ["DSABF_TSPLIT"], "rtw_kout", kout)
    bf.close()
print("seed", os.environ.get("SEED", "1"), "cases", N, "mismatches", bad, "distinct (instantiation, n_ipo, paired, mode, launch) combinations", len(classes),
      "distinct instantiations", len({k[0] for k in classes}),
      "cases on 8-wave workgroups", sum(v for k, v in classes.items() if k[4] == "waves8"),
      "on 8 slots per wave", sum(v for k, v in classes.items() if k[4] == "slots8"),
      "on fusedg_kernel", sum(v for k, v in classes.items() if k[4] == "generic"))
