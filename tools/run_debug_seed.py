#!/usr/bin/env python3
"""Re-runs one seed of the random DEBUG flow (tests/test_gpu_round4.py) many times and reports how often, and where, its table
differs from the oracle's.  GPU box, repo root:  python tools/run_debug_seed.py 4036 300"""
import os

os.environ.setdefault("DSABF_LAB", "1")   # a measurement tool: the library reads its A/B switches from the environment only in lab mode
import pathlib
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import dsabeamformer_amd as bfm  # noqa: E402
import oracle as orc  # noqa: E402
import test_gpu_round4 as t  # noqa: E402

seed, reps = int(sys.argv[1]), int(sys.argv[2])
bad = 0
orig = np.array_equal


def spy(a, b):
    ok = orig(a, b)
    if not ok and getattr(a, "ndim", 0) == 2 and a.shape == getattr(b, "shape", None):
        rows = np.nonzero((a != b).any(axis=1))[0]
        print("   rows that differ:", rows.tolist(), "of", a.shape[0], "| all-zero rows:", [int(r) for r in rows if not a[r].any()], flush=True)
    return ok


np.array_equal = spy
for rep in range(reps):
    with tempfile.TemporaryDirectory() as d:
        try:
            t.test_debug_flow_with_random_catalogues_geometries_and_launch_patterns(bfm, orc, pathlib.Path(d), seed)
        except AssertionError as e:
            bad += 1
            print("rep", rep, "FAILED", str(e)[:120], flush=True)
print("seed %d: %d of %d runs differ" % (seed, bad, reps))
