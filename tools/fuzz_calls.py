#!/usr/bin/env python3
"""The randomised walk over the streaming entry points (tests/test_gpu_round4.py::test_random_call_sequences_...) for many
seeds.  GPU box, repo root:  SEED=100 CASES=300 python tools/fuzz_calls.py
FUZZ=loops: test_production_loop_to_file_under_random_launch_patterns (the whole observation loop to a file sink) instead;
FUZZ=debug: test_debug_flow_with_random_catalogues_geometries_and_launch_patterns."""
import os

os.environ.setdefault("DSABF_LAB", "1")   # a measurement tool: the library reads its A/B switches from the environment only in lab mode
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch  # noqa: E402

import dsabeamformer_amd as bfm  # noqa: E402
import oracle as orc  # noqa: E402
import test_gpu_round4 as t  # noqa: E402

seed0, cases = int(os.environ.get("SEED", "100")), int(os.environ.get("CASES", "200"))
bad = 0
if os.environ.get("FUZZ") == "loops":      # run_observation to a file sink under random block / queue counts and launch patterns
    import pathlib
    import tempfile

    class Env:                              # (pytest's monkeypatch, as much of it as the test uses)
        def __init__(self):
            self.set = []

        def setenv(self, k, v):
            os.environ[k] = v
            self.set.append(k)

        def undo(self):
            for k in self.set:
                os.environ.pop(k, None)

    for s in range(seed0, seed0 + cases):
        env = Env()
        with tempfile.TemporaryDirectory() as d:
            try:
                t.test_production_loop_to_file_under_random_launch_patterns(bfm, orc, pathlib.Path(d), env, s)
            except AssertionError as e:
                bad += 1
                print("seed", s, "FAILED:", str(e)[:200], flush=True)
            finally:
                env.undo()
    print("observation loops, seeds %d..%d: %d runs, %d parity failures" % (seed0, seed0 + cases - 1, cases, bad))
    sys.exit(0)
if os.environ.get("FUZZ") == "dmloops":    # run_observation with the DM stage + detected sink under random launch patterns (round 5)
    import pathlib
    import tempfile

    import test_gpu_round5 as t5

    class Env5:
        def __init__(self):
            self.set = []

        def setenv(self, k, v):
            os.environ[k] = v
            self.set.append(k)

        def undo(self):
            for k in self.set:
                os.environ.pop(k, None)

    for s in range(seed0, seed0 + cases):
        env = Env5()
        with tempfile.TemporaryDirectory() as d:
            try:
                t5.test_production_loop_with_the_dm_stage_under_random_launch_patterns(bfm, orc, pathlib.Path(d), env, s)
            except AssertionError as e:
                bad += 1
                print("seed", s, "FAILED:", str(e)[:200], flush=True)
            finally:
                env.undo()
    print("observation loops with the DM stage, seeds %d..%d: %d runs, %d parity failures" % (seed0, seed0 + cases - 1, cases, bad))
    sys.exit(0)
if os.environ.get("FUZZ") == "debug":      # the DEBUG flow end to end on random catalogues / geometries / launch patterns
    import pathlib
    import tempfile

    for s in range(seed0, seed0 + cases):
        with tempfile.TemporaryDirectory() as d:
            try:
                t.test_debug_flow_with_random_catalogues_geometries_and_launch_patterns(bfm, orc, pathlib.Path(d), s)
            except AssertionError as e:
                bad += 1
                print("seed", s, "FAILED:", str(e)[:200], flush=True)
    print("DEBUG flows, seeds %d..%d: %d runs, %d parity failures" % (seed0, seed0 + cases - 1, cases, bad))
    sys.exit(0)
for s in range(seed0, seed0 + cases):
    try:
        t.test_random_call_sequences_leave_every_host_buffer_with_its_own_units_bits(torch, bfm, orc, s)
    except AssertionError as e:
        import traceback

        where = traceback.extract_tb(sys.exc_info()[2])[-1].line or ""
        if "len(expect_out) > 10" in where:
            continue                 # (a walk with too few host buffers: not a parity failure)
        bad += 1
        print("seed", s, "FAILED:", str(e)[:200], flush=True)
print("seeds %d..%d: %d walks, %d parity failures" % (seed0, seed0 + cases - 1, cases, bad))
