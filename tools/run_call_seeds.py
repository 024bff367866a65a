import os, sys, traceback

os.environ.setdefault("DSABF_LAB", "1")   # a measurement tool: the library reads its A/B switches from the environment only in lab mode
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
import dsabeamformer_amd as bfm
import oracle as orc
import test_gpu_round4 as t
for s in [int(x) for x in sys.argv[1:]]:
    for rep in range(4):
        try:
            t.test_random_call_sequences_leave_every_host_buffer_with_its_own_units_bits(torch, bfm, orc, s)
            print("seed", s, "rep", rep, "ok", flush=True)
        except AssertionError as e:
            tb = traceback.extract_tb(sys.exc_info()[2])[-1]
            print("seed", s, "rep", rep, "FAILED line", tb.lineno, tb.line, "|", str(e)[:300], flush=True)
