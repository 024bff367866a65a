"""Worker of tests/test_gpu_multirank.py::test_sharded_observation_loop_under_random_shapes: ONE frequency shard of a sharded
observation (dsabf::run_observation with a communicator) as its own process (TEST INFRASTRUCTURE).
usage: shard_loop_worker.py rank workdir       (DSABF_RCCL_LIB must point at tests/support/libfakerccl.so)

The parent wrote <workdir>/problem.json (geometry, world, transport, receiver, DM stage) and delays.npy; rank 0 draws the
communicator id.  The shard runs bfh_run_observation_junk_sharded and leaves its source ring and counters for the parent."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

rank, work = int(sys.argv[1]), sys.argv[2]
import dsabeamformer_amd as bfm  # noqa: E402
from dsabeamformer_amd import api, host  # noqa: E402

p = json.load(open(os.path.join(work, "problem.json")))
world = p["world"]
idfile = os.path.join(work, "id")
if rank == 0:
    uid = api.comm_unique_id()
    open(idfile + ".tmp", "wb").write(uid)
    os.rename(idfile + ".tmp", idfile)
else:
    t0 = time.time()
    while not os.path.exists(idfile):
        assert time.time() - t0 < 120
        time.sleep(0.05)
    uid = open(idfile, "rb").read()
for k, v in p.get("env", {}).items():
    os.environ[k] = str(v)
cfg = bfm.production_config(n_beams=p["n_beams"], n_freq=p["n_freq_local"], n_avg=p["n_avg"], n_out_per_gemm=p["n_out"])
cfg.n_gemms_per_block, cfg.n_streams = p["n_units"], p["n_streams"]
delays = np.load(os.path.join(work, "delays.npy")) if p["n_dm"] else None
holds = p["gather_root"] < 0 or p["gather_root"] == rank
det_path = os.path.join(work, "det.%d" % rank)
if p.get("bad_rank", -1) == rank:       # this shard's own preparations fail: its sink cannot be opened
    det_path = os.path.join(work, "no_such_directory", "det.%d" % rank)
r = host.run_observation_junk_sharded(
    cfg, p["n_blocks"], rank, world, uid, gather_root=p["gather_root"], staged=p["staged"], delays=delays,
    split_trials=p["split"], detected_path=det_path if holds else None,
    dm_path=os.path.join(work, "dm.%d" % rank) if (holds and delays is not None) else None,
    ring_blocks=p["ring_blocks"], seed=p["seed"], gpu=p["gpu"])
np.savez(os.path.join(work, "rank%d.npz" % rank), ring=r["ring"], dm_times=r["dm_times"])
print("rank", rank, "done", r["ms"])
