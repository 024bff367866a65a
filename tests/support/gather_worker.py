"""Worker of tests/test_gpu_multirank.py: ONE rank of a frequency partition, as its own process (TEST INFRASTRUCTURE).
usage: gather_worker.py rank world workdir          (DSABF_RCCL_LIB must point at tests/support/libfakerccl.so)

Every rank builds the same seeded full-band problem, beamforms ITS frequency shard on GPU 0 and takes part in
bf_gather_detected for every receiver mode and layout; what it received is saved for the parent to compare with the
oracle's full-band result."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

rank, world, work = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
import torch  # noqa: E402

import dsabeamformer_amd as bfm  # noqa: E402
from dsabeamformer_amd import api  # noqa: E402

F, B, A, n_avg, n_out, n_units = 8 * world, 64, 64, 16, 2, 4
if os.environ.get("GATHER_SHAPE") == "c4":
    # BASELINE configs[3] at its TRUE per-rank workload: 256 channels over 8 ranks = 32 per rank x 256 beams x 64 antennas,
    # N_TIME 512 (16 outputs x n_ipo 32), 2 gemm-units; the reference's own steering fan (each rank's slice is
    # conjugate-symmetric: the pair kernel, as in the bench)
    assert world == 8
    F, B, A, n_avg, n_out, n_units = 256, 256, 64, 16, 16, 2
fl = F // world
rng = np.random.default_rng(2026)
if os.environ.get("GATHER_SHAPE") == "c4":
    from dsabeamformer_amd import host

    w = host.make_weights_default(n_beams=B, n_ant=A, n_freq_total=256, gpu=0)
else:
    w = rng.integers(-127, 128, size=(F, A, B, 2), dtype=np.int8)
n_time = n_out * 2 * n_avg
packed = rng.integers(0, 256, size=(n_units, F, n_time, A), dtype=np.uint8)
np.savez(os.path.join(work, "problem.npz"), w=w, packed=packed) if rank == 0 else None

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
comm = api.Comm(rank, world, uid, device=0)

cfg = bfm.production_config(n_beams=B, n_ant=A, n_freq=fl, n_avg=n_avg, n_out_per_gemm=n_out)
bf = bfm.Beamformer(cfg)
bf.set_weights(np.ascontiguousarray(w[rank * fl:(rank + 1) * fl]))
if os.environ.get("GATHER_SHAPE") == "c4":
    assert "PAIRED" in bf.kernel_info(n_units)["kernel"] and cfg.n_freq == 32 and bf.n_time == 512
d_in = torch.from_numpy(np.ascontiguousarray(packed[:, rank * fl:(rank + 1) * fl])).cuda()
n_rows, row_floats = n_units * n_out, fl * B
d_local = torch.empty(n_rows * row_floats, dtype=torch.float32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
bf.beamform(d_in, n_units, d_local, s)
res = {}
for root in (0, world - 1, api.GATHER_ROOT_ALL, api.GATHER_ROOT_DISTRIBUTED):
    for layout in (api.GATHER_FREQ_MAJOR, api.GATHER_RANK_MAJOR):
        held = comm.rows_held(n_rows, root)
        d_full = torch.full((max(held, 1) * world * row_floats,), float("nan"), dtype=torch.float32, device="cuda")
        comm.gather(d_local, n_rows, row_floats, root, layout, d_full if held else None, s)
        torch.cuda.synchronize()
        if held:
            res["root%d_layout%d" % (root, layout)] = d_full.cpu().numpy()
    # the staged transport of the freq-major layout: rank-major on the wire, one device re-layout pass (bf_gather_detected_staged)
    held = comm.rows_held(n_rows, root)
    d_full = torch.full((max(held, 1) * world * row_floats,), float("nan"), dtype=torch.float32, device="cuda")
    d_stage = torch.full((max(held, 1) * world * row_floats,), float("nan"), dtype=torch.float32, device="cuda")
    comm.gather_staged(d_local, n_rows, row_floats, root, d_full if held else None, d_stage if held else None, s)
    torch.cuda.synchronize()
    if held:
        res["root%d_staged" % root] = d_full.cpu().numpy()
# rank 0: DM-0 sum and a DM ladder over the WHOLE band, on the freq-major gathered series (rows = time samples)
comm.gather(d_local, n_rows, row_floats, 0, api.GATHER_FREQ_MAJOR,
            d_band := (torch.empty(n_rows * world * row_floats, dtype=torch.float32, device="cuda") if rank == 0 else None), s)
if rank == 0:
    d_ded = torch.empty(B, dtype=torch.float32, device="cuda")
    bf.dedisperse_band(d_band, F, d_ded, s)                      # unit 0, output 0 = row 0
    delays = np.load(os.path.join(work, "delays.npy"))         # [n_dm][F], written by the parent
    n_dm, n_t_out = delays.shape[0], n_rows - int(delays.max())
    d_dm = torch.empty(n_dm * n_t_out * B, dtype=torch.float32, device="cuda")
    bf.dedisperse_dm_band(d_band, n_rows, F, torch.from_numpy(delays).cuda(), n_dm, n_t_out, d_dm, s)
    torch.cuda.synchronize()
    res["band_ded0"] = d_ded.cpu().numpy()
    res["band_dm"] = d_dm.cpu().numpy().reshape(n_dm, n_t_out, B)
np.savez(os.path.join(work, "rank%d.npz" % rank), local=d_local.cpu().numpy(), **res)
comm.close()
bf.close()
print("rank", rank, "done")
