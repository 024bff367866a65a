"""Multi-process CPU test (gloo, world sizes 2 / 4 / 8) of the multi-GPU path: frequency sharding + the detected-power gather
must reassemble exactly the single-device result (no arithmetic happens in the collective).  The messages exchanged are the
PRODUCT's: every rank asks libdsabf.so for its bf_gather_plan and executes it with torch.distributed point-to-point
operations -- the same walk bf_gather_detected does with ncclSend / ncclRecv (csrc/bf_comm.cpp).  The oracle stands in for
the device (this box has no GPU); tests/test_gpu_multirank.py runs the same partition with the HIP kernel."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["REPO_ROOT"])
sys.path.insert(0, os.path.join(os.environ["REPO_ROOT"], "tests"))
import numpy as np, torch, torch.distributed as dist
import oracle as orc
from support import shard_layouts as shard
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
g = orc.Geom(n_beams=32, n_ant=16, n_freq=8, n_avg=16, n_out_per_gemm=2)
rng = np.random.default_rng(5)
w = rng.integers(-127, 128, size=(g.n_freq, g.n_ant, g.n_beams, 2), dtype=np.int8)
packed = rng.integers(0, 256, size=(4, g.n_freq, g.n_time, g.n_ant), dtype=np.uint8)
full = orc.beamform(g, w, packed).reshape(-1, g.n_freq, g.n_beams)          # [og][F][B], the single-device answer
# what THIS rank's GPU would compute: its frequency shard (the oracle stands in for the device in this CPU test)
gl = orc.Geom(n_beams=g.n_beams, n_ant=g.n_ant, n_freq=g.n_freq // world, n_avg=g.n_avg, n_out_per_gemm=g.n_out_per_gemm)
f0, f1 = shard.freq_range(rank, world, g.n_freq)
lw = np.ascontiguousarray(shard.shard_weights(w, rank, world))
lp = shard.shard_packed(packed, rank, world)
assert lw.shape[0] == f1 - f0 and lp.shape == (4, f1 - f0, g.n_time, g.n_ant)
local = torch.from_numpy(orc.beamform(gl, lw, lp).reshape(-1, gl.n_freq, g.n_beams))
assert np.array_equal(local.numpy(), full[:, f0:f1])
og = local.shape[0]
# ---- the PRODUCT's gather plan (bf_gather_plan, csrc/bf_comm.cpp) executed between real processes: every SEND / RECV / COPY
# message of this rank's plan, in plan order, as gloo point-to-point operations on CPU tensors -- posted non-blocking and
# completed together, which is what bf_gather_detected does with ncclSend / ncclRecv inside one ncclGroup.  Gloo, like RCCL,
# matches the messages of one (sender, receiver) pair in issue order; nothing else orders them.
import ctypes as C
from dsabeamformer_amd._lib import BfGatherMsg, load
lib = load()
SEND, RECV, COPY = 0, 1, 2
row_floats = gl.n_freq * g.n_beams
def plan_of(layout, root):
    n = lib.bf_gather_plan(layout, og, row_floats, world, rank, root, None, 0)
    arr = (BfGatherMsg * max(n, 1))()
    assert lib.bf_gather_plan(layout, og, row_floats, world, rank, root, arr, n) == n
    return [(m.kind, m.peer, m.local_offset, m.full_offset, m.count) for m in arr[:n]]
def gather(layout, root, src_flat):
    held = lib.bf_gather_rows_held(og, world, rank, root)
    buf = torch.full((world * held * row_floats,), float("nan")) if held else None
    reqs = []
    for kind, peer, loff, foff, count in plan_of(layout, root):
        if kind == SEND:
            reqs.append(dist.isend(src_flat[loff:loff + count], dst=peer))
        elif kind == RECV:
            reqs.append(dist.irecv(buf[foff:foff + count], src=peer))
        else:
            assert kind == COPY and peer == rank
            buf[foff:foff + count] = src_flat[loff:loff + count]
    for q in reqs:
        q.wait()
    return held, buf
shards = np.stack([full[:, a * gl.n_freq:(a + 1) * gl.n_freq] for a in range(world)])     # [rank][row][f_local][b]
for layout in (0, 1):                                   # BF_GATHER_LAYOUT_FREQ_MAJOR / _RANK_MAJOR
    for root in (0, world - 1, -1, -2):                  # one owner (first / last rank), everybody, distributed owners
        for it in (1, 2):                                # back to back: a second gather must not overtake the first
            src = (local * float(it)).reshape(-1).contiguous()
            held, buf = gather(layout, root, src)
            want_held = og // world if root == -2 else og if root in (-1, rank) else 0
            assert held == want_held, (layout, root, rank, held)
            if not held:
                assert buf is None
                continue
            first = rank * held if root == -2 else 0
            got = buf.numpy()
            assert not np.isnan(got).any(), (layout, root, rank)
            if layout == 0:                              # the reference's [o][f][b] over the whole band
                assert np.array_equal(got.reshape(held, g.n_freq, g.n_beams), full[first:first + held] * np.float32(it)), (layout, root, rank)
            else:                                        # sub-band-major [rank][row][f_local][b]
                assert np.array_equal(got.reshape(world, held, gl.n_freq, g.n_beams), shards[:, first:first + held] * np.float32(it)), (layout, root, rank)
        dist.barrier()
# ---- the STAGED transport of the freq-major layout (bf_gather_detected_staged): the wire is the rank-major plan above; what the
# device re-layout pass then does is full[bf_gather_offset(FREQ_MAJOR, rank, row)] = stage[bf_gather_offset(RANK_MAJOR, rank, row)]
# for every (rank, row) -- walked here on the CPU with the product's own offset arithmetic
for root in (0, world - 1, -1, -2):
    src = local.reshape(-1).contiguous()
    held, stage = gather(1, root, src)
    if held:
        first = rank * held if root == -2 else 0
        relaid = torch.full_like(stage, float("nan"))
        for r in range(world):
            for row in range(held):
                a = lib.bf_gather_offset(1, held, row_floats, world, r, row)
                b = lib.bf_gather_offset(0, held, row_floats, world, r, row)
                relaid[b:b + row_floats] = stage[a:a + row_floats]
        assert np.array_equal(relaid.numpy().reshape(held, g.n_freq, g.n_beams), full[first:first + held]), (root, rank)
    dist.barrier()
# sub-band dedispersion: each rank sums its own channels (delays against the band-wide reference frequency), the
# partials are added in rank order -> identical on every rank and equal to the banded oracle sum
series = full                                                       # [t][F][B] detected series, t = og outputs
freq = np.array([orc.freq_weights(0, 8 * c) for c in range(g.n_freq)], np.float32)
delays = orc.dm_delays(np.array([0.0, 20.0, 40.0]), freq, float(freq[0]), 0.131)
assert 0 < delays.max() < og
n_t_out = og - int(delays.max())
mine = torch.from_numpy(orc.dedisperse_dm(np.ascontiguousarray(series[:, f0:f1]), delays[:, f0:f1], n_t_out))
tot = shard.reduce_dedispersed(torch, dist, mine)
want = None
for r in range(world):
    a, b = shard.freq_range(r, world, g.n_freq)
    part = orc.dedisperse_dm(np.ascontiguousarray(series[:, a:b]), delays[:, a:b], n_t_out)
    want = part if want is None else want + part
assert np.array_equal(tot.numpy(), want), rank
single = orc.dedisperse_dm(series, delays, n_t_out)                  # one device, whole band: same up to fp32 grouping
assert np.allclose(tot.numpy(), single, rtol=1e-6, atol=0)
dist.barrier()
dist.destroy_process_group()
print("rank %d ok" % rank)
'''


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 4, 8])
def test_frequency_shard_and_gather(tmp_path, world):
    """1 process per (pretend) GPU, gloo: world sizes of the driver's scaling runs (8 frequencies -> 4, 2, 1 per rank)."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, REPO_ROOT=ROOT, OMP_NUM_THREADS="1" if world > 2 else "2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    for r in range(world):
        assert "rank %d ok" % r in out.stdout


def test_freq_range_rules():
    from support import shard_layouts as shard

    assert [shard.freq_range(r, 8, 256) for r in (0, 3, 7)] == [(0, 32), (96, 128), (224, 256)]
    with pytest.raises(ValueError):
        shard.freq_range(0, 3, 256)
