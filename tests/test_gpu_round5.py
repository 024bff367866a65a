"""GPU tests (-m gpu) of the round-5 surface; every call goes through the C-ABI of libdsabf.so."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch():
    import torch as t

    assert t.cuda.is_available(), "these tests need a GPU"
    return t


@pytest.fixture(scope="module")
def bfmod():
    import dsabeamformer_amd as m

    return m


def _streaming_handle(bfmod, orc, seed, n_freq=48, n_beams=256, n_units=8, n_streams=4):
    g = orc.Geom(n_beams=n_beams, n_ant=64, n_freq=n_freq, n_avg=16, n_out_per_gemm=4)
    cfg = bfmod.production_config(n_avg=g.n_avg, n_out_per_gemm=g.n_out_per_gemm, n_freq=g.n_freq)
    cfg.n_beams, cfg.n_gemms_per_block, cfg.n_blocks_on_gpu, cfg.n_streams = g.n_beams, n_units, 2, n_streams
    rng = np.random.default_rng(seed)
    w = rng.integers(-127, 128, size=(g.n_freq, g.n_ant, g.n_beams, 2), dtype=np.int8)
    blocks = rng.integers(0, 256, size=(cfg.n_blocks_on_gpu, n_units, g.n_freq, g.n_time, g.n_ant), dtype=np.uint8)
    bf = bfmod.Beamformer(cfg)
    bf.set_weights(w)
    want = np.stack([orc.beamform(g, w, blocks[s]) for s in range(cfg.n_blocks_on_gpu)])     # [slot][unit][o][f][b]
    return g, cfg, bf, blocks, want


def test_a_raw_stream_sync_covers_the_coalesced_units_of_that_queue(torch, bfmod, orc):
    """with coalescing on, a unit enqueued on queue i is launched on one of the two flush queues.  A caller
    that orders on the RAW stream of queue i -- hipStreamSynchronize on the hipStream_t of bf_queue_stream, its own event, a
    collective chained on it -- must still see the unit's kernel AND host copy complete, as when the unit itself ran there:
    every queue that had a unit in a flush waits for the flush's end.  (Round 4: the raw stream was idle and the sync returned
    at once; the host buffers were stale, with no error.)"""
    g, cfg, bf, blocks, want = _streaming_handle(bfmod, orc, 51)
    per, n_st, n_u = bf.floats_per_detect, cfg.n_streams, cfg.n_gemms_per_block
    pinned_in = torch.from_numpy(blocks).pin_memory()
    for rep in range(6):                                    # (six times: a race that is lost only sometimes must never be lost)
        slot = rep % 2
        bf.submit_block(slot, pinned_in[slot], blocks[slot].nbytes)
        bf.sync(-1)
        host = torch.full((n_u, per), -1.0, dtype=torch.float32).pin_memory()
        for u in range(n_u):
            bf.enqueue_gemm_unit(u % n_st, slot, u, host[u])
        assert bf.counter("queued_units") == n_u
        q = 1 + rep % (n_st - 1)                            # never only the queue the flush itself may run on
        raw = bf.queue_stream(q)                            # launches what is queued, hands out the queue's hipStream_t
        assert bf.counter("queued_units") == 0
        torch.cuda.ExternalStream(raw).synchronize()        # the caller's own ordering: nothing but the raw stream
        for u in range(q, n_u, n_st):                       # the units THIS queue carried are complete on the host
            assert np.array_equal(host[u].numpy().reshape(want.shape[2:]), want[slot, u]), (rep, q, u)
        # ... and a copy the caller chains on that queue reads the finished powers (bf_enqueue_d2h: "behind everything enqueued
        # on that queue before the call, gemm-units included")
        bf.sync(-1)
    bf.close()


def test_destroy_launches_what_is_still_queued(torch, bfmod, orc):
    """gemm-units accepted with a host destination and never followed by an event or a sync were dropped by
    bf_destroy (round 4 cleared the queue); the literal pattern would have run them.  Destroy flushes, then drains the queues."""
    g, cfg, bf, blocks, want = _streaming_handle(bfmod, orc, 52, n_freq=8, n_beams=64)
    per, n_u = bf.floats_per_detect, cfg.n_gemms_per_block
    pinned_in = torch.from_numpy(blocks).pin_memory()
    bf.submit_block(0, pinned_in[0], blocks[0].nbytes)
    bf.sync(-1)
    host = torch.full((n_u, per), -1.0, dtype=torch.float32).pin_memory()
    for u in range(n_u):
        bf.enqueue_gemm_unit(u % cfg.n_streams, 0, u, host[u])
    assert bf.counter("queued_units") == n_u
    bf.close()
    assert np.array_equal(host.numpy().reshape(want[0].shape), want[0])


def test_two_late_dm0_requests_of_one_queue_run_in_order_on_that_queue(torch, bfmod, orc):
    """the direct path of bf_enqueue_dedisperse (the unit was launched already) runs on queue stream_idx itself,
    behind whichever queue produced the unit: repeated requests cannot overtake each other or overwrite d_ded under a copy."""
    g, cfg, bf, blocks, want = _streaming_handle(bfmod, orc, 53, n_freq=8, n_beams=64)
    n_st, n_u = cfg.n_streams, cfg.n_gemms_per_block
    pinned_in = torch.from_numpy(blocks).pin_memory()
    bf.submit_block(0, pinned_in[0], blocks[0].nbytes)
    bf.sync(-1)
    rows = torch.full((3 * n_st, g.n_beams), -1.0, dtype=torch.float32).pin_memory()
    for rnd in range(2):
        for u in range(n_u):
            bf.enqueue_gemm_unit(u % n_st, 0, u, None)
        bf.queue_stream(0)                                   # launched; every later DM-0 request takes the direct path
        for k in range(3):
            for st in range(n_st):
                bf.enqueue_dedisperse(st, rows[k * n_st + st])
        bf.sync(-1)
        for k in range(3):
            for st in range(n_st):
                last_u = n_u - n_st + st                     # the queue's most recent unit
                assert np.array_equal(rows[k * n_st + st].numpy(), orc.dedisperse(g, want[0, last_u, 0])), (rnd, k, st)
        rows.fill_(-1.0)
    bf.close()


def test_plain_bench_prints_its_one_line(torch):
    """The N = 1 command of the driver, shortened: one JSON line, status 0, the roofline object says what binds the kernel."""
    import json
    import subprocess
    import sys

    from conftest import ROOT

    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "2", "--min-warm-seconds", "0.2",
                        "--no-extras", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    roof = d["roofline"]
    assert d["value"] > 0 and d["n_gpus"] == 1 and roof["bound"] == "mfma" and 0.3 < roof["frac"] < 0.7
    assert roof["bound_measured"] == "simd-issue" and 15 < roof["valu_per_mfma"] < 19 and 0.9 < roof["issue_occupancy"] < 1.1
    assert abs(roof["issue_model_cycles_per_mfma"] - (13 + 2.45 * roof["valu_per_mfma"])) < 1e-6 and 1.8 < roof["clock_ghz_under_load"] < 2.5
    assert {"valu_per_mfma", "issue_occupancy", "bound_measured", "clock_ghz_under_load"} <= set(roof["from_committed_profile"])
