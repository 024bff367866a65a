"""GPU tests (-m gpu) of the round-4 surface; every call goes through the C-ABI of libdsabf.so."""
import os

import numpy as np
import pytest

from conftest import LONG, sweep

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


def test_dm_calls_in_flight_on_two_streams_of_one_handle_do_not_share_scratch(torch, bfmod, orc):
    """bf_dedisperse_dm_device keeps, per call, which trial groups the shared-window kernel took, for
    the per-thread-window kernel that follows on the same stream.  Two calls in flight on different streams of ONE handle --
    a fine ladder (every group fits a window) and a coarse one (none does) -- used to share that scratch: a group could be
    skipped by both kernels and its output left unwritten.  The scratch is now per stream (and zeroed on that stream).
    Both results must be the oracle's, every round, with the two calls racing."""
    rng = np.random.default_rng(11)
    n_t, n_f, n_b, n_dm = 520, 32, 128, 96
    fine = (np.arange(n_dm)[:, None] * np.linspace(0.4, 0.0, n_f)[None, :]).astype(np.int32)
    coarse = (np.arange(n_dm)[:, None] * np.linspace(2.0, 0.0, n_f)[None, :] * 1.0).astype(np.int32)
    coarse[1::2] += 240                                  # neighbouring trials 240 rows apart: no 32-trial group fits a window (<= 224 rows)
    series = (rng.random((n_t, n_f, n_b), dtype=np.float32) * 1e3).astype(np.float32)
    n_out = {"fine": n_t - int(fine.max()), "coarse": n_t - int(coarse.max())}
    assert min(n_out.values()) > 16
    want = {"fine": orc.dedisperse_dm(series, fine, n_out["fine"]), "coarse": orc.dedisperse_dm(series, coarse, n_out["coarse"])}
    bf = bfmod.Beamformer(bfmod.debug_config(n_beams=n_b, n_freq=n_f))
    d_series = torch.from_numpy(series).cuda()
    d_del = {"fine": torch.from_numpy(np.ascontiguousarray(fine)).cuda(), "coarse": torch.from_numpy(np.ascontiguousarray(coarse)).cuda()}
    streams = {"fine": torch.cuda.Stream(), "coarse": torch.cuda.Stream()}
    torch.cuda.synchronize()
    for rnd in range(6):
        outs = {k: torch.full((n_dm, n_out[k], n_b), float("nan"), dtype=torch.float32, device="cuda") for k in ("fine", "coarse")}
        torch.cuda.synchronize()
        order = ("fine", "coarse") if rnd % 2 == 0 else ("coarse", "fine")
        for rep in range(3):                              # several calls back to back per stream: keep both queues busy
            for k in order:
                bf.dedisperse_dm(d_series, n_t, d_del[k], n_dm, n_out[k], outs[k], streams[k].cuda_stream)
        torch.cuda.synchronize()
        for k in ("fine", "coarse"):
            got = outs[k].cpu().numpy()
            assert not np.isnan(got).any(), (k, rnd, "a trial group was written by neither kernel")
            assert np.array_equal(got, want[k]), (k, rnd)
    bf.close()


def test_dm_calls_on_more_streams_than_the_handle_keeps_scratch_for(torch, bfmod, orc):
    """The per-stream DM scratch is evicted when a caller has used 64 streams (a device-wide sync first: nothing of the handle
    may still read it): 70 streams, mixed fine / coarse ladders in flight, every result the oracle's."""
    rng = np.random.default_rng(12)
    n_t, n_f, n_b, n_dm = 200, 16, 64, 40
    fine = (np.arange(n_dm)[:, None] * np.linspace(0.4, 0.0, n_f)[None, :]).astype(np.int32)
    coarse = (np.arange(n_dm)[:, None] * np.linspace(1.0, 0.0, n_f)[None, :]).astype(np.int32)
    coarse[1::2] += 120
    series = (rng.random((n_t, n_f, n_b), dtype=np.float32) * 1e3).astype(np.float32)
    lad = {"fine": fine, "coarse": coarse}
    n_out = {k: n_t - int(v.max()) for k, v in lad.items()}
    want = {k: orc.dedisperse_dm(series, v, n_out[k]) for k, v in lad.items()}
    bf = bfmod.Beamformer(bfmod.debug_config(n_beams=n_b, n_freq=n_f))
    d_series = torch.from_numpy(series).cuda()
    d_del = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in lad.items()}
    # raw HIP streams (torch.cuda.Stream() hands out 32 pooled ones): the runtime torch already loaded
    import ctypes

    hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))

    class _S:                                        # (just enough of torch.cuda.Stream for the loop below)
        def __init__(self):
            h = ctypes.c_void_p()
            assert hip.hipStreamCreate(ctypes.byref(h)) == 0
            self.cuda_stream = h.value

    streams = [_S() for _ in range(70)]
    outs = []
    for i in range(len(streams)):
        k = "fine" if i % 3 else "coarse"
        outs.append((k, torch.full((n_dm, n_out[k], n_b), float("nan"), dtype=torch.float32, device="cuda")))
    torch.cuda.synchronize()                         # (the fills ran on torch's current stream)
    assert len({st.cuda_stream for st in streams}) == 70
    for st, (k, o) in zip(streams, outs):
        bf.dedisperse_dm(d_series, n_t, d_del[k], n_dm, n_out[k], o, st.cuda_stream)
    for i in (0, 5, 69, 64, 1):                      # ... and again on streams on both sides of the eviction
        k, o = outs[i]
        bf.dedisperse_dm(d_series, n_t, d_del[k], n_dm, n_out[k], o, streams[i].cuda_stream)
    torch.cuda.synchronize()
    for i, (k, o) in enumerate(outs):
        assert np.array_equal(o.cpu().numpy(), want[k]), (i, k)
    bf.close()
    for st in streams:
        assert hip.hipStreamDestroy(ctypes.c_void_p(st.cuda_stream)) == 0


def test_switches_are_per_handle_and_checked(torch, bfmod):
    from dsabeamformer_amd._lib import DsabfError

    a = bfmod.Beamformer(bfmod.production_config(n_freq=8))
    b = bfmod.Beamformer(bfmod.production_config(n_freq=8))
    base = a.kernel_info(32)["grid"]
    a.set_switch("tsplit", 1)
    assert a.kernel_info(32)["grid"] == 8 * 1 * 1 and b.kernel_info(32)["grid"] == base      # n_freq x beam groups x 1 split
    a.set_switch("tsplit", 0)
    assert a.kernel_info(32)["grid"] == base
    lds = a.kernel_info(32)["lds_bytes"]
    a.set_switch("lds_pad", 160 * 1024)                   # clamped: a launch can never ask for more than the CU has
    assert a.kernel_info(32)["lds_bytes"] == 160 * 1024 and b.kernel_info(32)["lds_bytes"] == lds
    a.set_switch("lds_pad", 0)
    for name, value in (("tsplit", -1), ("lds_pad", -4), ("lds_pad", 1 << 20), ("no_such_switch", 1)):
        with pytest.raises(DsabfError):
            a.set_switch(name, value)
    a.close()
    b.close()


def _small_streaming_handle(bfmod, orc, seed, paired=False, n_ant=64, n_avg=16):
    g = orc.Geom(n_beams=64, n_ant=n_ant, n_freq=6, n_avg=n_avg, n_out_per_gemm=2)
    cfg = bfmod.production_config(n_avg=g.n_avg, n_out_per_gemm=g.n_out_per_gemm, n_freq=g.n_freq)
    cfg.n_ant = n_ant
    cfg.n_beams, cfg.n_gemms_per_block, cfg.n_blocks_on_gpu, cfg.n_streams = g.n_beams, 8, 3, 4
    rng = np.random.default_rng(seed)
    w = rng.integers(-127, 128, size=(g.n_freq, g.n_ant, g.n_beams, 2), dtype=np.int8)
    if paired:
        w[:, :, 32:, 0] = w[:, :, :32, 0][:, :, ::-1]
        w[:, :, 32:, 1] = -w[:, :, :32, 1][:, :, ::-1]
    blocks = rng.integers(0, 256, size=(cfg.n_blocks_on_gpu, cfg.n_gemms_per_block, g.n_freq, g.n_time, g.n_ant), dtype=np.uint8)
    bf = bfmod.Beamformer(cfg)
    bf.set_weights(w)
    want = np.stack([orc.beamform(g, w, blocks[s]) for s in range(cfg.n_blocks_on_gpu)])     # [slot][unit][o][f][b]
    return g, cfg, bf, blocks, want


@pytest.mark.parametrize("paired", [False, True])
def test_the_references_unit_loop_is_coalesced_into_one_launch_per_block(torch, bfmod, orc, paired):
    """a drop-in caller keeps the reference's loop (src/beamformer.cu:454-519: one K1-K4 enqueue per
    gemm-unit, round-robin over the N_STREAMS queues, the D2H of every unit into beam_out[stream], K5 for the units that need
    it, generate_analysis_event after the block) -- and gets ONE fused launch per block.  Same bits as the literal pattern
    (bf_set_switch "coalesce" 0) and as the oracle; beam_out[stream] ends up holding the LAST unit of that stream, as the
    reference's overwriting copies leave it."""
    from dsabeamformer_amd import api

    g, cfg, bf, blocks, want = _small_streaming_handle(bfmod, orc, 41, paired)
    assert ("PAIRED" in bf.kernel_info(8)["kernel"]) == paired
    per = bf.floats_per_detect
    n_st, n_u = cfg.n_streams, cfg.n_gemms_per_block
    pinned_in = torch.from_numpy(blocks).pin_memory()
    for slot in range(cfg.n_blocks_on_gpu):
        bf.submit_block(slot, pinned_in[slot], blocks[slot].nbytes)
    bf.sync(-1)
    results = {}
    for mode in ("coalesced", "literal"):
        bf.set_switch("coalesce", 1 if mode == "coalesced" else 0)
        beam_out = torch.zeros((n_st, per), dtype=torch.float32).pin_memory()            # the reference's beam_out (:249)
        rows = torch.full((cfg.n_blocks_on_gpu * n_u, g.n_beams), -1.0, dtype=torch.float32).pin_memory()   # dedispersed_out (:212)
        evs = [api.event_create() for _ in range(cfg.n_blocks_on_gpu)]
        before = bf.counter("fused_launches")
        for slot in range(cfg.n_blocks_on_gpu):
            ts_of = list(range(n_st))                                                    # timeSlice[], :319
            for part in range(n_u // n_st):
                for st in range(n_st):
                    bf.enqueue_gemm_unit(st, slot, ts_of[st], beam_out[st])               # :464-488
                    if (slot * n_u + ts_of[st]) % 5 != 3:                                # check_ready_for_dh2_transfer: not every unit
                        bf.enqueue_dedisperse(st, rows[slot * n_u + ts_of[st]])          # :498-510
                    ts_of[st] = (ts_of[st] + n_st) % n_u                                 # :515-519
            if mode == "coalesced":
                assert bf.counter("queued_units") == n_u          # nothing launched before the event orders it
            bf.record_analysis_event(evs[slot])                                          # :525
            assert bf.counter("queued_units") == 0
        for e in evs:
            while api.event_query(e) != 0:
                pass
        launches = bf.counter("fused_launches") - before
        assert launches == (cfg.n_blocks_on_gpu if mode == "coalesced" else cfg.n_blocks_on_gpu * n_u), (mode, launches)
        results[mode] = (beam_out.numpy().copy(), rows.numpy().copy())
        for e in evs:
            api.event_destroy(e)
    last = cfg.n_blocks_on_gpu - 1
    for mode, (bo, rw) in results.items():
        for st in range(n_st):      # the last unit each stream carried: time slice n_u - n_st + st of the last block
            assert np.array_equal(bo[st].reshape(want.shape[2:]), want[last, n_u - n_st + st]), (mode, st)
        for slot in range(cfg.n_blocks_on_gpu):
            for u in range(n_u):
                k = slot * n_u + u
                if k % 5 != 3:
                    assert np.array_equal(rw[k], orc.dedisperse(g, want[slot, u, 0])), (mode, k)
                else:
                    assert (rw[k] == -1.0).all(), (mode, k)     # never asked for: never written
    bf.close()


def test_coalescing_handles_any_enqueue_order(torch, bfmod, orc):
    """Units out of order, the same time slice twice, more than a block's worth before the event, several slots mixed, a
    collapse requested after its unit was already launched, a single queue synchronised: every host buffer holds its own
    unit's bits."""
    g, cfg, bf, blocks, want = _small_streaming_handle(bfmod, orc, 43)
    per = bf.floats_per_detect
    pinned_in = torch.from_numpy(blocks).pin_memory()
    for slot in range(cfg.n_blocks_on_gpu):
        bf.submit_block(slot, pinned_in[slot], blocks[slot].nbytes)
    bf.sync(-1)
    seq = [(0, 0, 0), (1, 0, 2), (2, 0, 4), (3, 0, 1), (0, 0, 3),            # (queue, slot, time slice): strides and gaps
           (1, 1, 3),                                                       # slice 3 again, another slot: a clash -> flush first
           (2, 1, 4), (3, 1, 5), (0, 2, 6), (1, 2, 7),                      # a run that changes slot half way
           (2, 0, 5), (3, 0, 6), (0, 0, 7), (1, 1, 0), (2, 1, 1), (3, 1, 2), (0, 2, 0), (1, 2, 1)]   # > 8 queued before any event
    outs = torch.zeros((len(seq), per), dtype=torch.float32).pin_memory()
    rows = torch.zeros((len(seq), g.n_beams), dtype=torch.float32).pin_memory()
    before = bf.counter("fused_launches")
    for k, (q, slot, ts) in enumerate(seq):
        bf.enqueue_gemm_unit(q, slot, ts, outs[k])
        if k % 2 == 0:
            bf.enqueue_dedisperse(q, rows[k])
    bf.sync(1)                                      # one queue only: still everything that was queued is launched ...
    assert bf.counter("queued_units") == 0
    bf.sync(-1)
    assert bf.counter("fused_launches") - before < len(seq)
    for k, (q, slot, ts) in enumerate(seq):
        assert np.array_equal(outs[k].numpy().reshape(want.shape[2:]), want[slot, ts]), k
        if k % 2 == 0:
            assert np.array_equal(rows[k].numpy(), orc.dedisperse(g, want[slot, ts, 0])), k
    # a collapse asked for AFTER its unit was launched (the event came first): the unit's powers are still where it left them
    late = torch.zeros(g.n_beams, dtype=torch.float32).pin_memory()
    bf.enqueue_gemm_unit(2, 1, 6, None)
    bf.sync(-1)
    bf.enqueue_dedisperse(2, late)
    bf.sync(2)
    assert np.array_equal(late.numpy(), orc.dedisperse(g, want[1, 6, 0]))
    bf.close()


# (the 48x walks: a queue's last unit overwritten in the block buffer before its late DM-0 request; 2118, round 5: ... UNDER a late
#  DM-0 read that ran on another HIP queue than the flush that overwrote it)
@pytest.mark.parametrize("seed", sweep([1, 2, 3, 4, 5, 6, 484, 486, 489, 498, 2118], [486, 2118]))
def test_random_call_sequences_leave_every_host_buffer_with_its_own_units_bits(torch, bfmod, orc, seed):
    """Randomised walk over the streaming entry points -- per-unit enqueues (with and without a host destination), DM-0 requests,
    block launches, analysis events, queue and device syncs, the coalesce switch flipped mid-stream -- with a private host
    buffer per request: whatever the order, every buffer ends up with the bits of the unit it was asked for."""
    from dsabeamformer_amd import api

    # seeds from 10000 on (tools/fuzz_calls.py) walk other kernel families too: two k-steps, dword rows, the deep class, fusedg_kernel,
    # run-time windows
    n_ant, n_avg = ((64, 16), (100, 16), (192, 16), (132, 16), (64, 3), (320, 8))[(seed // 7) % 6] if seed >= 10000 else (64, 16)
    g, cfg, bf, blocks, want = _small_streaming_handle(bfmod, orc, 50 + seed, paired=bool(seed & 1), n_ant=n_ant, n_avg=n_avg)
    per = bf.floats_per_detect
    n_u, n_q, n_slots = cfg.n_gemms_per_block, cfg.n_streams, cfg.n_blocks_on_gpu
    pinned_in = torch.from_numpy(blocks).pin_memory()
    for slot in range(n_slots):
        bf.submit_block(slot, pinned_in[slot], blocks[slot].nbytes)
    bf.sync(-1)
    rng = np.random.default_rng(seed)
    n_ops = 60
    outs = torch.zeros((n_ops * n_u, per), dtype=torch.float32).pin_memory()
    rows = torch.zeros((n_ops * n_u, g.n_beams), dtype=torch.float32).pin_memory()
    # a second weight set, swapped in mid-stream by some walks: a request is answered under the weights in force when it was made
    w_sets = [None, np.random.default_rng(900 + seed).integers(-127, 128, size=(g.n_freq, g.n_ant, g.n_beams, 2), dtype=np.int8)]
    wants = [want, None]
    which = 0
    expect_out, expect_row = {}, {}          # buffer index -> (slot, unit, weight set)
    nxt = [0, 0]                             # next free out / row buffer
    last_unit = {}                           # queue -> (slot, unit) of its latest gemm-unit (what a DM-0 request collapses)
    events = []
    for _ in range(n_ops):
        op = rng.choice(["unit", "unit", "unit", "unit_nohost", "ded", "ded", "block", "event", "sync_q", "sync_all", "switch"]
                        + (["weights"] if seed % 3 == 0 else []))
        q = int(rng.integers(n_q))
        if op in ("unit", "unit_nohost"):
            slot, ts = int(rng.integers(n_slots)), int(rng.integers(n_u))
            dst = None
            if op == "unit":
                dst = outs[nxt[0]]
                expect_out[nxt[0]] = (slot, ts, which)
                nxt[0] += 1
            bf.enqueue_gemm_unit(q, slot, ts, dst)
            last_unit[q] = (slot, ts, which)
        elif op == "ded" and q in last_unit:
            bf.enqueue_dedisperse(q, rows[nxt[1]])
            expect_row[nxt[1]] = last_unit[q]
            nxt[1] += 1
        elif op == "block":
            slot, first = int(rng.integers(n_slots)), int(rng.integers(n_u))
            n = int(rng.integers(1, n_u - first + 1))
            dsts = []
            for u in range(first, first + n):
                dsts.append(outs[nxt[0]])
                expect_out[nxt[0]] = (slot, u, which)
                nxt[0] += 1
            bf.enqueue_block(q, slot, first, n, dsts)
            last_unit.pop(q, None)           # (a DM-0 request behind a block launch is bf_enqueue_block_dedisperse's business)
            if rng.integers(3) == 0:         # ... the DM-0 rows of a stretch of the block just launched, in one launch
                a = first + int(rng.integers(n))
                m = int(rng.integers(1, first + n - a + 1))
                bf.enqueue_block_dedisperse(q, a, m, rows[nxt[1]:nxt[1] + m])
                for u in range(a, a + m):
                    expect_row[nxt[1]] = (slot, u, which)
                    nxt[1] += 1
        elif op == "event":
            ev = api.event_create()
            bf.record_analysis_event(ev)
            events.append(ev)
        elif op == "sync_q":
            bf.sync(q)
        elif op == "sync_all":
            bf.sync(-1)
        elif op == "switch":
            bf.set_switch("coalesce", int(rng.integers(2)))
        elif op == "weights":
            if w_sets[0] is None:            # (first swap: remember the set the handle was built with, and the other set's answers)
                w_sets[0] = np.random.default_rng(50 + seed).integers(-127, 128, size=w_sets[1].shape, dtype=np.int8)   # (= _small_streaming_handle's)
                if seed & 1:
                    w_sets[0][:, :, 32:, 0] = w_sets[0][:, :, :32, 0][:, :, ::-1]
                    w_sets[0][:, :, 32:, 1] = -w_sets[0][:, :, :32, 1][:, :, ::-1]
                wants[1] = np.stack([orc.beamform(g, w_sets[1], blocks[sl]) for sl in range(n_slots)])
            which ^= 1
            bf.set_weights(w_sets[which])
    bf.sync(-1)
    assert bf.counter("queued_units") == 0
    for ev in events:
        assert api.event_query(ev) == 0
        api.event_destroy(ev)
    for k, (slot, u, ws) in expect_out.items():
        assert np.array_equal(outs[k].numpy().reshape(want.shape[2:]), wants[ws][slot, u]), (seed, k, slot, u, ws)
    for k, (slot, u, ws) in expect_row.items():
        assert np.array_equal(rows[k].numpy(), orc.dedisperse(g, wants[ws][slot, u, 0])), (seed, k, slot, u, ws)
    assert len(expect_out) > 10
    bf.close()


@pytest.mark.parametrize("seed", sweep(range(12), [9]))
def test_production_loop_to_file_under_random_launch_patterns(bfmod, orc, tmp_path, monkeypatch, seed):
    """run_observation with a file sink under random block / queue counts and every launch pattern it offers (whole blocks,
    sub-block launches, the reference's per-unit loop coalesced or literal): the detected stream in the file is the oracle's,
    every gemm-unit, in index order."""
    from dsabeamformer_amd import host

    rng = np.random.default_rng(700 + seed)
    n_u = int(rng.choice([4, 8, 16]))
    n_st = int(rng.choice([s for s in (1, 2, 4, 8) if s <= n_u]))
    n_blocks, ring_blocks = int(rng.integers(3, 10)), int(rng.integers(2, 5))
    pattern = rng.choice(["block", "sub", "units", "units_literal"])
    if pattern == "sub":
        monkeypatch.setenv("DSABF_UNITS_PER_LAUNCH", str(int(rng.choice([1, 2, n_u // 2]))))
    elif pattern in ("units", "units_literal"):
        monkeypatch.setenv("DSABF_UNIT_LAUNCH", "1")
        if pattern == "units_literal":
            monkeypatch.setenv("DSABF_COALESCE", "0")
    n_avg = int(rng.choice([16, 8]))
    cfg = bfmod.production_config(n_freq=8, n_avg=n_avg, n_out_per_gemm=int(rng.choice([2, 4])))
    cfg.n_beams, cfg.n_gemms_per_block, cfg.n_streams = 64, n_u, n_st
    path = str(tmp_path / "detected.bin")
    r = host.run_observation_junk_to_file(cfg, n_blocks, path, ring_blocks=ring_blocks, seed=1000 + seed, gpu=1)
    assert r["gemms_written"] == n_blocks * n_u, (pattern, n_u, n_st)
    hdr, data = host.read_detected_file(path)
    assert data.shape == (n_blocks * n_u, cfg.n_out_per_gemm, 8, 64)
    g = orc.Geom(n_beams=64, n_ant=64, n_freq=8, n_avg=n_avg, n_out_per_gemm=cfg.n_out_per_gemm)
    w = orc.make_weights(g, orc.default_positions(64), orc.default_directions(64), 1)
    for blk in range(n_blocks):
        want = orc.beamform(g, w, r["ring"][blk % ring_blocks])
        assert np.array_equal(data[blk * n_u:(blk + 1) * n_u].reshape(want.shape), want), (pattern, n_u, n_st, n_blocks, ring_blocks, blk)


# ---- fusedg_kernel (csrc/bf_fusedg.hip): the reference's whole geometry contract ------------------------------------------------
def _cfg_of(bfmod, g, **over):
    cfg = bfmod.debug_config(n_beams=g.n_beams, n_ant=g.n_ant, n_freq=g.n_freq, n_pol=g.n_pol, n_avg=g.n_avg,
                             n_out_per_gemm=g.n_out_per_gemm)
    for k, v in over.items():
        setattr(cfg, k, v)
    return cfg


def _beamform(torch, bf, packed, n_floats):
    d_in = torch.from_numpy(packed).cuda()
    d_out = torch.full((n_floats,), float("nan"), dtype=torch.float32, device="cuda")
    bf.beamform(d_in, packed.shape[0], d_out, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return d_out.cpu().numpy()


@pytest.mark.parametrize("n_ant", [132, 160, 192, 256, 260, 320, 512])
@pytest.mark.parametrize("n_avg", [16, 1])
def test_any_antenna_count_beyond_128_bit_exact(torch, bfmod, orc, n_ant, n_avg):
    """VERDICT r03 item 4 / src/beamformer.hh:48,156: N_ANTENNAS is any multiple of 4.  More than two k-steps of 64 run
    fusedg_kernel (accumulator-stationary, one staged plane per k-step); bit-exact against the oracle, full-range weights and
    all 16 nibble codes (|n| reaches the seed trick's range only with true nibbles: 2032 * 512 < 2^22)."""
    g = orc.Geom(n_beams=96, n_ant=n_ant, n_freq=3, n_avg=n_avg, n_out_per_gemm=4 if n_avg > 1 else 8)
    rng = np.random.default_rng(n_ant + n_avg)
    w = rng.integers(-127, 128, size=(g.n_freq, g.n_ant, g.n_beams, 2), dtype=np.int8)
    packed = rng.integers(0, 256, size=(3, g.n_freq, g.n_time, g.n_ant), dtype=np.uint8)
    if n_ant == 512:          # the extreme sums: every voltage -8 - 8j, every weight -127 - 127j in one beam / frequency
        packed[1, 0] = 0x88
        w[0, :, 5] = -127
    bf = bfmod.Beamformer(_cfg_of(bfmod, g))
    bf.set_weights(w)
    name = bf.kernel_info(3)["kernel"]
    deep = n_ant <= 256 and n_avg == 16          # the three / four k-step classes of fused16_kernel (any multiple of 4: round 5)
    assert ("fused16_kernel<ANT=%d" % n_ant in name and "WAVES=8" in name) if deep else \
        ("fusedg_kernel" in name and ("%d k-steps" % -(-n_ant // 64)) in name), name
    want = orc.beamform(g, w, packed)
    got = _beamform(torch, bf, packed, want.size).reshape(want.shape)
    assert np.array_equal(got, want)
    bf.close()


@pytest.mark.parametrize("n_pol,n_avg", [(2, 3), (2, 5), (2, 12), (2, 7), (2, 20), (2, 48), (2, 50), (1, 3), (1, 1), (2, 64), (1, 37)])
@pytest.mark.parametrize("n_ant", [64, 100, 192])
def test_any_accumulation_window_bit_exact(torch, bfmod, orc, n_pol, n_avg, n_ant):
    """src/beamformer.hh:55-60: N_AVERAGING is any positive integer (the reference ships 1 and 16).  n_ipo = n_pol * n_avg that
    is not a power of two <= 64 runs fusedg_kernel with run-time stream boundaries: windows that tile a 32-row run with padding
    (6, 10, 24), windows longer than a chunk (40, 96, 100, 128), odd windows (3, 1, 37); several gemm-units, a ragged last stream
    group, the detect in all three readings."""
    g = orc.Geom(n_beams=64, n_ant=n_ant, n_freq=2, n_pol=n_pol, n_avg=n_avg, n_out_per_gemm=5)
    rng = np.random.default_rng(97 * n_avg + n_pol + n_ant)
    w = rng.integers(-127, 128, size=(g.n_freq, g.n_ant, g.n_beams, 2), dtype=np.int8)
    packed = rng.integers(0, 256, size=(3, g.n_freq, g.n_time, g.n_ant), dtype=np.uint8)
    # up to 128 antennas: the run-time-window instantiations of fused16_kernel; beyond: fusedg_kernel
    expect = "fused16_kernel<ANT=%d,NIPO=%d(run-time)" % (n_ant, g.n_ipo) if n_ant <= 128 else "fusedg_kernel"
    for mode, contract in ((0, orc.CONTRACT_NONE), (2, orc.CONTRACT_NVCC)):
        bf = bfmod.Beamformer(_cfg_of(bfmod, g, detect_mode=mode))
        bf.set_weights(w)
        assert expect in bf.kernel_info(3)["kernel"], bf.kernel_info(3)["kernel"]
        with orc.detect_contract(contract):
            want = orc.beamform(g, w, packed)
        got = _beamform(torch, bf, packed, want.size).reshape(want.shape)
        assert np.array_equal(got, want), (mode,)
        bf.close()
    bf = bfmod.Beamformer(_cfg_of(bfmod, g, detect_mode=1))          # fast: the stated tolerance against the exact value
    bf.set_weights(w)
    got = _beamform(torch, bf, packed, want.size).reshape(want.shape).astype(np.float64)
    exact = orc.beamform_exact(g, w, packed)
    rel = np.abs(got - exact) / np.maximum(exact, 1e-300)
    assert rel.max() <= (g.n_ipo + 1) * 2.0 ** -23
    bf.close()


@pytest.mark.parametrize("shape", [(64, 16, 4), (64, 1, 8), (64, 4, 4), (100, 16, 2), (128, 32, 2), (48, 8, 4), (36, 2, 8)])
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_generic_kernel_gives_the_bits_of_the_specialised_kernels(torch, bfmod, orc, monkeypatch, shape, mode):
    """On geometries both cover, fusedg_kernel (DSABF_GENERIC=1 at bf_create) and fused16_kernel (general and conjugate-pair)
    produce the same bits in every detect mode -- x16-scaled operands + one fma vs true nibbles + one fma is the same float."""
    n_ant, n_avg, n_out = shape
    g = orc.Geom(n_beams=96, n_ant=n_ant, n_freq=3, n_avg=n_avg, n_out_per_gemm=n_out)
    rng = np.random.default_rng(n_ant * 7 + n_avg)
    w = rng.integers(-127, 128, size=(g.n_freq, g.n_ant, g.n_beams, 2), dtype=np.int8)
    w[:, :, 48:, 0] = w[:, :, :48, 0][:, :, ::-1]          # conjugate-symmetric: the specialised side runs the pair kernel
    w[:, :, 48:, 1] = -w[:, :, :48, 1][:, :, ::-1]
    packed = rng.integers(0, 256, size=(5, g.n_freq, g.n_time, g.n_ant), dtype=np.uint8)
    res = {}
    for generic in (False, True):
        if generic:
            monkeypatch.setenv("DSABF_GENERIC", "1")
        else:
            monkeypatch.delenv("DSABF_GENERIC", raising=False)
        bf = bfmod.Beamformer(_cfg_of(bfmod, g, detect_mode=mode))
        bf.set_weights(w)
        assert ("fusedg_kernel" in bf.kernel_info(5)["kernel"]) == generic
        res[generic] = _beamform(torch, bf, packed, 5 * g.out_per_gemm)
        bf.close()
    monkeypatch.delenv("DSABF_GENERIC", raising=False)
    assert np.array_equal(res[True], res[False])
    if mode == 0:
        assert np.array_equal(res[True].reshape(-1), orc.beamform(g, w, packed).reshape(-1))


def test_generic_kernel_stage_parity_beams_and_time_splits(torch, bfmod, orc, monkeypatch):
    """bf_gemm_device (the scaled complex GEMM result, the reference's d_C) through fusedg_kernel; beams that fill neither a
    32-beam wave nor a 256-beam workgroup (plain stores, inactive waves); several workgroups along time (tsplit) with windows
    longer than a chunk; 1 frequency; the streaming entry points on top of it."""
    g = orc.Geom(n_beams=332, n_ant=196, n_freq=1, n_avg=21, n_out_per_gemm=3)      # n_ipo 42: two chunks per stream group
    rng = np.random.default_rng(8)
    w = rng.integers(-127, 128, size=(g.n_freq, g.n_ant, g.n_beams, 2), dtype=np.int8)
    packed = rng.integers(0, 256, size=(7, g.n_freq, g.n_time, g.n_ant), dtype=np.uint8)
    want = orc.beamform(g, w, packed)
    for tsplit in ("0", "1", "3"):
        monkeypatch.setenv("DSABF_TSPLIT", tsplit)
        bf = bfmod.Beamformer(_cfg_of(bfmod, g))
        bf.set_weights(w)
        got = _beamform(torch, bf, packed, want.size).reshape(want.shape)
        assert np.array_equal(got, want), tsplit
        if tsplit == "0":
            d_unit = torch.from_numpy(packed[2]).cuda()
            d_c = torch.full((g.n_freq, g.n_time, g.n_beams, 2), float("nan"), dtype=torch.float32, device="cuda")
            bf.gemm(d_unit, d_c, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            c_want = orc.gemm(g, w, orc.expand(packed[2]))
            assert np.array_equal(d_c.cpu().numpy().reshape(-1), np.asarray(c_want).reshape(-1))
        bf.close()
    monkeypatch.delenv("DSABF_TSPLIT", raising=False)


@pytest.mark.parametrize("per_unit", [False, True])
def test_debug_flow_end_to_end_on_a_geometry_only_the_generic_kernel_covers(bfmod, orc, tmp_path, per_unit):
    """The reference's `make debug` main() (generator -> H2D -> fused kernel -> DM-0 collapse -> data.py) on 132 antennas and
    N_AVERAGING 3: through the scheduler and the streaming entry points into fusedg_kernel, with block launches and with the
    reference's per-gemm-unit loop (coalesced); the whole table equals the oracle's."""
    from dsabeamformer_amd import host

    n_ant, n_beams, n_src = 132, 96, 64
    rng = np.random.default_rng(77)
    pos = np.zeros((n_ant, 3), np.float32)
    pos[:, 0] = np.linspace(-400.0, 400.0, n_ant) + rng.uniform(-2, 2, n_ant)
    pos[:, 1] = rng.uniform(-30, 30, n_ant)
    dirs = np.stack([np.linspace(-3.0, 3.0, n_beams) * np.pi / 180, rng.uniform(-0.5, 0.5, n_beams) * np.pi / 180], 1).astype(np.float32)
    src = np.stack([np.linspace(-2.9, 2.9, n_src) * np.pi / 180, np.zeros(n_src)], 1).astype(np.float32)
    pfile, dfile, sfile = tmp_path / "pos.txt", tmp_path / "dir.txt", tmp_path / "src.txt"
    pfile.write_text("%d\n" % n_ant + "".join("%r %r %r\n" % (float(p[0]), float(p[1]), float(p[2])) for p in pos))
    dfile.write_text("%d\n" % n_beams + "".join("%r %r\n" % (float(d[0]), float(d[1])) for d in dirs))
    sfile.write_text("%d\n" % n_src + "".join("%r %r\n" % (float(s[0]), float(s[1])) for s in src))
    cfg = bfmod.debug_config(n_ant=n_ant, n_beams=n_beams, n_freq=16, n_avg=3, n_out_per_gemm=4, n_gemms_per_block=16,
                             n_blocks_on_gpu=4, n_streams=4)
    ded, _ms = host.run_debug_observation(cfg, gpu=0, positions=str(pfile), directions=str(dfile), sources=str(sfile),
                                          output=str(tmp_path / "data.py"), max_sources=n_src, per_unit_launches=per_unit)
    assert ded.shape == (n_src, n_beams)
    g = orc.Geom(n_beams=n_beams, n_ant=n_ant, n_freq=16, n_avg=3, n_out_per_gemm=4)
    p32, d32, s32 = orc.read_positions(str(pfile), n_ant), orc.read_directions(str(dfile), n_beams), orc.read_directions(str(sfile))
    w = orc.make_weights(g, p32, d32, 0)
    units = orc.generate_test_data(g, p32, s32, 0, 0, n_src)
    out = orc.beamform(g, w, units)
    want = np.stack([orc.dedisperse(g, out[u]) for u in range(n_src)])
    assert np.array_equal(ded, want)
    assert ded.max() > 0 and np.isfinite(ded).all()


@pytest.mark.parametrize("seed", sweep(range(10), [2, 7]))
def test_debug_flow_with_random_catalogues_geometries_and_launch_patterns(bfmod, orc, tmp_path, seed):
    """`make debug` end to end with a source catalogue that does not fill its last block, a random antenna class / window /
    block size / queue count, block launches or the reference's per-unit loop: the whole table equals the oracle's."""
    from dsabeamformer_amd import host

    rng = np.random.default_rng(1300 + seed)
    n_ant = int(rng.choice([64, 64, 100, 128, 132, 192]))
    n_beams = int(rng.choice([32, 64, 96]))
    n_avg, n_out = int(rng.choice([1, 3, 8, 16])), int(rng.choice([2, 4, 8]))
    n_u = int(rng.choice([4, 8, 16]))
    n_st = int(rng.choice([q for q in (1, 2, 4, 8) if q <= n_u]))
    n_src = int(rng.integers(1, 3 * n_u + 2))
    per_unit = bool(rng.integers(2))
    pos = np.zeros((n_ant, 3), np.float32)
    pos[:, 0] = np.linspace(-400.0, 400.0, n_ant) + rng.uniform(-2, 2, n_ant)
    pos[:, 1] = rng.uniform(-30, 30, n_ant)
    dirs = np.stack([np.linspace(-3.0, 3.0, n_beams) * np.pi / 180, rng.uniform(-0.5, 0.5, n_beams) * np.pi / 180], 1).astype(np.float32)
    src = np.stack([rng.uniform(-2.9, 2.9, n_src) * np.pi / 180, rng.uniform(-0.3, 0.3, n_src) * np.pi / 180], 1).astype(np.float32)
    pfile, dfile, sfile = tmp_path / "pos.txt", tmp_path / "dir.txt", tmp_path / "src.txt"
    pfile.write_text("%d\n" % n_ant + "".join("%r %r %r\n" % (float(p[0]), float(p[1]), float(p[2])) for p in pos))
    dfile.write_text("%d\n" % n_beams + "".join("%r %r\n" % (float(d[0]), float(d[1])) for d in dirs))
    sfile.write_text("%d\n" % n_src + "".join("%r %r\n" % (float(x[0]), float(x[1])) for x in src))
    cfg = bfmod.debug_config(n_ant=n_ant, n_beams=n_beams, n_freq=8, n_avg=n_avg, n_out_per_gemm=n_out, n_gemms_per_block=n_u,
                             n_blocks_on_gpu=int(rng.integers(2, 5)), n_streams=n_st)
    what = (n_ant, n_beams, n_avg, n_out, n_u, n_st, n_src, per_unit)
    ded, _ms = host.run_debug_observation(cfg, gpu=0, positions=str(pfile), directions=str(dfile), sources=str(sfile),
                                          output=str(tmp_path / "data.py"), max_sources=max(n_src, 1), per_unit_launches=per_unit)
    assert ded.shape == (n_src, n_beams), what
    g = orc.Geom(n_beams=n_beams, n_ant=n_ant, n_freq=8, n_avg=n_avg, n_out_per_gemm=n_out)
    p32, d32, s32 = orc.read_positions(str(pfile), n_ant), orc.read_directions(str(dfile), n_beams), orc.read_directions(str(sfile))
    w = orc.make_weights(g, p32, d32, 0)
    units = orc.generate_test_data(g, p32, s32, 0, 0, n_src)
    out = orc.beamform(g, w, units)
    want = np.stack([orc.dedisperse(g, out[u]) for u in range(n_src)])
    assert np.array_equal(ded, want), what


# ---- the deep classes of fused16_kernel: three / four k-steps, weights stationary (129 ... 256 antennas) -----------------------
@pytest.mark.parametrize("n_ant", [144, 192, 208, 256, 132, 188, 196, 252])      # (the last four: rows only dword-aligned, round 5)
@pytest.mark.parametrize("n_avg,paired", [(16, False), (8, False), (32, False), (16, True), (32, True), (16, 256), (8, 96)])
def test_deep_classes_bit_exact_and_equal_to_the_generic_kernel(torch, bfmod, orc, monkeypatch, n_ant, n_avg, paired):
    """129 ... 256 antennas with windows of 16 / 32 / 64 samples (rows that are only dword-aligned: 16 / 32) run fused16_kernel with three or four k-steps:
    8-wave workgroups, two output slots per wave (general) or two pair tiles (conjugate-symmetric weights, beams in groups
    of 512), true-nibble operands.  Bit-exact vs the oracle in both bit-exact readings, within tolerance in the fast one, and
    the same bits as fusedg_kernel (DSABF_DEEP=0) on the same handle geometry; several chunks per workgroup, a ragged tail."""
    # paired: True = 512 beams (two pair tiles per wave), a number = that many beams (one pair tile per wave, a ragged workgroup)
    n_beams = 512 if paired is True else paired if paired else 288
    paired = bool(paired)
    g = orc.Geom(n_beams=n_beams, n_ant=n_ant, n_freq=2, n_avg=n_avg, n_out_per_gemm=3)
    rng = np.random.default_rng(n_ant * 3 + n_avg + paired)
    w = rng.integers(-127, 128, size=(g.n_freq, g.n_ant, g.n_beams, 2), dtype=np.int8)
    if n_ant == 256:                  # the extreme sums: |n| = 2032 * 256 in one beam (and, when paired, in its mirror image)
        w[0, :, 7] = -127
    if paired:
        hb = n_beams // 2
        w[:, :, hb:, 0] = w[:, :, :hb, 0][:, :, ::-1]
        w[:, :, hb:, 1] = -w[:, :, :hb, 1][:, :, ::-1]
    packed = rng.integers(0, 256, size=(5, g.n_freq, g.n_time, g.n_ant), dtype=np.uint8)
    if n_ant == 256:
        packed[2, 0] = 0x88
    monkeypatch.setenv("DSABF_TSPLIT", "2")
    got = {}
    for mode, contract in ((0, orc.CONTRACT_NONE), (2, orc.CONTRACT_NVCC), (1, None)):
        bf = bfmod.Beamformer(_cfg_of(bfmod, g, detect_mode=mode))
        bf.set_weights(w)
        name = bf.kernel_info(5)["kernel"]
        if n_ant % 16 and g.n_ipo == 64:   # dword-aligned rows in windows of 64: the deep class would spill; fusedg_kernel keeps them
            assert "fusedg_kernel" in name, name
        else:
            assert "fused16_kernel<ANT=%d" % n_ant in name and ("PAIRED" in name) == paired and "WAVES=8" in name, name
        out = _beamform(torch, bf, packed, 5 * g.out_per_gemm)
        bf.close()
        if contract is not None:
            with orc.detect_contract(contract):
                want = orc.beamform(g, w, packed)
            assert np.array_equal(out.reshape(want.shape), want), (mode,)
        else:
            exact = orc.beamform_exact(g, w, packed)
            rel = np.abs(out.reshape(exact.shape).astype(np.float64) - exact) / np.maximum(exact, 1e-300)
            assert rel.max() <= (g.n_ipo + 1) * 2.0 ** -23
        got[mode] = out
    monkeypatch.setenv("DSABF_DEEP", "0")
    for mode in (0, 1, 2):
        bf = bfmod.Beamformer(_cfg_of(bfmod, g, detect_mode=mode))
        bf.set_weights(w)
        assert "fusedg_kernel" in bf.kernel_info(5)["kernel"]
        assert np.array_equal(_beamform(torch, bf, packed, 5 * g.out_per_gemm), got[mode]), mode
        bf.close()
    monkeypatch.delenv("DSABF_DEEP", raising=False)
    monkeypatch.delenv("DSABF_TSPLIT", raising=False)


def test_deep_class_stage_parity_and_streaming(torch, bfmod, orc):
    """bf_gemm_device on a deep-class handle (fusedg_kernel's stage-parity launch on the shared fragment image), and the
    streaming entry points (a block through bf_enqueue_block and through the coalesced per-unit loop) at 192 antennas."""
    g = orc.Geom(n_beams=96, n_ant=192, n_freq=3, n_avg=8, n_out_per_gemm=2)
    cfg = _cfg_of(bfmod, g, n_gemms_per_block=4, n_blocks_on_gpu=2, n_streams=2)
    rng = np.random.default_rng(12)
    w = rng.integers(-127, 128, size=(g.n_freq, g.n_ant, g.n_beams, 2), dtype=np.int8)
    block = rng.integers(0, 256, size=(4, g.n_freq, g.n_time, g.n_ant), dtype=np.uint8)
    bf = bfmod.Beamformer(cfg)
    bf.set_weights(w)
    assert "fused16_kernel<ANT=192" in bf.kernel_info(4)["kernel"]
    s = torch.cuda.current_stream().cuda_stream
    d_c = torch.full((g.n_freq, g.n_time, g.n_beams, 2), float("nan"), dtype=torch.float32, device="cuda")
    bf.gemm(torch.from_numpy(block[1]).cuda(), d_c, s)
    torch.cuda.synchronize()
    assert np.array_equal(d_c.cpu().numpy().reshape(-1), np.asarray(orc.gemm(g, w, orc.expand(block[1]))).reshape(-1))
    want = orc.beamform(g, w, block)
    pinned = torch.from_numpy(block).pin_memory()
    bf.submit_block(1, pinned, block.nbytes)
    bf.sync(-1)
    outs = torch.zeros((4, g.out_per_gemm), dtype=torch.float32).pin_memory()
    bf.enqueue_block(0, 1, 0, 4, [outs[u] for u in range(4)])
    bf.sync(-1)
    assert np.array_equal(outs.numpy().reshape(want.shape), want)
    outs2 = torch.zeros((4, g.out_per_gemm), dtype=torch.float32).pin_memory()
    for u in range(4):
        bf.enqueue_gemm_unit(u % 2, 1, u, outs2[u])
    bf.sync(-1)
    assert np.array_equal(outs2.numpy(), outs.numpy())
    bf.close()


def test_units_queued_before_a_weight_change_run_under_the_old_weights(torch, bfmod, orc):
    g, cfg, bf, blocks, want = _small_streaming_handle(bfmod, orc, 47)
    pinned_in = torch.from_numpy(blocks).pin_memory()
    bf.submit_block(0, pinned_in[0], blocks[0].nbytes)
    bf.sync(-1)
    outs = torch.zeros((2, bf.floats_per_detect), dtype=torch.float32).pin_memory()
    bf.enqueue_gemm_unit(0, 0, 3, outs[0])
    assert bf.counter("queued_units") == 1
    w2 = np.random.default_rng(48).integers(-127, 128, size=(g.n_freq, g.n_ant, g.n_beams, 2), dtype=np.int8)
    bf.set_weights(w2)                                  # launches what is queued, waits for it, then swaps the images
    assert bf.counter("queued_units") == 0
    bf.enqueue_gemm_unit(1, 0, 3, outs[1])
    bf.sync(-1)
    assert np.array_equal(outs[0].numpy().reshape(want.shape[2:]), want[0, 3])
    assert np.array_equal(outs[1].numpy().reshape(want.shape[2:]), orc.beamform(g, w2, blocks[0][3:4])[0])
    bf.close()


@pytest.mark.parametrize("n_ant", [64, 36, 100, 128, 80])
@pytest.mark.parametrize("n_pol,n_avg,n_out", [(2, 3, 5), (2, 12, 3), (2, 20, 2), (1, 7, 9), (2, 1, 3), (2, 50, 1), (1, 1, 5), (2, 4, 3)])
def test_run_time_window_instantiations_equal_the_generic_kernel(torch, bfmod, orc, monkeypatch, n_ant, n_pol, n_avg, n_out):
    """Windows without a compile-time instantiation -- and short windows in gemm-units that are not whole 16-sample runs (n_ipo 2
    with 3 outputs, n_ipo 8 with 3) -- run fused16_kernel<..., NIPO = 0> up to 128 antennas: every antenna class, general and
    conjugate-pair weights, all three detect modes, several chunks per workgroup; the bits of fusedg_kernel (DSABF_RTW=0) and,
    in the two bit-exact readings, of the oracle."""
    g = orc.Geom(n_beams=96, n_ant=n_ant, n_freq=2, n_pol=n_pol, n_avg=n_avg, n_out_per_gemm=n_out)
    rng = np.random.default_rng(n_ant * 11 + n_avg * 3 + n_out)
    w = rng.integers(-127, 128, size=(g.n_freq, g.n_ant, g.n_beams, 2), dtype=np.int8)
    if (n_ant + n_avg) % 2:                            # conjugate-symmetric weights: the pair kernel
        w[:, :, 48:, 0] = w[:, :, :48, 0][:, :, ::-1]
        w[:, :, 48:, 1] = -w[:, :, :48, 1][:, :, ::-1]
    n_units = max(2, -(-700 // g.n_time))
    packed = rng.integers(0, 256, size=(n_units, g.n_freq, g.n_time, g.n_ant), dtype=np.uint8)
    monkeypatch.setenv("DSABF_TSPLIT", "2")
    res = {}
    for rtw in (True, False):
        if rtw:
            monkeypatch.delenv("DSABF_RTW", raising=False)
        else:
            monkeypatch.setenv("DSABF_RTW", "0")
        for mode in (0, 1, 2):
            bf = bfmod.Beamformer(_cfg_of(bfmod, g, detect_mode=mode))
            bf.set_weights(w)
            name = bf.kernel_info(n_units)["kernel"]
            assert ("fused16_kernel" in name and "NIPO=%d(run-time)" % g.n_ipo in name) if rtw else "fusedg_kernel" in name, name
            res[rtw, mode] = _beamform(torch, bf, packed, n_units * g.out_per_gemm)
            bf.close()
    monkeypatch.delenv("DSABF_RTW", raising=False)
    monkeypatch.delenv("DSABF_TSPLIT", raising=False)
    for mode in (0, 1, 2):
        assert np.array_equal(res[True, mode], res[False, mode]), mode
    for mode, contract in ((0, orc.CONTRACT_NONE), (2, orc.CONTRACT_NVCC)):
        with orc.detect_contract(contract):
            assert np.array_equal(res[True, mode].reshape(-1), orc.beamform(g, w, packed).reshape(-1)), mode


def test_run_time_window_stage_parity(torch, bfmod, orc):
    """bf_gemm_device (the reference's d_C) through the run-time-window instantiation: n_ipo 6, 100 antennas."""
    g = orc.Geom(n_beams=40, n_ant=100, n_freq=3, n_avg=3, n_out_per_gemm=7)
    rng = np.random.default_rng(5)
    w = rng.integers(-127, 128, size=(g.n_freq, g.n_ant, g.n_beams, 2), dtype=np.int8)
    unit = rng.integers(0, 256, size=(g.n_freq, g.n_time, g.n_ant), dtype=np.uint8)
    bf = bfmod.Beamformer(_cfg_of(bfmod, g))
    bf.set_weights(w)
    d_c = torch.full((g.n_freq, g.n_time, g.n_beams, 2), float("nan"), dtype=torch.float32, device="cuda")
    bf.gemm(torch.from_numpy(unit).cuda(), d_c, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(d_c.cpu().numpy().reshape(-1), np.asarray(orc.gemm(g, w, orc.expand(unit))).reshape(-1))
    bf.close()
