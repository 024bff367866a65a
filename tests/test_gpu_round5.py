"""GPU tests (-m gpu) of the round-5 surface; every call goes through the C-ABI of libdsabf.so."""
import os

import numpy as np
import pytest

from conftest import sweep

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
    # the counters beside the timings are those of THIS build (the key stored in the committed summary = the library's own build id
    # + instantiation + launch; tests/test_bench_cpu.py holds the committed summaries to the tree's build id) ...
    from dsabeamformer_amd import build
    import dsabeamformer_amd as bfm

    assert ("kernels=%s " % build.kernel_build_id()) in roof["pmc_key"] and bfm.load().bf_version().decode().endswith(build.kernel_build_id() + ")")
    if not roof["pmc_stale"]:
        assert roof["pmc_source"].startswith("profiles/r06_") and roof["traffic"] is not None and roof["pmc_not_used"] is None
        assert roof["bound_measured"] == "simd-issue" and 15 < roof["valu_per_mfma"] < 19 and 0.9 < roof["issue_occupancy"] < 1.1
        assert abs(roof["issue_model_cycles_per_mfma"] - (13 + 2.45 * roof["valu_per_mfma"])) < 1e-6 and 1.8 < roof["clock_ghz_under_load"] < 2.5
        assert {"valu_per_mfma", "issue_occupancy", "bound_measured", "clock_ghz_under_load"} <= set(roof["from_committed_profile"])
    else:   # ... or none at all: a kernel change without a profile refresh never pairs new timings with old counters
        assert roof["traffic"] is None and roof["mfma_busy_frac"] is None and roof["pmc_source"] is None and roof["from_committed_profile"] == []


# ---- f4 as a pipeline stage: DM-trial dedispersion of the detected STREAM ---------------------------------
def _pulse_delays(n_dm, n_f, d_max):
    """A fine, monotone ladder: delay[dm][f] grows with the trial and falls with f (channel 0 = the highest frequency)."""
    d = (np.arange(n_dm)[:, None] * np.linspace(d_max / max(n_dm - 1, 1), 0.0, n_f)[None, :]).astype(np.int32)
    return np.ascontiguousarray(d)


@pytest.mark.parametrize("shape", ["fine_ladder", "mixed_groups"])
def test_dm_stream_chunks_are_the_whole_series_call_bit_for_bit(torch, bfmod, orc, shape):
    """bf_dm_stream: a detected series pushed in pieces of ANY size (1 row ... max_rows_per_push, on alternating HIP streams),
    the last max_delay rows carried over on the device.  The chunks, joined along t, are bit-equal to orc.dedisperse_dm over
    the whole series -- with the shared-window kernel and with the per-thread-window kernel alone -- chunk k starts where chunk
    k - 1 ended, nothing is emitted before max_delay rows have been seen, and a dispersed pulse that straddles push boundaries
    comes out at its own trial and time.  Round 6: the same with the ZERO-COPY feed (bf_dm_stream_reserve: the producer writes the
    rows into the stage's own buffer, the push only launches) and with both feeds mixed in one stream."""
    import ctypes as C

    from dsabeamformer_amd import _lib, api

    hip = _lib._preload_hip_runtime()

    rng = np.random.default_rng(77)
    if shape == "fine_ladder":
        n_t, n_f, n_b, n_dm, max_rows = 210, 48, 256, 40, 32
        delays = _pulse_delays(n_dm, n_f, 45)
    else:                                   # group 0 fits a window, group 1 is far too coarse (per-thread kernel), group 2 fits
        n_t, n_f, n_b, n_dm, max_rows = 260, 32, 132, 96, 40
        step = np.concatenate([np.full(32, 0.4), np.full(32, 2.5), np.full(32, 0.4)])
        delays = np.ascontiguousarray((np.cumsum(step)[:, None] * np.linspace(1.0, 0.05, n_f)[None, :]).astype(np.int32))
    D = int(delays.max())
    series = (rng.random((n_t, n_f, n_b), dtype=np.float32) * 1e3).astype(np.float32)
    k_true, t0 = n_dm // 2, 70                                  # a pulse dispersed at trial k_true, arriving at t0 in channel 0
    for f in range(n_f):
        series[t0 + delays[k_true, f], f, :] += np.float32(5e5)
    want = orc.dedisperse_dm(series, delays, n_t - D)          # [n_dm][n_t - D][n_b]: ONE call over the whole series
    assert want[k_true, t0].min() > want[k_true, t0 + 3].max() * 10 and int(want[:, :, 0].max(axis=1).argmax()) == k_true
    bf = bfmod.Beamformer(bfmod.debug_config(n_beams=n_b, n_freq=n_f))
    d_series = torch.from_numpy(series).cuda()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    row_bytes = n_f * n_b * 4
    # ring: the stage's buffer mapped twice back to back (nothing ever moves; the ring here holds 141 / 173 rows: a window wraps
    # around its end every few pushes) -- or the linear buffer whose carry slides (what a device without VMM gets)
    for wide, feed, ring in ((1, "copy", 1), (0, "copy", 0), (1, "reserve", 1), (0, "mixed", 1), (1, "mixed", 0), (0, "reserve", 0)):
        bf.set_switch("dm_wide", wide)
        bf.set_switch("dm_ring", ring)
        dm = api.DmStream(bf, delays, n_f, max_rows)
        assert bf.counter("dm_ring_stages") == ring            # an MI355X really gives the stage the twice-mapped ring
        assert dm.max_delay == D
        host = torch.full((n_dm * max_rows * n_b,), float("nan"), dtype=torch.float32).pin_memory()
        parts, at, pushed, k = [], 0, 0, 0
        sizes = [max_rows, 1, 7, max_rows, 3, 19, 2, max_rows, max_rows - 1, 11]
        while pushed < n_t:
            n = min(sizes[k % len(sizes)], n_t - pushed)
            st = streams[k % 2]
            src = d_series.data_ptr() + pushed * row_bytes
            if feed == "reserve" or (feed == "mixed" and k % 3):
                # the producer (here: a device copy standing in for the beamformer / the gather) writes the rows where the stage
                # wants them, on the stream the reservation was made on; the push finds them in place
                dst = dm.reserve(n, st.cuda_stream)
                with pytest.raises(bfmod.DsabfError, match="previous reservation"):
                    dm.reserve(n, st.cuda_stream)                                   # one at a time
                with pytest.raises(bfmod.DsabfError, match="push exactly those"):
                    dm.push(src, n, host, st.cuda_stream)                           # ... and pushed as reserved
                assert hip.hipMemcpyAsync(C.c_void_p(dst), C.c_void_p(src), C.c_size_t(n * row_bytes), 3, C.c_void_p(st.cuda_stream)) == 0
                src = dst
            first, n_out = dm.push(src, n, host, st.cuda_stream)
            assert first == at and n_out == max(0, pushed + n - D) - max(0, pushed - D)
            st.synchronize()
            if n_out:
                parts.append(host[:n_dm * n_out * n_b].numpy().reshape(n_dm, n_out, n_b).copy())
            at += n_out
            pushed += n
            k += 1
        got = np.concatenate(parts, axis=1)
        assert got.shape == want.shape and at == n_t - D
        assert np.array_equal(got, want), (shape, wide, feed, ring)
        assert int(got[:, :, 0].max(axis=1).argmax()) == k_true and int(got[k_true, :, 0].argmax()) == t0
        dm.close()
    bf.set_switch("dm_wide", 1)
    bf.set_switch("dm_ring", 1)
    bf.close()


def _observation_with_a_pulse(bfmod, orc, host, tmp_path, monkeypatch, wide):
    import threading

    monkeypatch.setenv("DSABF_DM_WIDE", "1" if wide else "0")      # read once, at bf_create inside run_observation
    cfg = bfmod.production_config(n_freq=16, n_out_per_gemm=2)
    cfg.n_beams, cfg.n_gemms_per_block, cfg.n_streams = 128, 4, 4
    n_ipo = cfg.n_pol * cfg.n_avg
    n_time = cfg.n_out_per_gemm * n_ipo
    rows_per_block = cfg.n_gemms_per_block * cfg.n_out_per_gemm                      # 8 beam-blocks per PSRDADA block
    n_blocks, n_dm = 9, 24
    delays = _pulse_delays(n_dm, cfg.n_freq, 21)                                      # the carry spans almost three blocks
    D = int(delays.max())
    rng = np.random.default_rng(9)
    blocks = rng.integers(0, 256, size=(n_blocks, cfg.n_gemms_per_block, cfg.n_freq, n_time, cfg.n_ant), dtype=np.uint8)
    # a pulse from the boresight (every antenna the same full-scale sample, BOGUS_DATA 0x70 = 7 + 0j), dispersed at trial
    # k_true, arriving in channel 0 two rows before the end of block 2: it crosses the boundaries of blocks 2 | 3 | 4 | 5
    k_true, t0 = 17, 3 * rows_per_block - 2
    flat = blocks.reshape(n_blocks * cfg.n_gemms_per_block, cfg.n_freq, cfg.n_out_per_gemm, n_ipo, cfg.n_ant)
    for f in range(cfg.n_freq):
        unit, o = divmod(t0 + int(delays[k_true, f]), cfg.n_out_per_gemm)
        flat[unit, f, o] = 0x70
    assert (t0 + int(delays[k_true, -1])) // rows_per_block == 2 and (t0 + D * k_true // (n_dm - 1)) // rows_per_block >= 4
    name = "dsabf_gpu_dm_%d_%d" % (os.getpid(), wide)
    block_bytes = blocks[0].size
    ring = host.ShmRing(name, n_blocks=3, block_size=block_bytes, header="HDR_SIZE 4096\n")

    def writer():
        for b in blocks:
            ring.write(b.reshape(-1))
        ring.write(np.zeros(0, np.uint8))

    t = threading.Thread(target=writer)
    t.start()
    det, dmf = str(tmp_path / ("det%d.bin" % wide)), str(tmp_path / ("dm%d.bin" % wide))
    try:
        r = host.run_observation_shm(cfg, name, path=det, gpu=1, delays=delays, dm_path=dmf)
        t.join(timeout=30)
    finally:
        ring.detach()
        ring.unlink()
    T = n_blocks * rows_per_block
    assert r["gemms"] == n_blocks * cfg.n_gemms_per_block and r["dm_times"] == T - D
    g = orc.Geom(n_beams=cfg.n_beams, n_ant=64, n_freq=cfg.n_freq, n_avg=16, n_out_per_gemm=cfg.n_out_per_gemm)
    wts = orc.make_weights(g, orc.default_positions(64), orc.default_directions(cfg.n_beams), 1)
    series = np.concatenate([orc.beamform(g, wts, blocks[b]).reshape(rows_per_block, g.n_freq, g.n_beams) for b in range(n_blocks)])
    _, data = host.read_detected_file(det)
    assert np.array_equal(data.reshape(series.shape), series)                       # the detected stream itself (a1-a4)
    want = orc.dedisperse_dm(series, delays, T - D)
    hdr, got, chunks = host.read_dm_file(dmf)
    assert int(hdr["N_DM"]) == n_dm and int(hdr["MAX_DELAY"]) == D and int(hdr["N_FREQUENCIES"]) == cfg.n_freq
    # blocks 0 and 1 complete nothing (16 rows < 21), block 2 the first 3 times, every later block its own 8
    assert chunks == [(0, 3)] + [(3 + 8 * i, 8) for i in range(n_blocks - 3)]
    assert np.array_equal(got, want), wide
    b0 = int(want[k_true, t0].argmax())                                               # the beam nearest the boresight
    assert int(got[:, :, b0].max(axis=1).argmax()) == k_true and int(got[k_true, :, b0].argmax()) == t0
    return got


def test_observation_loop_with_the_dm_stage_streams_a_pulse_across_block_boundaries(bfmod, orc, tmp_path, monkeypatch):
    """VERDICT r04 item 2, the done-criterion: the production loop (run_observation: ring slots, queues, events) fed 9 blocks
    through the shared-memory ring, DM stage on (where the reference's loop collapses frequency, src/beamformer.cu:492-511).
    A pulse dispersed at one trial crosses three block boundaries; the largest delay is longer than two blocks.  The streamed
    [dm][t][b] -- the chunks as the loop delivered them -- is BIT-EQUAL to orc.dedisperse_dm over the whole detected series,
    with both kernel selections; the detected stream written beside it is the oracle's; the pulse peaks at its trial and time."""
    from dsabeamformer_amd import host

    a = _observation_with_a_pulse(bfmod, orc, host, tmp_path, monkeypatch, wide=True)
    b = _observation_with_a_pulse(bfmod, orc, host, tmp_path, monkeypatch, wide=False)
    assert np.array_equal(a, b)


def _beam_ladder(host, dm_max, n_cap, n_freq, tsamp_ms):
    """What `beam -M dm_max -N n_cap -T tsamp_ms` computes (csrc/beam_main.cpp): the notebook's ladder, evenly picked."""
    dms = host.dm_trials(dm_max=dm_max)
    if len(dms) > n_cap:
        dms = np.array([dms[int(i * (len(dms) - 1) / (n_cap - 1))] for i in range(n_cap)])
    freq = np.array([host.channel_frequency(0, c) for c in range(n_freq)], np.float32)
    return dms, host.dm_delays(dms, freq, float(freq[0]), tsamp_ms)


def test_beam_cli_dm_stage_single_gpu_and_two_loopback_ranks(orc, tmp_path):
    """`beam -j 27 -M 40 -N 6 -T 0.02 -W dm.bin` (25 burn-in reads + 2 analysed blocks of the production geometry) on one GPU,
    and the same sub-band as `-R 2` shard processes (loopback stand-in for RCCL p2p): the gather root dedisperses the GATHERED
    band.  Each file is bit-equal to orc.dedisperse_dm over the whole detected series the run produced, ascending f over all
    256 channels -- one GPU or two.  With -X the band goes to every rank and the TRIALS are split across the shards."""
    from test_gpu_multirank import FAKE, SUPPORT  # noqa: F401  (built by that module's fixture; build here if it has not run)
    import subprocess

    from dsabeamformer_amd import build, host

    import dsabeamformer_amd as bfm

    src = os.path.join(SUPPORT, "fake_rccl.cpp")
    if not os.path.exists(FAKE) or os.path.getmtime(FAKE) < os.path.getmtime(src):
        obj = os.path.join(SUPPORT, "fake_rccl.o")
        subprocess.check_call([build.HIPCC, "-O2", "-std=c++17", "-fPIC", "-c", src, "-o", obj])
        cxx = os.path.join(os.path.dirname(os.path.realpath(build.HIPCC)), "..", "lib", "llvm", "bin", "clang++")
        subprocess.check_call([cxx if os.path.exists(cxx) else "g++", "-shared", "-fPIC", "-o", FAKE, obj, "-lpthread", "-lrt"])
    n_an, tsamp = 2, 0.02                                   # analysed blocks; a sample time that makes DM 40 span ~45 rows
    dms, delays = _beam_ladder(host, 40.0, 6, 256, tsamp)
    D = int(delays.max())
    assert len(dms) == 6 and 20 < D < 256
    pos, dirs = host.default_positions(64), host.default_directions(256)
    common = ["-j", str(25 + n_an), "-D", "0", "-M", "40", "-N", "6", "-T", str(tsamp)]
    for world in (1, 2):
        cfg = bfm.production_config(n_freq=256 // world)
        n_time = cfg.n_out_per_gemm * cfg.n_pol * cfg.n_avg
        out = tmp_path / ("dm_w%d.bin" % world)
        if world == 1:
            r = subprocess.run([build.BEAM] + common + ["-W", str(out)], capture_output=True, text=True, timeout=900)
            assert r.returncode == 0, r.stdout + r.stderr
            outs = [r.stdout]
        else:
            # (the shards' powers travel by the STAGED transport here -- one message per sender + the device re-layout pass inside
            #  run_observation; the in-place transport: tests/test_gpu_multirank.py::test_beam_sharded_over_two_ranks_...) -- and
            #  the root keeps the detected stream as well (-w): both sinks behind one loop
            env = dict(os.environ, DSABF_RCCL_LIB=FAKE, DSABF_GATHER_STAGED="1")
            det = tmp_path / "det_w2.bin"
            cmd = lambda rk: [build.BEAM] + common + ["-R", "2", "-r", str(rk), "-I", str(tmp_path / "id")] + (["-W", str(out), "-w", str(det)] if rk == 0 else [])  # noqa: E731
            procs = [subprocess.Popen(cmd(rk), env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for rk in (0, 1)]
            outs = [p.communicate(timeout=900)[0] for p in procs]
            assert all(p.returncode == 0 for p in procs), "\n".join(outs)
        assert "DM stage: 6 trials" in outs[0] and ("largest delay %d samples" % D) in outs[0]
        T = n_an * cfg.n_gemms_per_block * cfg.n_out_per_gemm
        assert ("Wrote %d dedispersed samples x 6 trials" % (T - D)) in outs[0]
        # the series the run detected: every shard reads the same junk bytes with ITS geometry and its own channels' weights
        ring = host.junk_bytes(cfg.n_ant * cfg.n_freq * n_time * cfg.n_gemms_per_block, 4, 0xD5A, cfg).reshape(
            4, cfg.n_gemms_per_block, cfg.n_freq, n_time, cfg.n_ant)
        g = orc.Geom(n_beams=256, n_ant=64, n_freq=cfg.n_freq, n_avg=16, n_out_per_gemm=8)
        band = []
        for rk in range(world):
            w = host.make_weights(pos, dirs, cfg.n_freq, chan0=cfg.n_freq * rk, gpu=0)
            band.append(np.concatenate([orc.beamform(g, w, ring[(25 + b) % 4]).reshape(-1, cfg.n_freq, 256) for b in range(n_an)]))
        series = np.concatenate(band, axis=1)                                         # [T][256 channels][256 beams]
        assert series.shape == (T, 256, 256)
        hdr, got, chunks = host.read_dm_file(str(out))
        assert int(hdr["N_FREQUENCIES"]) == 256 and sum(n for _, n in chunks) == T - D
        assert np.array_equal(got, orc.dedisperse_dm(series, delays, T - D)), world
        if world == 2:
            raw = np.fromfile(det, np.float32, offset=4096).reshape(T, 256, 256)      # [gemm][o] = time rows of the whole band
            assert np.array_equal(raw, series)
            # ---- -X: the same two shards, the powers gathered to EVERY rank (one all-gather), the ladder split: rank r dedisperses
            # trials [3 r, 3 r + 3) of the six and writes dm_x.bin.<r>; each file is the oracle's over ITS trials (a rank's window is
            # its own trials' largest delay, so the low-DM rank completes more output times)
            outx = tmp_path / "dm_x.bin"
            cmdx = lambda rk: [build.BEAM] + common + ["-R", "2", "-r", str(rk), "-I", str(tmp_path / "idx"), "-X", "-W", str(outx)]  # noqa: E731
            procs = [subprocess.Popen(cmdx(rk), env=dict(os.environ, DSABF_RCCL_LIB=FAKE), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
                     for rk in (0, 1)]
            outs = [p.communicate(timeout=900)[0] for p in procs]
            assert all(p.returncode == 0 for p in procs), "\n".join(outs)
            for rk in (0, 1):
                assert ("Shard %d dedisperses trials %d .. %d" % (rk, 3 * rk, 3 * rk + 2)) in outs[rk]
                mine = np.ascontiguousarray(delays[3 * rk:3 * rk + 3])
                d_r = int(mine.max())
                hdr, got, chunks = host.read_dm_file(str(outx) + ".%d" % rk)
                assert int(hdr["N_DM"]) == 3 and int(hdr["DM_FIRST_TRIAL"]) == 3 * rk and int(hdr["MAX_DELAY"]) == d_r
                assert np.array_equal(got, orc.dedisperse_dm(series, mine, T - d_r)), rk
            assert int(delays[:3].max()) < D                       # (rank 0's window really is the shorter one)


def test_beam_reads_a_psrdada_style_ring_through_the_dada_adapter(orc, tmp_path):
    """`beam -k baXX -w det.bin` built with -DDSABF_WITH_PSRDADA: the reference's own command line (src/beamformer.cu:66-75,132)
    taking the dada_block_source branch -- connect, lock_read, every ring block page-locked with hipHostRegister (the reference's
    dada_cuda_dbregister), 25 burn-in reads, the observation loop fed from the ring, the short block that ends it -- against
    tests/support/fake_psrdada (a functional stand-in for the libpsrdada calls; not PSRDADA).  Production geometry (128 MiB
    blocks); sampled gemm-units of the detected stream against the oracle."""
    import ctypes as C
    import subprocess
    import threading

    from test_host_cpu import _build_fake_psrdada

    import dsabeamformer_amd as bfm
    from dsabeamformer_amd import host

    lib_path, link = _build_fake_psrdada(tmp_path)
    from conftest import ROOT

    exe = link(str(tmp_path / "beam_dada"), os.path.join(ROOT, "dsabeamformer_amd", "csrc", "beam_main.cpp"))
    fake = C.CDLL(lib_path)
    fake.fakedada_create.restype = C.c_void_p
    fake.fakedada_create.argtypes = [C.c_uint, C.c_uint64, C.c_uint64, C.c_char_p]
    fake.fakedada_write.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    fake.fakedada_destroy.argtypes = [C.c_void_p]
    cfg = bfm.production_config()
    n_time = cfg.n_out_per_gemm * cfg.n_pol * cfg.n_avg
    block_bytes = cfg.n_gemms_per_block * cfg.n_freq * n_time * cfg.n_ant
    key, n_an = 0xBA00 + os.getpid() % 200, 3
    w = fake.fakedada_create(key, 4, block_bytes, b"HDR_SIZE 4096\n")
    assert w
    rng = np.random.default_rng(21)
    distinct = [rng.integers(0, 256, size=block_bytes, dtype=np.uint8) for _ in range(3)]
    rc = []

    def writer():
        for i in range(25 + n_an):                 # BURNIN reads first (src/beamformer.cu:348-355), then the analysed blocks
            rc.append(fake.fakedada_write(w, distinct[i % 3].ctypes.data_as(C.c_void_p), block_bytes))
        rc.append(fake.fakedada_write(w, None, 0))

    t = threading.Thread(target=writer)
    t.start()
    det = tmp_path / "det.bin"
    try:
        r = subprocess.run([exe, "-k", "%x" % key, "-w", str(det)], capture_output=True, text=True, timeout=900)
        t.join(timeout=60)
    finally:
        fake.fakedada_destroy(w)
    assert r.returncode == 0, r.stdout + r.stderr
    assert rc == [0] * (25 + n_an + 1)
    assert ("block size is: %d" % block_bytes) in r.stdout and "Burning IN" in r.stdout and "could not pin" not in r.stdout
    assert ("Wrote %d gemm-units" % (n_an * cfg.n_gemms_per_block)) in r.stdout
    _, data = host.read_detected_file(str(det))
    g = orc.Geom(n_beams=256, n_ant=64, n_freq=256, n_avg=16, n_out_per_gemm=8)
    wts = orc.make_weights(g, orc.default_positions(64), orc.default_directions(256), 0)
    for gemm in (0, 31, 45, n_an * 32 - 1):
        blk, ts = divmod(gemm, 32)
        unit = distinct[(25 + blk) % 3].reshape(32, 256, n_time, 64)[ts][None]
        assert np.array_equal(data[gemm], orc.beamform(g, wts, unit)[0]), gemm


def test_dm_stream_argument_errors(torch, bfmod):
    """bf_dm_stream_*: return codes + bf_last_error instead of undefined behaviour -- negative delays (a streamed dedispersion
    cannot look back before the first row), sizes out of range, NULL arguments."""
    import ctypes as C

    from dsabeamformer_amd import api
    from dsabeamformer_amd._lib import DsabfError, load

    bf = bfmod.Beamformer(bfmod.debug_config(n_beams=64, n_freq=8))
    good = np.zeros((3, 8), np.int32)
    for delays, n_f, rows, text in ((np.full((3, 8), -1, np.int32), 8, 4, "delays >= 0"), (good, 8, 0, "max_rows_per_push")):
        with pytest.raises(DsabfError) as e:
            api.DmStream(bf, delays, n_f, rows)
        assert text in str(e.value), str(e.value)
    lib = load()
    s = C.c_void_p()
    assert lib.bf_dm_stream_create(None, good.ctypes.data_as(C.c_void_p), 3, 8, 4, C.byref(s)) == -1 and not s.value
    dm = api.DmStream(bf, good, 8, 4)
    d_rows = torch.zeros(8 * 8 * 64, device="cuda")
    for n in (0, 5):                                            # more rows than max_rows_per_push
        with pytest.raises(DsabfError) as e:
            dm.push(d_rows, n)
        assert "n_rows must be 1 .. 4" in str(e.value)
    assert lib.bf_dm_stream_push(dm._s, None, 1, None, None, None, None) == -1
    first, n_out = dm.push(d_rows, 4)                           # all delays 0: every row is complete at once
    assert (first, n_out) == (0, 4) and dm.max_delay == 0 and dm.output_device() != 0
    torch.cuda.synchronize()
    dm.close()
    assert lib.bf_dm_stream_destroy(None) == 0 and lib.bf_dm_stream_max_delay(None) == -1
    # a handle that is destroyed first takes the stage's device memory with it: the stage answers BF_ERR_STATE, and can still be destroyed
    orphan = api.DmStream(bf, good, 8, 4)
    bf.close()
    with pytest.raises(DsabfError) as e:
        orphan.push(d_rows, 4)
    assert e.value.code == -4 and "has been destroyed" in str(e.value)
    orphan.close()


@pytest.mark.parametrize("seed", sweep(range(12), [4]))
def test_production_loop_with_the_dm_stage_under_random_launch_patterns(bfmod, orc, tmp_path, monkeypatch, seed):
    """run_observation with the DM stage AND the detected-stream sink under random block sizes, queue counts, sub-block launches
    (every launch is a push, on its own queue: the stream orders them), ladders whose window is shorter or much longer than a
    block, both kernel selections: the detected stream and the streamed [dm][t][b] are the oracle's over the whole observation."""
    from dsabeamformer_amd import host

    rng = np.random.default_rng(5200 + seed)
    n_u = int(rng.choice([4, 8, 16]))
    n_st = int(rng.choice([s for s in (1, 2, 4, 8) if s <= n_u]))
    n_blocks, ring_blocks = int(rng.integers(3, 9)), int(rng.integers(2, 5))
    if rng.integers(2):
        monkeypatch.setenv("DSABF_UNITS_PER_LAUNCH", str(int(rng.choice([1, 2, n_u // 2]))))
    monkeypatch.setenv("DSABF_DM_WIDE", str(int(rng.integers(2))))
    n_freq, n_out = int(rng.choice([4, 8, 12])), int(rng.choice([2, 4]))
    cfg = bfmod.production_config(n_freq=n_freq, n_avg=int(rng.choice([16, 8])), n_out_per_gemm=n_out)
    cfg.n_beams, cfg.n_gemms_per_block, cfg.n_streams = 64, n_u, n_st
    T = n_blocks * n_u * n_out
    n_dm = int(rng.integers(1, 50))
    d_max = int(rng.integers(0, max(1, min(T - 2, 3 * n_u * n_out))))                  # up to three blocks of window
    if rng.integers(3) == 0:
        delays = rng.integers(0, d_max + 1, size=(n_dm, n_freq)).astype(np.int32)       # no ladder at all (per-thread kernel)
    else:
        delays = _pulse_delays(n_dm, n_freq, d_max)
    D = int(delays.max())
    det, dmf = str(tmp_path / "det.bin"), str(tmp_path / "dm.bin")
    r = host.run_observation_junk_dm(cfg, n_blocks, delays, dmf, detected_path=det, ring_blocks=ring_blocks, seed=2000 + seed, gpu=1)
    assert r["dm_times"] == T - D, (seed, T, D)
    g = orc.Geom(n_beams=64, n_ant=64, n_freq=n_freq, n_avg=cfg.n_avg, n_out_per_gemm=n_out)
    w = orc.make_weights(g, orc.default_positions(64), orc.default_directions(64), 1)
    series = np.concatenate([orc.beamform(g, w, r["ring"][b % ring_blocks]).reshape(n_u * n_out, n_freq, 64) for b in range(n_blocks)])
    _, data = host.read_detected_file(det)
    assert np.array_equal(data.reshape(series.shape), series), seed
    hdr, got, chunks = host.read_dm_file(dmf)
    assert sum(n for _, n in chunks) == T - D and int(hdr["MAX_DELAY"]) == D
    assert np.array_equal(got, orc.dedisperse_dm(series, delays, T - D)), (seed, n_u, n_st, n_dm, D)


def test_dm_chunks_to_another_process_through_a_shared_memory_ring(bfmod, orc):
    """dsabf::dm_ring_sink (`beam -Q ring`): the DM stage's chunks handed to a consumer -- the downstream search -- through a
    shared-memory ring the loop creates, one block per chunk (record header + [dm][t][beam], whole blocks; a short block ends the
    data).  A 2-block ring for 7 chunks and a consumer that dawdles: the loop waits, nothing is dropped; joined along t the
    chunks are the oracle's over the whole detected series."""
    import threading
    import time

    from dsabeamformer_amd import host

    cfg = bfmod.production_config(n_freq=8, n_out_per_gemm=2)
    cfg.n_beams, cfg.n_gemms_per_block, cfg.n_streams = 64, 4, 2
    n_blocks, ring_blocks, n_dm = 8, 3, 10
    rows = cfg.n_gemms_per_block * cfg.n_out_per_gemm
    delays = _pulse_delays(n_dm, cfg.n_freq, 11)                    # longer than a block of 8 rows: the first block completes nothing
    D, T = int(delays.max()), n_blocks * rows
    name = "dsabf_dmout_%d" % os.getpid()
    got, err = [], []

    def consumer():
        try:
            ring = host.ShmRing(name, timeout_ms=20000)
            assert ring.block_size == 32 + n_dm * rows * 64 * 4 and "dedispersed_power" in ring.header
            assert "MAX_DELAY %d" % D in ring.header and "MAX_TIMES_PER_CHUNK %d" % rows in ring.header
            while True:
                data, bid = ring.read()
                if data.size < ring.block_size:
                    break
                first_t = int(data[:8].view("<u8")[0])
                n_t, nd, nb = (int(v) for v in data[8:20].view("<u4"))
                assert (nd, nb) == (n_dm, 64) and first_t == sum(p.shape[1] for p in got)
                got.append(data[32:32 + 4 * nd * n_t * nb].view(np.float32).reshape(nd, n_t, nb).copy())
                if bid == 2:
                    time.sleep(0.3)    # let the 2-block ring fill: the loop must wait for the consumer, not drop a chunk
            ring.detach()
        except Exception as e:  # noqa: BLE001
            err.append(e)

    t = threading.Thread(target=consumer)
    t.start()
    r = host.run_observation_junk_dm(cfg, n_blocks, delays, "ring:%s:2" % name, ring_blocks=ring_blocks, seed=77, gpu=1)
    t.join(timeout=60)
    assert not err, err
    assert r["dm_times"] == T - D and not os.path.exists("/dev/shm/" + name)      # drained, then removed by the sink
    g = orc.Geom(n_beams=64, n_ant=64, n_freq=8, n_avg=16, n_out_per_gemm=2)
    w = orc.make_weights(g, orc.default_positions(64), orc.default_directions(64), 1)
    series = np.concatenate([orc.beamform(g, w, r["ring"][b % ring_blocks]).reshape(rows, 8, 64) for b in range(n_blocks)])
    assert [p.shape[1] for p in got] == [2 * rows - D] + [rows] * (n_blocks - 2)
    assert np.array_equal(np.concatenate(got, axis=1), orc.dedisperse_dm(series, delays, T - D))


# ---- run-time accumulation windows (src/beamformer.hh:55-60: any N_AVERAGING): the stream length of the launch -------------------
def _small_cfg(bfmod, g, **over):
    cfg = bfmod.debug_config(n_beams=g.n_beams, n_ant=g.n_ant, n_freq=g.n_freq, n_pol=g.n_pol, n_avg=g.n_avg, n_out_per_gemm=g.n_out_per_gemm)
    for k, v in over.items():
        setattr(cfg, k, v)
    return cfg


@pytest.mark.parametrize("n_avg,n_ant", [(3, 64), (5, 64), (6, 100), (9, 64), (12, 64), (12, 100), (20, 64), (48, 64), (37, 36), (12, 132), (5, 192),
                                         (20, 260)])
def test_run_time_window_streams_of_any_length_bit_exact(torch, bfmod, orc, n_avg, n_ant):
    """A lane group's stream of a run-time-window launch is kout whole windows over ceil(kout L / 32) chunks; the library picks the
    kout with the least padding that still fills the chip (a big launch of L = 24 runs 4 windows over 3 chunks instead of 1 window in
    1 chunk, a quarter of it padding).  Every kout gives the oracle's bits -- windows that cross chunk boundaries inside a stream,
    ragged last streams, several chunk groups per workgroup, general and conjugate-pair weights, both bit-exact readings -- and
    the library's own choice for a launch that fills the chip is one without padding."""
    g = orc.Geom(n_beams=64, n_ant=n_ant, n_freq=2, n_avg=n_avg, n_out_per_gemm=5)
    L = g.n_ipo
    rng = np.random.default_rng(31 * n_avg + n_ant)
    w = rng.integers(-127, 128, size=(g.n_freq, g.n_ant, g.n_beams, 2), dtype=np.int8)
    if n_avg % 2:
        w[:, :, 32:, 0] = w[:, :, :32, 0][:, :, ::-1]
        w[:, :, 32:, 1] = -w[:, :, :32, 1][:, :, ::-1]
    n_units = 41                                         # 205 windows: several groups of 4 streams even at 16 windows per stream
    packed = rng.integers(0, 256, size=(n_units, g.n_freq, g.n_time, g.n_ant), dtype=np.uint8)
    import math
    k_full = 32 // math.gcd(L, 32)                      # the shortest stream that is whole chunks
    for mode, contract in ((0, orc.CONTRACT_NONE), (2, orc.CONTRACT_NVCC)):
        with orc.detect_contract(contract):
            want = orc.beamform(g, w, packed).reshape(-1)
        bf = bfmod.Beamformer(_small_cfg(bfmod, g, detect_mode=mode))
        bf.set_weights(w)
        name = bf.kernel_info(n_units)["kernel"]           # up to 128 antennas fused16_kernel's run-time-window class, beyond fusedg_kernel
        assert ("NIPO=%d(run-time)" % L in name) if n_ant <= 128 else ("fusedg_kernel" in name and "NIPO=%d" % L in name), name
        for kout in sorted({0, 1, 2, 3, 5, k_full, min(16, k_full + 1), 16}):
            bf.set_switch("rtw_kout", kout)
            for tsplit in (0, 2):
                bf.set_switch("tsplit", tsplit)
                d_in = torch.from_numpy(packed).cuda()
                d_out = torch.full((want.size,), float("nan"), dtype=torch.float32, device="cuda")
                bf.beamform(d_in, n_units, d_out, torch.cuda.current_stream().cuda_stream)
                torch.cuda.synchronize()
                assert np.array_equal(d_out.cpu().numpy(), want), (mode, kout, tsplit)
        with pytest.raises(bfmod.DsabfError):
            bf.set_switch("rtw_kout", 33)
        bf.close()


@pytest.mark.parametrize("n_avg", [5, 7, 19, 33])
def test_run_time_window_streams_of_odd_windows_one_polarisation(torch, bfmod, orc, n_avg):
    """N_POL = 1 (src/beamformer.hh:50) makes the window odd: whole-chunk streams are then 32 windows long.  Every stream length the
    library may pick (1 ... 32) gives the oracle's bits."""
    g = orc.Geom(n_beams=64, n_ant=64, n_freq=2, n_pol=1, n_avg=n_avg, n_out_per_gemm=9)
    rng = np.random.default_rng(n_avg)
    w = rng.integers(-127, 128, size=(g.n_freq, g.n_ant, g.n_beams, 2), dtype=np.int8)
    n_units = 31
    packed = rng.integers(0, 256, size=(n_units, g.n_freq, g.n_time, g.n_ant), dtype=np.uint8)
    want = orc.beamform(g, w, packed).reshape(-1)
    bf = bfmod.Beamformer(_small_cfg(bfmod, g))
    bf.set_weights(w)
    assert "NIPO=%d(run-time)" % n_avg in bf.kernel_info(n_units)["kernel"]
    for kout in (0, 1, 6, 17, 25, 31, 32):
        bf.set_switch("rtw_kout", kout)
        d_out = torch.full((want.size,), float("nan"), dtype=torch.float32, device="cuda")
        bf.beamform(torch.from_numpy(packed).cuda(), n_units, d_out, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert np.array_equal(d_out.cpu().numpy(), want), kout
    bf.close()


@pytest.mark.parametrize("n_avg,kout,rows", [(12, 4, 12288), (20, 4, 20480), (9, 3, 11008)])
def test_run_time_window_launch_that_fills_the_chip_takes_whole_chunk_streams(torch, bfmod, orc, n_avg, kout, rows):
    """The library's own choice at a size where it matters (64 channels x 512 windows): bf_rtw_plan says kout windows per stream --
    L = 24 and 40: whole chunks, not one padding row; L = 18: 16 windows per stream would leave too few groups for the chip, 3 (54
    of 64 rows) it is -- and the launch gives the oracle's bits over the whole output."""
    import ctypes as C

    from dsabeamformer_amd._lib import load

    g = orc.Geom(n_beams=64, n_ant=64, n_freq=64, n_avg=n_avg, n_out_per_gemm=16)
    cfg = _small_cfg(bfmod, g)
    n_units = 32
    k, chunks = C.c_int(), C.c_int()
    bf = bfmod.Beamformer(cfg)
    assert load().bf_rtw_plan(C.byref(cfg), n_units, 256, C.byref(k), C.byref(chunks)) == 0
    assert k.value == kout and chunks.value * 128 == rows >= n_units * g.n_time, (k.value, chunks.value)
    rng = np.random.default_rng(n_avg)
    w = rng.integers(-127, 128, size=(g.n_freq, g.n_ant, g.n_beams, 2), dtype=np.int8)
    bf.set_weights(w)
    packed = rng.integers(0, 256, size=(n_units, g.n_freq, g.n_time, g.n_ant), dtype=np.uint8)
    want = orc.beamform(g, w, packed).reshape(-1)
    d_out = torch.full((want.size,), float("nan"), dtype=torch.float32, device="cuda")
    bf.beamform(torch.from_numpy(packed).cuda(), n_units, d_out, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(d_out.cpu().numpy(), want)
    bf.close()
