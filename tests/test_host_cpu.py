"""CPU tests of the C++ host mirror (dsabeamformer_amd/csrc/bf_geometry.cpp, bf_generator.cpp, bf_scheduler.cpp, bf_sinks.cpp) -- the product's own weight generator,
config readers, data.py writer, test_data_generator and observation_loop_state -- against the oracle and the
reference's documented scheduler rules (README.md:122-140, src/observation_loop.hh)."""
import json
import os

import numpy as np
import pytest

from fake_events import make_obs

from conftest import CFG, GOLDEN, ROOT


@pytest.fixture(scope="module")
def host():
    from dsabeamformer_amd import build as b

    b.build()
    from dsabeamformer_amd import host as h

    return h


@pytest.fixture(scope="module")
def bfm():
    import dsabeamformer_amd as m

    return m


def test_defaults_and_frequency_table_match_oracle(host, orc):
    assert np.array_equal(host.default_positions(64), orc.default_positions(64))
    assert np.array_equal(host.default_directions(256), orc.default_directions(256))
    assert np.array_equal(host.default_positions(100), orc.default_positions(100))
    for gpu in range(8):
        for chan in (0, 1, 17, 255, 1023):
            assert host.channel_frequency(gpu, chan) == orc.freq_weights(gpu, chan)
            assert host.channel_frequency(gpu, chan, generator_variant=True) == orc.freq_generator(gpu, chan)


def test_config_readers_match_oracle(host, orc):
    for name, n in (("linear_positions.txt", 64), ("grid_positions.txt", 64), ("random_positions.txt", 64),
                    ("grid_positions.txt", 100)):
        p = os.path.join(CFG, name)
        assert np.array_equal(host.read_positions(p, n), orc.read_positions(p, n))
    for name, n in (("linear_directions.txt", 256), ("grid_beam_directions.txt", 256), ("grid_beam_directions.txt", 512),
                    ("linear_source_directions_1024.txt", None), ("grid_source_directions_3721.txt", None)):
        p = os.path.join(CFG, name)
        assert np.array_equal(host.read_directions(p, n), orc.read_directions(p, n))
    with pytest.raises(Exception):
        host.read_positions(os.path.join(CFG, "does_not_exist.txt"), 64)


def test_weights_bit_exact_vs_oracle_and_reference_hash(host, orc, linear_inputs, linear_weights):
    meta = json.load(open(os.path.join(GOLDEN, "golden.json")))
    pos, dirs, _ = linear_inputs
    w = host.make_weights(pos, dirs, 256, 0, 0)
    assert np.array_equal(w, linear_weights)
    assert "%016x" % orc.fnv1a64(w) == meta["weights_fnv1a64"]
    assert np.array_equal(host.make_weights_default(), linear_weights)
    # a frequency shard (rank 3 of 8: channels 96..127) is the matching slice
    assert np.array_equal(host.make_weights(pos, dirs, 32, 96, 0), linear_weights[96:128])
    # other sub-band and a 2-D array
    gpos = orc.read_positions(os.path.join(CFG, "grid_positions.txt"), 64)
    gdir = orc.read_directions(os.path.join(CFG, "grid_beam_directions.txt"), 256)
    g = orc.Geom(n_freq=6)
    assert np.array_equal(host.make_weights(gpos, gdir, 6, 0, 5), orc.make_weights(g, gpos, gdir, 5))


def test_python_file_writer(host, orc, tmp_path):
    x = np.array([[1.03932e8, 3003815.0, 1711951.38], [0.5, 1e-7, 12345678.0]], np.float32)
    p1, p2 = str(tmp_path / "a.py"), str(tmp_path / "b.py")
    host.write_python_file(x, p1)
    orc.write_python_file(x, p2)
    assert open(p1).read() == open(p2).read() == "A = [[1.03932e+08,3.00382e+06,1.71195e+06],\n[0.5,1e-07,1.23457e+07]]\n"
    gold = np.load(os.path.join(GOLDEN, "linear_debug.npz"))["dedispersed"][:16]
    host.write_python_file(gold, p1)
    orc.write_python_file(gold, p2)
    assert open(p1).read() == open(p2).read()


def test_generator_batch_bit_exact(host, orc, bfm, linear_inputs):
    meta = json.load(open(os.path.join(GOLDEN, "golden.json")))
    pos, _, _ = linear_inputs
    gen = host.TestDataGenerator(bfm.debug_config(), pin=False)
    assert gen.size() == 268435456
    assert gen.get_n_pt_sources() == 1024
    data = gen.data()
    assert data[0] == 0x70 and data[-1] == 0x70 and data[12345678] == 0x70  # BOGUS_DATA memset
    gen.read_in_source_directions(os.path.join(CFG, "linear_source_directions_1024.txt"))
    gen.generate_test_data(pos, 0)
    assert "%016x" % orc.fnv1a64(data) == meta["batch_fnv1a64"]
    gen.close()


def test_generator_batches_gating_and_overrun(host, orc, bfm, linear_inputs):
    pos, _, src = linear_inputs
    cfg = bfm.debug_config(n_freq=4)
    g = orc.Geom(n_freq=4)
    gen = host.TestDataGenerator(cfg, n_sources_per_batch=64, pin=False)
    # without a catalogue: nothing to generate, data is "ready" for the first batch only (hh:98-108)
    assert not gen.check_need_to_generate_more_input_data(0)
    assert gen.check_data_ready_for_transfer(0) and gen.check_data_ready_for_transfer(1)
    assert not gen.check_data_ready_for_transfer(2)  # 64 sources / 32 gemms per block = 2 blocks
    gen.close()
    gen = host.TestDataGenerator(cfg, n_sources_per_batch=64, pin=False)
    gen.set_source_directions(src[:100])
    assert gen.get_n_pt_sources() == 100
    assert gen.check_need_to_generate_more_input_data(0) and not gen.check_data_ready_for_transfer(0)
    gen.generate_test_data(pos, 0)
    assert np.array_equal(gen.data(), orc.generate_test_data(g, pos, src[:100], 0, 0, 64).ravel())
    assert not gen.check_need_to_generate_more_input_data(1) and gen.check_need_to_generate_more_input_data(2)
    assert gen.check_data_ready_for_transfer(1) and not gen.check_data_ready_for_transfer(2)
    gen.generate_test_data(pos, 0)  # second batch: sources 64..99 then zero bytes (hh:84-86)
    want = orc.generate_test_data(g, pos, src[:100], 0, 1, 64)
    assert np.array_equal(gen.data(), want.ravel())
    assert want[36:].max() == 0 and want[:36].max() > 0
    gen.close()


def _simulate(obs, gen_blocks, per_block=32, n_streams=8, transfer_lag=1, analysis_lag=2, max_iter=100000, rng=None):
    """The reference's DEBUG while-loop (src/beamformer.cu:364-534) with fake device completion.
    rng: every event gets its own random lag of 1 ... transfer_lag / analysis_lag iterations (completion stays in order)."""
    time_slice = list(range(n_streams))
    log = []
    pend_t, pend_a = [], []
    it = 0
    while not obs.check_observations_complete():
        it += 1
        assert it < max_iter, "scheduler did not terminate"
        c = obs.counters()
        # invariants (README.md:130-134)
        assert c["A"] <= c["AQ"] <= c["T"] <= c["TQ"]
        assert c["TQ"] - c["A"] <= 4 and c["TQ"] - c["T"] <= 2
        if obs.check_ready_for_transfer():
            if c["TQ"] < gen_blocks:
                log.append(("H2D", obs.get_next_gpu_transfer_block(), c["TQ"]))
                assert obs.get_next_gpu_transfer_block() == c["TQ"] % 8
                obs.generate_transfer_event()
                pend_t.append(it if rng is None else it + transfer_lag - int(rng.integers(1, transfer_lag + 1)))
            obs.check_transfers_complete()
        # device progress: a transfer completes `transfer_lag` iterations after it was queued
        done_t = 0
        while done_t < len(pend_t) and it - pend_t[done_t] >= transfer_lag:   # (in order: a stream's events complete FIFO)
            done_t += 1
        obs.fake_complete(done_t, 0)
        pend_t = pend_t[done_t:]
        obs.check_transfer_events()
        if obs.check_ready_for_analysis():
            blk = obs.get_next_gpu_analysis_block()
            for part in range(per_block // n_streams):
                for st in range(n_streams):
                    gemm = obs.get_current_analysis_gemm(time_slice[st])
                    log.append(("GEMM", blk, time_slice[st], gemm, obs.check_ready_for_dh2_transfer(time_slice[st])))
                    time_slice[st] += n_streams
                    if time_slice[st] >= per_block:
                        time_slice[st] -= per_block
            obs.generate_analysis_event()
            pend_a.append(it if rng is None else it + analysis_lag - int(rng.integers(1, analysis_lag + 1)))
        done_a = 0
        while done_a < len(pend_a) and it - pend_a[done_a] >= analysis_lag:
            done_a += 1
        obs.fake_complete(0, done_a)
        pend_a = pend_a[done_a:]
        obs.check_analysis_events()
    return log


@pytest.mark.parametrize("seed", list(range(6)))
def test_observation_loop_scheduler_under_random_completion_lags(host, bfm, seed):
    """The same loop with every transfer / analysis event completing after its own random lag: same order of work, same
    invariants (checked inside _simulate), every block once."""
    rng = np.random.default_rng(60 + seed)
    n_src = int(rng.integers(1, 700))
    cfg = bfm.debug_config()
    obs = make_obs(host, cfg, debug=True)
    obs.set_n_pt_sources(n_src)
    log = _simulate(obs, gen_blocks=10 ** 9, transfer_lag=int(rng.integers(1, 7)), analysis_lag=int(rng.integers(1, 9)), rng=rng)
    n_blocks = -(-n_src // 32)
    assert obs.counters() == {"A": n_blocks, "AQ": n_blocks, "T": n_blocks, "TQ": n_blocks}
    gemms = [e for e in log if e[0] == "GEMM"]
    assert [e[3] for e in gemms] == [b * 32 + p * 8 + s for b in range(n_blocks) for p in range(4) for s in range(8)]
    assert [e[3] for e in gemms if e[4]] == list(range(n_src))
    assert [e[2] for e in log if e[0] == "H2D"] == list(range(n_blocks))
    obs.close()


@pytest.mark.parametrize("n_src,tl,al", [(1024, 1, 2), (1024, 3, 1), (100, 1, 5), (33, 2, 2)])
def test_observation_loop_scheduler(host, bfm, n_src, tl, al):
    cfg = bfm.debug_config()
    obs = make_obs(host, cfg, debug=True)
    obs.set_n_pt_sources(n_src)
    assert obs.describe() == "A: 0, AQ: 0, T: 0, TQ: 0\ncurrent_gemm: 0, transfers_complete: 0"
    assert obs.check_ready_for_transfer() and not obs.check_ready_for_analysis()
    log = _simulate(obs, gen_blocks=10 ** 9, transfer_lag=tl, analysis_lag=al)
    n_blocks = -(-n_src // 32)
    c = obs.counters()
    assert c == {"A": n_blocks, "AQ": n_blocks, "T": n_blocks, "TQ": n_blocks}
    gemms = [e for e in log if e[0] == "GEMM"]
    # every gemm-unit of every block exactly once, in the reference's (part, stream) order, ring slot = block % 8
    assert [e[3] for e in gemms] == [b * 32 + p * 8 + s for b in range(n_blocks) for p in range(4) for s in range(8)]
    assert all(e[1] == (e[3] // 32) % 8 for e in gemms)
    # dedisperse/D2H only for gemm < n_pt_sources (src/beamformer.cu:492)
    assert [e[3] for e in gemms if e[4]] == list(range(n_src))
    assert [e[2] for e in log if e[0] == "H2D"] == list(range(n_blocks))
    assert "transfers_complete: 1" in obs.describe()
    obs.close()


def test_observation_loop_backpressure_rules(host, bfm):
    obs = make_obs(host, bfm.debug_config(), debug=False)
    # transfer separation: at most 2 un-transferred blocks in flight
    obs.generate_transfer_event()
    assert obs.check_ready_for_transfer()
    obs.generate_transfer_event()
    assert not obs.check_ready_for_transfer()
    obs.check_transfer_events()
    assert obs.counters()["T"] == 0 and not obs.check_ready_for_analysis()
    obs.fake_complete(1, 0)
    obs.check_transfer_events()
    assert obs.counters()["T"] == 1 and obs.check_ready_for_analysis() and obs.check_ready_for_transfer()
    # events complete in order only: a later finished event does not overtake an earlier pending one
    obs.fake_complete(1, 0)
    obs.check_transfer_events()
    for _ in range(2):
        obs.generate_transfer_event()
        obs.fake_complete(1, 0)
        obs.check_transfer_events()
    assert obs.counters()["TQ"] == 4 and obs.counters()["T"] == 4
    assert not obs.check_ready_for_transfer()  # total separation: TQ - A < 4 violated
    obs.generate_analysis_event()
    obs.fake_complete(0, 1)
    obs.check_analysis_events()
    assert obs.counters()["A"] == 1 and obs.check_ready_for_transfer()
    # production completion rule (src/observation_loop.hh:153-157)
    assert not obs.check_observations_complete()
    obs.set_transfers_complete(True)
    assert not obs.check_ready_for_transfer() and not obs.check_observations_complete()
    for _ in range(3):
        obs.generate_analysis_event()
    obs.fake_complete(0, 3)
    obs.check_analysis_events()
    assert obs.check_observations_complete()
    assert obs.get_current_transfer_gemm() == 4 * 32
    obs.close()


def test_observation_loop_event_errors_are_sticky_not_polled_forever(host, bfm):
    """A failed query / record / create must surface as an error, never as "not ready yet" (the reference exits inside
    gpuErrchk, src/beamformer.cuh:19-29; a loop that treats errors as not-ready spins forever on a dead device)."""
    from dsabeamformer_amd._lib import BF_ERR_DEVICE, BF_OK, DsabfError

    cfg = bfm.debug_config()
    # (1) a query that fails
    obs = make_obs(host, cfg, debug=False)
    obs.generate_transfer_event()
    assert obs.status() == BF_OK
    obs.fake.fail_query = True
    with pytest.raises(DsabfError) as e:
        obs.check_transfer_events()
    assert e.value.code == BF_ERR_DEVICE and obs.status() == BF_ERR_DEVICE
    obs.fake.fail_query = False
    before = obs.counters()
    with pytest.raises(DsabfError):      # sticky: nothing advances any more, every later call reports it
        obs.generate_transfer_event()
    with pytest.raises(DsabfError):
        obs.check_analysis_events()
    assert obs.counters() == before
    obs.close()
    # (2) a record that fails does not count as queued
    obs = make_obs(host, cfg, debug=False)
    obs.fake.fail_record = True
    with pytest.raises(DsabfError):
        obs.generate_analysis_event()
    assert obs.counters()["AQ"] == 0 and obs.status() == BF_ERR_DEVICE
    obs.close()
    # (3) the destroy-and-recreate step of check_*_events cannot create a new event
    obs = make_obs(host, cfg, debug=False)
    obs.generate_transfer_event()
    obs.fake_complete(1, 0)
    obs.fake.fail_create = True
    with pytest.raises(DsabfError):
        obs.check_transfer_events()
    assert obs.status() == BF_ERR_DEVICE
    obs.fake.fail_create = False
    obs.close()
    # (4) construction with a backend that cannot create events
    fake_cfg = bfm.debug_config()
    from fake_events import FakeEvents
    f = FakeEvents()
    f.fail_create = True
    o = host.ObservationLoopState(fake_cfg, debug=False, event_ops=f.ops)
    assert o.status() == BF_ERR_DEVICE
    o.close()


def test_detected_sink_ring_and_file_format(host, bfm, tmp_path):
    """dsabf::file_sink without a device: ring slots, in-order commit, back-pressure, header + payload layout."""
    cfg = bfm.debug_config()
    cfg.n_beams, cfg.n_freq, cfg.n_out_per_gemm, cfg.n_gemms_per_block = 32, 4, 2, 3
    per = cfg.n_out_per_gemm * cfg.n_freq * cfg.n_beams
    path = str(tmp_path / "detected.bin")
    sink = host.FileSink(cfg, path, gpu=3, slots=4)
    rng = np.random.default_rng(11)
    want = rng.standard_normal((9, per)).astype(np.float32)
    # fill the 4-slot ring out of order (streams finish gemm-units in time-slice order, not index order)
    for g in (2, 0, 3, 1):
        sink.acquire(g)[:] = want[g]
    assert sink.acquire(4) is None          # ring full: slot of gemm 0 not yet committed
    assert not sink.commit(1)               # commits are in order
    assert sink.commit(0) and sink.commit(1)
    assert not sink.commit(1)               # ... and happen once
    for g in (4, 5):
        sink.acquire(g)[:] = want[g]
    assert sink.acquire(6) is None
    for g in (2, 3, 4, 5):
        assert sink.commit(g)
    for g in (8, 7, 6):
        sink.acquire(g)[:] = want[g]
    for g in (6, 7, 8):
        assert sink.commit(g)
    assert sink.acquire(8) is None          # already delivered
    sink.close()
    hdr, data = host.read_detected_file(path)
    assert hdr["CONTENT"] == "detected_power" and hdr["DTYPE"] == "float32" and hdr["GPU"] == "3"
    assert (int(hdr["N_BEAMS"]), int(hdr["N_FREQUENCIES"]), int(hdr["N_OUTPUTS_PER_GEMM"])) == (32, 4, 2)
    assert os.path.getsize(path) == host.DETECTED_HEADER_BYTES + want.nbytes
    assert np.array_equal(data.reshape(9, per), want)


@pytest.mark.parametrize("seed", list(range(8)))
def test_detected_sink_under_random_acquire_and_commit_orders(host, bfm, tmp_path, seed):
    """dsabf::file_sink: gemm-units acquired in a random order inside the ring's window, committed in index order at random
    moments, acquires outside the window refused -- the file holds every unit once, in index order."""
    rng = np.random.default_rng(40 + seed)
    cfg = bfm.debug_config()
    cfg.n_beams, cfg.n_freq, cfg.n_out_per_gemm, cfg.n_gemms_per_block = 16, 3, int(rng.integers(1, 4)), 4
    per = cfg.n_out_per_gemm * cfg.n_freq * cfg.n_beams
    slots, n = int(rng.integers(2, 9)), int(rng.integers(5, 40))
    path = str(tmp_path / "detected.bin")
    sink = host.FileSink(cfg, path, gpu=1, slots=slots)
    want = rng.standard_normal((n, per)).astype(np.float32)
    committed, acquired = 0, set()
    while committed < n:
        window = [g for g in range(committed, min(n, committed + slots)) if g not in acquired]
        if window and (rng.integers(3) or committed not in acquired):
            g = int(rng.choice(window))
            view = sink.acquire(g)
            assert view is not None, (g, committed, slots)
            view[:] = want[g]
            acquired.add(g)
        elif committed in acquired:
            assert sink.commit(committed)
            committed += 1
        beyond = committed + slots + int(rng.integers(0, 3))
        assert sink.acquire(beyond) is None                     # no slot for a unit beyond the window
        if committed:
            assert sink.acquire(int(rng.integers(committed))) is None   # ... nor for one already delivered
            assert not sink.commit(committed - 1)
    sink.close()
    hdr, data = host.read_detected_file(path)
    assert os.path.getsize(path) == host.DETECTED_HEADER_BYTES + want.nbytes
    assert np.array_equal(data.reshape(n, per), want)


def _ring_name(tag):
    return "dsabf_test_%s_%d" % (tag, os.getpid())


def test_shm_ring_writer_process_to_reader(host, tmp_path):
    """the PSRDADA stand-in.  `junkdb` (separate process) creates the ring and writes 11 blocks through 3
    slots; the reader sees them in order with the junk source's bytes, then the short end-of-data block; the writer
    deletes the ring once it is drained.  Blocking in both directions is exercised by the ring being shorter than the
    stream and by a slow reader."""
    import subprocess
    import time

    from conftest import ROOT

    name, bs, n, distinct, seed = _ring_name("w"), 64 * 1024, 11, 4, 77
    hdr_file = os.path.join(CFG, "correlator_header_dsaX.txt")
    args = [os.path.join(ROOT, "dsabeamformer_amd", "junkdb"), "-k", name, "-n", str(n), "-r", "3", "-b", str(bs),
            "-d", str(distinct), "-s", str(seed)]
    if os.path.exists(hdr_file):
        args += ["-H", hdr_file]
    p = subprocess.Popen(args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    try:
        ring = host.ShmRing(name)
        assert (ring.n_blocks, ring.block_size) == (3, bs)
        assert "HDR_SIZE" in ring.header
        want = host.junk_bytes(bs, distinct, seed)
        assert len(np.unique(want)) == 256
        for i in range(n):
            if i == 4:
                time.sleep(0.3)  # the writer runs into a full ring
            data, bid = ring.read()
            assert bid == i and data.size == bs and np.array_equal(data, want[i % distinct]), i
        data, bid = ring.read()
        assert bid == n and data.size == 0   # short block = end of data (src/dada_handler.hh:105-113)
        ring.detach()
        out, err = p.communicate(timeout=30)
        assert p.returncode == 0, err
        assert "wrote 11 blocks" in out
    finally:
        if p.poll() is None:
            p.kill()
        host.shm_ring_unlink(name)
    assert not os.path.exists("/dev/shm/" + name)


def test_shm_ring_in_process_backpressure(host):
    """Writer and reader threads on a 2-slot ring: order, sizes, and that a full ring blocks the writer."""
    import threading

    name = _ring_name("t")
    ring = host.ShmRing(name, n_blocks=2, block_size=4096, header="HDR_SIZE 4096\nSOURCE TEST\n")
    try:
        reader = host.ShmRing(name)
        assert reader.header.startswith("HDR_SIZE 4096")
        wrote = []

        def writer():
            for i in range(6):
                ring.write(np.full(4096, i, np.uint8))
                wrote.append(i)
            ring.write(np.zeros(100, np.uint8))  # short block

        t = threading.Thread(target=writer)
        t.start()
        import time

        deadline = time.time() + 20
        while len(wrote) < 2 and time.time() < deadline:
            time.sleep(0.01)
        time.sleep(0.3)
        assert wrote == [0, 1]  # two slots: the third write is blocked until a block is read
        for i in range(6):
            data, bid = reader.read()
            assert bid == i and data.size == 4096 and (data == i).all()
        data, bid = reader.read()
        assert data.size == 100
        t.join(timeout=10)
        assert not t.is_alive()
        reader.detach()
    finally:
        ring.detach()
        ring.unlink()


def test_dm_trials_and_delays_match_the_executed_dispersion_notebook(host, orc):
    """8f-4, pinned by EXECUTION: tests/golden/make_dispersion_golden.py runs cells 1, 2 and 5 of the
    reference's sandbox/Dispersion Theory.ipynb as they stand and stores what they produced -- the whole 1627-trial DM ladder
    and the 2048 per-channel sample delays of the cell's DM-2000 pulse (read out twice: from the cell's int() calls and
    from the pulse positions left in its array C).  Host mirror == oracle == the notebook."""
    nb = np.load(os.path.join(GOLDEN, "dispersion_notebook.npz"))
    dms = host.dm_trials()
    assert np.array_equal(dms, orc.dm_trials())
    assert len(dms) == len(nb["dms"]) == 1627                          # "Number of trials = 1627" in the committed notebook
    # the ladder is a 1626-step recurrence in double; C and numpy group two products differently (x*d*d vs x*d**2):
    # agreement to a few units in the last place at every step, identical trial count
    assert np.abs(dms[1:] / nb["dms"][1:] - 1).max() <= 1e-13 and dms[0] == nb["dms"][0] == 0.0
    n_chan, _eps, _nu, _b, _ti, _tscat, _tsamp = nb["constants"]
    assert (n_chan, _eps, _ti, _tscat, _tsamp) == (2048, 1.25, 40.0, 0.0, 131.0)      # the defaults of dm_trials are cell 1
    # cell 5: channel i sits at 1.28 + 0.25/2048*i GHz, reference 1.53 GHz, 0.131*16 ms samples, DM 2000
    freq64 = np.array([1.28 + (1.53 - 1.28) / 2048 * i for i in range(2048)])
    assert float(nb["f_ref_ghz"][0]) == 1.53 and abs(float(nb["d_over_dm"][0]) - 4.15) < 1e-15
    # the delay law itself, in double on the notebook's own real numbers: exact
    law = 4.15 * 2000.0 * (-1.53 ** -2 + freq64 ** -2) / float(nb["tsamp_ms"][0])
    assert np.abs(law - nb["delay_args_dm2000"]).max() <= 1e-9 and np.array_equal(law.astype(np.int32), nb["delays_dm2000"])
    # the product's / oracle's table type is float32 (src/beamformer.hh: `float` frequencies): a channel whose real delay
    # lies within float32 rounding of an integer may truncate the other way -- never by more than one sample
    d = host.dm_delays([2000.0, 56.5], freq64.astype(np.float32), 1.53, float(nb["tsamp_ms"][0]))
    assert np.array_equal(d, orc.dm_delays(np.array([2000.0, 56.5]), freq64.astype(np.float32), 1.53, float(nb["tsamp_ms"][0])))
    diff = d[0] - nb["delays_dm2000"]
    assert np.abs(diff).max() <= 1 and np.count_nonzero(diff) <= 4, np.count_nonzero(diff)
    frac = nb["delay_args_dm2000"] - np.floor(nb["delay_args_dm2000"])
    assert all(min(frac[i], 1 - frac[i]) < 1e-4 for i in np.flatnonzero(diff))      # only where the real number is at an edge
    assert d[:, -1].tolist() == [0, 0] and (np.diff(d[0]) <= 0).all()
    assert int(nb["delays_dm2000"][0]) == 725 and int(nb["delays_dm2000"].sum()) == 675987
    # a different ladder (coarser channels, shorter span)
    assert np.array_equal(host.dm_trials(5.0, 300.0, 256, 1.5, 1.4, 0.9765625, 20.0, 5.0, 65.5),
                          orc.dm_trials(5.0, 300.0, 256, 1.5, 1.4, 0.9765625, 20.0, 5.0, 65.5))


def test_oracle_dedisperse_dm_properties(orc):
    """DM 0 (all delays 0) is the a8 column sum per time sample; a dispersed impulse is recovered at its DM."""
    rng = np.random.default_rng(8)
    n_t, n_f, n_b = 40, 16, 8
    series = rng.random((n_t, n_f, n_b), dtype=np.float32)
    zero = np.zeros((1, n_f), np.int32)
    out = orc.dedisperse_dm(series, zero, n_t)
    g = orc.Geom(n_beams=n_b, n_ant=64, n_freq=n_f, n_avg=1, n_out_per_gemm=1)
    for t in (0, 17, 39):
        assert np.array_equal(out[0, t], orc.dedisperse(g, series[t][None]))
    freq = np.linspace(1.4, 1.5, n_f).astype(np.float32)
    delays = orc.dm_delays(np.array([0.0, 300.0, 600.0]), freq, 1.5, 0.131 * 64)   # up to 19 samples
    assert 0 < delays[1].max() < delays[2].max() < n_t - 6
    pulse = np.zeros((n_t, n_f, n_b), np.float32)
    for f in range(n_f):
        pulse[5 + delays[1, f], f, :] = 1.0
    out = orc.dedisperse_dm(pulse, delays, n_t - int(delays.max()))
    assert out[1, 5, 0] == n_f and out[1].max() == n_f and out[0].max() < n_f and out[2].max() < n_f


def test_sink_and_ring_error_paths(host, bfm, tmp_path):
    """A sink on an unwritable path and a reader on a ring that does not exist fail with an error code, not a crash;
    an oversized write to a ring is rejected."""
    cfg = bfm.debug_config()
    with pytest.raises(Exception):
        host.FileSink(cfg, str(tmp_path / "no_such_dir" / "x.bin"))
    lib = host.load()
    import ctypes as C

    h = C.c_void_p()
    assert lib.bfh_shm_ring_attach(b"dsabf_no_such_ring", 50, C.byref(h)) < 0
    name = _ring_name("e")
    ring = host.ShmRing(name, n_blocks=2, block_size=1024)
    try:
        with pytest.raises(Exception):
            ring.write(np.zeros(2048, np.uint8))       # larger than a block
        assert host.ShmRing(name).block_size == 1024   # a second attach sees the same geometry
        with pytest.raises(Exception):
            host.ShmRing(name, n_blocks=host_max_blocks() + 1, block_size=64)
    finally:
        ring.detach()
        ring.unlink()


def host_max_blocks():
    return 64   # dsabf::kMaxRingBlocks


def test_observation_refuses_a_ring_of_the_wrong_block_size(host, bfm):
    """A ring built for another geometry: the reference prints "ERROR: block size ..." and carries on
    (src/beamformer.cu:336-339), copying the geometry's block size out of every ring block -- past the end of a smaller
    block, a prefix of a larger one.  run_observation refuses before it touches a device."""
    from dsabeamformer_amd._lib import BF_ERR_INVALID, DsabfError

    cfg = bfm.debug_config(n_freq=8, n_beams=64)
    want = cfg.n_ant * cfg.n_freq * cfg.n_out_per_gemm * cfg.n_pol * cfg.n_avg * cfg.n_gemms_per_block
    for wrong in (want // 2, want * 2):
        name = _ring_name("m")
        ring = host.ShmRing(name, n_blocks=2, block_size=wrong, header="HDR_SIZE 4096\n")
        try:
            with pytest.raises(DsabfError) as e:
                host.run_observation_shm(cfg, name)
            assert e.value.code == BF_ERR_INVALID and "block size" in str(e.value)
        finally:
            ring.detach()
            ring.unlink()


def test_psrdada_adapter_compiles_and_is_a_block_source(tmp_path):
    """SURVEY.md 8f-3 / VERDICT r02 item 7: csrc/bf_dada.cpp is the reference's dada_handler (src/dada_handler.hh:25-177)
    behind -DDSABF_WITH_PSRDADA.  libpsrdada is not in the image, so the check is compile-time only: the flag-on branch of
    the adapter and of `beam` type-checks against declarations of the PSRDADA calls the reference makes
    (tests/support/psrdada_api: declarations only, never linked), dada_block_source implements the very interface the loop
    reads from and that shm_block_source implements, and the default library carries no trace of it."""
    import subprocess

    from dsabeamformer_amd import build as b

    api = os.path.join(ROOT, "tests", "support", "psrdada_api")
    inc = ["-I" + api, "-I" + os.path.join(ROOT, "include")]
    src = os.path.join(ROOT, "dsabeamformer_amd", "csrc", "bf_dada.cpp")
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-DDSABF_WITH_PSRDADA"] + inc + [src],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    probe = tmp_path / "contract.cpp"
    probe.write_text(r'''
#include <type_traits>
#include <iostream>
#include "dsabf_host.hpp"
using namespace dsabf;
template <class S> constexpr bool is_source() {
    return std::is_base_of<block_source, S>::value && !std::is_abstract<S>::value &&
           std::is_same<decltype(std::declval<S&>().read()), char*>::value &&
           std::is_same<decltype(std::declval<S&>().check_transfers_complete()), bool>::value &&
           std::is_same<decltype(std::declval<S&>().ok()), bool>::value &&
           std::is_same<decltype(std::declval<S&>().is_pinned()), bool>::value &&
           std::is_same<decltype(std::declval<S&>().expect_block_bytes(0)), void>::value;
}
static_assert(is_source<shm_block_source>(), "shm_block_source");
static_assert(is_source<dada_block_source>(), "dada_block_source implements the same contract");
// the reference's constructor arguments: (name, core, in_key) -- src/dada_handler.hh:15, src/beamformer.cu:132
static_assert(std::is_constructible<dada_block_source, const char*, int, unsigned, std::ostream&>::value, "ctor");
int main() { return 0; }
''')
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-DDSABF_WITH_PSRDADA"] + inc + [str(probe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # `beam -k <hex>` takes the PSRDADA branch only in such a build
    r = subprocess.run([b.HIPCC, "-std=c++17", "-fsyntax-only", "-DDSABF_WITH_PSRDADA"] + inc +
                       [os.path.join(ROOT, "dsabeamformer_amd", "csrc", "beam_main.cpp")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    syms = subprocess.run(["nm", "-DC", os.path.join(ROOT, "dsabeamformer_amd", "libdsabf.so")], capture_output=True, text=True).stdout
    assert "dada_block_source" not in syms and "psrdada" not in syms     # off by default


def _build_fake_psrdada(tmp_path):
    """libfakepsrdada.so + the reader harness / a beam with the PSRDADA branch, linked against the in-tree libdsabf.so."""
    import subprocess

    from dsabeamformer_amd import build as b

    sup = os.path.join(ROOT, "tests", "support")
    api, fake = os.path.join(sup, "psrdada_api"), os.path.join(sup, "fake_psrdada")
    lib = str(tmp_path / "libfakepsrdada.so")
    r = subprocess.run(["gcc", "-std=c11", "-O2", "-Wall", "-Wextra", "-Werror", "-fPIC", "-shared", "-I" + api,
                        os.path.join(fake, "fake_psrdada.c"), "-o", lib, "-lrt"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    pkg = os.path.join(ROOT, "dsabeamformer_amd")

    def link(out, main_src):
        r = subprocess.run([b.HIPCC, "-O1", "-std=c++17", "-DDSABF_WITH_PSRDADA", "-I" + api, "-I" + os.path.join(ROOT, "include"), main_src,
                            os.path.join(pkg, "csrc", "bf_dada.cpp"), "-o", out, "-L" + pkg, "-ldsabf", "-L" + str(tmp_path), "-lfakepsrdada",
                            "-Wl,-rpath," + pkg, "-Wl,-rpath," + str(tmp_path)], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr
        return out

    return lib, link


def test_psrdada_adapter_runs_against_a_stand_in_library(tmp_path):
    """VERDICT r04 missing 3 (csrc/bf_dada.cpp was only ever type-checked): the adapter EXECUTES -- constructor (multilog,
    hdu create / set_key / connect / lock_read, the page-locking walk over the ring blocks, which fails here without a GPU and
    is survived), read_headers (header block read and cleared, block size), the read / check_transfers_complete / close cycle in
    write order with the right byte counts, the short block that ends the observation, the destructor's unlock -- against
    tests/support/fake_psrdada (a functional stand-in for the libpsrdada calls src/dada_handler.hh:25-177 makes; not PSRDADA)."""
    import ctypes as C
    import subprocess
    import threading

    lib_path, link = _build_fake_psrdada(tmp_path)
    exe = link(str(tmp_path / "dada_reader"), os.path.join(ROOT, "tests", "support", "fake_psrdada", "dada_reader_main.cpp"))
    fake = C.CDLL(lib_path)
    fake.fakedada_create.restype = C.c_void_p
    fake.fakedada_create.argtypes = [C.c_uint, C.c_uint64, C.c_uint64, C.c_char_p]
    fake.fakedada_write.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    fake.fakedada_blocks_read.restype = C.c_uint64
    fake.fakedada_blocks_read.argtypes = [C.c_void_p]
    fake.fakedada_header_cleared.argtypes = [C.c_void_p]
    fake.fakedada_destroy.argtypes = [C.c_void_p]
    key, bufsz, n_full = 0xBA00 + os.getpid() % 200, 1 << 16, 7
    w = fake.fakedada_create(key, 3, bufsz, b"HDR_SIZE 4096\nNCHAN 256\n")
    assert w
    rng = np.random.default_rng(12)
    blocks = [rng.integers(0, 256, size=bufsz, dtype=np.uint8) for _ in range(n_full)] + [rng.integers(0, 256, size=100, dtype=np.uint8)]
    rc = []

    def writer():
        for blk in blocks:                          # more blocks than buffers: the writer waits for the reader's closes
            rc.append(fake.fakedada_write(w, blk.ctypes.data_as(C.c_void_p), blk.size))

    t = threading.Thread(target=writer)
    t.start()
    try:
        r = subprocess.run([exe, "%x" % key, str(bufsz)], capture_output=True, text=True, timeout=120)
        t.join(timeout=60)
        assert r.returncode == 0, r.stdout + r.stderr
        assert rc == [0] * len(blocks) and fake.fakedada_blocks_read(w) == len(blocks) and fake.fakedada_header_cleared(w) == 1
    finally:
        fake.fakedada_destroy(w)
    out = r.stdout.splitlines()
    assert "block size is: %d" % bufsz in r.stdout and ("pinned 0 block_size %d" % bufsz) in out       # (no GPU: unpinned, carries on)
    assert "Error: could not pin dada buffer" in r.stdout

    def fnv(a):
        h = 1469598103934665603
        for v in a.tobytes():
            h = ((h ^ v) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        return "%x" % h

    got = [l for l in out if l.startswith("block ") and " bytes " in l]
    assert got == ["block %d bytes %d fnv %s%s" % (i, b.size, fnv(b), " last" if i == n_full else "") for i, b in enumerate(blocks)]
    assert "ERROR: Async, Bytes Read: 100, Should also be %d" % bufsz in r.stdout and out[-1] == "done"
    # a key nobody created: the reference's message, ok() false (the reference exits)
    r = subprocess.run([exe, "%x" % (key + 7), str(bufsz)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "Error: could not connect to dada buffer" in r.stdout


def test_dm_trial_share_covers_the_ladder_once():
    """dm_split_trials: the ranks' shares are contiguous, disjoint, cover every trial, differ by at most one."""
    import ctypes as C

    from dsabeamformer_amd._lib import load

    lib = load()
    for n_dm in (0, 1, 5, 6, 64, 97):
        for world in (1, 2, 3, 8):
            at, counts = 0, []
            for r in range(world):
                f, c = C.c_int(), C.c_int()
                assert lib.bfh_dm_trial_share(n_dm, world, r, C.byref(f), C.byref(c)) == 0
                assert f.value == at and c.value >= 0
                at += c.value
                counts.append(c.value)
            assert at == n_dm and max(counts) - min(counts) <= 1 and counts == sorted(counts, reverse=True)
    assert lib.bfh_dm_trial_share(4, 2, 2, None, None) != 0


def test_dm_chunk_sinks_file_and_ring_without_a_device(tmp_path):
    """The DM stage's sinks (round 5) on their own: the file of chunk records read back by host.read_dm_file (header keys, record
    framing, chunks joined along t, a gap refused), and the shared-memory ring to another process (whole blocks, the record in
    front of the payload, back-pressure on a 2-block ring, the short block that ends the data, the ring removed afterwards)."""
    import ctypes as C
    import threading

    from dsabeamformer_amd import api, host
    from dsabeamformer_amd._lib import load

    lib = load()
    cfg = api.production_config(n_freq=8, n_out_per_gemm=2)
    cfg.n_beams, cfg.n_gemms_per_block = 12, 4
    n_dm, rows, D = 5, 8, 3
    rng = np.random.default_rng(8)
    sizes = [5, 8, 8, 1]
    chunks = [rng.random((n_dm, n, 12), dtype=np.float32) for n in sizes]
    # ---- file ----
    path = str(tmp_path / "dm.bin")
    s = C.c_void_p()
    assert lib.bfh_dm_sink_create(C.byref(cfg), path.encode(), 8, n_dm, D, rows, 2, C.byref(s)) == 0
    at = 0
    for c in chunks:
        assert lib.bfh_dm_sink_deliver(s, at, c.shape[1], n_dm, 12, c.ctypes.data_as(C.c_void_p)) == 0
        at += c.shape[1]
    assert lib.bfh_dm_sink_deliver(s, at + 1, 1, n_dm, 12, chunks[0].ctypes.data_as(C.c_void_p)) != 0      # a gap: refused
    assert lib.bfh_dm_sink_destroy(s) == 0
    hdr, data, recs = host.read_dm_file(path)
    assert (int(hdr["N_DM"]), int(hdr["DM_FIRST_TRIAL"]), int(hdr["MAX_DELAY"]), int(hdr["N_FREQUENCIES"]), int(hdr["N_BEAMS"])) == (5, 2, 3, 8, 12)
    assert recs == [(0, 5), (5, 8), (13, 8), (21, 1)] and np.array_equal(data, np.concatenate(chunks, axis=1))
    assert lib.bfh_dm_sink_create(C.byref(cfg), b"/nonexistent/dir/dm.bin", 8, n_dm, D, rows, 0, C.byref(s)) != 0
    # ---- ring ----
    name = "dsabf_dmsink_%d" % os.getpid()
    got, err = [], []

    def consumer():
        try:
            ring = host.ShmRing(name, timeout_ms=20000)
            assert ring.block_size == 32 + n_dm * rows * 12 * 4 and "DM_FIRST_TRIAL 0" in ring.header
            while True:
                blk, bid = ring.read()
                if blk.size < ring.block_size:
                    break
                first_t, (n_t, nd, nb) = int(blk[:8].view("<u8")[0]), (int(v) for v in blk[8:20].view("<u4"))
                got.append((first_t, blk[32:32 + 4 * nd * n_t * nb].view(np.float32).reshape(nd, n_t, nb).copy()))
            ring.detach()
        except Exception as e:  # noqa: BLE001
            err.append(e)

    assert lib.bfh_dm_sink_create(C.byref(cfg), ("ring:%s:2" % name).encode(), 8, n_dm, D, rows, 0, C.byref(s)) == 0
    t = threading.Thread(target=consumer)
    t.start()
    at = 0
    for c in chunks:
        assert lib.bfh_dm_sink_deliver(s, at, c.shape[1], n_dm, 12, c.ctypes.data_as(C.c_void_p)) == 0   # (4 chunks, 2 blocks: waits)
        at += c.shape[1]
    assert lib.bfh_dm_sink_destroy(s) == 0          # the short block, the drain, the unlink
    t.join(timeout=30)
    assert not err, err
    assert [f for f, _ in got] == [0, 5, 13, 21] and all(np.array_equal(g, c) for (_, g), c in zip(got, chunks))
    assert not os.path.exists("/dev/shm/" + name)


@pytest.mark.parametrize("ring", [1, 2, 3, 4, 8])
def test_a_transfer_never_overwrites_a_ring_slot_that_is_still_being_analysed(host, bfm, ring):
    """Round 5, found by the random DEBUG-flow fuzz (2 ring slots, 1 run in 400 wrong): block j lives in slot j % n_blocks_on_gpu
    until its analysis has COMPLETED, so a transfer may be at most n_blocks_on_gpu blocks ahead of blocks_analyzed.  The
    reference's separations (MAX_TOTAL_SEP 4, MAX_TRANSFER_SEP 2) only guarantee that for its own ring of 8; with a run-time
    ring the scheduler holds them to the ring's size.  Slow analyses, instant transfers: every H2D must find its slot free."""
    cfg = bfm.debug_config(n_blocks_on_gpu=ring)
    obs = make_obs(host, cfg, debug=True)
    n_src = 32 * 20
    obs.set_n_pt_sources(n_src)
    busy = {}                    # slot -> block whose voltages it holds and whose analysis has not completed
    analysed_upto = 0
    pend_a, it = [], 0
    while not obs.check_observations_complete():
        it += 1
        assert it < 20000
        c = obs.counters()
        assert c["TQ"] - c["A"] <= min(4, ring) and c["TQ"] - c["T"] <= min(2, ring)
        if obs.check_ready_for_transfer():
            slot = obs.get_next_gpu_transfer_block()
            assert slot == c["TQ"] % ring
            assert slot not in busy, "block %d would overwrite slot %d under block %d" % (c["TQ"], slot, busy.get(slot, -1))
            busy[slot] = c["TQ"]
            obs.generate_transfer_event()
            obs.check_transfers_complete()
        obs.fake_complete(10, 0)                         # transfers land at once
        obs.check_transfer_events()
        if obs.check_ready_for_analysis():
            for ts in range(32):
                obs.get_current_analysis_gemm(ts)
            obs.generate_analysis_event()
            pend_a.append(it)
        if pend_a and it - pend_a[0] >= 7:               # an analysis takes 7 iterations
            obs.fake_complete(0, 1)
            pend_a.pop(0)
        obs.check_analysis_events()
        while analysed_upto < obs.counters()["A"]:       # completed analyses free their slots
            assert busy.pop(analysed_upto % ring) == analysed_upto
            analysed_upto += 1
    assert obs.counters()["A"] == 20 and not busy
    obs.close()


def test_run_time_window_plan_pads_nothing_when_the_launch_fills_the_chip():
    """bf_rtw_plan (include/dsabf_bench.h; no GPU): accumulation windows without a compile-time instantiation run streams of kout
    whole windows over ceil(kout L / 32) chunks.  A launch that fills the chip takes a kout whose streams are whole chunks -- no
    padding rows for the MFMAs -- unless a shorter stream costs no more; a one-unit launch keeps what fits one chunk (the finest
    split); compile-time windows are not this class."""
    import math

    import ctypes as C

    import dsabeamformer_amd as bfm
    from dsabeamformer_amd._lib import load

    lib = load()

    def plan(n_avg, n_units, n_ant=64):
        cfg = bfm.production_config(n_ant=n_ant, n_avg=n_avg, n_out_per_gemm=16)
        k, c = C.c_int(), C.c_int()
        assert lib.bf_rtw_plan(C.byref(cfg), n_units, 256, C.byref(k), C.byref(c)) == 0
        return k.value, c.value

    assert plan(16, 32)[0] == 0 and plan(1, 32)[0] == 0                       # n_ipo 32, 2: compile-time windows
    for n_avg in (3, 5, 6, 7, 9, 10, 12, 13, 20, 24, 37, 48, 100):
        L = 2 * n_avg
        k, chunks = plan(n_avg, 32)
        rows = 32 * 16 * L                                                    # samples per frequency of the launch
        k_full = 32 // math.gcd(L, 32)
        assert 1 <= k <= 16 and chunks * 128 >= rows
        if 32 * 16 >= 4 * k_full * 8:                                         # enough windows for whole-chunk streams, 8 groups at least
            assert chunks * 128 <= rows * 1.04 + 128 * (k * L // 32 + 1), (n_avg, k, chunks)     # at most a ragged last group of padding
        k1, c1 = plan(n_avg, 1)
        assert k1 == (32 // L if L <= 32 else 1), (n_avg, k1)                # a small launch: one chunk's worth
    assert plan(12, 32) == (4, 32 * 16 // 16 * 3)                             # L = 24: 4 windows = 96 rows = 3 chunks, 32 groups of 4 streams
    assert plan(12, 32, n_ant=100)[0] == 4
    assert lib.bf_rtw_plan(None, 1, 256, None, None) != 0
