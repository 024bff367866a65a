"""GPU tests (-m gpu) of the round-2 surface: the nvcc-contracted detect reading, the lifted geometry whitelist (any
n_ant % 4 <= 128, every n_ipo for every antenna count, n_beams % 4), block-granular launches, the executed-notebook
acceptance test on the HIP path, BASELINE config 5 at shard size.  Every call goes through the C-ABI of libdsabf.so."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import CFG, GOLDEN, LONG

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


def _cfg(bfmod, g, **over):
    kw = dict(n_beams=g.n_beams, n_ant=g.n_ant, n_freq=g.n_freq, n_pol=g.n_pol, n_avg=g.n_avg,
              n_out_per_gemm=g.n_out_per_gemm)
    kw.update(over)
    return bfmod.debug_config(**kw)


def _run(torch, bf, packed_np, n_out_floats):
    d_in = torch.from_numpy(packed_np).cuda()
    d_out = torch.full((n_out_floats,), float("nan"), dtype=torch.float32, device="cuda")
    bf.beamform(d_in, packed_np.shape[0], d_out, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return d_out.cpu().numpy()


def _conj_symmetric(w):
    w = w.copy()
    nb = w.shape[2]
    w[:, :, nb // 2:, 0] = w[:, :, :nb // 2, 0][:, :, ::-1]
    w[:, :, nb // 2:, 1] = -w[:, :, :nb // 2, 1][:, :, ::-1]
    return w


# ---- detect modes ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n_ant", [64, 100, 48])
@pytest.mark.parametrize("n_avg", [1, 4, 16, 32])
@pytest.mark.parametrize("paired", [False, True])
def test_contracted_detect_bit_identical_to_the_nvcc_reading(torch, bfmod, orc, n_ant, n_avg, paired):
    """BF_DETECT_CONTRACTED = acc + fma(x, x, y*y), what nvcc's default -fmad=true makes of src/beamformer.cuh:151.
    Bit-identical to the oracle's ORC_CONTRACT_NVCC reading, different from the canonical one, both inside the stated
    tolerance of the exact value."""
    from dsabeamformer_amd._lib import BF_DETECT_CONTRACTED

    g = orc.Geom(n_beams=64, n_ant=n_ant, n_freq=3, n_avg=n_avg, n_out_per_gemm=max(2, 16 // (2 * n_avg)))
    rng = np.random.default_rng(4000 + n_ant + n_avg)
    w = rng.integers(-127, 128, size=(g.n_freq, g.n_ant, g.n_beams, 2), dtype=np.int8)
    if paired:
        w = _conj_symmetric(w)
    packed = rng.integers(0, 256, size=(3, g.n_freq, g.n_time, g.n_ant), dtype=np.uint8)
    bf = bfmod.Beamformer(_cfg(bfmod, g, detect_mode=BF_DETECT_CONTRACTED))
    bf.set_weights(w)
    assert "CONTRACTED" in bf.kernel_info(3)["kernel"] and ("PAIRED" in bf.kernel_info(3)["kernel"]) == paired
    with orc.detect_contract(orc.CONTRACT_NVCC):
        want = orc.beamform(g, w, packed)
    canon = orc.beamform(g, w, packed)
    got = _run(torch, bf, packed, want.size).reshape(want.shape)
    assert np.array_equal(got, want)
    assert (got != canon).any()
    exact = orc.beamform_exact(g, w, packed)
    ok = exact > 0
    assert np.abs(got[ok] / exact[ok] - 1).max() <= (g.n_ipo + 4) * 2.0 ** -24
    bf.close()


@pytest.mark.parametrize("n_avg", [8, 16, 32])
def test_all_three_detect_modes_within_stated_tolerance_of_exact(torch, bfmod, orc, n_avg):
    from dsabeamformer_amd._lib import BF_DETECT_CANONICAL, BF_DETECT_CONTRACTED, BF_DETECT_FAST

    g = orc.Geom(n_beams=128, n_ant=64, n_freq=4, n_avg=n_avg, n_out_per_gemm=2)
    rng = np.random.default_rng(77 + n_avg)
    w = rng.integers(-127, 128, size=(g.n_freq, g.n_ant, g.n_beams, 2), dtype=np.int8)
    packed = rng.integers(0, 256, size=(2, g.n_freq, g.n_time, g.n_ant), dtype=np.uint8)
    packed[0, 0] = 0x88   # the largest magnitudes against ...
    w[0] = 127            # ... the largest weights
    exact = orc.beamform_exact(g, w, packed)
    outs = {}
    for name, mode, bound in (("canonical", BF_DETECT_CANONICAL, (g.n_ipo + 4) * 2.0 ** -24),
                              ("contracted", BF_DETECT_CONTRACTED, (g.n_ipo + 4) * 2.0 ** -24),
                              ("fast", BF_DETECT_FAST, (g.n_ipo + 1) * 2.0 ** -23)):
        bf = bfmod.Beamformer(_cfg(bfmod, g, detect_mode=mode))
        bf.set_weights(w)
        outs[name] = _run(torch, bf, packed, exact.size).reshape(exact.shape)
        assert np.abs(outs[name] / exact - 1).max() <= bound, name
        bf.close()
    assert np.array_equal(outs["canonical"], orc.beamform(g, w, packed))


# ---- the geometry contract of the reference: N_BEAMS % 4, N_ANTENNAS % 4 (src/beamformer.hh:155-156) ----------------------
@pytest.mark.parametrize("n_ant", [4, 12, 20, 36, 48, 60, 68, 80, 96, 112, 124])
@pytest.mark.parametrize("n_avg", [1, 16])
def test_any_antenna_count_divisible_by_four_bit_exact(torch, bfmod, orc, n_ant, n_avg):
    g = orc.Geom(n_beams=96, n_ant=n_ant, n_freq=3, n_avg=n_avg, n_out_per_gemm=max(2, 16 // (2 * n_avg)))
    rng = np.random.default_rng(5000 + n_ant + n_avg)
    w = rng.integers(-127, 128, size=(g.n_freq, g.n_ant, g.n_beams, 2), dtype=np.int8)
    packed = rng.integers(0, 256, size=(3, g.n_freq, g.n_time, g.n_ant), dtype=np.uint8)
    bf = bfmod.Beamformer(_cfg(bfmod, g))
    bf.set_weights(w)
    want = orc.beamform(g, w, packed)
    got = _run(torch, bf, packed, want.size).reshape(want.shape)
    assert np.array_equal(got, want)
    # stage parity entry point on the same class
    d_in = torch.from_numpy(packed[0]).cuda()
    d_c = torch.zeros(g.n_freq * g.n_time * g.n_beams * 2, dtype=torch.float32, device="cuda")
    bf.gemm(d_in, d_c, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(d_c.cpu().numpy().reshape(g.n_freq, g.n_time, g.n_beams, 2), orc.gemm(g, w, orc.expand(packed[0])))
    bf.close()


@pytest.mark.parametrize("n_ant", [16, 32, 100, 128, 52, 116])
@pytest.mark.parametrize("n_avg,n_out", [(2, 4), (4, 2), (8, 3), (32, 2)])
def test_every_n_ipo_for_every_antenna_class_bit_exact(torch, bfmod, orc, n_ant, n_avg, n_out):
    """Round 1 had n_ipo 4 / 8 / 16 / 64 for 64 antennas only."""
    g = orc.Geom(n_beams=64, n_ant=n_ant, n_freq=2, n_avg=n_avg, n_out_per_gemm=n_out)
    if g.n_ipo < 16 and g.n_time % 16:
        pytest.skip("n_time must be a multiple of 16 below n_ipo 16")
    rng = np.random.default_rng(6000 + n_ant + 7 * n_avg)
    w = _conj_symmetric(rng.integers(-127, 128, size=(g.n_freq, g.n_ant, g.n_beams, 2), dtype=np.int8))
    packed = rng.integers(0, 256, size=(2, g.n_freq, g.n_time, g.n_ant), dtype=np.uint8)
    want = orc.beamform(g, w, packed)
    for env in ("1", "0"):   # the conjugate-pair kernel and the general one
        os.environ["DSABF_PAIRED"] = env
        try:
            bf = bfmod.Beamformer(_cfg(bfmod, g))
            bf.set_weights(w)
            assert ("PAIRED" in bf.kernel_info(2)["kernel"]) == (env == "1")
            got = _run(torch, bf, packed, want.size).reshape(want.shape)
            assert np.array_equal(got, want), env
            bf.close()
        finally:
            os.environ.pop("DSABF_PAIRED", None)


@pytest.mark.parametrize("n_beams", [4, 20, 100, 252, 260, 300])
@pytest.mark.parametrize("n_avg", [1, 16])
def test_any_beam_count_divisible_by_four_bit_exact(torch, bfmod, orc, n_beams, n_avg):
    """The last 16-beam column tile is partly filled (zero weights, masked stores)."""
    g = orc.Geom(n_beams=n_beams, n_ant=64, n_freq=3, n_avg=n_avg, n_out_per_gemm=max(2, 16 // (2 * n_avg)))
    rng = np.random.default_rng(7000 + n_beams + n_avg)
    w = rng.integers(-127, 128, size=(g.n_freq, g.n_ant, g.n_beams, 2), dtype=np.int8)
    packed = rng.integers(0, 256, size=(2, g.n_freq, g.n_time, g.n_ant), dtype=np.uint8)
    bf = bfmod.Beamformer(_cfg(bfmod, g))
    bf.set_weights(w)
    want = orc.beamform(g, w, packed)
    d_in = torch.from_numpy(packed).cuda()
    guard = 64
    d_buf = torch.full((want.size + guard,), float("nan"), dtype=torch.float32, device="cuda")
    bf.beamform(d_in, 2, d_buf, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = d_buf.cpu().numpy()
    assert np.array_equal(got[:want.size].reshape(want.shape), want)
    assert np.isnan(got[want.size:]).all()        # nothing written behind the last beam
    bf.close()


def test_antenna_classes_by_their_instantiation(torch, bfmod, orc):
    """64 and 128 antennas run the run-time-count classes since round 6 (the compile-time ones measured inside the box noise
    and were folded, profiles/r06_class_fold_ab.txt), 100 antennas -- BASELINE config 5 -- keep theirs: the handle says which
    instantiation it launches, and the bits are the oracle's."""
    for n_ant, cls in ((64, -1), (100, 100), (128, -3), (96, -3), (48, -1), (52, -2), (108, -4)):
        g = orc.Geom(n_beams=64, n_ant=n_ant, n_freq=2, n_avg=16, n_out_per_gemm=4)
        rng = np.random.default_rng(8000 + n_ant)
        w = rng.integers(-127, 128, size=(g.n_freq, g.n_ant, g.n_beams, 2), dtype=np.int8)
        packed = rng.integers(0, 256, size=(2, g.n_freq, g.n_time, g.n_ant), dtype=np.uint8)
        want = orc.beamform(g, w, packed)
        bf = bfmod.Beamformer(_cfg(bfmod, g))
        bf.set_weights(w)
        assert bf.variant_key() == "fused16_kernel<%d, 32, false, 0, false, 4, 4>" % cls
        got = _run(torch, bf, packed, want.size).reshape(want.shape)
        assert np.array_equal(got, want)
        bf.close()


def test_geometry_fuzz_over_the_whole_contract(torch, bfmod, orc):
    rng = np.random.default_rng(20261003)
    for case in range(40 if LONG else 12):     # (the budgeted run: the first 12 geometries of the same seeded walk)
        n_ant = 4 * int(rng.integers(1, 33))
        n_beams = 4 * int(rng.integers(1, 80))
        n_avg = int(rng.choice([1, 2, 4, 8, 16, 32]))
        n_ipo = 2 * n_avg
        n_out = int(rng.integers(1, 5)) * (max(1, 16 // n_ipo))
        g = orc.Geom(n_beams=n_beams, n_ant=n_ant, n_freq=int(rng.integers(1, 5)), n_avg=n_avg, n_out_per_gemm=n_out)
        mode = int(rng.choice([0, 2]))
        w = rng.integers(-127, 128, size=(g.n_freq, g.n_ant, g.n_beams, 2), dtype=np.int8)
        if n_beams % 32 == 0 and rng.integers(0, 2):
            w = _conj_symmetric(w)
        n_units = int(rng.integers(1, 4))
        packed = rng.integers(0, 256, size=(n_units, g.n_freq, g.n_time, g.n_ant), dtype=np.uint8)
        bf = bfmod.Beamformer(_cfg(bfmod, g, detect_mode=mode))
        bf.set_weights(w)
        with orc.detect_contract(orc.CONTRACT_NVCC if mode == 2 else orc.CONTRACT_NONE):
            want = orc.beamform(g, w, packed)
        got = _run(torch, bf, packed, want.size).reshape(want.shape)
        assert np.array_equal(got, want), (case, g, mode)
        bf.close()


def test_geometry_refusals_name_the_reference_rule(bfmod):
    from dsabeamformer_amd._lib import DsabfError

    for kw, text in ((dict(n_beams=250), "N_BEAMS"), (dict(n_ant=66), "N_ANTENNAS"), (dict(n_ant=2052), "2048 antennas")):
        with pytest.raises(DsabfError) as e:
            bfmod.Beamformer(bfmod.debug_config(**kw))
        assert text in str(e.value), (kw, str(e.value))
    # round 4: what rounds 1-3 refused inside the reference's contract now has a kernel (csrc/bf_fusedg.hip)
    for kw, kern in ((dict(n_ant=132), "fusedg_kernel"), (dict(n_avg=3), "NIPO=6(run-time)"), (dict(n_avg=1, n_out_per_gemm=3), "NIPO=2(run-time)")):
        bf = bfmod.Beamformer(bfmod.debug_config(n_freq=2, n_beams=32, **kw))
        assert kern in bf.kernel_info(1)["kernel"]
        bf.close()


# ---- block-granular launches -----------------------------------------------------------------------------------------------
def test_enqueue_block_equals_per_unit_launches(torch, bfmod, orc):
    g = orc.Geom(n_beams=64, n_ant=64, n_freq=4, n_avg=16, n_out_per_gemm=2)
    cfg = _cfg(bfmod, g, n_gemms_per_block=8, n_blocks_on_gpu=2, n_streams=4)
    rng = np.random.default_rng(31)
    w = rng.integers(-127, 128, size=(g.n_freq, g.n_ant, g.n_beams, 2), dtype=np.int8)
    block = rng.integers(0, 256, size=(8, g.n_freq, g.n_time, g.n_ant), dtype=np.uint8)
    want = orc.beamform(g, w, block)
    bf = bfmod.Beamformer(cfg)
    bf.set_weights(w)
    from dsabeamformer_amd import api

    nbytes = block.nbytes
    pin_in = api.alloc_pinned(nbytes)
    C.memmove(pin_in, block.ctypes.data, nbytes)
    per = want[0].size
    pin_out = api.alloc_pinned(8 * per * 4)
    bf.submit_block(1, pin_in, nbytes)
    bf.sync(-1)
    # whole block in one launch, then a partial range with a hole in the destination list
    bf.enqueue_block(2, 1, 0, 8, [pin_out + u * per * 4 for u in range(8)])
    bf.sync(2)
    got = np.ctypeslib.as_array(C.cast(pin_out, C.POINTER(C.c_float)), shape=(8 * per,)).copy().reshape(want.shape)
    assert np.array_equal(got, want)
    C.memset(pin_out, 0xFF, 8 * per * 4)
    bf.enqueue_block(0, 1, 3, 4, [pin_out + 0 * per * 4, None, pin_out + 2 * per * 4, pin_out + 3 * per * 4])
    bf.sync(0)
    got = np.ctypeslib.as_array(C.cast(pin_out, C.POINTER(C.c_float)), shape=(8 * per,)).copy().reshape(want.shape)
    assert np.array_equal(got[0], want[3]) and np.array_equal(got[2], want[5]) and np.array_equal(got[3], want[6])
    assert np.isnan(got[1]).all()     # the NULL entry was skipped
    from dsabeamformer_amd._lib import DsabfError
    for args in ((0, 1, 6, 3), (0, 1, -1, 2), (0, 1, 0, 0), (9, 1, 0, 1), (0, 5, 0, 1)):
        with pytest.raises(DsabfError):
            bf.enqueue_block(*args)
    api.free_pinned(pin_in)
    api.free_pinned(pin_out)
    bf.close()


def test_observation_loop_block_launches_equal_unit_launches(bfmod, orc, tmp_path):
    """The production loop with the default launch granularity (one launch per PSRDADA block), with quarter / sixteenth
    blocks and with the reference's one launch per gemm-unit: the same detected stream, byte for byte."""
    from dsabeamformer_amd import host

    cfg = bfmod.production_config(n_freq=8, n_beams=64, n_out_per_gemm=2, n_gemms_per_block=8, n_blocks_on_gpu=4, n_streams=4)
    a = host.run_observation_junk_to_file(cfg, 6, str(tmp_path / "blk.bin"), ring_blocks=3)
    for env, val, name in (("DSABF_UNIT_LAUNCH", "1", "unit.bin"), ("DSABF_UNITS_PER_LAUNCH", "8", "quarter.bin"),
                           ("DSABF_UNITS_PER_LAUNCH", "2", "two.bin")):
        os.environ[env] = val
        try:
            b = host.run_observation_junk_to_file(cfg, 6, str(tmp_path / name), ring_blocks=3)
        finally:
            os.environ.pop(env)
        assert a["gemms_written"] == b["gemms_written"] == 6 * 8
        assert open(tmp_path / "blk.bin", "rb").read() == open(tmp_path / name, "rb").read(), name
    # and against the oracle
    g = orc.Geom(n_beams=64, n_ant=64, n_freq=8, n_avg=16, n_out_per_gemm=2)
    pos = host.default_positions(64)
    dirs = host.default_directions(64)
    w = host.make_weights(pos, dirs, 8, 0, 0)
    raw = np.fromfile(tmp_path / "blk.bin", np.float32, offset=4096).reshape(6 * 8, 2, 8, 64)
    ring = a["ring"]
    for blk in range(6):
        want = orc.beamform(g, w, ring[blk % 3])
        assert np.array_equal(raw[blk * 8:(blk + 1) * 8], want)


# ---- the reference's acceptance test on the HIP path -------------------------------------------------------------------------
def test_hip_data_py_table_passes_the_reference_acceptance_against_the_executed_notebook(bfmod, orc, tmp_path):
    """bin/data.py written by the HIP DEBUG flow vs the `out` array of the reference's notebook, executed cell by cell
    (tests/golden/make_notebook_golden.py).  README.md:202-211 / notebook cell 17: RMS 9.10e-4, mean 0.0435 %."""
    from dsabeamformer_amd import host

    nb = np.load(os.path.join(GOLDEN, "notebook_linear.npz"))
    ded, _ms = host.run_debug_observation(bfmod.debug_config(), gpu=0, positions=os.path.join(CFG, "linear_positions.txt"),
                                          directions=os.path.join(CFG, "linear_directions.txt"),
                                          sources=os.path.join(CFG, "linear_source_directions_1024.txt"),
                                          output=str(tmp_path / "data.py"))
    assert ded.shape == (1024, 256)
    ns = {}
    exec(open(tmp_path / "data.py").read(), ns)     # exactly how the notebook reads it: `import data; data.A`
    da = np.array(ns["A"])
    out = nb["nb2d_out"]
    b = np.abs((out.T - da) / out.T)
    rms, mean_pct = float(np.sqrt(np.sum(b ** 2) / (1024 * 256))), float(np.mean(b) * 100)
    assert rms <= 9.10e-4 and mean_pct <= 0.0435, (rms, mean_pct)
    assert float(b.max()) * 100 <= 1.2
    assert np.array_equal(out.argmax(0), da.argmax(1))
    # and the table itself is the oracle's, bit for bit (the 6-digit text of data.py is a rounding of it)
    assert np.array_equal(ded, np.load(os.path.join(GOLDEN, "linear_debug.npz"))["dedispersed"])


# ---- BASELINE config 5 at the size one rank runs it ----------------------------------------------------------------------------
def _grid_100():
    ax = np.linspace(-250, 250, 10)
    pos = np.zeros((100, 3), np.float32)
    pos[:, 0], pos[:, 1] = [v.ravel() for v in np.meshgrid(ax, ax)]
    th, ph = np.meshgrid(np.linspace(-3.5, 3.5, 32) * np.pi / 180, np.linspace(-3.5, 3.5, 16) * np.pi / 180)
    dirs = np.stack([th.ravel(), ph.ravel()], 1).astype(np.float32)
    return pos, dirs


def test_config5_rank_shard_at_full_size(torch, bfmod, orc):
    """One rank's share of BASELINE configs[4]: 128 of 1024 frequencies x 512 beams x 100 antennas, n_ipo 32, 8 gemm-units.
    Sampled frequencies bit for bit against the oracle, plus size-independent properties over the whole output."""
    from dsabeamformer_amd import host

    pos, dirs = _grid_100()
    rank = 5
    g = orc.Geom(n_beams=512, n_ant=100, n_freq=128, n_avg=16, n_out_per_gemm=4)
    w = host.make_weights(pos, dirs, g.n_freq, chan0=rank * 128, gpu=0)
    rng = np.random.default_rng(55)
    n_units = 8
    packed = rng.integers(0, 256, size=(n_units, g.n_freq, g.n_time, g.n_ant), dtype=np.uint8)
    packed[3] = packed[1]                    # a repeated unit must give a repeated output
    packed[5] = 0                            # silence in, exact zeros out
    bf = bfmod.Beamformer(_cfg(bfmod, g))
    bf.set_weights(w)
    assert "PAIRED" in bf.kernel_info(n_units)["kernel"]      # the symmetric 32 x 16 beam grid
    got = _run(torch, bf, packed, n_units * g.out_per_gemm).reshape(n_units, g.n_out_per_gemm, g.n_freq, g.n_beams)
    assert np.isfinite(got).all() and (got >= 0).all()
    assert np.array_equal(got[3], got[1]) and not got[5].any()
    for f in (0, 1, 63, 64, 127):           # sampled frequencies, every unit, every output, every beam
        gf = orc.Geom(n_beams=512, n_ant=100, n_freq=1, n_avg=16, n_out_per_gemm=4)
        want = orc.beamform(gf, w[f:f + 1], np.ascontiguousarray(packed[:, f:f + 1]))
        assert np.array_equal(got[:, :, f], want[:, :, 0]), f
    # general kernel on the same input: same bits (pairing is an optimisation, not a different function)
    os.environ["DSABF_PAIRED"] = "0"
    try:
        bf2 = bfmod.Beamformer(_cfg(bfmod, g))
        bf2.set_weights(w)
        got2 = _run(torch, bf2, packed, n_units * g.out_per_gemm).reshape(got.shape)
    finally:
        os.environ.pop("DSABF_PAIRED")
    assert np.array_equal(got2, got)
    bf.close()
    bf2.close()


def test_config5_debug_flow_on_the_named_catalogue(bfmod, orc, tmp_path):
    """configs[4] names grid_positions.txt + grid_source_directions_4096.txt.  The shipped grid_positions.txt holds 64
    antennas, so the 100-antenna run takes the synthesised 10 x 10 grid (SURVEY.md section 4's recipe) with the named
    catalogue: DEBUG flow end to end (generator -> H2D -> fused kernel -> dedisperse -> data.py), first 160 sources."""
    from dsabeamformer_amd import host

    pos, dirs = _grid_100()
    pfile, dfile, sfile = tmp_path / "pos100.txt", tmp_path / "dir512.txt", tmp_path / "src160.txt"
    pfile.write_text("100\n" + "".join("%r %r %r\n" % (float(p[0]), float(p[1]), float(p[2])) for p in pos))
    dfile.write_text("512\n" + "".join("%r %r\n" % (float(d[0]), float(d[1])) for d in dirs))
    src_all = orc.read_directions(os.path.join(CFG, "grid_source_directions_4096.txt"))
    assert src_all.shape == (4096, 2)
    src = src_all[::26][:160]
    sfile.write_text("160\n" + "".join("%r %r\n" % (float(s[0]), float(s[1])) for s in src))
    cfg = bfmod.debug_config(n_ant=100, n_beams=512, n_freq=32, n_gemms_per_block=16, n_blocks_on_gpu=4, n_streams=4)
    ded, _ms = host.run_debug_observation(cfg, gpu=0, positions=str(pfile), directions=str(dfile), sources=str(sfile),
                                          output=str(tmp_path / "data.py"), max_sources=160)
    assert ded.shape == (160, 512)
    g = orc.Geom(n_beams=512, n_ant=100, n_freq=32, n_avg=1, n_out_per_gemm=8)
    p32 = orc.read_positions(str(pfile), 100)
    d32 = orc.read_directions(str(dfile), 512)
    s32 = orc.read_directions(str(sfile))
    w = orc.make_weights(g, p32, d32, 0)
    units = orc.generate_test_data(g, p32, s32, 0, 0, 160)
    out = orc.beamform(g, w, units)
    want = np.stack([orc.dedisperse(g, out[u]) for u in range(160)])
    assert np.array_equal(ded, want)
    assert ded.max() > 0 and np.isfinite(ded).all()
    # (no 'brightest beam = nearest beam' check: a 10 x 10 grid at 55 m spacing has grating lobes every 0.2 degrees)


# ---- the gather behind the C-ABI, on one GPU ---------------------------------------------------------------------------------
@pytest.mark.parametrize("layout", [0, 1])
@pytest.mark.parametrize("root", [0, -1, -2])
@pytest.mark.parametrize("through_rccl", [False, True])
def test_gather_detected_world_one(torch, bfmod, orc, monkeypatch, layout, root, through_rccl):
    """bf_comm / bf_gather_detected with one rank.  through_rccl: a real one-rank RCCL communicator (unique id) and
    DSABF_GATHER_SELF_RCCL=1, so the rank's own rows travel through grouped ncclSend / ncclRecv -- the code path every
    message of a multi-GPU run takes -- instead of the device-to-device copy."""
    from dsabeamformer_amd import api

    g = orc.Geom(n_beams=64, n_ant=64, n_freq=4, n_avg=16, n_out_per_gemm=2)
    rng = np.random.default_rng(91)
    w = rng.integers(-127, 128, size=(g.n_freq, g.n_ant, g.n_beams, 2), dtype=np.int8)
    packed = rng.integers(0, 256, size=(3, g.n_freq, g.n_time, g.n_ant), dtype=np.uint8)
    bf = bfmod.Beamformer(_cfg(bfmod, g))
    bf.set_weights(w)
    want = orc.beamform(g, w, packed)
    d_in = torch.from_numpy(packed).cuda()
    d_local = torch.empty(want.size, dtype=torch.float32, device="cuda")
    d_full = torch.full((want.size,), float("nan"), dtype=torch.float32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    if through_rccl:
        monkeypatch.setenv("DSABF_GATHER_SELF_RCCL", "1")
        comm = api.Comm(0, 1, api.comm_unique_id(), device=0)
    else:
        comm = api.Comm(0, 1)
    n_rows, row_floats = 3 * g.n_out_per_gemm, g.n_freq * g.n_beams
    assert comm.rows_held(n_rows, root) == n_rows
    bf.beamform(d_in, 3, d_local, s)
    comm.gather(d_local, n_rows, row_floats, root, layout, d_full, s)
    torch.cuda.synchronize()
    assert np.array_equal(d_full.cpu().numpy().reshape(want.shape), want)   # one rank: both layouts are the identity
    comm.close()
    bf.close()


def _beam(*args, timeout=600):
    import subprocess

    from dsabeamformer_amd import build

    r = subprocess.run([build.BEAM] + [str(a) for a in args], capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


def test_beam_sharded_mode_and_replica_launcher_on_one_gpu(tmp_path):
    """`beam -R 1 -r 0 -I id` (a one-rank frequency partition: real RCCL communicator, gather path, root-side D2H) writes
    the same detected file as the unsharded run; `beam_replicas -n 1` (the reference's replica deployment) and
    `beam_replicas -n 1 -S` (the sharded one) drive the same binary."""
    import subprocess

    from dsabeamformer_amd import build

    plain, shard = tmp_path / "plain.bin", tmp_path / "shard.bin"
    assert "Wrote 64 gemm-units" in _beam("-j", 27, "-w", plain)      # 25 burn-in reads (BURNIN) + 2 analysed blocks
    out = _beam("-j", 27, "-w", shard, "-R", 1, "-r", 0, "-I", tmp_path / "id")
    assert "Shard 0 of 1: channels 0 .. 255" in out and "Wrote 64 gemm-units" in out
    a, b = open(plain, "rb").read(), open(shard, "rb").read()
    assert len(a) == len(b) == 4096 + 64 * 8 * 256 * 256 * 4 and a == b
    for extra, name in ((([], "rep_{i}.bin"), (["-S"], "shd_{i}.bin")) if LONG else ((["-S"], "shd_{i}.bin"),)):
        r = subprocess.run([build.REPLICAS, "-n", "1"] + extra + ["-j", "27", "-w", str(tmp_path / name)],
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        assert open(tmp_path / name.replace("{i}", "0"), "rb").read() == a
    r = subprocess.run([build.REPLICAS], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr


def test_beam_sharded_argument_errors(tmp_path):
    import subprocess

    from dsabeamformer_amd import build

    for args in (["-j", "1", "-R", "3", "-r", "0", "-I", str(tmp_path / "i")],      # 256 channels do not split 3 ways
                 ["-j", "1", "-R", "2", "-r", "2", "-I", str(tmp_path / "i")],      # rank out of range
                 ["-j", "1", "-R", "2", "-r", "0"]):                               # no id file
        r = subprocess.run([build.BEAM] + args, capture_output=True, text=True, timeout=120)
        assert r.returncode != 0 and "beam:" in r.stderr, args
