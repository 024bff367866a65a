"""GPU parity tests (run with -m gpu on an MI355X): every call goes through the C-ABI of libdsabf.so and is
checked against the CPU oracle and the committed golden fixtures.

Bar: BIT-EXACT everywhere -- 4-bit expand (integer), complex GEMM output (exact integer sums, one fp32 rounding),
detected powers (the kernel keeps the reference's sequential fp32 accumulation order in registers, so this holds
for n_ipo = 32 as well as the DEBUG n_ipo = 2) and dedispersed rows (ascending-f fp32 sum)."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import GOLDEN, LONG, sweep

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


def test_expand_bit_exact(torch, bfmod, orc):
    bf = bfmod.Beamformer(bfmod.debug_config())
    rng = np.random.default_rng(7)
    packed = rng.integers(0, 256, size=1 << 20, dtype=np.uint8)
    packed[:256] = np.arange(256, dtype=np.uint8)
    packed[256:260] = [0xD7, 0x25, 0xA8, 0x70]  # reference KATs (sandbox/kernelTest.cu:128, bitshift.cpp:5-6)
    d_in = torch.from_numpy(packed).cuda()
    d_out = torch.zeros(packed.size * 2, dtype=torch.int8, device="cuda")
    bf.expand(d_in, packed.size, d_out, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = d_out.cpu().numpy().reshape(-1, 2)
    assert got[256:260].tolist() == [[-3, 7], [2, 5], [-6, -8], [7, 0]]
    assert np.array_equal(got, orc.expand(packed))


@pytest.mark.parametrize("tag", ["p", "d"])
def test_fused_random_small_golden(torch, bfmod, orc, tag):
    gold = np.load(os.path.join(GOLDEN, "random_small.npz"))
    g = orc.Geom(*[int(x) for x in gold[tag + "_geom"]])
    bf = bfmod.Beamformer(_cfg(bfmod, g))
    bf.set_weights(np.ascontiguousarray(gold[tag + "_w"]))
    packed = np.ascontiguousarray(gold[tag + "_packed"])
    got = _run(torch, bf, packed, gold[tag + "_out"].size).reshape(gold[tag + "_out"].shape)
    assert np.array_equal(got, gold[tag + "_out"])
    assert np.array_equal(got, orc.beamform(g, gold[tag + "_w"], packed))


@pytest.mark.parametrize("n_ant,n_avg,n_units", [(64, 1, 1), (64, 1, 3), (64, 16, 2), (64, 2, 2), (64, 4, 1),
                                                  (64, 8, 1), (64, 32, 1), (16, 1, 2), (32, 16, 1), (128, 1, 3),
                                                  (128, 16, 2), (100, 1, 3), (100, 16, 2)])
def test_fused_geometries_bit_exact(torch, bfmod, orc, n_ant, n_avg, n_units):
    g = orc.Geom(n_beams=96, n_ant=n_ant, n_freq=5, n_avg=n_avg, n_out_per_gemm=max(2, 16 // (2 * n_avg)))
    rng = np.random.default_rng(1000 + n_ant + n_avg)
    w = rng.integers(-127, 128, size=(g.n_freq, g.n_ant, g.n_beams, 2), dtype=np.int8)
    packed = rng.integers(0, 256, size=(n_units, g.n_freq, g.n_time, g.n_ant), dtype=np.uint8)
    bf = bfmod.Beamformer(_cfg(bfmod, g))
    bf.set_weights(w)
    want = orc.beamform(g, w, packed)
    got = _run(torch, bf, packed, want.size).reshape(want.shape)
    assert np.array_equal(got, want)


def test_dsa100_scale_up_shape_bit_exact(torch, bfmod, orc):
    """BASELINE config 5 geometry on one frequency shard: 100 antennas (rows only dword aligned), 512 beams (two
    beam groups per frequency), production n_ipo 32; 10x10 grid positions / 32x16 grid beams synthesised with the
    notebook formulas (the shipped files are 64-antenna / 256-beam)."""
    from dsabeamformer_amd import host

    ax = np.linspace(-250, 250, 10)
    pos = np.zeros((100, 3), np.float32)
    pos[:, 0], pos[:, 1] = [v.ravel() for v in np.meshgrid(ax, ax)]
    th, ph = np.meshgrid(np.linspace(-3.5, 3.5, 32) * np.pi / 180, np.linspace(-3.5, 3.5, 16) * np.pi / 180)
    dirs = np.stack([th.ravel(), ph.ravel()], 1).astype(np.float32)
    g = orc.Geom(n_beams=512, n_ant=100, n_freq=6, n_avg=16, n_out_per_gemm=8)
    w = host.make_weights(pos, dirs, g.n_freq, chan0=128 * 3, gpu=0)  # shard of rank 3 of 8 of a 1024-channel band
    assert np.array_equal(w, np.stack([orc.make_weights(orc.Geom(n_beams=512, n_ant=100, n_freq=128 * 3 + 6), pos, dirs, 0)[128 * 3 + i]
                                       for i in range(6)]))
    rng = np.random.default_rng(100)
    packed = rng.integers(0, 256, size=(2, g.n_freq, g.n_time, g.n_ant), dtype=np.uint8)
    bf = bfmod.Beamformer(_cfg(bfmod, g))
    bf.set_weights(w)
    want = orc.beamform(g, w, packed)
    got = _run(torch, bf, packed, want.size).reshape(want.shape)
    assert np.array_equal(got, want)


def test_gemm_stage_bit_exact(torch, bfmod, orc):
    g = orc.Geom(n_beams=64, n_ant=64, n_freq=3, n_avg=1, n_out_per_gemm=8)
    rng = np.random.default_rng(11)
    w = rng.integers(-127, 128, size=(g.n_freq, g.n_ant, g.n_beams, 2), dtype=np.int8)
    packed = rng.integers(0, 256, size=(g.n_freq, g.n_time, g.n_ant), dtype=np.uint8)
    bf = bfmod.Beamformer(_cfg(bfmod, g))
    bf.set_weights(w)
    d_in = torch.from_numpy(packed).cuda()
    d_c = torch.zeros(g.n_freq * g.n_time * g.n_beams * 2, dtype=torch.float32, device="cuda")
    bf.gemm(d_in, d_c, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    want = orc.gemm(g, w, orc.expand(packed))
    assert np.array_equal(d_c.cpu().numpy().reshape(want.shape), want)


def test_linear_config_debug_geometry_bit_identical(torch, bfmod, orc, linear_inputs, linear_weights):
    """BASELINE config 2: 64 ant x 256 freq x 256 beams, N_TIME = 16, config/linear_* inputs."""
    pos, _, src = linear_inputs
    g = orc.DEBUG_GEOM
    gold = np.load(os.path.join(GOLDEN, "linear_debug.npz"))
    pick = [0, 100, 511, 512, 1023, 77, 900]
    units = orc.generate_test_data(g, pos, src[pick], 0, 0, len(pick))
    bf = bfmod.Beamformer(bfmod.debug_config())
    bf.set_weights(linear_weights)
    got = _run(torch, bf, units, len(pick) * g.out_per_gemm).reshape(len(pick), 8, 256, 256)
    assert np.array_equal(got[0, 0], gold["detected_src0_out0"])
    assert np.array_equal(got[2, 0], gold["detected_src511_out0"])
    assert np.array_equal(got, orc.beamform(g, linear_weights, units))
    # dedisperse (K5) on the device, output 0 of each unit
    d_out = torch.from_numpy(got).cuda()
    d_ded = torch.zeros(256, dtype=torch.float32, device="cuda")
    for i, s in enumerate(pick):
        bf.dedisperse(d_out[i], d_ded, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert np.array_equal(d_ded.cpu().numpy(), gold["dedispersed"][s])


def test_production_geometry_full_size_bit_exact(torch, bfmod, orc, linear_weights):
    """BASELINE config 3 shape: N_TIME = 512 (16 outputs x n_ipo 32), random nibbles incl. -8, 2 gemm-units."""
    g = orc.Geom(n_avg=16, n_out_per_gemm=16)
    rng = np.random.default_rng(0xD5A)
    packed = rng.integers(0, 256, size=(2, g.n_freq, g.n_time, g.n_ant), dtype=np.uint8)
    bf = bfmod.Beamformer(_cfg(bfmod, g))
    bf.set_weights(linear_weights)
    want = orc.beamform(g, linear_weights, packed)
    got = _run(torch, bf, packed, want.size).reshape(want.shape)
    assert np.array_equal(got, want)


def test_large_batch_properties_and_sampled_parity(torch, bfmod, orc, linear_weights):
    """32 gemm-units of production-shaped input (the bench workload): sampled frequencies bit-exact vs the oracle,
    plus size-independent properties: negating every voltage leaves the power unchanged, zero input gives zero,
    and a launch over all units equals unit-by-unit launches."""
    g = orc.Geom(n_avg=16, n_out_per_gemm=16)
    n_units = 32
    bf = bfmod.Beamformer(_cfg(bfmod, g))
    bf.set_weights(linear_weights)
    gen = torch.Generator(device="cuda").manual_seed(5)
    d_in = torch.randint(0, 256, (n_units, g.n_freq, g.n_time, g.n_ant), dtype=torch.uint8, device="cuda", generator=gen)
    # avoid the -8 code so that negation is representable: map nibble 8 -> 9 in both halves
    hi, lo = d_in >> 4, d_in & 15
    hi = torch.where(hi == 8, torch.full_like(hi, 9), hi)
    lo = torch.where(lo == 8, torch.full_like(lo, 9), lo)
    d_in = ((hi << 4) | lo).contiguous()
    d_neg = ((((16 - hi) & 15) << 4) | ((16 - lo) & 15)).to(torch.uint8).contiguous()
    n_out = n_units * g.out_per_gemm
    stream = torch.cuda.current_stream().cuda_stream
    d_a = torch.empty(n_out, dtype=torch.float32, device="cuda")
    d_b = torch.empty_like(d_a)
    bf.beamform(d_in, n_units, d_a, stream)
    bf.beamform(d_neg, n_units, d_b, stream)
    torch.cuda.synchronize()
    assert torch.equal(d_a, d_b)
    assert float(d_a.min()) >= 0.0 and bool(torch.isfinite(d_a).all())
    # unit-by-unit == batched
    d_c = torch.empty_like(d_a)
    for u in range(n_units):
        bf.beamform(d_in[u], 1, d_c[u * g.out_per_gemm:(u + 1) * g.out_per_gemm], stream)
    torch.cuda.synchronize()
    assert torch.equal(d_a, d_c)
    # zero input
    bf.beamform(torch.zeros_like(d_in), n_units, d_b, stream)
    torch.cuda.synchronize()
    assert float(d_b.abs().max()) == 0.0
    # sampled parity: frequencies 3, 130, 255 of units 0, 17, 31
    fs, us = [3, 130, 255], [0, 17, 31]
    gs = orc.Geom(n_avg=16, n_out_per_gemm=16, n_freq=len(fs))
    sub = d_in[us][:, fs].contiguous().cpu().numpy()
    want = orc.beamform(gs, np.ascontiguousarray(linear_weights[fs]), sub)
    got = d_a.view(n_units, g.n_out_per_gemm, g.n_freq, g.n_beams)[us][:, :, fs].cpu().numpy()
    assert np.array_equal(got, want)


def test_streaming_api_debug_flow(torch, bfmod, orc, linear_inputs, linear_weights):
    """The reference's DEBUG observation flow through the streaming entry points: pinned host batch ->
    bf_submit_block -> per-stream bf_enqueue_gemm_unit + bf_enqueue_dedisperse -> events (src/beamformer.cu:
    421-431,461-526).  First two PSRDADA-sized blocks (64 sources) reproduce the golden data.py rows."""
    from dsabeamformer_amd import api

    pos, _, src = linear_inputs
    g = orc.DEBUG_GEOM
    gold = np.load(os.path.join(GOLDEN, "linear_debug.npz"))
    cfg = bfmod.debug_config()
    bf = bfmod.Beamformer(cfg)
    bf.set_weights(linear_weights)
    n_blocks, per_block = 2, cfg.n_gemms_per_block
    nbytes = bf.bytes_per_block
    host = api.alloc_pinned(nbytes * n_blocks)
    batch = orc.generate_test_data(g, pos, src, 0, 0, per_block * n_blocks)
    C.memmove(host, batch.ctypes.data, batch.nbytes)
    ded_ptr = api.alloc_pinned(per_block * n_blocks * 256 * 4)
    out_ptr = api.alloc_pinned(bf.floats_per_detect * 4 * cfg.n_streams)
    evs = [api.event_create() for _ in range(n_blocks)]
    aev = [api.event_create() for _ in range(n_blocks)]
    for blk in range(n_blocks):
        bf.submit_block(blk, host + blk * nbytes, nbytes, evs[blk])
    for blk in range(n_blocks):
        while api.event_query(evs[blk]) != 0:
            pass
        for part in range(per_block // cfg.n_streams):
            for st in range(cfg.n_streams):
                ts = part * cfg.n_streams + st
                bf.enqueue_gemm_unit(st, blk, ts, out_ptr + st * bf.floats_per_detect * 4)
                bf.enqueue_dedisperse(st, ded_ptr + (blk * per_block + ts) * 256 * 4)
        bf.record_analysis_event(aev[blk])
    for blk in range(n_blocks):
        while api.event_query(aev[blk]) != 0:
            pass
    bf.sync()
    ded = np.ctypeslib.as_array(C.cast(ded_ptr, C.POINTER(C.c_float)), shape=(per_block * n_blocks, 256)).copy()
    assert np.array_equal(ded, gold["dedispersed"][:per_block * n_blocks])
    for e in evs + aev:
        api.event_destroy(e)
    for p in (host, ded_ptr, out_ptr):
        api.free_pinned(p)


def test_error_paths_on_device(torch, bfmod, orc):
    from dsabeamformer_amd._lib import DsabfError

    bf = bfmod.Beamformer(bfmod.debug_config(n_freq=2, n_beams=32))
    d = torch.zeros(2 * 16 * 64, dtype=torch.uint8, device="cuda")
    o = torch.zeros(8 * 2 * 32, dtype=torch.float32, device="cuda")
    with pytest.raises(DsabfError) as e:
        bf.beamform(d, 1, o)  # before set_weights
    assert e.value.code == -4
    w = np.zeros((2, 64, 32, 2), np.int8)
    w[1, 3, 5, 1] = -128
    with pytest.raises(DsabfError) as e:
        bf.set_weights(w)
    assert e.value.code == -1 and "-128" in str(e.value)
    w[1, 3, 5, 1] = -127
    bf.set_weights(w)
    bf.beamform(d, 1, o)
    torch.cuda.synchronize()


def test_beam_cli_writes_reference_data_py(tmp_path, orc):
    """The `beam` driver end to end = the reference's `make debug` run (src/beamformer.cu:12-621):
    bin/beam -p linear_positions -d linear_directions -s linear_source_directions_1024 -> data.py.
    The file must be byte-identical to the oracle's dedispersed table written by the reference's writer format."""
    import subprocess

    from conftest import CFG, ROOT

    out = tmp_path / "data.py"
    cmd = [os.path.join(ROOT, "dsabeamformer_amd", "beam"), "-g", "0", "-p", os.path.join(CFG, "linear_positions.txt"),
           "-d", os.path.join(CFG, "linear_directions.txt"), "-s", os.path.join(CFG, "linear_source_directions_1024.txt"),
           "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "obs Complete" in r.stdout and "Time per data chunk" in r.stdout
    assert "Code produced outputs for 8192 data chunks." in r.stdout  # 1024 sources x N_OUTPUTS_PER_GEMM
    gold = np.load(os.path.join(GOLDEN, "linear_debug.npz"))["dedispersed"]
    want = tmp_path / "want.py"
    orc.write_python_file(gold, str(want))
    assert out.read_text() == want.read_text()
    ns = {}
    exec(out.read_text(), ns)  # "written such that it can be imported into any python file" (beamformer.hh:291)
    assert np.array_equal(np.array(ns["A"], np.float32), np.array(eval(want.read_text()[4:]), np.float32))


def test_debug_observation_bogus_data(bfmod, orc):
    """No -s file: the generator's buffer stays BOGUS_DATA 0x70 = (7 + 0j) everywhere (test_data_generator.hh:8,36),
    n_pt_sources = 1024; every row of the dedispersed table is the same oracle-computable vector."""
    from dsabeamformer_amd import host

    cfg = bfmod.debug_config()
    ded, ms = host.run_debug_observation(cfg, gpu=0)
    assert ded.shape == (1024, 256) and ms > 0
    g = orc.DEBUG_GEOM
    w = orc.make_weights(g, orc.default_positions(64), orc.default_directions(256), 0)
    unit = np.full((1, g.n_freq, g.n_time, g.n_ant), 0x70, np.uint8)
    row = orc.dedisperse(g, orc.beamform(g, w, unit)[0])
    assert np.array_equal(ded, np.broadcast_to(row, ded.shape))


def test_whole_reference_batch_in_one_launch(torch, bfmod, orc, linear_inputs, linear_weights):
    """Maximum size of the reference's DEBUG flow: all 1024 gemm-units of one generator batch (256 MiB packed,
    2 GiB of detected output) in ONE launch; every unit's dedispersed row must equal the golden data.py table,
    and all 8 outputs of a unit must be identical (the reference's columns are)."""
    from dsabeamformer_amd import host

    pos, _, src = linear_inputs
    g = orc.DEBUG_GEOM
    gold = np.load(os.path.join(GOLDEN, "linear_debug.npz"))["dedispersed"]
    cfg = bfmod.debug_config()
    gen = host.TestDataGenerator(cfg)  # pinned 256 MiB, the product's own generator
    gen.set_source_directions(src)
    gen.generate_test_data(pos, 0)
    bf = bfmod.Beamformer(cfg)
    bf.set_weights(linear_weights)
    d_in = torch.from_numpy(gen.data()).cuda()
    n_units = 1024
    d_out = torch.empty(n_units * g.out_per_gemm, dtype=torch.float32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    bf.beamform(d_in, n_units, d_out, s)
    d_ded = torch.empty((n_units, g.n_beams), dtype=torch.float32, device="cuda")
    v = d_out.view(n_units, g.n_out_per_gemm, g.n_freq, g.n_beams)
    for u in range(n_units):
        bf.dedisperse(v[u], d_ded[u], s)
    torch.cuda.synchronize()
    assert np.array_equal(d_ded.cpu().numpy(), gold)
    assert bool((v[:, 1:] == v[:, :1]).all())
    gen.close()


def test_edge_geometries_and_arguments(torch, bfmod, orc):
    from dsabeamformer_amd._lib import DsabfError

    # one 32-beam tile (7 of the 8 waves of a workgroup idle), one frequency, one gemm-unit
    for n_avg, n_out in ((1, 8), (16, 1), (16, 3)):
        g = orc.Geom(n_beams=32, n_ant=64, n_freq=1, n_avg=n_avg, n_out_per_gemm=n_out)
        rng = np.random.default_rng(9)
        w = rng.integers(-127, 128, size=(1, 64, 32, 2), dtype=np.int8)
        packed = rng.integers(0, 256, size=(1, 1, g.n_time, 64), dtype=np.uint8)
        bf = bfmod.Beamformer(_cfg(bfmod, g))
        bf.set_weights(w)
        want = orc.beamform(g, w, packed)
        assert np.array_equal(_run(torch, bf, packed, want.size).reshape(want.shape), want)
        d = torch.zeros(16, dtype=torch.uint8, device="cuda")
        o = torch.zeros(16, dtype=torch.float32, device="cuda")
        with pytest.raises(DsabfError) as e:
            bf.beamform(d, 0, o)  # empty input is an error, not a silent no-op
        assert e.value.code == -1
        with pytest.raises(DsabfError):
            bf.beamform(d[1:], 1, o)  # misaligned input pointer


@pytest.mark.parametrize("launches", ["reference", "quarter-block", "default"])
def test_production_observation_loop_with_junk_source(bfmod, orc, monkeypatch, launches):
    """Observation (non-DEBUG) mode, src/beamformer.cu:364-534: production geometry (N_AVERAGING 16, 128 MiB blocks),
    blocks from the in-memory dada_junkdb stand-in, 8 compute queues.  After the run each queue's beam_out slot holds
    the detected powers of the last gemm-unit it processed: must equal the oracle on that gemm-unit's bytes.
    launches: the reference's one launch per gemm-unit, 8 gemm-units per launch, and the default: one launch per block."""
    from dsabeamformer_amd import host

    if launches == "reference":
        monkeypatch.setenv("DSABF_UNIT_LAUNCH", "1")
    elif launches == "quarter-block":
        monkeypatch.setenv("DSABF_UNITS_PER_LAUNCH", "8")
    cfg = bfmod.production_config()
    n_blocks, ring_blocks = 6, 3
    r = host.run_observation_junk(cfg, n_blocks, ring_blocks=ring_blocks, seed=7)
    assert r["ms"] > 0
    g = orc.PROD_GEOM
    w = orc.make_weights(g, orc.default_positions(64), orc.default_directions(256), 0)
    per_block = cfg.n_gemms_per_block
    last = r["last_gemm"].tolist()
    if launches == "reference":
        # every stream ends on the last block, time slices 24..31 (4 parts x 8 streams, src/beamformer.cu:454-519)
        assert sorted(last) == [(n_blocks - 1) * per_block + 24 + i for i in range(8)]
    elif launches == "quarter-block":
        # 4 launches of 8 gemm-units per block on consecutive queues: launch j on queue j % 8, its last unit 8 (j % 4) + 7
        assert last == [(16 + q) // 4 * per_block + 8 * ((16 + q) % 4) + 7 for q in range(8)]
    else:
        # whole blocks alternate between queues 0 and 1 (round 4; more queues only add concurrent host copies): queue 0 ends on
        # block 4, queue 1 on block 5, the others saw none
        assert last[:2] == [4 * per_block + 31, 5 * per_block + 31] and last[2:] == [-1] * 6
    for st in ((0, 1) if launches == "default" else (0, 3, 5)):
        blk, ts = divmod(int(last[st]), per_block)
        unit = r["ring"][blk % ring_blocks, ts][None]
        want = orc.beamform(g, w, unit)[0]
        assert np.array_equal(r["beam_out"][st], want), st


@pytest.mark.parametrize("n_avg", [8, 16, 32])
def test_fast_detect_mode_within_stated_tolerance(torch, bfmod, orc, n_avg):
    """BF_DETECT_FAST (opt-in): fma-contracted detect, 4 VALU ops per sample instead of 6.  Stated tolerance vs the
    canonical (bit-exact) result: 4 * n_ipo * 2^-24 relative (every term is non-negative, so the bound is on the sum)."""
    n_ipo = 2 * n_avg
    g = orc.Geom(n_beams=128, n_ant=64, n_freq=6, n_avg=n_avg, n_out_per_gemm=4)
    rng = np.random.default_rng(77 + n_avg)
    w = rng.integers(-127, 128, size=(g.n_freq, g.n_ant, g.n_beams, 2), dtype=np.int8)
    packed = rng.integers(0, 256, size=(3, g.n_freq, g.n_time, g.n_ant), dtype=np.uint8)
    want = orc.beamform(g, w, packed)
    bf = bfmod.Beamformer(_cfg(bfmod, g, detect_mode=1))
    assert "FAST" in bf.kernel_info()["kernel"]
    bf.set_weights(w)
    got = _run(torch, bf, packed, want.size).reshape(want.shape)
    tol = 4 * n_ipo * 2.0 ** -24
    rel = np.abs(got.astype(np.float64) - want) / np.maximum(want.astype(np.float64), 1e-30)
    assert rel.max() <= tol, (rel.max(), tol)
    assert not np.array_equal(got, want)  # it really is the other arithmetic
    # against exact arithmetic (integer voltages, fl(1/127)^2 applied in double): both modes round the running fp32 sum
    # once or twice per sample, so both stay within a few n_ipo * 2^-24 of the exact value
    v = orc.expand(packed).astype(np.int64)                                  # [unit][f][t][a][2]
    wc = w.astype(np.int64)
    cre = np.einsum("ufta,fab->uftb", v[..., 0], wc[..., 0]) - np.einsum("ufta,fab->uftb", v[..., 1], wc[..., 1])
    cim = np.einsum("ufta,fab->uftb", v[..., 0], wc[..., 1]) + np.einsum("ufta,fab->uftb", v[..., 1], wc[..., 0])
    p = (cre * cre + cim * cim).reshape(3, g.n_freq, g.n_out_per_gemm, n_ipo, g.n_beams).sum(3)      # exact integers
    exact = p.transpose(0, 2, 1, 3).astype(np.float64) * (float(np.float32(1.0 / 127.0)) ** 2)
    e_fast = np.abs(got.astype(np.float64) - exact.reshape(want.shape)) / exact.reshape(want.shape)
    e_canon = np.abs(want.astype(np.float64) - exact.reshape(want.shape)) / exact.reshape(want.shape)
    assert e_fast.max() <= 2 * n_ipo * 2.0 ** -24 and e_canon.max() <= 2 * n_ipo * 2.0 ** -24
    # the default mode on the same input stays bit-exact
    bf0 = bfmod.Beamformer(_cfg(bfmod, g))
    bf0.set_weights(w)
    assert np.array_equal(_run(torch, bf0, packed, want.size).reshape(want.shape), want)


@pytest.mark.parametrize("n_ant,n_avg", [(64, 1), (64, 2), (64, 4), (64, 8), (64, 16), (64, 32), (128, 1), (128, 16),
                                         (100, 16), (32, 16), (16, 1)])
@pytest.mark.parametrize("tsplit", ["1", "3"])
def test_many_chunks_per_workgroup_bit_exact(torch, bfmod, orc, monkeypatch, n_ant, n_avg, tsplit):
    """Long time ranges per workgroup (>= 5 LDS chunks, forced with the DSABF_TSPLIT tuning override): exercises the
    double-buffered staging, the deferred stores across chunk boundaries and outputs that span chunks (n_ipo 64).
    (Round 1 found a deferred-store bookkeeping bug for n_ipo = 16 that only shows with > 1 chunk per workgroup.)"""
    monkeypatch.setenv("DSABF_TSPLIT", tsplit)
    n_ipo = 2 * n_avg
    n_out = max(2, 16 // n_ipo)
    g = orc.Geom(n_beams=64, n_ant=n_ant, n_freq=3, n_avg=n_avg, n_out_per_gemm=n_out)
    n_units = max(3, -(-(5 * 128 * (2 if n_ipo == 64 else 1) + 64) // g.n_time))  # > 5 chunks, ragged end
    rng = np.random.default_rng(31 + n_ant + n_avg)
    w = rng.integers(-127, 128, size=(g.n_freq, g.n_ant, g.n_beams, 2), dtype=np.int8)
    packed = rng.integers(0, 256, size=(n_units, g.n_freq, g.n_time, g.n_ant), dtype=np.uint8)
    bf = bfmod.Beamformer(_cfg(bfmod, g))
    bf.set_weights(w)
    want = orc.beamform(g, w, packed)
    got = _run(torch, bf, packed, want.size).reshape(want.shape)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("n_ant,n_avg,n_out", [(64, 16, 24), (64, 1, 40), (64, 8, 20), (128, 16, 12), (100, 16, 12)])
def test_gemm_stage_many_chunks_bit_exact(torch, bfmod, orc, n_ant, n_avg, n_out):
    """Stage parity of a2 (the reference's d_C) over several LDS chunks and both MFMA shapes."""
    g = orc.Geom(n_beams=64, n_ant=n_ant, n_freq=2, n_avg=n_avg, n_out_per_gemm=n_out)
    rng = np.random.default_rng(5 + n_ant + n_avg)
    w = rng.integers(-127, 128, size=(g.n_freq, g.n_ant, g.n_beams, 2), dtype=np.int8)
    packed = rng.integers(0, 256, size=(g.n_freq, g.n_time, g.n_ant), dtype=np.uint8)
    bf = bfmod.Beamformer(_cfg(bfmod, g))
    bf.set_weights(w)
    d_in = torch.from_numpy(packed).cuda()
    d_c = torch.full((g.n_freq * g.n_time * g.n_beams * 2,), float("nan"), dtype=torch.float32, device="cuda")
    bf.gemm(d_in, d_c, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    want = orc.gemm(g, w, orc.expand(packed))
    assert np.array_equal(d_c.cpu().numpy().reshape(want.shape), want)


@pytest.mark.parametrize("n_freq,n_beams,n_avg", [(8, 512, 16), (16, 288, 1), (24, 512, 8), (8, 32, 32), (3, 320, 16),
                                                  (2, 576, 1), (4, 64, 4)])
def test_xcd_block_map_and_beam_groups_bit_exact(torch, bfmod, orc, n_freq, n_beams, n_avg):
    """n_freq % 8 == 0 takes the XCD-aware block decode; n_beams > 256 gives several beam groups per frequency,
    n_beams = 288 a partially filled last group; several time splits and chunks per workgroup."""
    n_ipo = 2 * n_avg
    g = orc.Geom(n_beams=n_beams, n_ant=64, n_freq=n_freq, n_avg=n_avg, n_out_per_gemm=max(2, 16 // n_ipo))
    n_units = -(-1100 // g.n_time)
    rng = np.random.default_rng(n_freq * 1000 + n_beams)
    w = rng.integers(-127, 128, size=(g.n_freq, g.n_ant, g.n_beams, 2), dtype=np.int8)
    packed = rng.integers(0, 256, size=(n_units, g.n_freq, g.n_time, g.n_ant), dtype=np.uint8)
    bf = bfmod.Beamformer(_cfg(bfmod, g))
    bf.set_weights(w)
    want = orc.beamform(g, w, packed)
    got = _run(torch, bf, packed, want.size).reshape(want.shape)
    assert np.array_equal(got, want)


def test_seeded_geometry_fuzz_bit_exact(torch, bfmod, orc):
    """Seeded fuzz over the supported geometry space (shape, units, ragged ends); every case bit-exact vs the oracle."""
    rng = np.random.default_rng(20261002)
    combos = [(64, a) for a in (1, 2, 4, 8, 16, 32)] + [(16, 1), (16, 16), (32, 1), (32, 16), (100, 1), (100, 16),
                                                        (128, 1), (128, 16)]
    for case in range(14):
        n_ant, n_avg = combos[int(rng.integers(len(combos)))]
        n_ipo = 2 * n_avg
        n_out = int(rng.integers(1, 5)) * max(1, 16 // n_ipo)
        g = orc.Geom(n_beams=32 * int(rng.integers(1, 11)), n_ant=n_ant, n_freq=int(rng.integers(1, 12)), n_avg=n_avg,
                     n_out_per_gemm=n_out)
        n_units = int(rng.integers(1, 1 + max(1, 700 // g.n_time)))
        w = rng.integers(-127, 128, size=(g.n_freq, g.n_ant, g.n_beams, 2), dtype=np.int8)
        packed = rng.integers(0, 256, size=(n_units, g.n_freq, g.n_time, g.n_ant), dtype=np.uint8)
        bf = bfmod.Beamformer(_cfg(bfmod, g))
        bf.set_weights(w)
        want = orc.beamform(g, w, packed)
        got = _run(torch, bf, packed, want.size).reshape(want.shape)
        assert np.array_equal(got, want), (case, g, n_units)
        bf.close()


def _conj_symmetric(w):
    """Make beam B-1-b the complex conjugate of beam b (what a beam set symmetric about the boresight produces)."""
    w = w.copy()
    B = w.shape[2]
    w[:, :, B // 2:, 0] = w[:, :, :B // 2, 0][:, :, ::-1]
    w[:, :, B // 2:, 1] = -w[:, :, :B // 2, 1][:, :, ::-1]
    return w


@pytest.mark.parametrize("n_avg", [1, 2, 4, 8, 16, 32])
@pytest.mark.parametrize("tsplit", ["1", "3"])
def test_conjugate_paired_kernel_bit_exact(torch, bfmod, orc, monkeypatch, n_avg, tsplit):
    """Conjugate-symmetric weights select fused16_kernel<..., PAIRED> (half the MFMA work); its results must be the
    same bits as the oracle's and as the general kernel's (DSABF_PAIRED=0), over several chunks per workgroup."""
    monkeypatch.setenv("DSABF_TSPLIT", tsplit)
    n_ipo = 2 * n_avg
    g = orc.Geom(n_beams=96, n_ant=64, n_freq=3, n_avg=n_avg, n_out_per_gemm=max(2, 16 // n_ipo))
    n_units = max(3, -(-(5 * 128 * (2 if n_ipo == 64 else 1) + 64) // g.n_time))
    rng = np.random.default_rng(77 + n_avg)
    w = _conj_symmetric(rng.integers(-127, 128, size=(g.n_freq, g.n_ant, g.n_beams, 2), dtype=np.int8))
    packed = rng.integers(0, 256, size=(n_units, g.n_freq, g.n_time, g.n_ant), dtype=np.uint8)
    want = orc.beamform(g, w, packed)
    bf = bfmod.Beamformer(_cfg(bfmod, g))
    bf.set_weights(w)
    assert "PAIRED" in bf.kernel_info(n_units)["kernel"]
    assert np.array_equal(_run(torch, bf, packed, want.size).reshape(want.shape), want)
    monkeypatch.setenv("DSABF_PAIRED", "0")
    bf.set_weights(w)
    assert "PAIRED" not in bf.kernel_info(n_units)["kernel"]
    assert np.array_equal(_run(torch, bf, packed, want.size).reshape(want.shape), want)


def test_pairing_is_decided_per_weight_set(torch, bfmod, orc):
    """One element off the symmetry -> the general kernel; symmetric again -> the paired one; both bit-exact.  The
    reference's own default geometry (linear fan about the boresight) is symmetric, so the product path pairs."""
    g = orc.Geom(n_beams=64, n_ant=64, n_freq=4, n_avg=16, n_out_per_gemm=2)
    rng = np.random.default_rng(5150)
    ws = _conj_symmetric(rng.integers(-127, 128, size=(g.n_freq, g.n_ant, g.n_beams, 2), dtype=np.int8))
    packed = rng.integers(0, 256, size=(3, g.n_freq, g.n_time, g.n_ant), dtype=np.uint8)
    bf = bfmod.Beamformer(_cfg(bfmod, g))
    for f, a, b, c in [(3, 63, 63, 1), (0, 0, 0, 0), (2, 17, 31, 1)]:
        wb = ws.copy()
        wb[f, a, b, c] = np.int8(wb[f, a, b, c] + 1 if wb[f, a, b, c] < 127 else 126)  # its partner stays
        bf.set_weights(wb)
        assert "PAIRED" not in bf.kernel_info(3)["kernel"], (f, a, b, c)
        want = orc.beamform(g, wb, packed)
        assert np.array_equal(_run(torch, bf, packed, want.size).reshape(want.shape), want)
    bf.set_weights(ws)
    assert "PAIRED" in bf.kernel_info(3)["kernel"]
    want = orc.beamform(g, ws, packed)
    assert np.array_equal(_run(torch, bf, packed, want.size).reshape(want.shape), want)
    from dsabeamformer_amd import host
    gd = orc.DEBUG_GEOM
    bfd = bfmod.Beamformer(_cfg(bfmod, gd))
    bfd.set_weights(host.make_weights_default(gd.n_beams, gd.n_ant, gd.n_freq))
    assert "PAIRED" in bfd.kernel_info(1)["kernel"]


@pytest.mark.parametrize("n_freq,n_beams,n_avg", [(8, 512, 16), (16, 288, 1), (24, 512, 8), (8, 32, 32), (5, 544, 16),
                                                  (3, 320, 16), (2, 576, 1), (4, 64, 4)])
def test_conjugate_paired_beam_groups_bit_exact(torch, bfmod, orc, n_freq, n_beams, n_avg):
    """Paired kernel across several beam groups (base beams of one workgroup pair with the far end of the beam axis),
    a partially filled last group, and the XCD-aware block decode."""
    n_ipo = 2 * n_avg
    g = orc.Geom(n_beams=n_beams, n_ant=64, n_freq=n_freq, n_avg=n_avg, n_out_per_gemm=max(2, 16 // n_ipo))
    n_units = -(-1100 // g.n_time)
    rng = np.random.default_rng(n_freq * 1000 + n_beams + 1)
    w = _conj_symmetric(rng.integers(-127, 128, size=(g.n_freq, g.n_ant, g.n_beams, 2), dtype=np.int8))
    packed = rng.integers(0, 256, size=(n_units, g.n_freq, g.n_time, g.n_ant), dtype=np.uint8)
    bf = bfmod.Beamformer(_cfg(bfmod, g))
    bf.set_weights(w)
    assert "PAIRED" in bf.kernel_info(n_units)["kernel"]
    want = orc.beamform(g, w, packed)
    assert np.array_equal(_run(torch, bf, packed, want.size).reshape(want.shape), want)


@pytest.mark.parametrize("n_avg", [8, 16, 32])
def test_fast_detect_paired_equals_fast_general(torch, bfmod, orc, monkeypatch, n_avg):
    """BF_DETECT_FAST on the paired kernel: the integer part is exact, so it gives the same bits as FAST on the general
    kernel (and both stay within the stated tolerance of the canonical result)."""
    g = orc.Geom(n_beams=64, n_ant=64, n_freq=3, n_avg=n_avg, n_out_per_gemm=2)
    rng = np.random.default_rng(909 + n_avg)
    w = _conj_symmetric(rng.integers(-127, 128, size=(g.n_freq, g.n_ant, g.n_beams, 2), dtype=np.int8))
    packed = rng.integers(0, 256, size=(5, g.n_freq, g.n_time, g.n_ant), dtype=np.uint8)
    want = orc.beamform(g, w, packed)
    bf = bfmod.Beamformer(_cfg(bfmod, g, detect_mode=1))
    bf.set_weights(w)
    assert "FAST,PAIRED" in bf.kernel_info(5)["kernel"]
    got_p = _run(torch, bf, packed, want.size).reshape(want.shape)
    monkeypatch.setenv("DSABF_PAIRED", "0")
    bf.set_weights(w)
    got_g = _run(torch, bf, packed, want.size).reshape(want.shape)
    assert np.array_equal(got_p, got_g)
    rel = np.abs(got_p.astype(np.float64) - want) / np.maximum(want.astype(np.float64), 1e-30)
    assert rel.max() <= 4 * g.n_ipo * 2.0 ** -24


def test_observation_detected_stream_to_file_bit_exact(bfmod, orc, tmp_path):
    """the production loop with a file sink keeps EVERY gemm-unit's detected powers (the reference
    overwrites beam_out and drops them).  Small geometry so the oracle can check the whole stream bit for bit; the
    gemm-units reach the file in index order although the 4 queues finish them in time-slice order."""
    from dsabeamformer_amd import host

    cfg = bfmod.production_config(n_freq=8)
    cfg.n_beams, cfg.n_gemms_per_block, cfg.n_streams = 64, 8, 4
    n_blocks, ring_blocks = 7, 3
    path = str(tmp_path / "detected.bin")
    r = host.run_observation_junk_to_file(cfg, n_blocks, path, ring_blocks=ring_blocks, seed=99, gpu=2)
    assert r["gemms_written"] == n_blocks * cfg.n_gemms_per_block
    hdr, data = host.read_detected_file(path)
    assert hdr["GPU"] == "2" and data.shape == (n_blocks * 8, cfg.n_out_per_gemm, 8, 64)
    g = orc.Geom(n_beams=64, n_ant=64, n_freq=8, n_avg=16, n_out_per_gemm=cfg.n_out_per_gemm)
    pos, dirs = orc.default_positions(64), orc.default_directions(64)
    w = orc.make_weights(g, pos, dirs, 2)
    for blk in range(n_blocks):
        want = orc.beamform(g, w, r["ring"][blk % ring_blocks])       # [8 units * n_out][f][b]
        got = data[blk * 8:(blk + 1) * 8].reshape(want.shape)
        assert np.array_equal(got, want), blk


def test_observation_production_size_to_file_sampled(bfmod, orc, tmp_path):
    """Same at the reference's production geometry (128 MiB blocks, 32 gemm-units of 2 MiB detected each, 8 queues):
    file size, and sampled gemm-units against the oracle."""
    from dsabeamformer_amd import host

    cfg = bfmod.production_config()
    n_blocks, ring_blocks = 3, 2
    path = str(tmp_path / "detected_prod.bin")
    r = host.run_observation_junk_to_file(cfg, n_blocks, path, ring_blocks=ring_blocks, seed=5)
    per = cfg.n_gemms_per_block
    assert r["gemms_written"] == n_blocks * per
    hdr, data = host.read_detected_file(path)
    assert data.shape == (n_blocks * per, cfg.n_out_per_gemm, 256, 256)
    g = orc.PROD_GEOM
    w = orc.make_weights(g, orc.default_positions(64), orc.default_directions(256), 0)
    for gi in (0, 31, 32, 50, n_blocks * per - 1):
        blk, ts = divmod(gi, per)
        want = orc.beamform(g, w, r["ring"][blk % ring_blocks, ts][None])
        assert np.array_equal(data[gi], want.reshape(data[gi].shape)), gi


def test_beam_cli_junk_mode_writes_detected_file(tmp_path):
    import subprocess

    from conftest import ROOT
    from dsabeamformer_amd import host

    exe = os.path.join(ROOT, "dsabeamformer_amd", "beam")
    path = str(tmp_path / "d.bin")
    # 25 of the 27 blocks go to the burn-in reads (BURNIN, src/beamformer.hh:45), 2 are analysed
    p = subprocess.run([exe, "-j", "27", "-w", path], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    assert "Wrote 64 gemm-units of detected powers" in p.stdout
    hdr, data = host.read_detected_file(path)
    assert data.shape[0] == 64 and np.isfinite(data).all() and data.max() > 0


@pytest.mark.parametrize("n_avg,n_out", [(32, 4), (32, 6), (32, 8), (16, 4), (16, 5), (16, 12), (8, 8), (8, 24), (4, 16),
                                         (1, 64), (1, 192), (2, 36)])
@pytest.mark.parametrize("paired", [False, True])
def test_scalar_and_generic_chunk_addressing_bit_exact(torch, bfmod, orc, monkeypatch, n_avg, n_out, paired):
    """fused16_kernel addresses a chunk through a scalar base when gemm-units are a multiple of the chunk span
    (128 samples; 256 for n_ipo = 64) and through the generic per-row path otherwise; power-of-two and other unit
    lengths, several units per workgroup range, ragged ends, both kernels."""
    monkeypatch.setenv("DSABF_TSPLIT", "2")
    g = orc.Geom(n_beams=64, n_ant=64, n_freq=2, n_avg=n_avg, n_out_per_gemm=n_out)
    n_units = -(-1700 // g.n_time)
    rng = np.random.default_rng(1000 * n_avg + n_out)
    w = rng.integers(-127, 128, size=(g.n_freq, g.n_ant, g.n_beams, 2), dtype=np.int8)
    if paired:
        w = _conj_symmetric(w)
    packed = rng.integers(0, 256, size=(n_units, g.n_freq, g.n_time, g.n_ant), dtype=np.uint8)
    bf = bfmod.Beamformer(_cfg(bfmod, g))
    bf.set_weights(w)
    assert ("PAIRED" in bf.kernel_info(n_units)["kernel"]) == paired
    want = orc.beamform(g, w, packed)
    assert np.array_equal(_run(torch, bf, packed, want.size).reshape(want.shape), want)


def test_observation_from_shared_memory_ring_bit_exact(bfmod, orc, tmp_path):
    """SURVEY 8f-3 end to end: `junkdb` (writer process) feeds production-size blocks (128 MiB) through a 3-slot
    shared-memory ring, `beam -k <ring> -w <file>` attaches, page-locks the blocks (hipHostRegister, the reference's
    dada_cuda_dbregister), runs the production observation loop until the short block and keeps the detected stream;
    sampled gemm-units must equal the oracle on the junk bytes."""
    import subprocess

    from conftest import ROOT
    from dsabeamformer_amd import host

    cfg = bfmod.production_config()
    name, n_blocks, distinct, seed = "dsabf_gpu_%d" % os.getpid(), 5, 2, 31
    path = str(tmp_path / "detected_shm.bin")
    w = subprocess.Popen([os.path.join(ROOT, "dsabeamformer_amd", "junkdb"), "-k", name, "-n", str(n_blocks), "-r", "3",
                          "-d", str(distinct), "-s", str(seed)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    try:
        r = subprocess.run([os.path.join(ROOT, "dsabeamformer_amd", "beam"), "-k", name, "-w", path],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "block size is: 134217728" in r.stdout
        assert "Wrote %d gemm-units" % (n_blocks * cfg.n_gemms_per_block) in r.stdout
        wout, werr = w.communicate(timeout=60)
        assert w.returncode == 0, werr
    finally:
        if w.poll() is None:
            w.kill()
        host.shm_ring_unlink(name)
    hdr, data = host.read_detected_file(path)
    per = cfg.n_gemms_per_block
    assert data.shape == (n_blocks * per, cfg.n_out_per_gemm, 256, 256)
    junk = host.junk_bytes(134217728, distinct, seed).reshape(distinct, per, 256, -1, 64)
    g = orc.PROD_GEOM
    wts = orc.make_weights(g, orc.default_positions(64), orc.default_directions(256), 0)
    for gi in (0, 33, 95, n_blocks * per - 1):
        blk, ts = divmod(gi, per)
        want = orc.beamform(g, wts, junk[blk % distinct, ts][None])
        assert np.array_equal(data[gi], want.reshape(data[gi].shape)), gi


def test_observation_shm_python_api_small_blocks(bfmod, orc, tmp_path):
    """Same path through the C wrapper with a small geometry: writer thread in this process, whole stream checked."""
    import threading

    from dsabeamformer_amd import host

    cfg = bfmod.production_config(n_freq=4)
    cfg.n_beams, cfg.n_gemms_per_block, cfg.n_streams = 64, 4, 2
    n_time = cfg.n_out_per_gemm * cfg.n_pol * cfg.n_avg
    block_bytes = cfg.n_gemms_per_block * cfg.n_freq * n_time * cfg.n_ant
    name, n_blocks = "dsabf_gpu_small_%d" % os.getpid(), 9
    rng = np.random.default_rng(3)
    blocks = rng.integers(0, 256, size=(n_blocks, cfg.n_gemms_per_block, cfg.n_freq, n_time, cfg.n_ant), dtype=np.uint8)
    ring = host.ShmRing(name, n_blocks=3, block_size=block_bytes, header="HDR_SIZE 4096\n")

    def writer():
        for b in blocks:
            ring.write(b.reshape(-1))
        ring.write(np.zeros(0, np.uint8))

    t = threading.Thread(target=writer)
    t.start()
    path = str(tmp_path / "d.bin")
    try:
        r = host.run_observation_shm(cfg, name, path=path, gpu=1)
        t.join(timeout=30)
        assert r["gemms"] == n_blocks * cfg.n_gemms_per_block
    finally:
        ring.detach()
        ring.unlink()
    hdr, data = host.read_detected_file(path)
    g = orc.Geom(n_beams=64, n_ant=64, n_freq=4, n_avg=16, n_out_per_gemm=cfg.n_out_per_gemm)
    wts = orc.make_weights(g, orc.default_positions(64), orc.default_directions(64), 1)
    for b in range(n_blocks):
        want = orc.beamform(g, wts, blocks[b])
        assert np.array_equal(data[b * 4:(b + 1) * 4].reshape(want.shape), want), b


@pytest.mark.parametrize("n_t,n_f,n_b,n_dm,tsamp", [(64, 16, 32, 5, 8.0), (300, 256, 256, 12, 0.131), (37, 5, 96, 3, 4.0),
                                                    (129, 32, 544, 7, 2.0)])
def test_dedisperse_dm_bit_exact(torch, bfmod, orc, n_t, n_f, n_b, n_dm, tsamp):
    """8f-4: out[dm][t][b] = sum_f series[t + delay[dm][f]][f][b], ascending-f fp32: bit-exact vs the oracle, for
    complete sums (n_t_out = n_t - max delay) and for the ragged tail (n_t_out = n_t: rows past the end are skipped)."""
    from dsabeamformer_amd import host

    g = orc.Geom(n_beams=n_b, n_ant=64, n_freq=n_f, n_avg=1, n_out_per_gemm=8)
    bf = bfmod.Beamformer(_cfg(bfmod, g))
    rng = np.random.default_rng(n_t * n_f)
    series = (rng.random((n_t, n_f, n_b), dtype=np.float32) * 1e5).astype(np.float32)
    freq = np.array([host.channel_frequency(0, c * (256 // n_f) if n_f <= 256 else c) for c in range(n_f)], np.float32)
    dms = np.linspace(0.0, 180.0, n_dm)
    delays = host.dm_delays(dms, freq, float(freq[0]), tsamp)
    assert delays.min() == 0 and 0 < delays.max() < n_t
    d_series = torch.from_numpy(series).cuda()
    d_delays = torch.from_numpy(delays).cuda()
    for n_t_out in (n_t - int(delays.max()), n_t):
        d_out = torch.full((n_dm, n_t_out, n_b), float("nan"), dtype=torch.float32, device="cuda")
        bf.dedisperse_dm(d_series, n_t, d_delays, n_dm, n_t_out, d_out, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        want = orc.dedisperse_dm(series, delays, n_t_out)
        assert np.array_equal(d_out.cpu().numpy(), want), n_t_out


@pytest.mark.parametrize("case", ["fine_ladder", "negative_delays", "many_channels", "one_wide_gap"])
def test_dedisperse_dm_regular_and_irregular_tiles_bit_exact(torch, bfmod, orc, case):
    """The kernel has a branch-free loop for regular tiles (every trial of a block inside the block's window at every
    channel: a fine DM ladder) and a general loop for the rest; both must give the oracle's bits, including where regular
    tiles meet the end of the series (windows partly and wholly past the last row, dropped by the buffer range check), a
    partial trial block, a partial beam group, delays that go negative (reference channel mid-band), more channels than the
    in-LDS offset table holds, and a ladder that is fine except for one gap."""
    from dsabeamformer_amd import host

    n_t, n_f, n_b, n_dm = 160, 64, 96, 13
    if case == "many_channels":
        n_t, n_f, n_b, n_dm = 40, 1040, 32, 9
    g = orc.Geom(n_beams=n_b, n_ant=64, n_freq=n_f if n_f <= 256 else 256, n_avg=1, n_out_per_gemm=8)
    cfg = _cfg(bfmod, g)
    cfg.n_freq = n_f
    bf = bfmod.Beamformer(cfg)
    rng = np.random.default_rng(7)
    series = (rng.random((n_t, n_f, n_b), dtype=np.float32) * 1e5).astype(np.float32)
    freq = np.linspace(1.53, 1.28, n_f).astype(np.float32)
    dms = np.arange(n_dm) * 0.25                        # neighbouring trials: <= 2 samples apart per channel
    if case == "one_wide_gap":
        dms[7:] += 15.0
    ref = float(freq[n_f // 2]) if case == "negative_delays" else float(freq[0])
    delays = host.dm_delays(dms, freq, ref, 0.131 if case != "many_channels" else 1.0)
    if case == "negative_delays":
        assert delays.min() < 0
    if case == "fine_ladder":
        assert (delays[1:] - delays[:-1]).max() <= 2 and (delays[3] - delays[0]).max() <= 8 and delays.max() > 8   # regular tiles
    d_series = torch.from_numpy(series).cuda()
    d_delays = torch.from_numpy(delays).cuda()
    for n_t_out in (max(1, n_t - int(delays.max())), n_t):
        d_out = torch.full((n_dm, n_t_out, n_b), float("nan"), dtype=torch.float32, device="cuda")
        bf.dedisperse_dm(d_series, n_t, d_delays, n_dm, n_t_out, d_out, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        want = orc.dedisperse_dm(series, delays, n_t_out)
        assert np.array_equal(d_out.cpu().numpy(), want), (case, n_t_out)


def test_dedisperse_dm_recovers_dispersed_pulse_from_detected_stream(torch, bfmod, orc):
    """End to end on the product path: a source whose voltage burst arrives later at lower frequencies (the sample
    delays of DM 120) goes through the fused kernel; the DM ladder's matched trial collects the whole burst in one
    sample of the right beam, DM 0 does not.  Every trial equals the oracle on the same detected series."""
    from dsabeamformer_amd import host

    g = orc.Geom(n_beams=64, n_ant=64, n_freq=32, n_avg=16, n_out_per_gemm=8)
    n_units, n_t = 12, 96
    pos, dirs = orc.default_positions(64), orc.default_directions(64)
    w = orc.make_weights(g, pos, dirs, 0)
    freq = np.array([host.channel_frequency(0, c) for c in range(g.n_freq)], np.float32)
    tsamp_ms = 0.131                                                       # the production output sample time
    dms = np.array([0.0, 60.0, 120.0, 180.0])
    delays = host.dm_delays(dms, freq, float(freq[0]), tsamp_ms)
    assert delays[2].max() >= 5
    src = np.array([[dirs[40][0], 0.0]], np.float32)   # a source in the direction of beam 40
    tone = orc.generate_test_data(g, pos, src, 0, n_units=1)[0]           # [f][t][a], the same column at every t
    packed = np.zeros((n_units, g.n_freq, g.n_time, g.n_ant), np.uint8)
    t_burst = 20
    for f in range(g.n_freq):
        o = t_burst + int(delays[2, f])                                    # output sample in which channel f is lit
        u, oo = divmod(o, g.n_out_per_gemm)
        packed[u, f, oo * g.n_ipo:(oo + 1) * g.n_ipo] = tone[f, 0]
    bf = bfmod.Beamformer(_cfg(bfmod, g))
    bf.set_weights(w)
    want_series = orc.beamform(g, w, packed)                               # [n_t][f][b]
    d_in = torch.from_numpy(packed).cuda()
    d_series = torch.empty(n_t * g.n_freq * g.n_beams, dtype=torch.float32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    bf.beamform(d_in, n_units, d_series, s)
    n_t_out = n_t - int(delays.max())
    d_out = torch.empty((len(dms), n_t_out, g.n_beams), dtype=torch.float32, device="cuda")
    bf.dedisperse_dm(d_series, n_t, torch.from_numpy(delays).cuda(), len(dms), n_t_out, d_out, s)
    torch.cuda.synchronize()
    got = d_out.cpu().numpy()
    assert np.array_equal(d_series.cpu().numpy().reshape(want_series.shape), want_series)
    assert np.array_equal(got, orc.dedisperse_dm(want_series.reshape(n_t, g.n_freq, g.n_beams), delays, n_t_out))
    dm_best, t_best, b_best = np.unravel_index(np.argmax(got), got.shape)
    assert (dm_best, t_best) == (2, t_burst) and abs(int(b_best) - 40) <= 1
    assert got[2, t_burst, b_best] > 3 * got[0].max()


@pytest.mark.parametrize("n_ant,n_avg", [(128, 16), (128, 1), (100, 16), (100, 1), (32, 16), (32, 1), (16, 16), (16, 1)])
@pytest.mark.parametrize("paired", [False, True])
@pytest.mark.parametrize("tsplit", ["1", "3"])
def test_wide_antenna_16x16_kernel_bit_exact(torch, bfmod, orc, monkeypatch, n_ant, n_avg, paired, tsplit):
    """Antenna counts other than 64 on fused16_kernel: 100 and 128 antennas are two k-steps of 64 (two LDS planes,
    chained MFMAs), 100-byte rows are staged in dwords, 16 / 32 / 100 antennas have zero weights behind the last
    antenna; general and conjugate-pair variants, several chunks per workgroup, several beam groups, ragged end.
    Extreme weights and voltages exercise the +-2^22 accumulator range."""
    monkeypatch.setenv("DSABF_TSPLIT", tsplit)
    n_ipo = 2 * n_avg
    g = orc.Geom(n_beams=288, n_ant=n_ant, n_freq=3, n_avg=n_avg, n_out_per_gemm=max(2, 16 // n_ipo) * (3 if n_avg > 1 else 1))
    n_units = max(2, -(-700 // g.n_time))
    rng = np.random.default_rng(n_ant + n_avg + 17 * paired)
    w = rng.integers(-127, 128, size=(g.n_freq, g.n_ant, g.n_beams, 2), dtype=np.int8)
    w[0, :, 5] = 127                      # all-max weights on one beam ...
    if paired:
        w = _conj_symmetric(w)
    packed = rng.integers(0, 256, size=(n_units, g.n_freq, g.n_time, g.n_ant), dtype=np.uint8)
    packed[0, 0, :4] = 0x88               # ... against all (-8, -8) voltages: the largest |sum| the path can see
    bf = bfmod.Beamformer(_cfg(bfmod, g))
    bf.set_weights(w)
    name = bf.kernel_info(n_units)["kernel"]
    assert "fused16_kernel<ANT=%d" % n_ant in name and ("PAIRED" in name) == paired
    want = orc.beamform(g, w, packed)
    assert np.array_equal(_run(torch, bf, packed, want.size).reshape(want.shape), want)


def test_wide_antenna_fast_detect_tolerance(torch, bfmod, orc):
    g = orc.Geom(n_beams=64, n_ant=128, n_freq=2, n_avg=16, n_out_per_gemm=4)
    rng = np.random.default_rng(4)
    w = _conj_symmetric(rng.integers(-127, 128, size=(g.n_freq, g.n_ant, g.n_beams, 2), dtype=np.int8))
    packed = rng.integers(0, 256, size=(3, g.n_freq, g.n_time, g.n_ant), dtype=np.uint8)
    bf = bfmod.Beamformer(_cfg(bfmod, g, detect_mode=1))
    bf.set_weights(w)
    assert "FAST,PAIRED" in bf.kernel_info(3)["kernel"]
    want = orc.beamform(g, w, packed)
    got = _run(torch, bf, packed, want.size).reshape(want.shape)
    rel = np.abs(got.astype(np.float64) - want) / np.maximum(want.astype(np.float64), 1e-30)
    assert rel.max() <= 4 * g.n_ipo * 2.0 ** -24


@pytest.mark.parametrize("gpu", sweep([0, 5], [5]))
def test_grid_configuration_debug_flow_bit_identical(bfmod, orc, gpu):
    """The reference's 2-D configuration end to end through the DEBUG flow (`beam -p grid_positions -d
    grid_beam_directions -s grid_source_directions_4096 -g <gpu>`): 8x8 antenna grid, 16x16 beam grid (conjugate-
    symmetric -> the paired kernel), 4096 point sources = 4 generator batches with re-generation gating
    (src/test_data_generator.hh:44-61), a non-zero sub-band (-g 5: integer-division channel offset,
    src/beamformer.cu:233).  The dedispersed table [4096][256] must equal the oracle's, row for row."""
    from conftest import CFG
    from dsabeamformer_amd import host

    cfg = bfmod.debug_config()
    pos_f, dir_f = os.path.join(CFG, "grid_positions.txt"), os.path.join(CFG, "grid_beam_directions.txt")
    src_f = os.path.join(CFG, "grid_source_directions_4096.txt")
    ded, ms = host.run_debug_observation(cfg, gpu=gpu, positions=pos_f, directions=dir_f, sources=src_f, max_sources=4096)
    assert ded.shape == (4096, 256) and ms > 0
    g = orc.DEBUG_GEOM
    pos, dirs, src = orc.read_positions(pos_f, 64), orc.read_directions(dir_f, 256), orc.read_directions(src_f)
    assert src.shape == (4096, 2)
    w = orc.make_weights(g, pos, dirs, gpu)
    for batch in range(4):
        packed = orc.generate_test_data(g, pos, src, gpu, batch_counter=batch)
        picks = list(range(0, 1024, 37)) + [1023]                          # every 37th source of the batch + the last
        out = orc.beamform(g, w, np.ascontiguousarray(packed[picks]))      # (the oracle beamforms the sampled units only)
        for k, u in enumerate(picks):
            assert np.array_equal(ded[batch * 1024 + u], orc.dedisperse(g, out[k])), (batch, u)


@pytest.mark.parametrize("paired", sweep(["default", "0"], ["default"]))
def test_random_array_debug_flow_bit_identical(bfmod, orc, monkeypatch, paired):
    """The reference's remaining fixtures end to end through the DEBUG flow: config/random_positions.txt (uniform +-250 m, an
    array without any symmetry) with the 16x16 beam grid and the 61x61 source catalogue grid_source_directions_3721.txt --
    3721 sources = 3 full generator batches + a partial fourth (sources beyond the catalogue are zero voltages,
    src/test_data_generator.hh:77-90).  Symmetric beam SETS make the weights conjugate-symmetric whatever the array is, so
    the default run takes the pair kernel; DSABF_PAIRED=0 sends the same observation through the general kernel.  Either way
    the dedispersed table equals the oracle's row for row."""
    from conftest import CFG
    from dsabeamformer_amd import host

    if paired == "0":
        monkeypatch.setenv("DSABF_PAIRED", "0")
    cfg = bfmod.debug_config()
    pos_f, dir_f = os.path.join(CFG, "random_positions.txt"), os.path.join(CFG, "grid_beam_directions.txt")
    src_f = os.path.join(CFG, "grid_source_directions_3721.txt")
    ded, ms = host.run_debug_observation(cfg, gpu=3, positions=pos_f, directions=dir_f, sources=src_f, max_sources=3721)
    assert ded.shape == (3721, 256) and ms > 0
    g = orc.DEBUG_GEOM
    pos, dirs, src = orc.read_positions(pos_f, 64), orc.read_directions(dir_f, 256), orc.read_directions(src_f)
    assert src.shape == (3721, 2)
    w = orc.make_weights(g, pos, dirs, 3)
    for batch in range(4):
        packed = orc.generate_test_data(g, pos, src, 3, batch_counter=batch)
        n_here = min(1024, 3721 - batch * 1024)
        picks = list(range(0, n_here, 41)) + [n_here - 1]
        out = orc.beamform(g, w, np.ascontiguousarray(packed[picks]))      # (the oracle beamforms the sampled units only)
        for k, u in enumerate(picks):
            assert np.array_equal(ded[batch * 1024 + u], orc.dedisperse(g, out[k])), (batch, u)


@pytest.mark.parametrize("mode,layout", sweep([("alltoall", "rank"), ("root", "rank"), ("alltoall", "freq"), ("root", "freq")],
                                               [("alltoall", "rank")]))
def test_bench_rccl_gather_plumbing_on_one_gpu(mode, layout):
    """bench.py --force-dist: the multi-GPU path (process group, a one-rank RCCL communicator behind bf_comm_create,
    bf_gather_detected on its side stream, double buffering) with world size 1 -- what can be exercised of it on a 1-GPU
    box.  The gathered result of the last steps must be the kernel's output (identity for one rank), the JSON line well
    formed, and every gather mode reported side by side."""
    import json
    import subprocess
    import sys

    from conftest import ROOT

    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + (os.getpid() % 300)))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--gather", mode, "--layout", layout,
                        "--units", "4", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--min-warm-seconds", "0.1"] +
                       ([] if LONG else ["--no-extras"]),      # (the side-by-side modes: the multi-rank tests and the LONG run)
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["config"]["gather"].startswith(mode) and "C-ABI" in d["config"]["gather"]
    assert "gather_note" not in d["config"]          # the C-ABI communicator was created (no torch.distributed fallback)
    assert d["value"] > 0 and d["roofline"]["frac"] > 0.05 and d["roofline"]["unit"] == "TOP/s"
    key = "%s_%s_major" % (mode, layout)
    assert d["gather_modes"][key]["verified"] is True and d["gather_modes"][key]["headline"] is True and d["gather_modes"]["none"]["value"] > 0
    assert d["rccl"]["ranks"] == 1 and d["rccl"]["version"] > 0 and "rccl" in d["rccl"]["lib"]      # the real library, one rank
    if LONG:
        assert set(d["gather_modes"]) >= {"none", "root_rank_major", "root_freq_major", "alltoall_rank_major", "alltoall_freq_major"}


def test_bench_watchdog_prints_the_headline_when_the_supplementary_records_overrun():
    """bench.py --budget-seconds: whatever happens after the headline measurement (a gather mode that never returns on a real
    8-GPU node, a supplementary record that overruns), the ONE JSON line still appears, complete up to the roofline, marked
    truncated, exit status 0."""
    import json
    import subprocess
    import sys

    from conftest import ROOT

    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2", "--min-warm-seconds", "0.2",
                        "--budget-seconds", "0.05"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert "truncated" in d and d["value"] > 0 and d["roofline"]["frac"] > 0.05 and d["n_gpus"] == 1
    assert "cpu_baseline" not in d      # it comes last and takes 20 s: the budget expired long before


@pytest.mark.parametrize("n_ant", [100, 128])
def test_observation_loop_wide_antenna_geometry_bit_exact(bfmod, orc, tmp_path, n_ant):
    """The production observation loop (ring slots, 4 queues, sink) on a DSA100-style geometry: 100 / 128 antennas go
    through the two-k-step kernel via bf_enqueue_gemm_unit, one gemm-unit per launch, straight from PSRDADA-style
    blocks; the whole detected stream is checked against the oracle (default linear geometry -> the paired kernel)."""
    from dsabeamformer_amd import host

    cfg = bfmod.production_config(n_freq=4)
    cfg.n_ant, cfg.n_beams, cfg.n_gemms_per_block, cfg.n_streams = n_ant, 128, 8, 4
    n_blocks, ring_blocks = 5, 2
    path = str(tmp_path / "wide.bin")
    r = host.run_observation_junk_to_file(cfg, n_blocks, path, ring_blocks=ring_blocks, seed=n_ant)
    assert r["gemms_written"] == n_blocks * 8
    hdr, data = host.read_detected_file(path)
    assert int(hdr["N_ANTENNAS"]) == n_ant
    g = orc.Geom(n_beams=128, n_ant=n_ant, n_freq=4, n_avg=16, n_out_per_gemm=cfg.n_out_per_gemm)
    w = orc.make_weights(g, orc.default_positions(n_ant), orc.default_directions(128), 0)
    for blk in range(n_blocks):
        want = orc.beamform(g, w, r["ring"][blk % ring_blocks])
        assert np.array_equal(data[blk * 8:(blk + 1) * 8].reshape(want.shape), want), blk


def test_observation_detected_stream_to_output_ring(bfmod, orc):
    """The output side of the PSRDADA picture (README.md:149 "not yet implemented" in the reference): the loop hands every
    gemm-unit to a consumer through a shared-memory ring (dsabf::ring_sink, 3 slots for 40 gemm-units, so the loop also
    blocks on a slow consumer); the consumer thread checks order, sizes, the end-of-data block and every value."""
    import threading
    import time

    from dsabeamformer_amd import host

    cfg = bfmod.production_config(n_freq=4)
    cfg.n_beams, cfg.n_gemms_per_block, cfg.n_streams = 64, 8, 4
    n_blocks, ring_blocks = 5, 2
    name = "dsabf_out_%d" % os.getpid()
    per = cfg.n_out_per_gemm * cfg.n_freq * cfg.n_beams
    got, err = [], []

    def consumer():
        try:
            ring = host.ShmRing(name, timeout_ms=20000)
            assert ring.block_size == per * 4 and "detected_power" in ring.header
            while True:
                data, bid = ring.read()
                if data.size < ring.block_size:
                    break
                assert bid == len(got)
                got.append(data.view(np.float32).copy())
                if bid == 7:
                    time.sleep(0.2)   # let the 3-slot ring fill: the producer must wait, not drop
            ring.detach()
        except Exception as e:  # noqa: BLE001
            err.append(e)

    t = threading.Thread(target=consumer)
    t.start()
    r = host.run_observation_junk_to_ring(cfg, n_blocks, name, out_ring_blocks=3, ring_blocks=ring_blocks, seed=12)
    t.join(timeout=60)
    assert not err, err
    assert r["gemms_written"] == n_blocks * 8 == len(got)
    assert not os.path.exists("/dev/shm/" + name)   # the sink removed its ring after the consumer drained it
    g = orc.Geom(n_beams=64, n_ant=64, n_freq=4, n_avg=16, n_out_per_gemm=cfg.n_out_per_gemm)
    w = orc.make_weights(g, orc.default_positions(64), orc.default_directions(64), 0)
    for blk in range(n_blocks):
        want = orc.beamform(g, w, r["ring"][blk % ring_blocks]).reshape(8, per)
        for u in range(8):
            assert np.array_equal(got[blk * 8 + u], want[u]), (blk, u)


@pytest.mark.parametrize("name", ["minimal", "dm_stream"])
def test_plain_c_example_runs(tmp_path, name):
    """examples/minimal.c (C99, streaming entry points of the C-ABI) and examples/dm_stream.c (the DM stage behind block launches,
    every emitted sum checked in C against the host's own ascending-f chain) built with hipcc against libdsabf.so and run."""
    import subprocess

    from conftest import ROOT

    exe = str(tmp_path / name)
    pkg = os.path.join(ROOT, "dsabeamformer_amd")
    b = subprocess.run(["/opt/rocm/bin/hipcc", "-x", "c", "-std=c99", "-I" + os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "examples", name + ".c"), "-o", exe, "-L" + pkg, "-ldsabf",
                        "-Wl,-rpath," + pkg], capture_output=True, text=True, timeout=300)
    assert b.returncode == 0, b.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr


def test_device_resident_weights_and_host_registration(torch, bfmod, orc):
    """bf_set_weights_device (weights already in HBM, e.g. a sharded slice): same image, same pairing decision, same
    -128 rejection as the host path; bf_host_register / bf_host_unregister on caller-owned memory (the PSRDADA blocks'
    dada_cuda_dbregister) feeding bf_submit_block."""
    import ctypes as C

    from dsabeamformer_amd import _lib

    g = orc.Geom(n_beams=64, n_ant=64, n_freq=3, n_avg=16, n_out_per_gemm=2)
    rng = np.random.default_rng(2718)
    w = rng.integers(-127, 128, size=(g.n_freq, g.n_ant, g.n_beams, 2), dtype=np.int8)
    packed = rng.integers(0, 256, size=(2, g.n_freq, g.n_time, g.n_ant), dtype=np.uint8)
    bf = bfmod.Beamformer(_cfg(bfmod, g))
    s = torch.cuda.current_stream().cuda_stream
    for paired in (False, True):
        ww = _conj_symmetric(w) if paired else w
        bf.set_weights_device(torch.from_numpy(ww).cuda(), s)
        assert ("PAIRED" in bf.kernel_info(2)["kernel"]) == paired
        want = orc.beamform(g, ww, packed)
        assert np.array_equal(_run(torch, bf, packed, want.size).reshape(want.shape), want)
    bad = w.copy()
    bad[1, 5, 7, 1] = -128
    with pytest.raises(Exception):
        bf.set_weights_device(torch.from_numpy(bad).cuda(), s)
    with pytest.raises(Exception):
        bf.beamform(torch.from_numpy(packed).cuda(), 2, torch.empty(want.size, device="cuda"), s)  # no valid weights now
    # page-lock ordinary host memory and stream a block from it
    lib = _lib.load()
    cfg = _cfg(bfmod, g, n_gemms_per_block=2, n_blocks_on_gpu=2, n_streams=2)
    bf2 = bfmod.Beamformer(cfg)
    bf2.set_weights(w)
    block = np.ascontiguousarray(packed)                        # exactly one block of 2 gemm-units
    assert lib.bf_host_register(C.c_void_p(block.ctypes.data), block.nbytes) == 0
    try:
        out = np.zeros((2,) + (g.n_out_per_gemm, g.n_freq, g.n_beams), np.float32)
        bf2.submit_block(0, block, block.nbytes)
        for ts in range(2):
            bf2.enqueue_gemm_unit(ts, 0, ts, out[ts])
        bf2.sync(-1)
        assert np.array_equal(out.reshape(-1), orc.beamform(g, w, packed).reshape(-1))
    finally:
        assert lib.bf_host_unregister(C.c_void_p(block.ctypes.data)) == 0
    assert lib.bf_host_register(None, 16) < 0


def test_remaining_abi_entry_points(torch, bfmod, orc):
    """bf_device_name, bf_get_config, bf_record_transfer_event, bf_event_synchronize: the entry points the C++ host
    mirror uses internally, called directly."""
    import ctypes as C

    from dsabeamformer_amd import _lib

    lib = _lib.load()
    buf = C.create_string_buffer(256)
    assert lib.bf_device_name(0, buf, 256) == 0 and b"gfx950" in buf.value
    assert lib.bf_device_name(99, buf, 256) < 0
    g = orc.Geom(n_beams=64, n_ant=64, n_freq=2, n_avg=16, n_out_per_gemm=2)
    cfg = _cfg(bfmod, g, n_gemms_per_block=1, n_blocks_on_gpu=2, n_streams=1)
    bf = bfmod.Beamformer(cfg)
    back = _lib.BfConfig()
    assert lib.bf_get_config(bf._h, C.byref(back)) == 0
    assert (back.n_beams, back.n_ant, back.n_freq, back.n_avg, back.n_out_per_gemm) == (64, 64, 2, 16, 2)
    from dsabeamformer_amd import api

    ev = api.event_create()
    block = np.zeros(cfg.n_gemms_per_block * g.n_freq * g.n_time * g.n_ant, np.uint8)
    pinned = C.c_void_p()
    assert lib.bf_alloc_pinned(C.byref(pinned), block.nbytes) == 0
    C.memmove(pinned, block.ctypes.data, block.nbytes)
    bf.submit_block(1, pinned, block.nbytes)
    assert lib.bf_record_transfer_event(bf._h, ev) == 0
    assert lib.bf_event_synchronize(ev) == 0
    assert api.event_query(ev) == 0            # BF_OK: complete
    api.event_destroy(ev)
    assert lib.bf_free_pinned(pinned) == 0
