"""The instantiation census on the GPU: EVERY fused kernel compiled into libdsabf.so is launched once, on the
smallest geometry of the reference's contract that selects it (tools/census.py), and held to the oracle:

  * the handle says which instantiation it launches (bf_handle_variant_key) -- it must be the one the case is meant to cover;
  * canonical reading: np.array_equal with orc.beamform (the g++ reading of src/beamformer.cuh:150-152);
  * contracted reading: np.array_equal with orc.beamform under ORC_CONTRACT_NVCC;
  * fast detect (the product's own option): np.array_equal with the oracle's restatement of it AND inside the stated tolerance
    (n_ipo + 1) * 2^-23 of the exact value;
  * stage-parity store (bf_gemm_device): np.array_equal with orc.gemm.

Every case runs its kernel's STEADY STATE, not only its prologue: the launch is held to one workgroup per (frequency, beam group)
(`tsplit` 1) over at least five 128-sample chunks (ten for windows of 64 samples and more) plus a ragged tail, so the LDS double
buffer, the prefetch two chunks ahead and the parked stores of the previous chunk all happen in every instantiation; the
stage-parity cases take the largest gemm-unit (more outputs) that still selects the same instantiation.

A wrong bit in any one of the 382 instantiations -- the reference has ONE kernel per stage (src/beamformer.cuh:66-155), every one
of these replaces them for some geometry and is selected silently -- fails here, in the driver-observed run.  CPU side of the same
census: tests/test_census_cpu.py (compiled set == reachable set)."""
import os
import sys
import time

import numpy as np
import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu


def _conj_symmetric(w):
    w = w.copy()
    nb = w.shape[2]
    w[:, :, nb // 2:, 0] = w[:, :, :nb // 2, 0][:, :, ::-1]
    w[:, :, nb // 2:, 1] = -w[:, :, :nb // 2, 1][:, :, ::-1]
    return w


def run_case(torch, bfm, orc, key, rec, n_freq, whole_spans=False):
    """One instantiation on its smallest geometry; returns a failure text or None.  whole_spans: gemm-units of 256 samples instead
    -- a chunk's sample span then never straddles a gemm-unit and the kernel takes its scalar chunk addressing (fused16_kernel's
    `fast_addr`, a run-time branch of every compile-time-window instantiation: the path every production geometry runs)."""
    g = orc.Geom(n_beams=rec["n_beams"], n_ant=rec["n_ant"], n_freq=n_freq, n_pol=rec["n_pol"], n_avg=rec["n_avg"],
                 n_out_per_gemm=(256 // (rec["n_pol"] * rec["n_avg"])) if whole_spans else rec["n_out"])
    seed = sum(ord(c) * (i + 1) for i, c in enumerate(key)) % (1 << 31)
    rng = np.random.default_rng(seed)
    w = rng.integers(-127, 128, size=(g.n_freq, g.n_ant, g.n_beams, 2), dtype=np.int8)
    if rec["paired"]:
        w = _conj_symmetric(w)
    def config(geom):
        return bfm.production_config(n_beams=geom.n_beams, n_ant=geom.n_ant, n_freq=geom.n_freq, n_pol=geom.n_pol, n_avg=geom.n_avg,
                                     n_out_per_gemm=geom.n_out_per_gemm, n_gemms_per_block=1, n_blocks_on_gpu=1, n_streams=1,
                                     detect_mode=rec["mode"])

    if rec["write_c"]:
        # bf_gemm_device takes ONE gemm-unit: the longest one (more outputs per unit) that still selects this instantiation
        for n_out in (max(1, -(-640 // g.n_ipo)), max(1, -(-384 // g.n_ipo))):
            if n_out % 2 != rec["n_out"] % 2:
                n_out += 1                   # (short windows: whether a unit is whole 16-sample runs decides the class)
            big = orc.Geom(n_beams=g.n_beams, n_ant=g.n_ant, n_freq=g.n_freq, n_pol=g.n_pol, n_avg=g.n_avg, n_out_per_gemm=n_out)
            if n_out > g.n_out_per_gemm and bfm.variant_key(config(big), bool(rec["paired"]), True) == key:
                g = big
                break
    cfg = config(g)
    stream = torch.cuda.current_stream().cuda_stream
    with bfm.Beamformer(cfg) as bf:
        bf.set_weights(w)
        ran = bf.variant_key(bool(rec["write_c"]))
        if ran != key:
            return "the handle launches %s" % ran
        bf.set_switch("tsplit", 1)           # one workgroup per (frequency, beam group) walks every chunk of the launch
        if rec["write_c"]:
            packed = rng.integers(0, 256, size=(g.n_freq, g.n_time, g.n_ant), dtype=np.uint8)
            d_in = torch.from_numpy(packed).cuda()
            d_c = torch.full((g.n_freq * g.n_time * g.n_beams * 2,), float("nan"), dtype=torch.float32, device="cuda")
            bf.gemm(d_in, d_c, stream)
            torch.cuda.synchronize()
            want = orc.gemm(g, w, orc.expand(packed))
            got = d_c.cpu().numpy().reshape(want.shape)
            return None if np.array_equal(got, want) else "bf_gemm_device differs from orc.gemm in %d of %d values" % ((got != want).sum(), want.size)
        n_units = -(-(1280 if g.n_ipo >= 64 else 640) // g.n_time) + 1     # >= 5 (10) chunks of 128 samples and a ragged tail
        packed = rng.integers(0, 256, size=(n_units, g.n_freq, g.n_time, g.n_ant), dtype=np.uint8)
        packed[0, 0, :, :] = 0x88            # the extremes: all (-8, -8) ...
        packed[1, 0, 0, :] = 0x77            # ... and (7, 7)
        d_in = torch.from_numpy(packed).cuda()
        d_out = torch.full((n_units * g.out_per_gemm + 64,), float("nan"), dtype=torch.float32, device="cuda")
        bf.beamform(d_in, n_units, d_out, stream)
        torch.cuda.synchronize()
        got = d_out.cpu().numpy()
        if not np.isnan(got[n_units * g.out_per_gemm:]).all():
            return "wrote behind the last output"
        got = got[:n_units * g.out_per_gemm].reshape(n_units, g.n_out_per_gemm, g.n_freq, g.n_beams)
        if rec["mode"] == 1 and g.n_ipo >= 16:     # BF_DETECT_FAST (below 16 samples the library runs the canonical detect)
            want = orc.beamform_fast(g, w, packed)
            exact = orc.beamform_exact(g, w, packed)
            ok = exact > 0
            if ok.any() and np.abs(got[ok] / exact[ok] - 1).max() > (g.n_ipo + 1) * 2.0 ** -23:
                return "fast detect outside its stated tolerance"
        else:
            with orc.detect_contract(orc.CONTRACT_NVCC if rec["mode"] == 2 else orc.CONTRACT_NONE):
                want = orc.beamform(g, w, packed)
        if not np.array_equal(got, want):
            return "differs from the oracle in %d of %d values" % ((got != want).sum(), want.size)
    return None


def test_every_compiled_instantiation_against_the_oracle():
    import torch

    import census
    import dsabeamformer_amd as bfm
    import oracle as orc

    assert torch.cuda.is_available(), "these tests need a GPU"
    # every instantiation on the smallest geometry of EACH of its two store paths the contract reaches (beams dealt round robin to a
    # wave's column tiles: vector stores of whole lines; or tile by tile: scalar stores, a partly filled last tile)
    comp, reach = census.compiled(), census.reachable(both_store_paths=True)
    assert comp == {k for k, _ in reach}, (sorted(comp - {k for k, _ in reach}), sorted({k for k, _ in reach} - comp))
    failures, lines, t_all, n_spans = [], [], time.perf_counter(), 0
    for i, (key, inter) in enumerate(sorted(reach)):
        t0 = time.perf_counter()
        # two frequency counts: 3 (a frequency per block index) and 8 (the XCD-aware block map, decode_block)
        err = run_case(torch, bfm, orc, key, reach[(key, inter)], 8 if i % 4 == 0 else 3)
        lines.append("%-54s %-11s %6.1f ms  %s" % (key, "interleaved" if inter else "by tile", (time.perf_counter() - t0) * 1e3, err or "ok"))
        if err:
            failures.append((key, reach[(key, inter)], err))
        # ... and once more with gemm-units of whole 256-sample spans (the scalar chunk addressing of the compile-time windows)
        targs = [t.strip() for t in key[key.index("<") + 1:-1].split(",")]          # fused16_kernel<AIN, NIPO, WRITE_C, ...>
        compile_time_window = key.startswith("fused16_kernel") and int(targs[1]) != 0 and targs[2] == "false"
        if compile_time_window and (not inter or (key, False) not in reach):
            t0 = time.perf_counter()
            err = run_case(torch, bfm, orc, key, reach[(key, inter)], 3, whole_spans=True)
            lines.append("%-54s %-11s %6.1f ms  %s" % (key, "whole spans", (time.perf_counter() - t0) * 1e3, err or "ok"))
            n_spans += 1
            if err:
                failures.append((key, "whole spans", err))
    total = time.perf_counter() - t_all
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "census_gpu.txt"), "w") as fp:
        fp.write("# tests/test_gpu_census.py: %d instantiations, %d (instantiation, store path) cases + %d with whole-span gemm-units, %d failures, %.1f s\n"
                 % (len({k for k, _ in reach}), len(reach), n_spans, len(failures), total))
        fp.write("\n".join(lines) + "\n")
    assert not failures, failures[:10]
    assert len(reach) >= 300                     # (the walk really found the library's kernels)


def test_a_dm_stage_survives_more_than_64_streams_on_its_handle(orc):
    """the 65th distinct HIP stream passed to bf_dedisperse_dm_device used to release every live bf_dm_stream of
    the handle (a stray line in the scratch cache's eviction).  A DM stage created before, pushed after, destroyed in both orders."""
    import torch

    import dsabeamformer_amd as bfm
    from dsabeamformer_amd.api import DmStream

    n_freq, n_beams, n_dm = 8, 64, 4
    rng = np.random.default_rng(65)
    delays = np.sort(rng.integers(0, 6, size=(n_dm, n_freq)), axis=1)[:, ::-1].astype(np.int32).copy()
    series = rng.random((40, n_freq, n_beams), dtype=np.float32)
    for order in ("stream first", "handle first"):
        bf = bfm.Beamformer(bfm.production_config(n_freq=n_freq, n_beams=n_beams, n_gemms_per_block=1, n_blocks_on_gpu=1, n_streams=1))
        ds = DmStream(bf, delays, n_freq, 16)
        d_series = torch.from_numpy(series).cuda()
        d_delays = torch.from_numpy(delays).cuda()
        d_out = torch.empty(n_dm * 30 * n_beams, dtype=torch.float32, device="cuda")
        host = np.empty(n_dm * 16 * n_beams, np.float32)     # a push's chunk arrives as [n_dm][n_t_out][beam], contiguous
        first, n0 = ds.push(d_series[:16], 16, host)
        torch.cuda.synchronize()
        got = [host[:n_dm * n0 * n_beams].reshape(n_dm, n0, n_beams).copy()]
        streams = [torch.cuda.Stream() for _ in range(70)]
        for s in streams:                             # 70 distinct streams: the cache of 64 is evicted on the way
            bf.dedisperse_dm(d_series, 40, d_delays, n_dm, 30, d_out, s.cuda_stream)
        torch.cuda.synchronize()
        want_whole = orc.dedisperse_dm(series, delays, 30)
        assert np.array_equal(d_out.cpu().numpy().reshape(want_whole.shape), want_whole)
        first2, n1 = ds.push(d_series[16:32], 16, host)   # the stage is still alive ...
        torch.cuda.synchronize()
        got.append(host[:n_dm * n1 * n_beams].reshape(n_dm, n1, n_beams).copy())
        chunks = np.concatenate(got, axis=1)
        assert first == 0 and first2 == n0
        assert np.array_equal(chunks, orc.dedisperse_dm(series[:32], delays, n0 + n1))   # ... and still right
        if order == "stream first":
            ds.close()
            bf.close()
        else:
            bf.close()
            ds.close()


def test_real_rccl_receives_straight_into_the_dm_stages_ring(orc, monkeypatch):
    """What a holder of a sharded run does since round 6: bf_gather_detected receives the band INTO the rows bf_dm_stream_reserve
    handed out -- memory that is mapped twice (hipMemCreate + 2 x hipMemMap), the rows of some pushes running across the seam.
    Here with the REAL RCCL library: a one-rank communicator whose own rows travel through grouped ncclSend / ncclRecv
    (DSABF_GATHER_SELF_RCCL=1, the code path every message of a multi-GPU run takes).  Chunks bit-equal to the oracle over the whole
    series; the ring is small (D + 3 pushes = 41 rows) so that 12 pushes wrap it several times."""
    import torch

    import dsabeamformer_amd as bfm
    from dsabeamformer_amd import api

    n_f, n_b, n_dm, rows = 16, 64, 6, 8
    rng = np.random.default_rng(606)
    delays = np.ascontiguousarray((np.arange(n_dm)[:, None] * np.linspace(17 / (n_dm - 1), 0.0, n_f)[None, :]).astype(np.int32))
    D = int(delays.max())
    n_t = 12 * rows
    series = (rng.random((n_t, n_f, n_b), dtype=np.float32) * 1e3).astype(np.float32)
    want = orc.dedisperse_dm(series, delays, n_t - D)
    monkeypatch.setenv("DSABF_GATHER_SELF_RCCL", "1")
    bf = bfm.Beamformer(bfm.production_config(n_freq=n_f, n_beams=n_b, n_gemms_per_block=1, n_blocks_on_gpu=1, n_streams=1))
    comm = api.Comm(0, 1, api.comm_unique_id(), device=0)
    assert comm.info()["lib"] and "fakerccl" not in comm.info()["lib"]
    dm = api.DmStream(bf, delays, n_f, rows)
    assert bf.counter("dm_ring_stages") == 1
    d_series = torch.from_numpy(series).cuda()
    st = torch.cuda.Stream()
    host = torch.full((n_dm * rows * n_b,), float("nan"), dtype=torch.float32).pin_memory()
    parts, seam = [], 0
    lo = None
    for k in range(12):
        dst = dm.reserve(rows, st.cuda_stream)
        lo = dst if lo is None else min(lo, dst)
        seam += int(dst + rows * n_f * n_b * 4 > lo + (D + 3 * rows) * n_f * n_b * 4)     # this push's rows run past the ring's end
        with torch.cuda.stream(st):
            comm.gather(d_series[k * rows:(k + 1) * rows], rows, n_f * n_b, 0, api.GATHER_FREQ_MAJOR, dst, st.cuda_stream)
        first, n_out = dm.push(dst, rows, host, st.cuda_stream)
        st.synchronize()
        if n_out:
            parts.append(host[:n_dm * n_out * n_b].numpy().reshape(n_dm, n_out, n_b).copy())
    assert seam >= 1, "no push crossed the seam of the double mapping: the test does not test what it says"
    assert np.array_equal(np.concatenate(parts, axis=1), want)
    dm.close()
    comm.close()
    bf.close()
