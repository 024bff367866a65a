"""GPU tests (-m gpu) of the round-4 surface; every call goes through the C-ABI of libdsabf.so."""
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
