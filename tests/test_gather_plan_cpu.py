"""CPU tests of the gather's layout arithmetic (bf_gather_offset / bf_gather_plan, include/dsabf.h): the device part of
bf_gather_detected only walks this plan with ncclSend / ncclRecv, so simulating the plan with numpy for R = 1 ... 8 ranks
proves where every float of every rank lands.  SURVEY.md 8e: rank r's row o sits at o*F*B + r*(F/R)*B in [o][f][b]."""
import ctypes as C

import numpy as np
import pytest

from dsabeamformer_amd._lib import BfGatherMsg, load

FREQ_MAJOR, RANK_MAJOR = 0, 1
SEND, RECV, COPY = 0, 1, 2


def _plan(lib, layout, n_rows, row_floats, world, rank, root):
    n = lib.bf_gather_plan(layout, n_rows, row_floats, world, rank, root, None, 0)
    arr = (BfGatherMsg * max(n, 1))()
    assert lib.bf_gather_plan(layout, n_rows, row_floats, world, rank, root, arr, n) == n
    return [(m.kind, m.peer, m.local_offset, m.full_offset, m.count) for m in arr[:n]]


@pytest.mark.parametrize("world", [1, 2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("layout", [FREQ_MAJOR, RANK_MAJOR])
@pytest.mark.parametrize("root", [0, -1, "last", -2])
def test_simulated_gather_reproduces_the_full_band(world, layout, root):
    """root 0 / last: gather to one owner; -1: everybody receives everything; -2: distributed owners (all-to-all)."""
    lib = load()
    root = world - 1 if root == "last" else root
    n_out, n_units, f_local, n_beams = 4, (world if root == -2 else 4), 5, 12      # (distributed owners: rows divisible by the ranks)
    n_rows, row_floats = n_units * n_out, f_local * n_beams
    rng = np.random.default_rng(world * 10 + layout)
    full_ofb = rng.random((n_rows, world * f_local, n_beams)).astype(np.float32)      # the whole band, [o][f][b]
    local = [np.ascontiguousarray(full_ofb[:, r * f_local:(r + 1) * f_local]) for r in range(world)]   # rank r's shard
    plans = [_plan(lib, layout, n_rows, row_floats, world, r, root) for r in range(world)]
    # point-to-point semantics: between one (sender, receiver) pair messages match in issue order
    sends = {(r, m[1]): [x for x in plans[r] if x[0] == SEND and x[1] == m[1]] for r in range(world) for m in plans[r] if m[0] == SEND}
    got = {}
    for dst in range(world):
        held = lib.bf_gather_rows_held(n_rows, world, dst, root)
        assert held == (n_rows // world if root == -2 else n_rows if root in (-1, dst) else 0)
        receives = held > 0
        first = dst * held if root == -2 else 0
        recvs = [m for m in plans[dst] if m[0] in (RECV, COPY)]
        assert bool(recvs) == receives
        if not receives:
            assert all(m[0] == SEND and m[1] == root for m in plans[dst])
            continue
        buf = np.full(world * held * row_floats, np.nan, np.float32)
        cursor = {}
        for kind, peer, loff, foff, count in recvs:
            if kind == COPY:
                assert peer == dst
                src = local[dst].ravel()[loff:loff + count]
            else:
                q = sends[(peer, dst)]
                i = cursor.get(peer, 0)
                cursor[peer] = i + 1
                skind, sdst, sloff, _, scount = q[i]
                assert scount == count and sloff == loff      # the receive's local_offset names the sender's offset
                src = local[peer].ravel()[sloff:sloff + scount]
            assert np.isnan(buf[foff:foff + count]).all()       # nothing is written twice
            buf[foff:foff + count] = src
        for peer, i in cursor.items():
            assert i == len(sends[(peer, dst)])                 # every send has its receive
        assert not np.isnan(buf).any()                          # ... and nothing is left out
        got[dst] = buf
        if layout == FREQ_MAJOR:     # the reference's [o][f][b] over the whole band, rows first .. first + held
            assert np.array_equal(buf.reshape(held, world * f_local, n_beams), full_ofb[first:first + held])
        else:
            assert np.array_equal(buf.reshape(world, held, f_local, n_beams), np.stack([x[first:first + held] for x in local]))
    # ---- the STAGED transport of the freq-major layout (bf_gather_detected_staged, round 5): what arrived rank-major is moved by
    # one device pass, full[bf_gather_offset(FREQ, rank, row)] = stage[bf_gather_offset(RANK, rank, row)] for every sender but the
    # receiver itself, whose rows are copied straight from its own output -- the result is the reference's [o][f][b]
    if layout == RANK_MAJOR:
        for dst, stage in got.items():
            held = lib.bf_gather_rows_held(n_rows, world, dst, root)
            first = dst * held if root == -2 else 0
            full = np.full(world * held * row_floats, np.nan, np.float32)
            for r in range(world):
                for row in range(held):
                    to = lib.bf_gather_offset(FREQ_MAJOR, held, row_floats, world, r, row)
                    if r == dst:                              # own rows: never staged
                        full[to:to + row_floats] = local[dst][first + row].ravel()
                    else:
                        frm = lib.bf_gather_offset(RANK_MAJOR, held, row_floats, world, r, row)
                        full[to:to + row_floats] = stage[frm:frm + row_floats]
            assert np.array_equal(full.reshape(held, world * f_local, n_beams), full_ofb[first:first + held])
    # message counts: one per sender for the rank-major layout, one per (row, sender) for [o][f][b]
    r0 = root if root >= 0 else 0
    n_recv = len([m for m in plans[r0] if m[0] == RECV])
    held0 = lib.bf_gather_rows_held(n_rows, world, r0, root)
    assert n_recv == (world - 1) * (1 if layout == RANK_MAJOR else held0)


def test_offsets_are_the_survey_formula():
    lib = load()
    n_out, F, B = 16, 256, 256
    for world in (2, 4, 8):
        fl = F // world
        for r in range(world):
            for o in (0, 1, 7, n_out - 1):
                assert lib.bf_gather_offset(FREQ_MAJOR, n_out, fl * B, world, r, o) == o * F * B + r * fl * B
                assert lib.bf_gather_offset(RANK_MAJOR, n_out, fl * B, world, r, o) == (r * n_out + o) * fl * B
    assert lib.bf_gather_plan(FREQ_MAJOR, 4, 8, 2, 2, 0, None, 0) == 0     # rank out of range
    assert lib.bf_gather_plan(FREQ_MAJOR, 5, 8, 2, 0, -2, None, 0) == 0    # distributed owners: rows not divisible
    assert lib.bf_gather_plan(FREQ_MAJOR, 4, 8, 2, 0, -3, None, 0) == 0
    assert lib.bf_gather_plan(FREQ_MAJOR, 4, 8, 2, 0, 2, None, 0) == 0     # root out of range


def test_comm_world_one_needs_no_rccl_and_argument_checks():
    from dsabeamformer_amd._lib import DsabfError, check

    lib = load()
    c = C.c_void_p()
    check(lib.bf_comm_create(0, 1, None, 0, C.byref(c)))
    assert lib.bf_comm_rank(c) == 0 and lib.bf_comm_world(c) == 1
    check(lib.bf_comm_destroy(c))
    for rank, world in ((1, 1), (-1, 2), (0, 0)):
        with pytest.raises(DsabfError):
            check(lib.bf_comm_create(rank, world, None, 0, C.byref(c)))
    with pytest.raises(DsabfError):
        check(lib.bf_comm_create(0, 2, None, 0, C.byref(c)))    # world > 1 needs the unique id
