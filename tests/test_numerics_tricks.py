"""CPU proofs (exhaustive, numpy) of the exact-arithmetic identities the fused HIP kernel relies on
(dsabeamformer_amd/csrc/bf_kernels.hip header).  If any of these failed, bit-exact parity could not hold."""
import numpy as np

K_BITS = 0x4B400000
K = np.float32(12582912.0)
C127 = np.float32(1.0 / 127.0)          # the reference's alpha, src/beamformer.cu:191
C16 = np.float32(C127 * np.float32(0.0625))
NEG_KC16 = np.float32(-(K * C16))


def test_nibble_to_int8_times_16():
    b = np.arange(256, dtype=np.uint8)
    re = (b.astype(np.int8) >> 4).astype(np.int32)                       # src/beamformer.cuh:96
    im = ((b << 4).astype(np.uint8).astype(np.int8) >> 4).astype(np.int32)  # :97-98
    assert np.array_equal((b & 0xF0).astype(np.int8).astype(np.int32), 16 * re)
    assert np.array_equal(((b.astype(np.uint32) << 4) & 0xF0).astype(np.uint8).astype(np.int8).astype(np.int32), 16 * im)


def test_magic_seed_is_exact_int_to_float():
    n = np.arange(-130048, 130049, dtype=np.int64)  # |sum| <= 64*2*127*8
    bits = (K_BITS + 16 * n).astype(np.uint32)
    assert np.array_equal(bits.view(np.float32).astype(np.float64), 12582912.0 + 16.0 * n)


def test_single_fma_equals_convert_then_scale():
    assert float(K) * float(C16) == float(np.float32(K * C16)), "K*alpha/16 must be exactly representable"
    n = np.arange(-130048, 130049, dtype=np.int64)
    m = (12582912.0 + 16.0 * n)                              # exact in float32 (previous test)
    # fma(m, c16, -K*c16): the double product and difference are exact (<= 49 significant bits), so one
    # rounding to float32 is exactly the fma result
    fma = (m * np.float64(C16) + np.float64(NEG_KC16)).astype(np.float32)
    want = n.astype(np.float32) * C127                       # oracle: (float)n * alpha
    assert np.array_equal(fma.view(np.uint32) & 0x7FFFFFFF, want.view(np.uint32) & 0x7FFFFFFF)
    assert np.array_equal(fma == 0, want == 0)


def _sext4x4(nib):
    return (((nib ^ np.uint32(0x88888888)) - np.uint32(0x08080808)) ^ np.uint32(0x80808080)).astype(np.uint32)


def _perm(s0, s1, sel):
    """v_perm_b32: result byte i = byte sel_i of the 8-byte value {s0 (bytes 4-7), s1 (bytes 0-3)}."""
    out = np.zeros_like(s0)
    for i in range(4):
        k = (sel >> (8 * i)) & 0xFF
        src = s1 if k < 4 else s0
        out |= ((src >> np.uint32(8 * (k & 3))) & np.uint32(0xFF)) << np.uint32(8 * i)
    return out


def test_standalone_expand_bit_tricks():
    rng = np.random.default_rng(3)
    w = rng.integers(0, 2 ** 32, size=4096, dtype=np.uint64).astype(np.uint32)
    w[:64] = (np.arange(64, dtype=np.uint32) * 4 + np.arange(4, dtype=np.uint32)[:, None] * 0).ravel()[:64]
    w[:256] = np.arange(256, dtype=np.uint32) * 0x01010101
    hi = _sext4x4((w >> np.uint32(4)) & np.uint32(0x0F0F0F0F))
    lo = _sext4x4(w & np.uint32(0x0F0F0F0F))
    e0 = _perm(lo, hi, 0x05010400)
    e1 = _perm(lo, hi, 0x07030602)
    got = np.stack([e0, e1], 1).view(np.int8).reshape(-1, 8)
    b = w.view(np.uint8).reshape(-1, 4)
    want = np.empty((len(w), 8), np.int8)
    want[:, 0::2] = b.astype(np.int8) >> 4
    want[:, 1::2] = (b << 4).astype(np.uint8).astype(np.int8) >> 4
    assert np.array_equal(got, want)


def _sample_in_half(i, e, nipo):
    return (2 * (i // nipo) + e) * nipo + (i % nipo)


def _lds_row_of_mfma_row(r, nipo):
    """Mirror of lds_row_of_mfma_row<NIPO> in bf_kernels.hip: D row r -> row of the tile's LDS image."""
    reg, h = (r & 3) + 4 * (r >> 3), (r >> 2) & 1
    i, e = reg >> 1, reg & 1
    if nipo >= 16:
        return (2 * h + e) * 8 + i
    return 16 * h + _sample_in_half(i, e, nipo)


def test_mfma_row_mapping_pairs_two_outputs_per_register_pair():
    """D row of (half h, reg) is (reg&3) + 8*(reg>>2) + 4*h (cdna guide section 3).  The mapping must (a) be a
    bijection on the 32 rows of a tile, (b) put the SAME window position of two different outputs in the two
    elements of every register pair, (c) walk each output's window in increasing time order over the pairs."""
    for nipo in (2, 4, 8, 16, 32, 64):
        rows = {}
        for h in range(2):
            for reg in range(16):
                d_row = (reg & 3) + 8 * (reg >> 2) + 4 * h
                rows[(h, reg >> 1, reg & 1)] = _lds_row_of_mfma_row(d_row, nipo)
        assert sorted(rows.values()) == list(range(32))
        for h in range(2):
            if nipo >= 16:
                # stream (h, e) = 8 contiguous samples, pair i = sample i
                for e in range(2):
                    assert [rows[(h, i, e)] for i in range(8)] == list(range((2 * h + e) * 8, (2 * h + e) * 8 + 8))
            else:
                for i in range(8):
                    s0, s1 = rows[(h, i, 0)] - 16 * h, rows[(h, i, 1)] - 16 * h
                    assert s0 // nipo != s1 // nipo and s0 % nipo == s1 % nipo == i % nipo   # (b)
                    assert s1 // nipo == s0 // nipo + 1 and (s0 // nipo) % 2 == 0
                for u in range(16 // nipo):                                                    # (c)
                    pos = [(i, e) for i in range(8) for e in range(2) if (rows[(h, i, e)] - 16 * h) // nipo == u]
                    assert [rows[(h, i, e)] - 16 * h - u * nipo for i, e in pos] == list(range(nipo))


def test_lds_swizzle_is_bank_conflict_free():
    """ds_read_b128: 4 groups of 16 lanes, bank = (addr/4) % 64; ds_write_b128: 8 groups of 8 contiguous lanes,
    bank = (addr/4) % 32 (MI355X_MICROARCH.md LDS table).  Conflict-free <=> the 16-byte slots are distinct."""
    groups_r = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
                list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
    groups_r += [[l + 32 for l in g] for g in groups_r]
    for nipo in (2, 4, 8, 16, 32, 64):
        run = 8 if nipo >= 16 else 16
        for rbc, nks in ((8, 4), (8, 2), (16, 8)):
            rb = rbc * 16
            swz = (lambda c, row: c ^ (((row >> 1) & 7) ^ ((row & 1) << 2))) if rbc == 8 else (lambda c, row: c ^ (row & 15))
            for tile in range(4):
                for ks in range(nks):
                    for g in groups_r:
                        slots = set()
                        for lane in g:
                            hl, lc = lane >> 5, lane & 31
                            row = tile * 32 + _lds_row_of_mfma_row(lc, nipo)
                            addr = row * rb + 16 * swz(hl * (rbc // 2) + ks, row)
                            slots.add((addr // 16) % 16)
                        assert len(slots) == 16, (nipo, rbc, tile, ks)
            # staging writes: thread tid writes piece pc: row = rr*run + (pi // nks), ks = pi % nks
            if nks == 2:
                continue  # half-empty rows: 8 consecutive pieces span 4 rows (2-way at most, irrelevant sizes)
            for pc0 in range(0, 128 * nks, 8):
                for comp in range(2):
                    slots = set()
                    for pc in range(pc0, pc0 + 8):
                        rr, pi = divmod(pc, run * nks)
                        row, ks = rr * run + pi // nks, pi % nks
                        addr = row * rb + 16 * swz(comp * (rbc // 2) + ks, row)
                        slots.add((addr // 16) % 8)
                    assert len(slots) == 8, (nipo, rbc, pc0, comp)
