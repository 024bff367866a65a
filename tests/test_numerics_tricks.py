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


N_MAX = 128 * 2 * 127 * 8   # largest |sum|: 128 antennas x (re, im) x |w| <= 127 x |v| <= 8  (64 antennas: half)


def test_magic_seed_is_exact_int_to_float():
    n = np.arange(-N_MAX, N_MAX + 1, dtype=np.int64)
    assert 16 * N_MAX < 2 ** 22                      # K + 16 n stays inside [2^23, 2^24): unit spacing
    bits = (K_BITS + 16 * n).astype(np.uint32)
    assert np.array_equal(bits.view(np.float32).astype(np.float64), 12582912.0 + 16.0 * n)


def test_single_fma_equals_convert_then_scale():
    assert float(K) * float(C16) == float(np.float32(K * C16)), "K*alpha/16 must be exactly representable"
    n = np.arange(-N_MAX, N_MAX + 1, dtype=np.int64)
    m = (12582912.0 + 16.0 * n)                              # exact in float32 (previous test)
    # fma(m, c16, -K*c16): the double product and difference are exact (<= 49 significant bits), so one
    # rounding to float32 is exactly the fma result
    fma = (m * np.float64(C16) + np.float64(NEG_KC16)).astype(np.float32)
    want = n.astype(np.float32) * C127                       # oracle: (float)n * alpha
    assert np.array_equal(fma.view(np.uint32) & 0x7FFFFFFF, want.view(np.uint32) & 0x7FFFFFFF)
    assert np.array_equal(fma == 0, want == 0)


def test_conjugate_pair_combinations_stay_exact():
    """Paired kernel: P1, P3 carry the seed K, P2, P4 do not; (K + 16 P1) -+ 16 P2 as int32 adds must still be the bit
    pattern of the float K + 16 (P1 -+ P2), i.e. stay inside [2^23, 2^24) for every reachable pair.  |P1| + |P2| is
    bounded by the same 128*2*127*8 as the full sum (they split its terms), so the extreme cases are enough."""
    half = N_MAX // 2
    for p1, p2 in ((half, half), (half, -half), (-half, half), (-half, -half), (N_MAX, 0), (-N_MAX, 0), (0, N_MAX), (0, -N_MAX)):
        for sgn in (1, -1):
            bits = np.uint32((K_BITS + 16 * p1 + sgn * 16 * p2) & 0xFFFFFFFF)
            assert float(bits.view(np.float32)) == 12582912.0 + 16.0 * (p1 + sgn * p2)


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


def _lds_row16(t8, rho, nipo):
    """Mirror of lds_row16<NIPO> in bf_kernels.hip: A row rho of 16-row tile t8 -> row of the 128-row chunk image."""
    if nipo >= 32:
        return (rho >> 2) * 32 + 4 * t8 + (rho & 3)
    return (t8 >> 2) * 64 + (rho >> 2) * 16 + 4 * (t8 & 3) + (rho & 3)


def _swz16(piece, row, nipo):
    """Mirror of swz16<NIPO>: XOR swizzle of the 16-byte piece index (0..7) inside a 128-byte row."""
    lr = 32 if nipo >= 32 else 16
    return piece ^ ((((row >> 1) & 1) | (((row // lr) & 3) << 1)) ^ ((row & 1) << 2))


def test_mfma_row_mapping_keeps_each_output_in_one_lane_in_time_order():
    """v_mfma_i32_16x16x64_i8: lane (column c = lane & 15, group g = lane >> 4) holds D rows 4g + r in registers
    r = 0..3 (cdna guide section 3).  The chunk image is the chunk's 128 samples in time order (n_ipo <= 32), so the
    mapping must (a) be a bijection of (tile, D row) onto the 128 rows, (b) give every lane group ONE stream per tile
    whose samples are consecutive over the registers and over the tiles, so the fp32 sum runs in the reference's order
    inside a lane, (c) make streams = whole outputs (n_ipo >= 16) or 16-sample runs of whole outputs (n_ipo < 16)."""
    for nipo in (2, 4, 8, 16, 32, 64):
        rows = {(t8, rho): _lds_row16(t8, rho, nipo) for t8 in range(8) for rho in range(16)}
        assert sorted(rows.values()) == list(range(128))                                    # (a)
        stream_len = 32 if nipo >= 32 else 16
        for g in range(4):
            for t8 in range(8):
                r4 = [rows[(t8, 4 * g + r)] for r in range(4)]
                assert r4 == list(range(r4[0], r4[0] + 4))                                  # registers = consecutive samples
                assert r4[0] % 4 == 0 and r4[0] // stream_len == r4[3] // stream_len        # inside one stream
            if nipo >= 32:   # stream g = rows [32g, 32g+32): tiles t8 = 0..7 walk it in order
                assert [rows[(t8, 4 * g)] for t8 in range(8)] == [32 * g + 4 * t8 for t8 in range(8)]
            else:            # streams (t8 >> 2, g) = 16 rows each: tiles 0-3 and 4-7 walk one stream each
                for hi in range(2):
                    assert [rows[(4 * hi + q, 4 * g)] for q in range(4)] == [64 * hi + 16 * g + 4 * q for q in range(4)]
        assert stream_len % min(nipo, stream_len) == 0                                       # (c)


def test_lds_swizzle_is_bank_conflict_free():
    """ds_read_b128: 4 groups of 16 lanes, bank = (addr/4) % 64; ds_write_b128: 8 groups of 8 contiguous lanes,
    bank = (addr/4) % 32 (MI355X_MICROARCH.md LDS table).  Conflict-free <=> the 16-byte slots are distinct.
    Mirrors fused16_kernel: fragment reads (lane -> row lds_row16(t8, lane & 15), piece swz16(lane >> 4 [+4])) and the
    staging writes of the 64- and 128-antenna layouts (plane 1 keeps its re / im halves swapped for that reason)."""
    groups_r = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
                list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
    groups_r += [[l + 32 for l in g] for g in groups_r]
    plane = 128 * 128
    for nipo in (2, 4, 8, 16, 32, 64):
        for t8 in range(8):
            for h in range(2):
                for comp in range(2):
                    for g in groups_r:
                        slots = set()
                        for lane in g:
                            row = _lds_row16(t8, lane & 15, nipo)
                            piece = (lane >> 4) + 4 * ((h & 1) ^ comp)
                            slots.add(((h * plane + row * 128 + 16 * _swz16(piece, row, nipo)) // 16) % 16)
                        assert len(slots) == 16, (nipo, t8, h, comp)
        for ppr in (4, 8):                      # 16-byte pieces per time sample: 64 and 128 antennas
            for k in range(128 * ppr // 256):
                for l0 in range(0, 256, 8):
                    for comp in range(2):
                        slots = set()
                        for tid in range(l0, l0 + 8):
                            row, pi = divmod(tid + 256 * k, ppr)
                            h, kp = divmod(pi, 4)
                            addr = h * plane + row * 128 + 16 * _swz16(kp + 4 * (h & 1), row, nipo)
                            slots.add(((addr ^ (64 * comp)) // 16) % 8)
                        assert len(slots) == 8, (nipo, ppr, k, l0, comp)
        # 100 antennas: 4-byte pieces (ds_write_b32, 64 lanes, bank = (addr/4) % 64): at most 2 lanes per bank
        for k in range(13):
            for w in range(4):
                for comp in range(2):
                    banks = {}
                    for lane in range(64):
                        pc = w * 64 + lane + 256 * k
                        if pc >= 3200:
                            continue
                        row, pi = divmod(pc, 25)
                        h, kp = divmod(pi // 4, 4)
                        addr = (h * plane + row * 128 + 16 * _swz16(kp + 4 * (h & 1), row, nipo) + 4 * (pi % 4)) ^ (64 * comp)
                        banks.setdefault((addr // 4) % 64, set()).add(addr)
                    assert max(len(v) for v in banks.values()) <= 2 if banks else True
