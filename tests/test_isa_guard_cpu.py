"""CPU-box guard on the code-object properties the measured speed depends on (VERDICT r03 "weak" 7): the shipped gfx950 objects
(dsabeamformer_amd/build/*.hip.o, compiled by build.py with its per-file LLVM scheduling strategies and launch_bounds budgets)
are unbundled and read with llvm-readelf / llvm-objdump.  A compiler bump or an innocent edit that adds scratch, drops a wave
per SIMD or duplicates / loses MFMAs in the unrolled chunk loop fails HERE instead of showing up as a slower GPU bench."""
import os
import re
import sys
import tempfile

import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_report  # noqa: E402


@pytest.fixture(scope="module")
def objects():
    from dsabeamformer_amd import build

    build.build()
    wd = tempfile.mkdtemp(prefix="isaguard")
    cos = {}
    for obj in isa_report.shipped_objects():
        co = isa_report.code_object(obj, wd)
        if co:
            cos[os.path.basename(obj)[:-len(".hip.o")]] = (co, isa_report.kernels(co))
    return cos


def fused(ant, nipo, mode, paired, waves, ns, write_c=False):
    a = "Li%dE" % ant if ant > 0 else "Lin%dE" % -ant
    return "_ZN5dsabf14fused16_kernelIL%sLi%dELb%dELi%dELb%dELi%dELi%dEEEvNS_9FusedArgsE" % (
        a[1:], nipo, int(write_c), mode, int(paired), waves, ns)


# (translation unit, kernel, VGPR budget = 512 / waves per SIMD the launch shape counts on, MFMAs in the unrolled chunk loop)
#   MFMAs per 128-sample chunk and wave: 8 row tiles x column tiles x (general: 4 = 2 chains x re | im; pair: 4 real products
#   per PAIR tile) x k-steps
HOT = [
    # (round 6: 64 antennas run the run-time class k1p16 = template argument -1, profiles/r06_class_fold_ab.txt)
    ("bf_fused16_k1p16", fused(-1, 32, 0, True, 4, 4), 128, 8 * 2 * 4 * 1),       # C3 headline: conjugate-pair kernel, 4 waves / SIMD
    ("bf_fused16_k1p16", fused(-1, 32, 0, False, 4, 4), 128, 8 * 4 * 4 * 1),      # C3 general kernel (calibrated weights), 4 waves / SIMD
    ("bf_fused16_k1p16", fused(-1, 32, 2, True, 4, 4), 128, 8 * 2 * 4 * 1),       # contracted readings
    ("bf_fused16_k1p16", fused(-1, 32, 2, False, 4, 4), 128, 8 * 4 * 4 * 1),
    ("bf_fused16_k1p16", fused(-1, 2, 0, True, 4, 4), 168, 8 * 2 * 4 * 1),        # C2 DEBUG geometry: 3 waves / SIMD
    ("bf_fused16_a100_s8", fused(100, 32, 0, True, 4, 8), 256, 8 * 4 * 4 * 2),  # C5 headline: 8 output slots per wave, 2 waves / SIMD
    ("bf_fused16_a100_w8", fused(100, 32, 0, False, 8, 4), 256, 8 * 4 * 4 * 2),  # C5 general kernel on 8-wave workgroups
    ("bf_fused16_k2p16_s8", fused(-3, 32, 0, True, 4, 8), 256, 8 * 4 * 4 * 2, 16),   # 128 antennas x 512 beams, pair: 3 dwords of scratch at 256
                                                                                      # registers, and as fast as the compile-time class was
    ("bf_fused16_k2p16_w8", fused(-3, 32, 0, False, 8, 4), 256, 8 * 4 * 4 * 2),
    ("bf_fused16_k4p16", fused(-5, 32, 0, False, 8, 2), 256, 8 * 2 * 4 * 4),      # deep general: 2 slots per wave, 4 k-steps
    ("bf_fused16_k4p16", fused(-5, 32, 0, True, 8, 4), 256, 8 * 2 * 4 * 4),       # deep pair: 2 pair tiles per wave
    ("bf_fused16_k3p16", fused(-6, 32, 0, False, 8, 2), 256, 8 * 2 * 4 * 3),
]


@pytest.mark.parametrize("hot", HOT, ids=[h[1][25:60] for h in HOT])
def test_hot_instantiations_keep_their_registers_no_scratch_and_their_mfma_count(objects, hot):
    unit, name, budget, mfmas = hot[:4]
    scratch_ok = hot[4] if len(hot) > 4 else 0        # bytes of scratch per lane this instantiation is known to carry (0: none)
    co, ks = objects[unit]
    assert name in ks, "instantiation missing from %s: %s" % (unit, name)
    k = ks[name]
    assert k["private_segment_fixed_size"] <= scratch_ok, "scratch (a spill) in %s" % name
    assert k.get("sgpr_spill_count", 0) == 0 and (scratch_ok or k.get("vgpr_spill_count", 0) == 0)
    assert k["vgpr_count"] + k["agpr_count"] <= budget, (name, k)
    ops = isa_report.disassembly(co, name)
    assert scratch_ok or isa_report.count(ops, "scratch_") == 0
    assert isa_report.count(ops, "v_mfma_i32_16x16x64_i8") == mfmas, "the unrolled chunk loop holds %d MFMAs" % isa_report.count(ops, "v_mfma")
    assert isa_report.count(ops, "v_mfma") == mfmas                    # ... and no other matrix instruction
    # the detect stays on plain fp32 VALU ops: packed f32 issues beside MFMAs at twice the price (MI355X_MICROARCH.md), and
    # -ffp-contract=off / -fno-slp-vectorize are what keep the compiler from forming them
    assert isa_report.count(ops, "v_pk_(add|mul|fma)_f32") == 0
    if "ELi0ELb" in name:   # canonical reading: x*x + y*y must not be contracted (bit-exact contract with the oracle)
        assert isa_report.count(ops, r"v_fma_f32|v_fmac_f32") == 0, "a contracted multiply-add in the canonical detect"


def test_spills_stay_where_they_are_known(objects):
    """No fused kernel of the 16-byte-staged run-time classes (64, 128, 192, 256 antennas among them) or of the wide (8-slot /
    8-wave) 100-antenna launches carries scratch.  The classes that do are listed: the dword-staged run-time
    classes (13 staging pieces per thread), 100 antennas on 4-wave workgroups at n_ipo 2 / 4 / 64 (general kernel; the wide
    launches replace it from n_ipo 16 on), one 8-slot run-time instantiation -- a few dwords each, bounded here so that growth
    shows."""
    clean_units = ("bf_fused16_a100_s8", "bf_fused16_a100_w8", "bf_fused16_a100_w8p", "bf_fused16_k1p16", "bf_fused16_k2p16",
                   "bf_fused16_k2p16_w8", "bf_fused16_k2p16_w8p", "bf_fused16_k2p4_w8p", "bf_fused16_k3p16", "bf_fused16_k4p16")
    bad, worst = [], 0
    for unit, (co, ks) in objects.items():
        for name, k in ks.items():
            if "fused16_kernel" not in name:
                continue
            sc = k["private_segment_fixed_size"]
            if unit in ("bf_fused16_k3p4", "bf_fused16_k4p4"):
                # the dword-staged deep classes (round 5: 132, 140, ... antennas; 12 / 16 staging pieces per thread beside 3 / 4 k-steps
                # of weight fragments): three k-steps fit, four carry up to 432 bytes and still beat fusedg_kernel by 17 ... 96 %
                # (profiles/r05_deep_p4_perf.txt); bounded on their own so that growth shows
                assert sc <= (0 if unit == "bf_fused16_k3p4" else 448), (unit, name, sc)
                continue
            worst = max(worst, sc)
            rtw = re.search(r"fused16_kernelIL[in0-9]+ELi0E", name) is not None   # run-time-window instantiations: not hot, may carry a few dwords
            if sc and unit in clean_units and not rtw:
                bad.append((unit, name, sc))
    assert not bad, bad
    assert worst <= 320, "a fused kernel now spills %d bytes per lane" % worst


def test_generic_kernel_budget(objects):
    """fusedg_kernel (every geometry outside the specialised classes): two 4-wave workgroups per CU = 2 waves per SIMD = 256
    registers, no scratch in the detect instantiations (a scratch access is a vmcnt event in the middle of the MFMA stream),
    128 MFMAs = two copies of the 64-MFMA plane (the B register sets swap roles), no conditional branch between the MFMAs of a
    plane's tile loop that would hide loads from the s_waitcnt pass: no `s_waitcnt vmcnt(0)` in front of an MFMA."""
    co, ks = objects["bf_fusedg"]
    names = [n for n in ks if "fusedg_kernel" in n]
    assert len(names) == 8
    for n in names:
        k = ks[n]
        write_c = n.endswith("Lb1EEEvNS0_7GenArgsE")
        assert k["vgpr_count"] + k["agpr_count"] <= 256, (n, k)
        if not write_c:
            assert k["private_segment_fixed_size"] == 0, (n, k)
            ops = isa_report.disassembly(co, n)
            assert isa_report.count(ops, "v_mfma_i32_16x16x64_i8") == 128
            assert isa_report.count(ops, "v_pk_(add|mul|fma)_f32") == 0
    # the wait in front of an MFMA group is for LDS data only (lgkmcnt); vmcnt(0) there means a load got under a branch again
    import subprocess

    txt = subprocess.check_output([os.path.join(isa_report.LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn",
                                   "--disassemble-symbols=" + [n for n in names if "Lb1ELi0ELb0E" in n][0], co], text=True)
    lines = [l.strip() for l in txt.splitlines()]
    bad = 0
    for i, l in enumerate(lines):
        if l.startswith("v_mfma") and not lines[i - 1].startswith("v_mfma"):
            j = i - 1
            while j > 0 and lines[j].startswith(("s_nop", "v_mov", "v_add", "s_add", "s_mov")):
                j -= 1
            if lines[j].startswith("s_waitcnt") and "vmcnt(0)" in lines[j]:
                bad += 1
    assert bad <= 2, "%d MFMA groups wait for vmcnt(0)" % bad     # (the first group of a plane may wait for its B fragments)


def test_dm_kernels(objects):
    co, ks = objects["bf_dm_wide"]
    name = [n for n in ks if "dedisperse_dm_wide_kernel" in n][0]
    k = ks[name]
    assert k["private_segment_fixed_size"] == 0 and k["vgpr_count"] <= 128       # 16-wave workgroups: 4 waves per SIMD
    ops = isa_report.disassembly(co, name)
    assert isa_report.count(ops, "ds_read2_b64") == 0     # the 8-byte-aligned pair read measured slower (profiles/r03_variants_log.txt)
    assert isa_report.count(ops, "scratch_") == 0
    # The kernel is bound by instruction issue (every instruction per wave and channel ~0.7 % of its time): its loop
    # must keep the scalar-only staging path (buffer-addressed LDS-DMA), the unrolling over the ring's three positions (five copies
    # of the 160-add body, a barrier per copy) and its size -- a compiler that re-inflates the bookkeeping shows up here first.
    assert isa_report.count(ops, "buffer_load_dwordx4") >= 15 and isa_report.count(ops, "v_pk_add_f32") == 5 * 160
    assert isa_report.count(ops, "s_barrier") >= 8
    assert len(ops) <= 4400 and sum(1 for o in ops if o.startswith("s_")) <= 2050, (len(ops), "instructions")
    co, ks = objects["bf_kernels"]
    name = [n for n in ks if "dedisperse_dm_kernel" in n][0]
    assert ks[name]["private_segment_fixed_size"] == 0 and ks[name]["vgpr_count"] <= 128
    name = [n for n in ks if "expand_kernel" in n][0]
    assert ks[name]["vgpr_count"] <= 32 and ks[name]["private_segment_fixed_size"] == 0


def test_gather_relayout_kernel_streams_whole_16_byte_pieces(objects):
    """The staged gather transport's device pass (round 5) is an HBM-bound copy: its loop must be 16-byte nontemporal loads and
    stores, four loads in flight before the first store, nothing spilled -- a compiler that splits the pieces or serialises
    load -> store shows up here, not as 0.4 of the HBM roofline on the first multi-GPU node."""
    co, ks = objects["bf_kernels"]
    name = [n for n in ks if "gather_relayout_kernel" in n][0]
    k = ks[name]
    assert k["private_segment_fixed_size"] == 0 and k["vgpr_count"] <= 64
    import subprocess

    txt = subprocess.check_output([os.path.join(isa_report.LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", "--disassemble-symbols=" + name, co],
                                  text=True)
    ops = [l.strip() for l in txt.splitlines() if l.startswith("\t")]
    loads = [o for o in ops if o.startswith("global_load_dwordx4")]
    stores = [o for o in ops if o.startswith("global_store_dwordx4")]
    assert len(loads) >= 5 and len(stores) >= 5                    # the unrolled body (4 + 4) and the tail loop (1 + 1)
    assert all(" nt" in o for o in loads + stores), (loads[:2], stores[:2])
    assert not [o for o in ops if o.startswith(("global_load_dword ", "global_store_dword ", "global_load_dwordx2", "global_store_dwordx2"))]
    # four loads issued back to back before the first store of the unrolled body
    first_store = next(i for i, o in enumerate(ops) if o.startswith("global_store_dwordx4"))
    before = [o for o in ops[:first_store] if o.startswith("global_load_dwordx4")]
    assert len(before) >= 4


def test_generated_dm_body_is_what_the_generator_writes():
    """csrc/bf_dm_wide_body.inc is generated (tools/gen_dm_body.py) and committed: the two must not drift apart."""
    import gen_dm_body

    assert open(gen_dm_body.PATH).read() == gen_dm_body.render()
