"""The instantiation census, CPU side: the set of fused-kernel instantiations the reference's geometry
contract can select (bf_variant_key walked over N_ANTENNAS % 4, N_BEAMS % 4 -- src/beamformer.hh:155-156 -- windows, outputs per
gemm-unit, detect readings, general / conjugate-symmetric weights, detect and stage-parity launch; tools/census.py) IS the set
compiled into libdsabf.so.  Compiled but unreachable = dead code nobody can test: delete it.  Reachable but not compiled = a
geometry bf_create accepts and the first launch refuses: a bug.  The GPU side (tests/test_gpu_census.py) launches every one."""
import os
import re
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import census  # noqa: E402


@pytest.fixture(scope="module")
def sets():
    from dsabeamformer_amd import build

    build.build()
    return census.compiled(), census.reachable()


def test_reachable_set_is_the_compiled_set(sets):
    comp, reach = sets
    assert sorted(comp - set(reach)) == [], "compiled, but no geometry of the contract selects them"
    assert sorted(set(reach) - comp) == [], "selected by a geometry of the contract, but not in the library"
    assert len([k for k in comp if k.startswith("fusedg_kernel")]) == 8
    # the reference's own geometries and BASELINE config 5 select what DESIGN.md section 3 says they do
    by_geo = {(r["n_ant"], r["n_beams"], r["n_pol"] * r["n_avg"], r["mode"], r["paired"], r["write_c"]): k for k, r in reach.items()}
    import dsabeamformer_amd as bfm

    assert bfm.variant_key(bfm.production_config(), True) == "fused16_kernel<-1, 32, false, 0, true, 4, 4>"             # C3 / C4, the bench line
    assert bfm.variant_key(bfm.production_config(), False) == "fused16_kernel<-1, 32, false, 0, false, 4, 4>"
    assert bfm.variant_key(bfm.debug_config(), True) == "fused16_kernel<-1, 2, false, 0, true, 4, 4>"                    # C1 / C2
    c5 = bfm.production_config(n_ant=100, n_beams=512, n_freq=1024)
    assert bfm.variant_key(c5, True) == "fused16_kernel<100, 32, false, 0, true, 4, 8>"                                  # C5: 8 output slots per wave
    assert bfm.variant_key(c5, False) == "fused16_kernel<100, 32, false, 0, false, 8, 4>"                                # ... general: 8-wave workgroups
    assert by_geo  # (used above only to show the shape of the records)


def test_census_file_is_current(sets):
    """profiles/r06_instantiations.txt is the committed census: same instantiations as the library built from the tree."""
    comp, _ = sets
    path = os.path.join(ROOT, "profiles", "r06_instantiations.txt")
    listed = set()
    for line in open(path):
        m = re.match(r"(fused(?:16|g)_kernel<[^>]*>)\s", line)
        if m:
            assert "UNREACHABLE" not in line and "NOT COMPILED" not in line, line
            listed.add(m.group(1))
    assert listed == comp, (sorted(listed - comp)[:5], sorted(comp - listed)[:5])


def test_no_measurement_switch_reaches_the_library_outside_lab_mode(monkeypatch):
    """north_star: "no dual code paths".  Every getenv of csrc/ is on the production allow-list (documented in INTEGRATION.md) or
    goes through lab_getenv, which answers only in a process that says DSABF_LAB=1."""
    allow = {"DSABF_RCCL_LIB", "DSABF_THREADS", "DSABF_COALESCE", "DSABF_PAIRED", "DSABF_LAB"}
    csrc = os.path.join(ROOT, "dsabeamformer_amd", "csrc")
    seen = set()
    for f in sorted(os.listdir(csrc)):
        if not f.endswith((".cpp", ".hip", ".hpp", ".h", ".inc")):
            continue
        for m in re.finditer(r"(?<![_a-z])getenv\(\s*\"([A-Z0-9_]+)\"", open(os.path.join(csrc, f)).read()):
            seen.add(m.group(1))
            assert m.group(1) in allow, "%s reads %s from the environment outside DSABF_LAB mode" % (f, m.group(1))
    assert {"DSABF_RCCL_LIB", "DSABF_THREADS", "DSABF_COALESCE", "DSABF_PAIRED", "DSABF_LAB"} <= seen
    integration = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for name in allow:
        assert name in integration, name + " is not documented in INTEGRATION.md"
    # ... and the gate works: the same switch selects another launch only with DSABF_LAB=1
    import dsabeamformer_amd as bfm

    c5 = bfm.production_config(n_ant=100, n_beams=512, n_freq=1024)
    monkeypatch.setenv("DSABF_COL_TILES", "4")
    monkeypatch.setenv("DSABF_GENERIC", "1")
    assert bfm.variant_key(c5, True).startswith("fusedg_kernel")                 # (the test process runs in lab mode: conftest.py)
    monkeypatch.delenv("DSABF_LAB")
    assert bfm.variant_key(c5, True) == "fused16_kernel<100, 32, false, 0, true, 4, 8>"
    monkeypatch.setenv("DSABF_LAB", "10")                                        # only the exact value "1" is lab mode
    assert bfm.variant_key(c5, True) == "fused16_kernel<100, 32, false, 0, true, 4, 8>"
