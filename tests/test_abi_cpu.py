"""CPU-side checks of the C-ABI boundary: the library loads, exports every symbol include/dsabf.h declares,
and the geometry helpers reproduce the reference's compile-time constants.  No compute calls (no GPU here)."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def lib():
    from dsabeamformer_amd import build as b
    from dsabeamformer_amd import _lib

    b.build()
    return _lib.load()


HEADERS = ("dsabf.h", "dsabf_bench.h", "dsabf_host.h")


def _declared_symbols(headers=HEADERS):
    names = set()
    for hdr in headers:
        text = open(os.path.join(ROOT, "include", hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(bfh?_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


CITATION = re.compile(r"(src/[\w.]+|README\.md|makefile|config/[\w.]+):\d+")


def test_every_export_of_the_boundary_header_cites_the_reference_call_site_it_replaces():
    """include/dsabf.h is the drop-in boundary and nothing else -- every function it declares must say,
    in the comment block in front of it (shared by the declarations that follow one comment) or in a comment on its own line,
    which lines of the reference it replaces (file:line).  The measurement knobs live in include/dsabf_bench.h, which includes
    the boundary header, cites nothing and is named by no call-site map."""
    text = open(os.path.join(ROOT, "include", "dsabf.h")).read()
    pieces = re.split(r"(/\*.*?\*/)", text, flags=re.S)       # comment, code, comment, code, ...
    last_comment, checked, prev_code = "", [], "\n"
    for piece in pieces:
        if piece.startswith("/*"):
            if prev_code.rsplit("\n", 1)[-1].strip() == "":      # a block of its own (not trailing a declaration on its line)
                last_comment = piece
            continue
        prev_code = piece
        for m in re.finditer(r"\b(bf_[a-z0-9_]+)\s*\([^;{]*\)\s*;", piece, flags=re.S):
            checked.append(m.group(1))
            # the comment in front of this run of declarations; a trailing comment belongs to the declaration before it
            tail = text[text.index(m.group(0)) + len(m.group(0)):].split("\n", 1)[0]
            assert CITATION.search(last_comment) or CITATION.search(tail), \
                "%s: neither the comment in front of it nor one on its line cites a reference call site" % m.group(1)
    assert sorted(set(checked)) == _declared_symbols(("dsabf.h",)) and len(checked) >= 45
    bench_only = set(_declared_symbols(("dsabf_bench.h",))) - set(_declared_symbols(("dsabf.h",)))
    assert bench_only == {"bf_set_switch", "bf_get_counter", "bf_mfma_peak_device", "bf_launch_plan", "bf_kernel_info", "bf_rtw_plan", "bf_kernel_name",
                          "bf_gather_relayout_device", "bf_variant_key", "bf_handle_variant_key"}
    assert not bench_only & set(checked)
    integ = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for name in bench_only:                                   # the call-site map binds the boundary, not the lab bench
        assert name not in integ.split("## 2")[0] or "dsabf_bench.h" in integ, name


def test_library_exports_every_declared_symbol(lib):
    from dsabeamformer_amd import _lib

    names = _declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), "libdsabf.so does not export %s" % n
    # and the ctypes signature table covers exactly the header
    assert sorted(_lib.SIGNATURES) == names


def test_default_configs_match_reference_constants(lib):
    from dsabeamformer_amd import BfConfig

    dbg, prod = BfConfig(), BfConfig()
    assert lib.bf_config_default(C.byref(dbg), 1) == 0 and lib.bf_config_default(C.byref(prod), 0) == 0
    # SURVEY.md section 0 table (src/beamformer.hh:47-152)
    for cfg, n_avg, n_ipo, n_time, per_gemm, per_block in ((dbg, 1, 2, 16, 256 << 10, 8 << 20),
                                                           (prod, 16, 32, 256, 4 << 20, 128 << 20)):
        assert (cfg.n_beams, cfg.n_ant, cfg.n_freq, cfg.n_pol, cfg.n_avg) == (256, 64, 256, 2, n_avg)
        assert (cfg.n_out_per_gemm, cfg.n_gemms_per_block, cfg.n_blocks_on_gpu, cfg.n_streams) == (8, 32, 8, 8)
        assert lib.bf_n_inputs_per_output(C.byref(cfg)) == n_ipo
        assert lib.bf_n_timesteps_per_gemm(C.byref(cfg)) == n_time
        assert lib.bf_bytes_per_gemm(C.byref(cfg)) == per_gemm
        assert lib.bf_bytes_per_block(C.byref(cfg)) == per_block
        assert lib.bf_floats_per_detect(C.byref(cfg)) == 524288
        assert cfg.detect_mode == 0  # BF_DETECT_CANONICAL: bit-exact by default


def test_error_convention_without_gpu(lib):
    """Return codes + bf_last_error instead of exit() (src/beamformer.cuh:19-29 exits)."""
    from dsabeamformer_amd import BfConfig

    cfg = BfConfig()
    lib.bf_config_default(C.byref(cfg), 1)
    h = C.c_void_p()
    cfg.n_beams = 250  # not a multiple of 32 -> invalid before any device work
    assert lib.bf_create(C.byref(cfg), 0, C.byref(h)) == -1
    assert b"n_beams" in lib.bf_last_error() or b"N_BEAMS" in lib.bf_last_error()
    cfg.n_beams = 256
    cfg.n_ant = 2052  # the exactly convertible range of the int32 sums ends at 2048 antennas
    assert lib.bf_create(C.byref(cfg), 0, C.byref(h)) == -1
    assert b"2048 antennas" in lib.bf_last_error()
    assert not h.value
    cfg.n_ant, cfg.n_avg = 64, 3  # n_ipo = 6: since round 4 a supported geometry -> the refusal is the missing GPU
    assert lib.bf_create(C.byref(cfg), 0, C.byref(h)) == -3
    assert not h.value
    # which kernel a geometry runs is host arithmetic (bf_launch_plan): the reference's whole contract has one
    name = C.create_string_buffer(200)
    for n_ant, n_avg, expect in ((64, 16, b"fused16_kernel"), (100, 16, b"fused16_kernel"), (64, 3, b"fused16_kernel<ANT=64,NIPO=6(run-time)>"),
                                 (132, 16, b"fused16_kernel<ANT=132,NIPO=32,WAVES=8>"), (132, 32, b"fusedg_kernel<ANT=132 (3 k-steps, 4-byte staging)"),
                                 (324, 16, b"(6 k-steps, 4-byte staging)"), (320, 16, b"(5 k-steps, 16-byte staging)"),
                                 (256, 16, b"fused16_kernel<ANT=256,NIPO=32,WAVES=8>"), (144, 8, b"fused16_kernel<ANT=144,NIPO=16,WAVES=8>")):
        cfg.n_ant, cfg.n_avg = n_ant, n_avg
        assert lib.bf_launch_plan(C.byref(cfg), 0, 4, 256, None, None, None, name, 200) == 0
        assert expect in name.value, name.value
    assert lib.bf_beamform_device(None, None, 1, None, None) == -1
    assert lib.bf_destroy(None) == 0


def test_product_does_not_reference_the_oracle():
    """The product path must never import, link or fall back to oracle/."""
    pkg = os.path.join(ROOT, "dsabeamformer_amd")
    for dirpath, _, files in os.walk(pkg):
        if os.path.basename(dirpath) in ("build", "__pycache__"):
            continue
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hip", ".h", ".hpp")):
                src = open(os.path.join(dirpath, fn)).read()
                assert "liborc" not in src and "import oracle" not in src and "dsabf_oracle" not in src, fn
    import subprocess

    out = subprocess.run(["ldd", os.path.join(pkg, "libdsabf.so")], capture_output=True, text=True).stdout
    assert "liborc" not in out


def test_headers_and_example_compile_as_plain_c(tmp_path):
    """The boundary is a C ABI: dsabf.h / dsabf_host.h and the worked example must compile as pedantic C99 with gcc."""
    import subprocess

    from conftest import ROOT

    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I" + os.path.join(ROOT, "include"), "-x", "c",
                        "-fsyntax-only", os.path.join(ROOT, "include", "dsabf_bench.h")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    for name in ("minimal", "sharded", "dm_stream"):     # single GPU; one rank of a frequency-sharded run; the DM stage of the loop
        obj = tmp_path / (name + ".o")
        r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I" + os.path.join(ROOT, "include"),
                            "-c", os.path.join(ROOT, "examples", name + ".c"), "-o", str(obj)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def test_product_library_ships_no_test_double_and_no_hard_runtime_dependencies():
    """libdsabf.so exports what the headers declare and nothing test-only (round 1 shipped a fake event backend);
    it has no DT_NEEDED on a HIP runtime or on RCCL (both bind to what the process already has)."""
    import subprocess

    lib = os.path.join(ROOT, "dsabeamformer_amd", "libdsabf.so")
    syms = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True).stdout
    assert "fake" not in syms.lower()
    exported_c = set(re.findall(r" T (bfh?_[a-z0-9_]+)$", syms, flags=re.M))
    assert exported_c == set(_declared_symbols())
    needed = subprocess.run(["readelf", "-d", lib], capture_output=True, text=True).stdout
    assert "amdhip" not in needed and "rccl" not in needed


def test_unloadable_rccl_is_an_error_code_not_a_crash():
    """`dlerror() ? dlerror() : ...` called dlerror twice -- the second call returns NULL and std::string + NULL
    segfaults -- so a librccl that cannot be loaded killed the process instead of returning BF_ERR_DEVICE."""
    import subprocess
    import sys

    code = ("import sys; sys.path.insert(0, %r)\n"
            "import dsabeamformer_amd as bfm\nfrom dsabeamformer_amd import api\n"
            "try:\n"
            "    api.comm_unique_id()\n"
            "except bfm.DsabfError as e:\n"
            "    print('DsabfError:', e)\n"
            "    sys.exit(0)\n"
            "sys.exit(3)\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, DSABF_RCCL_LIB="/nonexistent/librccl.so"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "DSABF_RCCL_LIB=/nonexistent/librccl.so could not be loaded" in r.stdout


def test_bench_has_one_gather_path():
    """a scaling line must never come from a second code path -- bench.py imports no torch.distributed
    gather, and the gloo statement of the layouts lives under tests/ only."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "shard_layouts" not in src and "DetectedGather" not in src and "import shard" not in src
    assert not os.path.exists(os.path.join(ROOT, "dsabeamformer_amd", "shard.py"))


def test_launch_plan_picks_the_launch_of_every_class_without_a_device(monkeypatch):
    """bf_launch_plan: the host arithmetic behind fused_launch_shape / fused_wg_waves / fused_col_tiles (csrc/bf_kernels.hip) --
    which kernel, workgroup size and grid a configuration runs -- needs no GPU.  BASELINE configs[1] (C3), configs[4] (C5) and
    its rank shard, and the geometries on either side of every rule."""
    import dsabeamformer_amd as bfm

    def plan(paired, n_units, **kw):
        return bfm.launch_plan(bfm.production_config(**kw), paired, n_units)

    c3 = plan(True, 128, n_out_per_gemm=16)
    assert c3 == {"kernel": "dsabf::fused16_kernel<ANT=64,NIPO=32,PAIRED> (v_mfma_i32_16x16x64_i8)", "grid": 6656, "block": 256,
                  "lds_bytes": 32768}
    assert plan(False, 128, n_out_per_gemm=16)["grid"] == 6656
    c5 = dict(n_ant=100, n_beams=512, n_freq=1024, n_out_per_gemm=8)
    # conjugate-pair kernel, two k-steps, beams in whole groups of 512: 8 output slots per wave, one set of long workgroups
    p = plan(True, 16, **c5)
    assert "PAIRED,SLOTS=8" in p["kernel"] and (p["grid"], p["block"], p["lds_bytes"]) == (1024, 256, 65536)
    # the general kernel (calibrated weights) cannot hold 8 slots: 8-wave workgroups, one per CU
    g = plan(False, 16, **c5)
    assert g["kernel"].endswith("NIPO=32,WAVES=8> (v_mfma_i32_16x16x64_i8)") and (g["grid"], g["block"]) == (1024, 512)
    # one rank's shard of C5 (128 channels): two resident 8-slot workgroups per CU -> 512; one 8-wave workgroup per CU -> 256
    assert plan(True, 16, **dict(c5, n_freq=128))["grid"] == 512 and plan(False, 16, **dict(c5, n_freq=128))["grid"] == 256
    # 8 slots need whole groups of 512 beams and an instantiation that fits its registers; otherwise 8 waves, or the plain launch
    assert "WAVES=8" in plan(True, 16, **dict(c5, n_beams=480))["kernel"]            # ragged: 8-wave workgroups
    assert "WAVES=8" in plan(True, 16, **dict(c5, n_ant=108))["kernel"]              # run-time dword-staged class
    assert "SLOTS=8" in plan(True, 16, **dict(c5, n_ant=112))["kernel"]              # run-time 16-byte-staged class
    assert "SLOTS=8" in plan(True, 16, **dict(c5, n_ant=128))["kernel"]
    assert "WAVES=8" in plan(True, 16, **dict(c5, n_avg=32))["kernel"]               # 100 antennas at n_ipo 64 would spill
    assert "SLOTS=8" in plan(True, 16, **dict(c5, n_ant=128, n_avg=32))["kernel"]
    for kw in (dict(c5, n_beams=256), dict(c5, n_beams=768), dict(c5, n_avg=4, n_out_per_gemm=32), dict(c5, n_ant=64)):
        q = plan(True, 16, **kw)                                                     # odd group count / short window / one k-step
        assert "S=8" not in q["kernel"] and q["block"] == 256, kw
    # the test switches give the plain launch back (read once per handle: here per call)
    monkeypatch.setenv("DSABF_COL_TILES", "4")
    assert "WAVES=8" in plan(True, 16, **c5)["kernel"]
    monkeypatch.setenv("DSABF_WG_WAVES", "4")
    q = plan(True, 16, **c5)
    assert "S=8" not in q["kernel"] and (q["grid"], q["block"]) == (4096, 256)
    monkeypatch.delenv("DSABF_COL_TILES")
    monkeypatch.delenv("DSABF_WG_WAVES")
    # errors: a geometry outside the reference's contract, a non-positive unit count
    import ctypes as C
    lib = bfm.load()
    bad = bfm.production_config(n_beams=6)
    assert lib.bf_launch_plan(C.byref(bad), 0, 1, 256, None, None, None, None, 0) != 0 and b"N_BEAMS" in lib.bf_last_error()
    assert lib.bf_launch_plan(C.byref(bfm.production_config()), 0, 0, 256, None, None, None, None, 0) != 0
