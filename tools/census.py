#!/usr/bin/env python3
"""Instantiation census: every fused kernel compiled into libdsabf.so against what the reference's geometry
contract can select.

  compiled()   the kernel instantiations IN the shipped library: the `__device_stub__` symbols of `nm -C libdsabf.so`, as
               "fused16_kernel<-1, 32, false, 0, true, 4, 4>" / "fusedg_kernel<true, 0, false>"
  reachable()  {instantiation: smallest geometry that selects it}: bf_variant_key (host arithmetic, no GPU) walked over the
               contract -- N_ANTENNAS % 4, N_BEAMS % 4 (src/beamformer.hh:155-156), any accumulation window n_pol * n_avg, any
               outputs per gemm-unit, the three detect readings, general / conjugate-symmetric weights, the detect launch and the
               stage-parity launch (bf_gemm_device) -- with NO measurement switch set

tests/test_census_cpu.py asserts reachable == compiled; tests/test_gpu_census.py launches every reachable instantiation on its
smallest geometry, asserts the handle runs THAT instantiation and compares with the oracle.

  python tools/census.py > profiles/r06_instantiations.txt     # the census file: one line per instantiation
"""
from __future__ import annotations

import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# The walk.  Antennas: every multiple of 4 up to 256 (all classes of fused16_kernel) and a few beyond (fusedg_kernel).  Beams: what
# switches the launch shape -- below / at / above one workgroup's 256, % 32 (pairing), % 512 (8 output slots, deep pair tiles), odd and
# even counts of 256-beam groups (8-wave workgroups).  Windows: every compile-time one, run-time ones short and long.  Outputs per
# gemm-unit: gemm-units that are / are not whole 16-sample runs.
ANTS = list(range(4, 257, 4)) + [260, 272, 320, 512, 1024]
BEAMS = [4, 16, 32, 36, 64, 96, 256, 288, 384, 512, 768, 1024]
WINDOWS = [(1, 1), (2, 1), (1, 3), (2, 2), (2, 3), (2, 4), (2, 6), (2, 8), (2, 12), (2, 16), (2, 20), (2, 32), (2, 48), (2, 64)]   # (n_pol, n_avg)
OUTS = [1, 2, 3, 8]
MODES = [0, 2, 1]   # BF_DETECT_CANONICAL, _CONTRACTED, _FAST


def compiled(lib_path: str | None = None) -> set[str]:
    lib_path = lib_path or os.path.join(ROOT, "dsabeamformer_amd", "libdsabf.so")
    txt = subprocess.check_output(["nm", "-C", lib_path], text=True)
    out = set()
    for m in re.finditer(r"__device_stub__(fused(?:16|g)_kernel<[^>]*>)\(", txt):
        out.add(m.group(1))
    return out


def other_kernels(lib_path: str | None = None) -> list[str]:
    """The library's kernels that are not instantiations of the two fused templates (one each: nothing to select)."""
    lib_path = lib_path or os.path.join(ROOT, "dsabeamformer_amd", "libdsabf.so")
    txt = subprocess.check_output(["nm", "-C", lib_path], text=True)
    return sorted({m.group(1) for m in re.finditer(r"__device_stub__([a-z_0-9]+)\(", txt)})


def interleaved(key: str, n_beams: int) -> bool:
    """Does a launch of instantiation `key` over n_beams deal the beams to a wave's column tiles round robin (16- / 8-byte vector
    stores, whole 128-byte lines) or tile by tile (scalar stores, the last tile partly filled)?  A run-time argument of the kernel
    (FusedArgs::interleave; bf_kernels.hip interleaved(), bf_fusedg.hip generic_interleave()): both store paths live in every
    instantiation, and which one runs depends on the beam count alone."""
    if key.startswith("fusedg_kernel"):
        return n_beams % 32 == 0
    ns = int(key.rstrip(">").split(",")[-1])
    return n_beams % (16 * ns) == 0


def reachable(both_store_paths: bool = False) -> dict:
    """{instantiation: smallest geometry that selects it}; both_store_paths: {(instantiation, interleaved stores?): smallest geometry}
    -- an instantiation appears once or twice, as the contract reaches one or both of its store paths."""
    import ctypes as C

    import dsabeamformer_amd as bfm

    assert not any(os.environ.get(k) for k in ("DSABF_GENERIC", "DSABF_DEEP", "DSABF_RTW", "DSABF_WG_WAVES", "DSABF_COL_TILES")), \
        "the census walks the contract without measurement switches"
    lib = bfm.load()
    buf = C.create_string_buffer(120)
    cfg = bfm.production_config(n_freq=2, n_gemms_per_block=1, n_blocks_on_gpu=1, n_streams=1)
    best: dict[str, dict] = {}
    for n_ant in ANTS:
        cfg.n_ant = n_ant
        for n_beams in BEAMS:
            cfg.n_beams = n_beams
            for n_pol, n_avg in WINDOWS:
                cfg.n_pol, cfg.n_avg = n_pol, n_avg
                for n_out in OUTS:
                    cfg.n_out_per_gemm = n_out
                    cost = n_ant * n_beams * n_pol * n_avg * n_out
                    for mode in MODES:
                        cfg.detect_mode = mode
                        for paired in (0, 1):
                            for write_c in (0, 1):
                                if write_c and (paired or mode):
                                    continue   # (the stage-parity launch ignores both: nothing new to find)
                                rc = lib.bf_variant_key(C.byref(cfg), paired, write_c, buf, 120)
                                if rc != 0:
                                    continue   # outside the product's contract (e.g. beyond 2048 antennas): bf_create refuses too
                                key = buf.value.decode()
                                assert key, (n_ant, n_beams, n_pol, n_avg, n_out, mode, paired, write_c)
                                if both_store_paths:
                                    key = (key, interleaved(key, n_beams))
                                rec = best.get(key)
                                if rec is None or cost < rec["cost"]:
                                    best[key] = dict(n_ant=n_ant, n_beams=n_beams, n_pol=n_pol, n_avg=n_avg, n_out=n_out, mode=mode,
                                                     paired=paired, write_c=write_c, cost=cost)
    return best


def resources() -> dict[str, dict]:
    """{instantiation key: {unit, vgprs, agprs, scratch}} from the shipped objects (tools/isa_report.py)."""
    import tempfile

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_report

    wd = tempfile.mkdtemp(prefix="census")
    out = {}
    for obj in isa_report.shipped_objects():
        co = isa_report.code_object(obj, wd)
        if not co:
            continue
        ks = isa_report.kernels(co)
        names = [n for n in ks if "fused16_kernel" in n or "fusedg_kernel" in n]
        if not names:
            continue
        dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
        for n, d in zip(names, dem):
            m = re.search(r"(fused(?:16|g)_kernel<[^>]*>)\(", d)
            if m:
                out[m.group(1)] = dict(unit=os.path.basename(obj)[:-len(".hip.o")], vgprs=ks[n].get("vgpr_count", -1), agprs=ks[n].get("agpr_count", 0),
                                       scratch=ks[n].get("private_segment_fixed_size", 0))
    return out


def main() -> int:
    comp, reach, res = compiled(), reachable(), resources()
    lib = os.path.join(ROOT, "dsabeamformer_amd", "libdsabf.so")
    print("# instantiation census of dsabeamformer_amd/libdsabf.so (%d bytes): tools/census.py" % os.path.getsize(lib))
    print("# fused16_kernel<antenna class, window (0: run-time), stage-parity store, detect mode, conjugate-pair, waves / workgroup, output slots / wave>")
    print("#   antenna class: 100 = compile-time; -1 k1p16, -2 k1p4 (<= 64 antennas, 16- / 4-byte rows), -3 k2p16, -4 k2p4 (<= 128), -6 k3p16, -8 k3p4 (<= 192), -5 k4p16, -7 k4p4 (<= 256)")
    print("# fusedg_kernel<16-byte rows, detect mode, stage-parity store>;  detect mode 0 canonical, 1 fast, 2 contracted")
    print("# compiled %d, reachable over the contract %d, compiled but unreachable %d, reachable but not compiled %d"
          % (len(comp), len(reach), len(comp - set(reach)), len(set(reach) - comp)))
    print("# other kernels (one instantiation each): " + ", ".join(other_kernels()))
    print("# %-52s %-22s %5s %7s  smallest geometry that selects it" % ("instantiation", "translation unit", "vgprs", "scratch"))
    for key in sorted(comp | set(reach)):
        r, g = res.get(key, {}), reach.get(key)
        geo = ("ant %d beams %d n_pol %d n_avg %d n_out %d mode %d %s%s" % (g["n_ant"], g["n_beams"], g["n_pol"], g["n_avg"], g["n_out"], g["mode"],
                                                                            "paired" if g["paired"] else "general", " (bf_gemm_device)" if g["write_c"] else "")) if g else "UNREACHABLE"
        print("%-54s %-22s %5s %7s  %s%s" % (key, r.get("unit", "?"), r.get("vgprs", "?") if not r.get("agprs") else "%d+%d" % (r["vgprs"], r["agprs"]),
                                            r.get("scratch", "?"), geo, "" if key in comp else "  NOT COMPILED"))
    return 0 if comp == set(reach) else 1


if __name__ == "__main__":
    sys.exit(main())
