#!/usr/bin/env python3
"""Resource usage and instruction counts of the kernels in the gfx950 code objects dsabeamformer_amd/build.py produced.

    python tools/isa_report.py                      # every kernel of every object: VGPRs, AGPRs, SGPRs, scratch, LDS
    python tools/isa_report.py fused16_kernelILi64ELi32ELb0ELi0ELb1E   # + instruction counts of the kernels that match

The objects are the SHIPPED ones (dsabeamformer_amd/build/*.hip.o, compiled with build.flags_for(src): the per-file LLVM
scheduling strategies included), so the figures describe what libdsabf.so runs -- unlike a re-compile with hand-copied
flags.  tests/test_isa_guard_cpu.py asserts the budgets of the hot instantiations from the same functions.
"""
from __future__ import annotations

import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"


def code_object(obj: str, workdir: str) -> str | None:
    """The gfx950 code object embedded in a host object / shared library (its .hip_fatbin section, unbundled); None for an
    object without device code (a translation unit whose instantiations are all compiled out)."""
    base = os.path.join(workdir, os.path.basename(obj))
    r = subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + base + ".fat", obj,
                        base + ".copy"], capture_output=True, text=True)   # (an output name: never rewrite the input in place)
    if r.returncode != 0:
        if "not found" in r.stderr:
            return None
        raise RuntimeError(r.stderr)
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=" + TARGET,
                           "--input=" + base + ".fat", "--output=" + base + ".co"])
    return base + ".co"


FIELDS = ("agpr_count", "sgpr_count", "vgpr_count", "private_segment_fixed_size", "group_segment_fixed_size",
          "max_flat_workgroup_size", "sgpr_spill_count", "vgpr_spill_count")


def kernels(co: str) -> dict[str, dict[str, int]]:
    """{mangled kernel name: {metadata field: value}} from the code object's amdhsa.kernels note."""
    txt = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", co], text=True)
    out: dict[str, dict[str, int]] = {}
    for block in txt.split("  - .agpr_count:")[1:]:
        block = ".agpr_count:" + block
        name = re.search(r"\.name:\s+(\S+)", block).group(1)
        rec = {}
        for f in FIELDS:
            m = re.search(r"\." + f + r":\s+(\d+)", block)
            if m:
                rec[f] = int(m.group(1))
        out[name] = rec
    return out


def disassembly(co: str, name: str) -> list[str]:
    """Instruction mnemonics of one kernel."""
    txt = subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn",
                                   "--disassemble-symbols=" + name, co], text=True)
    ops = []
    for line in txt.splitlines():
        m = re.match(r"^\s+([a-z_0-9]+)\b", line)
        if m and not line.lstrip().startswith("//"):
            ops.append(m.group(1))
    return ops


def count(ops: list[str], pattern: str) -> int:
    r = re.compile(pattern)
    return sum(1 for o in ops if r.match(o))


def shipped_objects() -> list[str]:
    d = os.path.join(ROOT, "dsabeamformer_amd", "build")
    return sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith(".hip.o"))


def main() -> None:
    pat = sys.argv[1] if len(sys.argv) > 1 else None
    with tempfile.TemporaryDirectory() as wd:
        for obj in shipped_objects():
            co = code_object(obj, wd)
            if co is None:
                continue
            for name, rec in sorted(kernels(co).items()):
                if pat and pat not in name:
                    continue
                line = "%-28s %s vgpr %3d agpr %3d sgpr %3d scratch %4d lds %6d" % (
                    os.path.basename(obj)[:-6], name, rec.get("vgpr_count", -1), rec.get("agpr_count", -1), rec.get("sgpr_count", -1),
                    rec.get("private_segment_fixed_size", -1), rec.get("group_segment_fixed_size", -1))
                if pat:
                    ops = disassembly(co, name)
                    line += "  | insts %d mfma %d valu_f32 %d v_pk %d ds_read %d ds_write %d scratch_ops %d s_waitcnt %d" % (
                        len(ops), count(ops, "v_mfma"), count(ops, r"v_(fma|fmac|fmaak|fmamk|mul|add|sub)_f32"), count(ops, "v_pk_"),
                        count(ops, "ds_read"), count(ops, "ds_write"), count(ops, "scratch_"), count(ops, "s_waitcnt"))
                print(line)


if __name__ == "__main__":
    main()
