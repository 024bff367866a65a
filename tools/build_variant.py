#!/usr/bin/env python3
"""Measurement builds beside the product library: `python tools/build_variant.py NAME "-DDSABF_X=1 ..."` compiles every
source of libdsabf.so with the extra flags into variants/NAME/libdsabf.so (git-ignored; travels to the GPU box).  A process
selects it with DSABF_LIB_PATH=variants/NAME/libdsabf.so (dsabeamformer_amd/_lib.py); the in-tree product library and its
flag stamp are never touched, so tests / bench / profiles cannot pick up an experiment by accident.
Built HERE (no GPU needed), so that A/B runs on the GPU box spend their minutes measuring, not compiling."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dsabeamformer_amd import build as b  # noqa: E402


def main():
    name, extra = sys.argv[1], (sys.argv[2].split() if len(sys.argv) > 2 else [])
    only = os.environ.get("VARIANT_ONLY", "").split()     # e.g. "bf_fused16_a64.hip bf_kernels.hip": reuse the product's other objects
    out = os.path.join(b.ROOT, "variants", name)
    os.makedirs(out, exist_ok=True)
    objs, procs = [], []
    for src in b.sources():
        base = os.path.basename(src)
        if only and base not in only:
            objs.append(os.path.join(b.PKG, "build", base + ".o"))
            continue
        obj = os.path.join(out, base + ".o")
        objs.append(obj)
        procs.append(subprocess.Popen([b.HIPCC] + b.flags_for(src) + extra + ["-c", src, "-o", obj]))
    if any(p.wait() for p in procs):
        sys.exit("compile failed")
    cxx = os.path.join(os.path.dirname(os.path.realpath(b.HIPCC)), "..", "lib", "llvm", "bin", "clang++")
    lib = os.path.join(out, "libdsabf.so")
    subprocess.check_call([cxx, "-shared", "-fPIC", "-o", lib] + objs + ["-lpthread", "-lrt"])
    for o in objs:
        if o.startswith(out):
            os.remove(o)
    print(lib)


if __name__ == "__main__":
    main()
