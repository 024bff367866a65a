"""Builds libdsabf.so (HIP kernels + C-ABI runtime + C++ host mirror) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels to the GPU box with the snapshot.
"""
from __future__ import annotations

import glob
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libdsabf.so")
ROOT = os.path.dirname(PKG)

HIPCC = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
# -ffp-contract=off: the detect stage's x*x + y*y must be two multiplies and one add (bit-exact contract).
# -fno-slp-vectorize: keep the detect epilogue on plain fp32 VALU ops (packed v_pk_*_f32 do not co-execute with MFMA).
# -amdgpu-sched-strategy=max-ilp: the fused kernel's unrolled tile loop schedules 1-2 % faster than with the default
#   occupancy-driven strategy (profiles/r01_variants_paired_log.txt); register counts stay inside every variant's budget.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize", "-Wall",
         "-Wno-unused-function", "-mllvm", "-amdgpu-sched-strategy=" + os.environ.get("DSABF_SCHED", "max-ilp"),
         "-I" + os.path.join(ROOT, "include")] + os.environ.get("DSABF_EXTRA_FLAGS", "").split()
# The 8-wave-workgroup kernels of the two-k-step classes schedule better with other LLVM strategies than max-ilp, so they live in
# translation units of their own (profiles/r03_ab_sched.txt, one box, interleaved): the general kernel (csrc/bf_fused16_*_w8.hip)
# with iterative-maxocc -4.3 % (C5), -5.0 % contracted, -6.3 % on a C5 rank shard (iterative-ilp: -2.5 / -2.2 / -3.6 %); the
# conjugate-pair kernel (bf_fused16_*_w8p.hip) with iterative-ilp -4.2 % (iterative-maxocc: -1.8 %).  Every other kernel is
# neutral or worse with either (C3 +-0.5 ... +1.1 %, the 8-slot pair kernel +0.2 ... +0.5 %) and keeps max-ilp.
# Round 4: the deep classes (three / four k-steps, 8-wave workgroups; bf_fused16_{a192,a256,k3p16,k4p16}.hip) with iterative-ilp: general
# kernel -6.3 % at 256 antennas (iterative-maxocc -3.8 %), conjugate-pair kernel -3.1 % (profiles/r04_deep_sched.txt).
# Round 6: the compile-time classes a192 / a256 were folded into k3p16 / k4p16 (profiles/r06_class_fold_ab.txt), which already had the strategy.
SCHED_BY_SUFFIX = {"_w8.hip": "iterative-maxocc", "_w8p.hip": "iterative-ilp",
                   "_k3p16.hip": "iterative-ilp", "_k4p16.hip": "iterative-ilp", "_k3p4.hip": "iterative-ilp", "_k4p4.hip": "iterative-ilp"}


def flags_for(src: str) -> list[str]:
    """FLAGS with the scheduling strategy of this source (an explicit DSABF_SCHED in the environment wins everywhere)."""
    if "DSABF_SCHED" in os.environ:
        return list(FLAGS)
    for suffix, sched in SCHED_BY_SUFFIX.items():
        if src.endswith(suffix):
            return [("-amdgpu-sched-strategy=" + sched) if f.startswith("-amdgpu-sched-strategy=") else f for f in FLAGS]
    return list(FLAGS)


# Optional: the real PSRDADA input adapter (csrc/bf_dada.cpp).  libpsrdada is not in this image, so it is off
# by default; DSABF_WITH_PSRDADA=1 [PSRDADA_INCLUDE=dir PSRDADA_LIB=dir] builds it and links -lpsrdada (makefile:5-6,13).
WITH_PSRDADA = os.environ.get("DSABF_WITH_PSRDADA") == "1"
if WITH_PSRDADA:
    FLAGS += ["-DDSABF_WITH_PSRDADA"] + (["-I" + os.environ["PSRDADA_INCLUDE"]] if os.environ.get("PSRDADA_INCLUDE") else [])
PSRDADA_LINK = ((["-L" + os.environ["PSRDADA_LIB"]] if os.environ.get("PSRDADA_LIB") else []) + ["-lpsrdada"]) if WITH_PSRDADA else []


BEAM = os.path.join(PKG, "beam")                      # the `beam` CLI driver (reference: bin/beam)
BEAM_SRC = os.path.join(CSRC, "beam_main.cpp")
JUNKDB = os.path.join(PKG, "junkdb")                  # writer side of the shared-memory input ring (dada_db + dada_junkdb)
JUNKDB_SRC = os.path.join(CSRC, "junkdb_main.cpp")
REPLICAS = os.path.join(PKG, "beam_replicas")          # one `beam` per GPU: the reference's 8 replicas, or one sharded sub-band
REPLICAS_SRC = os.path.join(CSRC, "beam_replicas_main.cpp")
MAINS = {BEAM: BEAM_SRC, JUNKDB: JUNKDB_SRC, REPLICAS: REPLICAS_SRC}


def sources() -> list[str]:
    """Sources of libdsabf.so (everything under csrc/ except the CLI driver)."""
    return sorted(p for p in glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp"))
                  if os.path.abspath(p) not in [os.path.abspath(m) for m in MAINS.values()])


STAMP = os.path.join(PKG, "build", "flags.stamp")   # the flag set the in-tree library was built with
KID_STAMP = os.path.join(PKG, "build", "kernel_id.stamp")   # the kernel build id compiled into bf_version()
RUNTIME_SRC = os.path.join(CSRC, "bf_runtime.cpp")


def kernel_build_id() -> str:
    """Identity of the device code: a hash over every kernel source (csrc/*.hip, *.hpp, *.h, *.inc) and the flag set they are compiled
    with.  Compiled into the library (bf_version() ends in "kernels <id>") and written into every rocprofv3 counter summary under
    profiles/ (tools/pmc.sh), so that bench.py pairs live timings only with counters of THE SAME kernels."""
    import hashlib

    h = hashlib.sha256(_stamp().encode())
    for p in sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(os.path.join(CSRC, "*.h")) +
                    glob.glob(os.path.join(CSRC, "*.inc"))):
        h.update(os.path.basename(p).encode())
        h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def _stamp() -> str:
    # (the tree's own location is not part of the flag set: the same tree under another path -- the GPU box's scratch copy, a
    #  checkout elsewhere -- is the same build, and bf_version()'s kernel build id must not depend on where it was compiled)
    return (" ".join(FLAGS) + " | " + repr(sorted(SCHED_BY_SUFFIX.items()))).replace(ROOT, "<root>")


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    # a library built with other flags (an experiment's DSABF_EXTRA_FLAGS, tools/variants.sh) is stale even if it is newer
    # than every source: never let tests / bench / profiles silently run a variant build
    if not os.path.exists(STAMP) or open(STAMP).read() != _stamp():
        return True
    if not os.path.exists(KID_STAMP) or open(KID_STAMP).read() != kernel_build_id():
        return True
    t = os.path.getmtime(LIB)
    if not all(os.path.exists(b) for b in MAINS):
        return True
    deps = list(MAINS.values()) + sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.hpp")) + \
        glob.glob(os.path.join(ROOT, "include", "*.h*")) + [os.path.abspath(__file__)]
    return any(os.path.getmtime(p) > t for p in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every source under csrc/ into dsabeamformer_amd/libdsabf.so."""
    if not force and not _stale():
        return LIB
    objs = []
    objdir = os.path.join(PKG, "build")
    os.makedirs(objdir, exist_ok=True)
    procs = []
    kid = kernel_build_id()
    kid_same = os.path.exists(KID_STAMP) and open(KID_STAMP).read() == kid
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        flags_same = os.path.exists(STAMP) and open(STAMP).read() == _stamp()
        is_runtime = os.path.abspath(src) == os.path.abspath(RUNTIME_SRC)   # carries the kernel build id: rebuilt whenever that changes
        if not force and flags_same and (kid_same or not is_runtime) and os.path.exists(obj) and os.path.getmtime(obj) > max(
                [os.path.getmtime(src)] + [os.path.getmtime(p) for p in glob.glob(os.path.join(CSRC, "*.h*"))] +
                [os.path.getmtime(p) for p in glob.glob(os.path.join(ROOT, "include", "*.h*"))]):
            continue
        # .hip: device + host; .cpp: plain host C++ (HIP host API only), also through hipcc for the include paths
        cmd = [HIPCC] + flags_for(src) + (['-DDSABF_KERNEL_BUILD_ID="%s"' % kid] if is_runtime else []) + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for stale in glob.glob(os.path.join(objdir, "*.o")):   # objects of sources that no longer exist (tools link build/*.o by glob)
        if stale not in objs:
            os.remove(stale)
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), out))
        if verbose and out.strip():
            print(out)
    # Link WITHOUT a DT_NEEDED on libamdhip64: hip* symbols stay undefined and bind to the HIP runtime that is
    # already in the process (a C++ application links -lamdhip64 itself; Python callers get torch's bundled runtime
    # preloaded by _lib.load()).  Linking /opt/rocm's libamdhip64.so.7 here would put a SECOND HIP runtime next to
    # torch's, and streams/events created by one are meaningless to the other.
    cxx = os.path.join(os.path.dirname(os.path.realpath(HIPCC)), "..", "lib", "llvm", "bin", "clang++")
    if not os.path.exists(cxx):
        cxx = shutil.which("g++") or "g++"
    cmd = [cxx, "-shared", "-fPIC", "-o", LIB] + objs + ["-lpthread", "-lrt"] + PSRDADA_LINK
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    with open(STAMP, "w") as fp:
        fp.write(_stamp())
    with open(KID_STAMP, "w") as fp:
        fp.write(kid)
    # the CLI programs are ordinary HIP applications: they link libdsabf.so AND the HIP runtime
    for exe, src in MAINS.items():
        if src == REPLICAS_SRC:   # a plain launcher: no HIP, no libdsabf (it must not initialise a GPU before exec)
            cmd = [shutil.which("g++") or "g++", "-O2", "-std=c++17", src, "-o", exe]
        else:
            cmd = [HIPCC, "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"), src, "-o", exe, "-L" + PKG, "-ldsabf",
                   "-Wl,-rpath,$ORIGIN"] + (["-DDSABF_WITH_PSRDADA"] if WITH_PSRDADA else [])
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in os.sys.argv, verbose=True))
