#!/usr/bin/env python3
"""Generates tests/golden/notebook_linear.npz by EXECUTING the reference's own Python beamformer.

The only implementation of beamform + detect + frequency collapse the reference ships that can run without an
NVIDIA GPU is its validation notebooks (the ones its README's accuracy figures come from, README.md:202-211):

* ``sandbox/2D Beamformer.ipynb``  cells 1, 3, 5, 6, 8  (imports; antenna positions; Fourier-coefficient matrix A;
  ``np.sum(A)``; quantised source signals and the detected, frequency-summed output ``out[beam, source]``)
* ``sandbox/Beamformer Theory.ipynb`` cells 1, 2, 3     (the 1-D predecessor of the same computation)

This script opens the notebooks as JSON and ``exec``s those cells' source text as it stands.  Nothing of the notebooks is
restated or stored here; the cell text is read from /root/reference at run time, so the script only works in the build
container (the reference does not travel to the GPU box -- the .npz it writes does).  The notebooks are Python 2 /
numpy-1.x code, so the following MECHANICAL fixes are applied to the cell text before ``exec`` (each is a syntax or
removed-alias fix, none touches a formula):

  1. ``print "x"`` / ``print x``  statements                 -> ``print(x)``
  2. ``N_FREQ = tot_channels / n_gpus``                     -> ``//``   (Python 2 integer division: range(N_FREQ))
  3. ``gpu * tot_channels/(n_gpus-1)``                      -> ``//``   (same; gpu = 0, so the value is 0 either way)
  4. ``np.complex``                                         -> ``complex`` (alias removed from numpy >= 1.24)
  5. IPython ``%magic`` lines                                -> dropped
  6. matplotlib runs on the ``Agg`` backend and ``plt.show`` does nothing (there is no display)
  7. 2D notebook, between cell 5 and cell 8: ``multiproc = 0``.  Cell 5 runs in the multiprocessing mode the committed
     notebook output shows ("Make Array / Starting / Done", double-precision ``RawArray('d')``); cell 8's multiprocessing
     branch cannot run at all (it slices a ``RawArray`` into a list and ``+=`` an ndarray onto it, and reinterprets a
     real array of N_BEAMS*n_angles doubles as (n_angles, N_BEAMS) complex) -- the committed notebook loaded ``out`` from
     a cached .npy instead -- so cell 8 takes the notebook's own serial branch, selected by the notebook's own flag.
  8. Theory notebook cell 0 (``sys.path.append(os.chdir(...))``, imports) is replaced by the equivalent imports.

Run from the repo root:  python tests/golden/make_notebook_golden.py      (about 3 minutes, 8 processes for cell 5)
"""
import json
import os
import re
import sys
import tempfile
import types

import numpy as np

REF = "/root/reference/sandbox"
HERE = os.path.dirname(os.path.abspath(__file__))


def cell_source(nb_name, idx):
    with open(os.path.join(REF, nb_name)) as fp:
        nb = json.load(fp)
    cell = nb["cells"][idx]
    assert cell["cell_type"] == "code", (nb_name, idx)
    return "".join(cell["source"])


def mechanical_fixes(src):
    out = []
    for line in src.splitlines():
        if line.lstrip().startswith("%"):                       # 5
            continue
        m = re.match(r"^(\s*)print\s+(?!\()(.*)$", line)        # 1 (statement form only)
        if m:
            line = "%sprint(%s)" % (m.group(1), m.group(2))
        line = line.replace("N_FREQ = tot_channels / n_gpus", "N_FREQ = tot_channels // n_gpus")   # 2
        line = line.replace("gpu * tot_channels/(n_gpus-1)", "gpu * tot_channels//(n_gpus-1)")     # 3
        line = line.replace("np.complex)", "complex)")                                             # 4
        out.append(line)
    return "\n".join(out) + "\n"


def new_namespace(name):
    """A real module, registered in sys.modules, so functions defined by exec() pickle by reference for Pool.map."""
    mod = types.ModuleType(name)
    sys.modules[name] = mod
    return mod


def run_cells(mod, nb_name, indices):
    for idx in indices:
        src = mechanical_fixes(cell_source(nb_name, idx))
        print("--- executing %s cell %d (%d lines)" % (nb_name, idx, len(src.splitlines())), flush=True)
        exec(compile(src, "%s[cell %d]" % (nb_name, idx), "exec"), mod.__dict__)


def main():
    os.environ["MPLBACKEND"] = "Agg"                                # 6
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    plt.show = lambda *a, **k: None
    work = tempfile.mkdtemp(prefix="nbgolden_")
    os.makedirs(os.path.join(work, "bin"))
    os.chdir(work)                                                  # cell 8 saves its cache under ./bin

    # ---- 2D Beamformer.ipynb -------------------------------------------------------------------------------
    nb = "2D Beamformer.ipynb"
    m2 = new_namespace("nb2d_cells")
    run_cells(m2, nb, [1, 3, 5])
    assert m2.multiproc == 1 and m2.A.dtype == np.complex128 and m2.A.shape == (256, 256, 64)
    sum_a = complex(np.sum(m2.A))                                   # cell 6 is the bare expression np.sum(A)
    m2.multiproc = 0                                                # 7
    devnull = open(os.devnull, "w")
    stdout, sys.stdout = sys.stdout, devnull                        # cell 8 prints one line per (freq, source)
    try:
        run_cells(m2, nb, [8])
    finally:
        sys.stdout = stdout
    out2d = np.array(m2.out, dtype=np.float64)
    assert out2d.shape == (256, 1024) and m2.N_AVERAGING == 1 and m2.Antenna_positions == "Linear"
    last_signal = np.array(m2.signal)                               # source 1023 at frequency 255 (last loop iteration)
    a127 = np.round(m2.A * 127.0)                                   # the integers the notebook divided by 127
    assert np.abs(a127 / 127.0 - m2.A).max() < 1e-15

    # ---- Beamformer Theory.ipynb ---------------------------------------------------------------------------
    nb = "Beamformer Theory.ipynb"
    mt = new_namespace("nbtheory_cells")
    mt.__dict__.update(sys=sys, os=os, np=np, plt=plt)              # 8
    run_cells(mt, nb, [1, 2, 3])
    out_theory = np.array(mt.out, dtype=np.float64)
    assert out_theory.shape == (256, 1024) and mt.A.dtype == np.complex64

    def cplx_i8(x):   # complex integers -> int8 [..., 2]
        return np.stack([x.real, x.imag], axis=-1).astype(np.int8)

    np.savez_compressed(
        os.path.join(HERE, "notebook_linear.npz"),
        nb2d_out=out2d,                                             # out[beam, source], float64, sum over 256 freqs
        nb2d_sum_A=np.array([sum_a.real, sum_a.imag]),
        nb2d_A127_f0=cplx_i8(a127[0]), nb2d_A127_f128=cplx_i8(a127[128]), nb2d_A127_f255=cplx_i8(a127[255]),   # [beam][ant]
        nb2d_freq=np.array(m2.freq, dtype=np.float64),
        nb2d_pos=np.array(m2.pos, dtype=np.float64), nb2d_theta=np.array(m2.theta, dtype=np.float64),
        nb2d_source_angles=np.array(m2.T_angles, dtype=np.float64),
        nb2d_signal_src1023_f255=cplx_i8(last_signal),
        theory_out=out_theory.astype(np.float32),                   # float32 is 1e-4 of the tolerance it is used at
        theory_sum_A=np.array([complex(np.sum(mt.A)).real, complex(np.sum(mt.A)).imag]),
    )
    print("np.sum(A) (2D notebook) =", sum_a, " notebook prints 13295.149606299225 (SURVEY.md 8c)")
    print("out[0, 0..2] =", out2d[0, :3], " out.max =", out2d.max())
    print("wrote", os.path.join(HERE, "notebook_linear.npz"))


if __name__ == "__main__":
    main()
