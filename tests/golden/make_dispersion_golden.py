#!/usr/bin/env python3
"""Generates tests/golden/dispersion_notebook.npz by EXECUTING the reference's dispersion design notebook.

``sandbox/Dispersion Theory.ipynb`` is the only statement of the dedispersion step the reference holds:
cell 1 (instrument constants), cell 2 (the DM trial ladder, "Number of trials = 1627" in the committed output) and cell 5
(a DM-2000 pulse drawn into a [2048 channel][1000 sample] array: the per-channel sample delays).  This script opens the
notebook as JSON and ``exec``s those cells' text as it stands -- nothing is restated here, the cell text is read from
/root/reference at run time (build container only; the .npz travels).  Mechanical steps, none touching a formula:

  1. IPython ``%magic`` lines dropped; matplotlib on the ``Agg`` backend, ``plt.show`` / ``plt.imshow`` left to draw into it.
  2. ``np.random.seed(20260)`` before cell 5, so that the uniform noise the cell draws into ``C`` can be drawn again and
     subtracted: what is left of ``C`` is the pulse (2.0 spread over three samples by the cell's own ``np.convolve``), and
     its position per channel is the delay the CELL used.
  3. The name ``int`` in the cell namespace is a recording stand-in for the builtin (it returns ``int(x)`` and keeps x):
     cell 5 calls it twice per channel on ``d*(-f1**(-2) + f**(-2))/(0.131*16)``.  Two independent read-outs of the same
     2048 delays -- the recorded ``int()`` results and the pulse positions recovered from ``C`` -- must agree.

Run from the repo root:  python tests/golden/make_dispersion_golden.py
"""
import builtins
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference/sandbox/Dispersion Theory.ipynb"
HERE = os.path.dirname(os.path.abspath(__file__))


# The notebook is untrusted upstream content and its cells are exec()'d below: it is pinned by content.  A notebook that differs
# from the one this script was written against (and reviewed: cells 1, 2 and 5 only compute numpy arrays) is refused.
REF_SHA256 = "bfff3221ead7c8b6cfae71cedbeb475fb8c886286611997ae0d9d59fe2fe32fc"


def cell_source(idx):
    import hashlib

    raw = open(REF, "rb").read()
    if hashlib.sha256(raw).hexdigest() != REF_SHA256:
        raise SystemExit("refusing to execute %s: its SHA-256 is not the reviewed notebook's" % REF)
    nb = json.loads(raw.decode("utf-8"))
    cell = nb["cells"][idx]
    assert cell["cell_type"] == "code"
    return "\n".join(l for l in "".join(cell["source"]).splitlines() if not l.lstrip().startswith("%")) + "\n"   # 1


class RecordingInt:
    def __init__(self):
        self.args, self.results = [], []

    def __call__(self, x):
        r = builtins.int(x)
        self.args.append(float(x))
        self.results.append(r)
        return r


def main():
    os.environ["MPLBACKEND"] = "Agg"
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    plt.show = lambda *a, **k: None
    ns = types.ModuleType("dispersion_cells").__dict__
    ns.update(np=np, plt=plt)
    for idx in (1, 2):
        exec(compile(cell_source(idx), "Dispersion Theory.ipynb[cell %d]" % idx, "exec"), ns)
    dms = np.array(ns["dms"], dtype=np.float64)
    assert len(dms) == 1627                                            # the committed notebook prints "Number of trials = 1627"
    rec = RecordingInt()
    ns["int"] = rec                                                    # 3
    seed = 20260
    np.random.seed(seed)                                               # 2
    exec(compile(cell_source(5), "Dispersion Theory.ipynb[cell 5]", "exec"), ns)
    del ns["int"]
    C = np.array(ns["C"])
    nchan, max_time, start = ns["Nchan"], ns["max_time"], ns["start_index"]
    assert C.shape == (2048, max_time) and nchan == 2048 and len(rec.results) % 2 == 0
    # read-out A: the int() results (two calls per channel while the pulse is inside the array)
    calls_per_chan = len(rec.results) // nchan
    assert calls_per_chan == 2 and len(rec.results) == 2 * nchan, "every channel's pulse lies inside the array"
    delays_int = np.array(rec.results[0::2], dtype=np.int32)
    assert np.array_equal(delays_int, np.array(rec.results[1::2]))
    args = np.array(rec.args[0::2], dtype=np.float64)
    # read-out B: the pulse positions in C once the (re-drawn) noise is taken out
    np.random.seed(seed)
    noise = np.random.uniform(size=(2048, max_time))
    delays_c = np.empty(nchan, np.int32)
    for i in range(nchan):
        pulse = C[i] - np.convolve(noise[i], np.ones(3), "same")
        hot = np.flatnonzero(pulse > 1.0)
        assert len(hot) == 3 and hot[2] - hot[0] == 2 and np.allclose(pulse[hot], 2.0), (i, hot)
        delays_c[i] = hot[1] - start
    assert np.array_equal(delays_c, delays_int)
    np.savez_compressed(
        os.path.join(HERE, "dispersion_notebook.npz"),
        dms=dms,                                                       # cell 2: the whole ladder, 1627 trials to DM 2000
        constants=np.array([ns["Nchan"], ns["epsilon"], ns["nu"], ns["B"], ns["ti"], ns["tscat"], ns["tsamp"]], dtype=np.float64),
        delays_dm2000=delays_int,                                      # cell 5: samples, channel i at 1.28 + 0.25/2048*i GHz
        delay_args_dm2000=args,                                        # the real numbers int() truncated
        f_ref_ghz=np.array([ns["f1"]]), d_over_dm=np.array([ns["d"] / 2000.0]), tsamp_ms=np.array([0.131 * 16]),
    )
    print("trials %d, dm[1] %.17g, dm[-1] %.17g; delays: chan 0 %d, chan 2047 %d, sum %d"
          % (len(dms), dms[1], dms[-1], delays_int[0], delays_int[-1], int(delays_int.sum())))
    print("wrote", os.path.join(HERE, "dispersion_notebook.npz"))


if __name__ == "__main__":
    main()
