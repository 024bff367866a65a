import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
CFG = os.path.join(GOLDEN, "config")


# The library's measurement / test switches are read from the environment only in a process that says DSABF_LAB=1 (round 6: a
# production host's stray DSABF_* variable selects nothing).  The A/B tests set them with monkeypatch; whatever the CALLER's
# environment held is dropped first, so that only what a test sets takes effect.  tests/test_abi_cpu.py checks the gate itself.
LAB_SWITCHES = ("DSABF_WG_WAVES", "DSABF_COL_TILES", "DSABF_TSPLIT", "DSABF_LDS_PAD", "DSABF_DM_WIDE", "DSABF_GENERIC", "DSABF_DEEP",
                "DSABF_RTW", "DSABF_UNIT_LAUNCH", "DSABF_UNITS_PER_LAUNCH", "DSABF_GATHER_STAGED", "DSABF_SINK_THREADS",
                "DSABF_GATHER_SELF_RCCL")
for _k in LAB_SWITCHES:
    os.environ.pop(_k, None)
os.environ["DSABF_LAB"] = "1"


# ---- the GPU suite runs on a budget (<= 300 s on a fresh driver box) -------------------------------------
# The default `-m gpu` run keeps a spread of at most SWEEP_CAP cases of every parametrised GPU test (first, last and evenly
# between, in collection order -- every kernel family, antenna class and config keeps its representatives; tests/README.md maps
# SURVEY.md section 8's rows and BASELINE's configs to the tests that stay); DSABF_LONG_TESTS=1 runs every case (tools/
# refresh_profiles_r05.sh does, and commits the tail under profiles/).  A test that must always run in full says
# @pytest.mark.sweep_cap(n) with its own number.  CPU tests are never thinned.
LONG = os.environ.get("DSABF_LONG_TESTS") == "1"
SWEEP_CAP = 4


def sweep(every, default):
    """Parameter lists that are expensive per case: `default` in the budgeted run, `every` under DSABF_LONG_TESTS=1."""
    return list(every) if LONG else list(default)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "sweep_cap(n): cases of this parametrised GPU test kept in the budgeted run (default %d)" % SWEEP_CAP)


def pytest_collection_modifyitems(config, items):
    if LONG:
        return
    groups = {}
    for it in items:
        if it.get_closest_marker("gpu") is None:
            continue
        groups.setdefault((str(it.fspath), getattr(it, "originalname", None) or it.name), []).append(it)
    drop = set()
    for group in groups.values():
        m = group[0].get_closest_marker("sweep_cap")
        cap = int(m.args[0]) if m and m.args else SWEEP_CAP
        if len(group) <= cap:
            continue
        keep = {round(i * (len(group) - 1) / max(cap - 1, 1)) for i in range(cap)}
        drop.update(id(it) for k, it in enumerate(group) if k not in keep)
    if drop:
        gone = [it for it in items if id(it) in drop]
        items[:] = [it for it in items if id(it) not in drop]
        config.hook.pytest_deselected(items=gone)


@pytest.fixture(scope="session")
def orc():
    import oracle

    return oracle


@pytest.fixture(scope="session")
def linear_inputs(orc):
    g = orc.DEBUG_GEOM
    pos = orc.read_positions(os.path.join(CFG, "linear_positions.txt"), g.n_ant)
    dirs = orc.read_directions(os.path.join(CFG, "linear_directions.txt"), g.n_beams)
    src = orc.read_directions(os.path.join(CFG, "linear_source_directions_1024.txt"))
    return pos, dirs, src


@pytest.fixture(scope="session")
def linear_weights(orc, linear_inputs):
    pos, dirs, _ = linear_inputs
    return orc.make_weights(orc.DEBUG_GEOM, pos, dirs, 0)


@pytest.fixture(scope="session")
def notebook_integers(orc, linear_inputs, linear_weights):
    """The integers the reference's notebook itself used (tests/golden/make_notebook_golden.py, round 3): its A * 127 for all
    256 frequencies and its quantised signal for every (source, frequency, antenna), rebuilt from the committed differences
    to the oracle's a5 / a6 output and CHECKED against the SHA-256 of the notebook's full arrays.
    Returns (weights int8 [f][a][b][2], packed uint8 [source][f][a], notebook out float64 [beam][source], diff counts)."""
    import hashlib

    import numpy as np

    nb = np.load(os.path.join(GOLDEN, "notebook_linear.npz"))
    pos, _, src = linear_inputs
    w = linear_weights.copy()
    w.reshape(-1)[nb["nb2d_A127_diff_index"]] = nb["nb2d_A127_diff_value"]
    assert hashlib.sha256(w.tobytes()).digest() == nb["nb2d_A127_sha256"].tobytes(), "not the notebook's A * 127"
    g = orc.DEBUG_GEOM
    batch = orc.generate_test_data(g, pos, src, 0, 0, 1024)            # [source][f][t][a]; every t column is the same
    col = np.ascontiguousarray(batch[:, :, 0, :])
    col.reshape(-1)[nb["nb2d_packed_diff_index"]] = nb["nb2d_packed_diff_value"]
    assert hashlib.sha256(col.tobytes()).digest() == nb["nb2d_packed_sha256"].tobytes(), "not the notebook's signals"
    # the 16 raw sample sources say the same without the oracle
    assert np.array_equal(col[nb["nb2d_packed_sample_sources"]], nb["nb2d_packed_sample"])
    return w, col, nb["nb2d_out"], (len(nb["nb2d_A127_diff_index"]), len(nb["nb2d_packed_diff_index"]))


# a2 / a3 / a8 against the executed notebook ON IDENTICAL INTEGERS: what is left is fp32 rounding only.  Per frequency
# term at most (n_ipo + 4) * 2^-24 relative (include/dsabf.h: 4 roundings per sample term + n_ipo - 1 accumulate
# roundings), plus 255 roundings of the ascending-f sum of 256 non-negative terms.  Measured: 1.5e-6.
NOTEBOOK_INTEGER_TOL = (255 + 2 + 4) * 2.0 ** -24
