import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
CFG = os.path.join(GOLDEN, "config")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
