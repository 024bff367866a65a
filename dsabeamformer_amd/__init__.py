"""dsabeamformer_amd -- MI355X-native (gfx950) DSA beamformer hot path.

The product is ``libdsabf.so`` (hand-written HIP kernels behind the C-ABI of ``include/dsabf.h``) plus the C++
host mirror of the reference's ``observation_loop_state`` / ``test_data_generator``.  This Python package is the
thin ctypes harness used by the tests, ``bench.py`` and the multi-GPU launcher; it never computes anything itself
and has no CPU fallback.
"""
from ._lib import BfConfig, DsabfError, load  # noqa: F401
from .api import Beamformer, debug_config, launch_plan, production_config, variant_key  # noqa: F401
