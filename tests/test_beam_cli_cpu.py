"""CPU checks of the `beam` driver's command line (reference: src/beamformer.cu:41-127, usage() beamformer.hh:222-243)."""
import os
import subprocess

from conftest import ROOT

BEAM = os.path.join(ROOT, "dsabeamformer_amd", "beam")


def _build():
    from dsabeamformer_amd import build as b

    b.build()


def test_usage_text_matches_reference():
    _build()
    r = subprocess.run([BEAM, "-h"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0
    assert r.stdout == ("dsaX_beamformer_DEBUG_MODE [options]\n"
                        " -g gpu                  select a predefined frequency range\n"
                        " -p position_filename    file where the antenna positions are stored\n"
                        " -d direction_filename   file where the beam directions are stored\n"
                        " -s source_filename      file where the source directions are stored\n"
                        " -h                      print usage\n")


def test_fails_loudly_without_a_gpu():
    import torch

    if torch.cuda.is_available():
        return  # on the GPU box the real run is covered by tests/test_gpu_parity.py
    _build()
    r = subprocess.run([BEAM], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "GPUassert" in r.stderr  # no CPU fallback


def test_extended_usage_lists_what_this_build_adds():
    """`beam -h` is the reference's text, byte for byte; `beam -H` appends the options this build adds (observation sources and
    sinks, sharding, the DM stage), so that nobody has to read the driver's source to find them."""
    r = subprocess.run([BEAM, "-H"], capture_output=True, text=True, timeout=60)
    h = subprocess.run([BEAM, "-h"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and r.stdout.startswith(h.stdout) and "extensions of this build" in r.stdout
    for flag in ("-j n_blocks", "-R world -r rank -I id", "-M dm_max", "-W file | -Q ring", "-X ", "-w file | -K ring"):
        assert flag in r.stdout, flag
