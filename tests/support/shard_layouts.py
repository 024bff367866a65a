"""TEST INFRASTRUCTURE (tests/test_shard_gloo.py): the frequency partition of SURVEY.md section 8e stated for numpy / torch
arrays, and the rank-ordered sum of sub-band dedispersion partials.  The product's gather is bf_comm_create /
bf_gather_detected behind the C-ABI (csrc/bf_comm.cpp), and it is the PRODUCT's plan (bf_gather_plan) that the gloo test
executes between its processes; nothing in dsabeamformer_amd/ or bench.py imports this file.

The reference scales by running one process per GPU on a different 256-channel slice (`-g`, README.md:168,
src/beamformer.cu:233) and has no communication at all.  Here rank r of R owns frequencies [r*F/R, (r+1)*F/R) of every
gemm-unit -- its slice of the weights ([f][a][b] is f-major) and of every input block ([unit][f][t][a]: one contiguous run
per unit).
"""
from __future__ import annotations


def freq_range(rank: int, world: int, n_freq_total: int) -> tuple[int, int]:
    if n_freq_total % world:
        raise ValueError("n_freq_total (%d) must be divisible by the number of ranks (%d)" % (n_freq_total, world))
    n = n_freq_total // world
    return rank * n, (rank + 1) * n


def shard_weights(w, rank: int, world: int):
    """w: [F][ant][beam][2] -> this rank's [F/R][ant][beam][2] slice (a view)."""
    f0, f1 = freq_range(rank, world, w.shape[0])
    return w[f0:f1]


def shard_packed(packed, rank: int, world: int):
    """packed: [unit][F][time][ant] -> this rank's [unit][F/R][time][ant] slice (copy into contiguous memory)."""
    f0, f1 = freq_range(rank, world, packed.shape[1])
    sl = packed[:, f0:f1]
    return sl.contiguous() if hasattr(sl, "contiguous") else sl.copy()


def reduce_dedispersed(torch, dist, partial, group=None):
    """Sub-band dedispersion across frequency shards (SURVEY.md section 8f-4 on the section 8e partition).

    Each rank dedisperses ITS channels (bf_dedisperse_dm_device on its local detected series, delays computed for its
    channels against the band-wide reference frequency): ``partial`` [n_dm][n_t][n_beams] float32.  The band-wide
    result is the sum of the partials; to keep it bit-reproducible (fp32 addition is not associative and a ring
    all-reduce fixes no order a caller can name) the partials are all-gathered and added in rank order -- ascending
    frequency, the same order the single-device kernel uses within a band.  Every rank returns the full result.
    """
    world = dist.get_world_size(group)
    if world == 1:
        return partial.clone()
    flat = partial.contiguous().reshape(-1)
    gathered = torch.empty(world * flat.numel(), dtype=partial.dtype, device=partial.device)
    dist.all_gather_into_tensor(gathered, flat, group=group)
    parts = gathered.reshape((world,) + tuple(partial.shape))
    out = parts[0].clone()
    for r in range(1, world):
        out += parts[r]
    return out
