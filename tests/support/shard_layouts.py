"""TEST INFRASTRUCTURE (tests/test_shard_gloo.py): the gloo-testable statement of the frequency partition and of the two
gathered layouts.  The product's gather is bf_comm_create / bf_gather_detected behind the C-ABI (csrc/bf_comm.cpp); nothing
in dsabeamformer_amd/ or bench.py imports this file.

Frequency sharding across the GPUs of one node and the ONE collective of the path: the detected-power gather.

The reference scales by running one process per GPU on a different 256-channel slice (`-g`, README.md:168,
src/beamformer.cu:233) and has no communication at all.  Here rank r of R owns frequencies
[r*F/R, (r+1)*F/R) of every gemm-unit -- its slice of the weights ([f][a][b] is f-major) and of every input block
([unit][f][t][a]: one contiguous run per unit) -- runs the fused kernel on it, and the detected powers
[output][f_local][beam] are brought together with RCCL over xGMI (torch.distributed backend "nccl"):

  * ``alltoall`` (default): every rank becomes the owner of the FULL band for 1/R of the outputs (time slices).
    xGMI is point-to-point, so this uses all 7 links of every GPU in both directions at once instead of funnelling
    R-1 shards into one GPU's links, and it is the layout a downstream dedispersion search wants (all frequencies
    of a time range on one device).
  * ``root``: plain gather of every shard to rank 0 (what SURVEY.md section 8e calls "gather to the output owner").

The functions below are backend-agnostic (gloo on CPU in the tests, nccl on GPUs in bench.py).
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


class DetectedGather:
    """Double-buffered, asynchronous gather of detected powers.

    local layout  : [n_outputs][f_local][n_beams] float32 (what bf_beamform_device writes for a frequency shard)
    alltoall mode : result [n_outputs/R][F][n_beams] on every rank -- outputs r*n/R .. (r+1)*n/R-1 of the full band
    root mode     : result [n_outputs][F][n_beams] on rank 0, None elsewhere
    """

    def __init__(self, torch, dist, mode: str, n_outputs: int, n_freq_local: int, n_beams: int, device, group=None,
                 slots: int = 2):
        if mode not in ("alltoall", "root"):
            raise ValueError(mode)
        self.torch, self.dist, self.mode, self.group = torch, dist, mode, group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        if mode == "alltoall" and n_outputs % self.world:
            raise ValueError("n_outputs (%d) must be divisible by the number of ranks (%d)" % (n_outputs, self.world))
        self.no, self.fl, self.nb = n_outputs, n_freq_local, n_beams
        self.F = n_freq_local * self.world
        f32 = torch.float32
        if mode == "alltoall":
            self.recv = [torch.empty(n_outputs * n_freq_local * n_beams, dtype=f32, device=device) for _ in range(slots)]
            self.full = [torch.empty((n_outputs // self.world, self.F, n_beams), dtype=f32, device=device)
                         for _ in range(slots)]
        else:
            self.recv = [[torch.empty(n_outputs * n_freq_local * n_beams, dtype=f32, device=device)
                          for _ in range(self.world)] if self.rank == 0 else None for _ in range(slots)]
            self.full = [torch.empty((n_outputs, self.F, n_beams), dtype=f32, device=device) if self.rank == 0 else None
                         for _ in range(slots)]
        self.pending = [None] * slots
        # On GPUs the wait for the collective and the re-layout into [o][f][b] run on a side stream, so the compute
        # stream never waits for RCCL or for the (HBM-bound) permute copy of a step it is not about to overwrite.
        self.on_gpu = str(device).startswith("cuda")
        if self.on_gpu:
            self.side = torch.cuda.Stream(device=device)
            self.done = [torch.cuda.Event() for _ in range(slots)]

    def _assemble(self, slot: int):
        R = self.world
        if self.mode == "alltoall":
            # received [src rank = frequency shard][o_local][f_local][b] -> [o_local][shard][f_local][b]
            src = self.recv[slot].view(R, self.no // R, self.fl, self.nb)
            self.full[slot].view(self.no // R, R, self.fl, self.nb).copy_(src.permute(1, 0, 2, 3))
        elif self.rank == 0:
            dst = self.full[slot].view(self.no, R, self.fl, self.nb)
            for r in range(R):
                dst[:, r].copy_(self.recv[slot][r].view(self.no, self.fl, self.nb))

    def start(self, slot: int, local_out):
        """Launch the collective for `local_out` (flat or [n_outputs][f_local][n_beams]); returns immediately."""
        assert self.pending[slot] is None, "finish(slot) before reusing it"
        flat = local_out.reshape(-1)
        if self.mode == "alltoall":
            # chunk j of the send buffer = outputs j*n/R.. of MY frequencies -> rank j
            work = self.dist.all_to_all_single(self.recv[slot], flat, group=self.group, async_op=True)
        else:
            work = self.dist.gather(flat, self.recv[slot] if self.rank == 0 else None, dst=0,
                                    group=self.group, async_op=True)
        self.pending[slot] = work
        if self.on_gpu:
            with self.torch.cuda.stream(self.side):
                work.wait()            # stream-level: the side stream waits for RCCL, the host does not block
                self._assemble(slot)
                self.done[slot].record(self.side)

    def finish(self, slot: int):
        """Make the caller's stream wait for the collective of `slot` and return the assembled tensor in the reference
        layout [o][f][b] (also: `local_out` of that slot may be overwritten by work queued after this call)."""
        if self.pending[slot] is None:
            return self.full[slot]
        if self.on_gpu:
            self.torch.cuda.current_stream().wait_event(self.done[slot])
        else:
            self.pending[slot].wait()
            self._assemble(slot)
        self.pending[slot] = None
        return self.full[slot]


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
