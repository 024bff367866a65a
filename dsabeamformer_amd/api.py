"""Object wrapper over the C-ABI (include/dsabf.h) for Python callers that own device memory through torch.

Names follow the reference: a *gemm-unit* is the work of one cublasGemmStridedBatchedEx call
(src/beamformer.cu:470-477), a *block* is a PSRDADA block of ``n_gemms_per_block`` gemm-units, a *beam-block* is
one detected ``[n_freq][n_beams]`` float32 output.
"""
from __future__ import annotations

import ctypes as C

from ._lib import BfConfig, check, load


def debug_config(**over) -> BfConfig:
    """The reference's ``make debug`` geometry (N_AVERAGING = 1, src/beamformer.hh:55-57)."""
    cfg = BfConfig()
    check(load().bf_config_default(C.byref(cfg), 1))
    for k, v in over.items():
        setattr(cfg, k, v)
    return cfg


def production_config(**over) -> BfConfig:
    """The reference's production geometry (N_AVERAGING = 16, src/beamformer.hh:59)."""
    cfg = BfConfig()
    check(load().bf_config_default(C.byref(cfg), 0))
    for k, v in over.items():
        setattr(cfg, k, v)
    return cfg


def launch_plan(cfg: BfConfig, paired: bool, n_units: int = 1, n_cus: int = 256) -> dict:
    """Which fused kernel and launch shape ``cfg`` would run (bf_launch_plan: host arithmetic, no device needed)."""
    g, b, l = C.c_int(), C.c_int(), C.c_int()
    name = C.create_string_buffer(200)
    check(load().bf_launch_plan(C.byref(cfg), int(paired), n_units, n_cus, C.byref(g), C.byref(b), C.byref(l), name, 200))
    return {"kernel": name.value.decode(), "grid": g.value, "block": b.value, "lds_bytes": l.value}


def variant_key(cfg: BfConfig, paired: bool, write_c: bool = False) -> str:
    """The compiled kernel instantiation ``cfg`` selects, as its demangled symbol spells the template arguments
    (bf_variant_key: host arithmetic; the census of tests/test_census_cpu.py)."""
    buf = C.create_string_buffer(120)
    check(load().bf_variant_key(C.byref(cfg), int(paired), int(write_c), buf, 120))
    return buf.value.decode()


def _ptr(x) -> C.c_void_p:
    """Accept ints, ctypes pointers, numpy arrays (host) and torch tensors (device or host)."""
    if x is None:
        return C.c_void_p(0)
    if isinstance(x, int):
        return C.c_void_p(x)
    if hasattr(x, "data_ptr"):
        return C.c_void_p(x.data_ptr())
    if hasattr(x, "ctypes"):
        return C.c_void_p(x.ctypes.data)
    return x


class Beamformer:
    """One handle = one GPU = one frequency shard (the reference runs one process per GPU, README.md:168)."""

    def __init__(self, cfg: BfConfig, device: int = 0):
        self._lib = load()
        self._h = C.c_void_p()
        self.cfg = cfg
        check(self._lib.bf_create(C.byref(cfg), device, C.byref(self._h)))

    # -- geometry -----------------------------------------------------------------------------------------
    @property
    def n_ipo(self) -> int:
        return self._lib.bf_n_inputs_per_output(C.byref(self.cfg))

    @property
    def n_time(self) -> int:
        return self._lib.bf_n_timesteps_per_gemm(C.byref(self.cfg))

    @property
    def bytes_per_gemm(self) -> int:
        return self._lib.bf_bytes_per_gemm(C.byref(self.cfg))

    @property
    def bytes_per_block(self) -> int:
        return self._lib.bf_bytes_per_block(C.byref(self.cfg))

    @property
    def floats_per_detect(self) -> int:
        return self._lib.bf_floats_per_detect(C.byref(self.cfg))

    # -- setup --------------------------------------------------------------------------------------------
    def set_weights(self, w_host) -> None:
        """``w_host``: int8 host array [freq][ant][beam][2] (reference layout)."""
        check(self._lib.bf_set_weights(self._h, _ptr(w_host)))

    def set_weights_device(self, d_w, stream: int = 0) -> None:
        check(self._lib.bf_set_weights_device(self._h, _ptr(d_w), C.c_void_p(stream)))

    # -- device-pointer entry points -------------------------------------------------------------------------
    def beamform(self, d_packed, n_units: int, d_out, stream: int = 0) -> None:
        check(self._lib.bf_beamform_device(self._h, _ptr(d_packed), int(n_units), _ptr(d_out), C.c_void_p(stream)))

    def expand(self, d_in, nbytes: int, d_out, stream: int = 0) -> None:
        check(self._lib.bf_expand_device(self._h, _ptr(d_in), int(nbytes), _ptr(d_out), C.c_void_p(stream)))

    def gemm(self, d_packed_unit, d_c, stream: int = 0) -> None:
        check(self._lib.bf_gemm_device(self._h, _ptr(d_packed_unit), _ptr(d_c), C.c_void_p(stream)))

    def dedisperse(self, d_out_unit, d_ded, stream: int = 0) -> None:
        check(self._lib.bf_dedisperse_device(self._h, _ptr(d_out_unit), _ptr(d_ded), C.c_void_p(stream)))

    def dedisperse_dm(self, d_series, n_t: int, d_delays, n_dm: int, n_t_out: int, d_out, stream: int = 0) -> None:
        """d_series float32 [n_t][freq][beam], d_delays int32 [n_dm][freq] -> d_out float32 [n_dm][n_t_out][beam]."""
        check(self._lib.bf_dedisperse_dm_device(self._h, _ptr(d_series), int(n_t), _ptr(d_delays), int(n_dm),
                                                int(n_t_out), _ptr(d_out), C.c_void_p(stream)))

    def dedisperse_band(self, d_out_unit, n_freq_total: int, d_ded, stream: int = 0) -> None:
        check(self._lib.bf_dedisperse_band_device(self._h, _ptr(d_out_unit), n_freq_total, _ptr(d_ded), C.c_void_p(stream)))

    def dedisperse_dm_band(self, d_series, n_t: int, n_freq_total: int, d_delays, n_dm: int, n_t_out: int, d_out,
                           stream: int = 0) -> None:
        check(self._lib.bf_dedisperse_dm_band_device(self._h, _ptr(d_series), n_t, n_freq_total, _ptr(d_delays), n_dm, n_t_out,
                                                     _ptr(d_out), C.c_void_p(stream)))

    # -- streaming entry points (the reference's observation loop) ---------------------------------------------
    def submit_block(self, slot: int, host, nbytes: int, event=None) -> None:
        check(self._lib.bf_submit_block(self._h, slot, _ptr(host), nbytes, _ptr(event)))

    def enqueue_gemm_unit(self, stream_idx: int, slot: int, time_slice: int, host_out=None) -> None:
        check(self._lib.bf_enqueue_gemm_unit(self._h, stream_idx, slot, time_slice, _ptr(host_out)))

    def enqueue_block(self, stream_idx: int, slot: int, first_unit: int, n_units: int, host_outs=None) -> None:
        """One launch over n_units consecutive gemm-units of a ring slot; host_outs: n_units host pointers (or None)."""
        arr = None
        if host_outs is not None:
            arr = (C.c_void_p * n_units)(*[_ptr(p).value for p in host_outs])
        check(self._lib.bf_enqueue_block(self._h, stream_idx, slot, first_unit, n_units, arr))

    def enqueue_block_to(self, stream_idx: int, slot: int, first_unit: int, n_units: int, d_dst, host_outs=None) -> None:
        """enqueue_block with the powers written to the device address d_dst ([unit][o][f][b]) instead of the queue's buffer."""
        arr = None
        if host_outs is not None:
            arr = (C.c_void_p * n_units)(*[_ptr(p).value for p in host_outs])
        check(self._lib.bf_enqueue_block_to(self._h, stream_idx, slot, first_unit, n_units, _ptr(d_dst), arr))

    def enqueue_block_dedisperse(self, stream_idx: int, first_unit: int, n_units: int, host_rows=None) -> None:
        check(self._lib.bf_enqueue_block_dedisperse(self._h, stream_idx, first_unit, n_units, _ptr(host_rows)))

    def enqueue_dedisperse(self, stream_idx: int, host_out_row=None) -> None:
        check(self._lib.bf_enqueue_dedisperse(self._h, stream_idx, _ptr(host_out_row)))

    def queue_stream(self, stream_idx: int) -> int:
        """bf_queue_stream: the hipStream_t of compute queue stream_idx (launches what is still only queued first)."""
        p = C.c_void_p()
        check(self._lib.bf_queue_stream(self._h, stream_idx, C.byref(p)))
        return p.value or 0

    def block_output_device(self, stream_idx: int) -> int:
        """bf_block_output_device: device pointer of the queue's block buffer [n_gemms_per_block][output][freq][beam]."""
        p = C.c_void_p()
        check(self._lib.bf_block_output_device(self._h, stream_idx, C.byref(p)))
        return p.value or 0

    def enqueue_d2h(self, stream_idx: int, d_src, host_dst, n_floats: int) -> None:
        check(self._lib.bf_enqueue_d2h(self._h, stream_idx, _ptr(d_src), _ptr(host_dst), n_floats))

    def record_analysis_event(self, event) -> None:
        check(self._lib.bf_record_analysis_event(self._h, _ptr(event)))

    def sync(self, stream_idx: int = -1) -> None:
        check(self._lib.bf_stream_sync(self._h, stream_idx))

    def timer_start(self) -> None:
        check(self._lib.bf_timer_start(self._h))

    def timer_stop(self) -> float:
        ms = C.c_float()
        check(self._lib.bf_timer_stop(self._h, C.byref(ms)))
        return ms.value

    def mfma_peak(self, d_operands, operand_bytes: int, d_scratch, scratch_bytes: int, iters: int, stream) -> float:
        """One launch of back-to-back v_mfma_i32_16x16x64_i8 on the caller's operand bytes; returns the int8 ops it executes."""
        ops = C.c_double()
        check(self._lib.bf_mfma_peak_device(self._h, _ptr(d_operands), operand_bytes, _ptr(d_scratch), scratch_bytes, iters,
                                            C.byref(ops), C.c_void_p(stream)))
        return ops.value

    def gather_relayout(self, d_stage, d_full, rows_held: int, world: int, row_floats: int, skip_rank: int, stream: int = 0) -> None:
        """bf_gather_relayout_device (dsabf_bench.h): the staged transport's device pass by itself."""
        check(self._lib.bf_gather_relayout_device(self._h, _ptr(d_stage), _ptr(d_full), rows_held, world, row_floats, skip_rank,
                                                  C.c_void_p(stream)))

    def set_switch(self, name: str, value: int) -> None:
        """bf_set_switch: a measurement / test switch of this handle ("tsplit", "lds_pad", "dm_wide", "paired")."""
        check(self._lib.bf_set_switch(self._h, name.encode(), int(value)))

    def counter(self, name: str) -> int:
        """bf_get_counter: "fused_launches", "queued_units"."""
        v = C.c_uint64()
        check(self._lib.bf_get_counter(self._h, name.encode(), C.byref(v)))
        return int(v.value)

    def kernel_info(self, n_units: int = 1) -> dict:
        g, b, l, v = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        check(self._lib.bf_kernel_info(self._h, n_units, C.byref(g), C.byref(b), C.byref(l), C.byref(v)))
        name = C.create_string_buffer(160)
        check(self._lib.bf_kernel_name(self._h, name, 160))
        return {"kernel": name.value.decode(), "grid": g.value, "block": b.value, "lds_bytes": l.value,
                "vgprs": v.value}

    def variant_key(self, write_c: bool = False) -> str:
        """The instantiation this handle launches (after set_weights decided general / conjugate-pair)."""
        buf = C.create_string_buffer(120)
        check(self._lib.bf_handle_variant_key(self._h, int(write_c), buf, 120))
        return buf.value.decode()

    def close(self) -> None:
        if self._h:
            self._lib.bf_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class DmStream:
    """bf_dm_stream: DM-trial dedispersion of the detected stream, block by block, with the delay window carried over on the
    device (include/dsabf.h).  delays: int32 host array [n_dm][n_freq_total]."""

    def __init__(self, bf: Beamformer, delays, n_freq_total: int, max_rows_per_push: int):
        import numpy as np

        self._lib = load()
        self._s = C.c_void_p()
        d = np.ascontiguousarray(delays, np.int32)
        assert d.ndim == 2 and d.shape[1] == n_freq_total
        self.n_dm, self.n_beams, self._bf = d.shape[0], bf.cfg.n_beams, bf
        check(self._lib.bf_dm_stream_create(bf._h, _ptr(d), d.shape[0], n_freq_total, max_rows_per_push, C.byref(self._s)))

    @property
    def max_delay(self) -> int:
        return self._lib.bf_dm_stream_max_delay(self._s)

    def push(self, d_rows, n_rows: int, host_out=None, stream: int = 0):
        """Returns (first_t, n_t_out) of the chunk this push emits ([n_dm][n_t_out][beam] into host_out, asynchronously)."""
        first, n = C.c_uint64(), C.c_int()
        check(self._lib.bf_dm_stream_push(self._s, _ptr(d_rows), n_rows, _ptr(host_out), C.byref(first), C.byref(n), C.c_void_p(stream)))
        return int(first.value), int(n.value)

    def reserve(self, n_rows: int, stream: int = 0) -> int:
        """Device address of the next n_rows rows' place inside the stage's buffer (bf_dm_stream_reserve): write them there on
        ``stream``, then push(that address, n_rows) -- no copy."""
        p = C.c_void_p()
        check(self._lib.bf_dm_stream_reserve(self._s, n_rows, C.byref(p), C.c_void_p(stream)))
        return p.value or 0

    def output_device(self) -> int:
        p = C.c_void_p()
        check(self._lib.bf_dm_stream_output_device(self._s, C.byref(p)))
        return p.value or 0

    def close(self) -> None:
        if self._s:
            self._lib.bf_dm_stream_destroy(self._s)
            self._s = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# events / pinned memory as free functions (they are not tied to a handle in the C-ABI)
GATHER_FREQ_MAJOR, GATHER_RANK_MAJOR = 0, 1
GATHER_ROOT_ALL, GATHER_ROOT_DISTRIBUTED = -1, -2


def comm_unique_id() -> bytes:
    """Rank 0: the 128-byte RCCL unique id to hand to the other ranks (bf_comm_unique_id)."""
    buf = C.create_string_buffer(128)
    check(load().bf_comm_unique_id(buf))
    return buf.raw


def comm_library_info() -> dict:
    """Which librccl bf_comm_create would bind, and its version -- before any communicator exists (bf_comm_library_info)."""
    v, path = C.c_int(), C.create_string_buffer(512)
    check(load().bf_comm_library_info(C.byref(v), path, 512))
    return {"version": v.value, "lib": path.value.decode(errors="replace")}


class Comm:
    """bf_comm: this rank's place in the frequency partition + the RCCL communicator behind bf_gather_detected."""

    def __init__(self, rank: int, world: int, unique_id: bytes | None = None, device: int = 0):
        self._lib = load()
        self._c = C.c_void_p()
        idbuf = C.create_string_buffer(unique_id, 128) if unique_id is not None else None
        check(self._lib.bf_comm_create(rank, world, idbuf, device, C.byref(self._c)))
        self.rank, self.world = rank, world

    def info(self) -> dict:
        """{"ranks": what the library reports (ncclCommCount), "version": ncclGetVersion, "lib": file it was loaded from}."""
        n, v, path = C.c_int(), C.c_int(), C.create_string_buffer(512)
        check(self._lib.bf_comm_info(self._c, C.byref(n), C.byref(v), path, 512))
        return {"ranks": n.value, "version": v.value, "lib": path.value.decode(errors="replace")}

    def rows_held(self, n_rows: int, root: int) -> int:
        return self._lib.bf_gather_rows_held(n_rows, self.world, self.rank, root)

    def gather(self, d_local, n_rows: int, row_floats: int, root: int, layout: int, d_full, stream: int = 0) -> None:
        check(self._lib.bf_gather_detected(self._c, _ptr(d_local), n_rows, row_floats, root, layout, _ptr(d_full),
                                           C.c_void_p(stream)))

    def gather_staged(self, d_local, n_rows: int, row_floats: int, root: int, d_full, d_stage, stream: int = 0) -> None:
        """bf_gather_detected_staged: freq-major result, rank-major on the wire + one device re-layout pass."""
        check(self._lib.bf_gather_detected_staged(self._c, _ptr(d_local), n_rows, row_floats, root, _ptr(d_full), _ptr(d_stage),
                                                  C.c_void_p(stream)))

    def close(self) -> None:
        if self._c:
            self._lib.bf_comm_destroy(self._c)
            self._c = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def event_create(bf=None) -> C.c_void_p:
    """An event on the caller's current device, or -- given a Beamformer -- on that handle's device (bf_event_create_on)."""
    ev = C.c_void_p()
    if bf is not None:
        check(load().bf_event_create_on(bf._h, C.byref(ev)))
    else:
        check(load().bf_event_create(C.byref(ev)))
    return ev


def event_query(ev) -> int:
    return check(load().bf_event_query(ev))


def event_destroy(ev) -> None:
    check(load().bf_event_destroy(ev))


def alloc_pinned(nbytes: int) -> int:
    p = C.c_void_p()
    check(load().bf_alloc_pinned(C.byref(p), nbytes))
    return p.value


def free_pinned(ptr: int) -> None:
    check(load().bf_free_pinned(C.c_void_p(ptr)))
