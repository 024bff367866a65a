"""ctypes access to the C++ host mirror (include/dsabf_host.h): weights, config readers, the data.py writer,
test_data_generator and observation_loop_state.  Thin wrappers only; the logic is C++ (csrc/bf_geometry.cpp, bf_generator.cpp, bf_scheduler.cpp, bf_sinks.cpp; C wrappers bf_host_c.cpp)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import BfhEventOps, BfConfig, check, load


def _p(a: np.ndarray) -> C.c_void_p:
    assert a.flags["C_CONTIGUOUS"]
    return C.c_void_p(a.ctypes.data)


def default_positions(n_ant: int) -> np.ndarray:
    pos = np.zeros((n_ant, 3), np.float32)
    check(load().bfh_default_positions(n_ant, _p(pos)))
    return pos


def default_directions(n_beams: int) -> np.ndarray:
    d = np.zeros((n_beams, 2), np.float32)
    check(load().bfh_default_directions(n_beams, _p(d)))
    return d


def read_positions(path: str, n_ant: int) -> np.ndarray:
    pos = np.zeros((n_ant, 3), np.float32)
    check(load().bfh_read_positions(path.encode(), n_ant, _p(pos)))
    return pos


def read_directions(path: str, expected: int | None = None) -> np.ndarray:
    if expected is None:
        expected = check(load().bfh_count_entries(path.encode()))
    d = np.zeros((expected, 2), np.float32)
    check(load().bfh_read_directions(path.encode(), expected, _p(d)))
    return d


def write_python_file(data: np.ndarray, path: str) -> None:
    data = np.ascontiguousarray(data, np.float32)
    check(load().bfh_write_python_file(_p(data), data.shape[0], data.shape[1], path.encode()))


def channel_frequency(gpu: int, chan: int, generator_variant: bool = False) -> float:
    return float(load().bfh_channel_frequency(1 if generator_variant else 0, gpu, chan))


def make_weights(pos: np.ndarray, dirs: np.ndarray, n_freq: int, chan0: int = 0, gpu: int = 0) -> np.ndarray:
    """src/beamformer.cu:230-241 -> int8 [n_freq][n_ant][n_beams][2] for channels chan0.. of sub-band gpu."""
    pos = np.ascontiguousarray(pos, np.float32)
    dirs = np.ascontiguousarray(dirs, np.float32)
    w = np.empty((n_freq, pos.shape[0], dirs.shape[0], 2), np.int8)
    check(load().bfh_make_weights(dirs.shape[0], pos.shape[0], n_freq, chan0, gpu, _p(pos), _p(dirs), _p(w)))
    return w


def make_weights_default(n_beams: int = 256, n_ant: int = 64, n_freq_total: int = 256, gpu: int = 0) -> np.ndarray:
    return make_weights(default_positions(n_ant), default_directions(n_beams), n_freq_total, 0, gpu)


class TestDataGenerator:
    """test_data_generator (src/test_data_generator.hh:11-108)."""

    __test__ = False

    def __init__(self, cfg: BfConfig, n_sources_per_batch: int = 1024, pin: bool = True):
        self._lib = load()
        self._g = C.c_void_p()
        self.cfg = cfg
        check(self._lib.bfh_gen_create(C.byref(cfg), n_sources_per_batch, 1 if pin else 0, C.byref(self._g)))

    def read_in_source_directions(self, path: str) -> None:
        check(self._lib.bfh_gen_read_sources(self._g, path.encode()))

    def set_source_directions(self, src: np.ndarray) -> None:
        src = np.ascontiguousarray(src, np.float32).reshape(-1, 2)
        check(self._lib.bfh_gen_set_sources(self._g, _p(src), src.shape[0]))

    def generate_test_data(self, pos: np.ndarray, gpu: int = 0) -> None:
        check(self._lib.bfh_gen_generate(self._g, _p(np.ascontiguousarray(pos, np.float32)), gpu))

    def get_n_pt_sources(self) -> int:
        return check(self._lib.bfh_gen_n_pt_sources(self._g))

    def data_ptr(self) -> int:
        return self._lib.bfh_gen_data(self._g)

    def size(self) -> int:
        return self._lib.bfh_gen_size(self._g)

    def data(self) -> np.ndarray:
        """View (no copy) of the generator's batch buffer as uint8."""
        buf = (C.c_uint8 * self.size()).from_address(self.data_ptr())
        return np.frombuffer(buf, dtype=np.uint8)

    def check_need_to_generate_more_input_data(self, blocks_transferred: int) -> bool:
        return bool(check(self._lib.bfh_gen_need_more(self._g, blocks_transferred)))

    def check_data_ready_for_transfer(self, blocks_transfer_queue: int) -> bool:
        return bool(check(self._lib.bfh_gen_ready(self._g, blocks_transfer_queue)))

    def close(self) -> None:
        if self._g:
            self._lib.bfh_gen_destroy(self._g)
            self._g = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ObservationLoopState:
    """observation_loop_state (src/observation_loop.hh:1-177) on HIP events of `handle`'s queues, or -- event_ops given --
    on a caller-supplied event backend (``BfhEventOps``; the CPU tests drive the scheduler that way)."""

    def __init__(self, cfg: BfConfig, handle=None, max_transfer_sep: int = 2, max_total_sep: int = 4, debug: bool = True,
                 event_ops=None):
        self._lib = load()
        self._o = C.c_void_p()
        self._ops = event_ops   # keep the callback table (and the Python callables behind it) alive
        if event_ops is not None:
            check(self._lib.bfh_obs_create_custom(max_transfer_sep, max_total_sep, C.byref(cfg), C.byref(event_ops),
                                                  1 if debug else 0, C.byref(self._o)))
        else:
            check(self._lib.bfh_obs_create(max_transfer_sep, max_total_sep, C.byref(cfg), handle, 1 if debug else 0,
                                           C.byref(self._o)))

    def status(self) -> int:
        """BF_OK or the first (sticky) error an event operation reported."""
        return int(self._lib.bfh_obs_status(self._o))

    def counters(self) -> dict:
        v = [C.c_uint64() for _ in range(4)]
        check(self._lib.bfh_obs_counters(self._o, *[C.byref(x) for x in v]))
        return dict(zip(("A", "AQ", "T", "TQ"), (x.value for x in v)))

    def generate_transfer_event(self):
        check(self._lib.bfh_obs_generate_transfer_event(self._o))

    def generate_analysis_event(self):
        check(self._lib.bfh_obs_generate_analysis_event(self._o))

    def check_transfer_events(self):
        check(self._lib.bfh_obs_check_transfer_events(self._o))

    def check_analysis_events(self):
        check(self._lib.bfh_obs_check_analysis_events(self._o))

    def check_ready_for_transfer(self) -> bool:
        return bool(check(self._lib.bfh_obs_check_ready_for_transfer(self._o)))

    def check_ready_for_analysis(self) -> bool:
        return bool(check(self._lib.bfh_obs_check_ready_for_analysis(self._o)))

    def check_ready_for_dh2_transfer(self, time_slice: int) -> bool:
        return bool(check(self._lib.bfh_obs_check_ready_for_dh2_transfer(self._o, time_slice)))

    def check_observations_complete(self) -> bool:
        return bool(check(self._lib.bfh_obs_check_observations_complete(self._o)))

    def check_transfers_complete(self) -> bool:
        return bool(check(self._lib.bfh_obs_check_transfers_complete(self._o)))

    def set_transfers_complete(self, v: bool):
        check(self._lib.bfh_obs_set_transfers_complete(self._o, 1 if v else 0))

    def set_n_pt_sources(self, n: int):
        check(self._lib.bfh_obs_set_n_pt_sources(self._o, n))

    def get_current_analysis_gemm(self, time_slice: int) -> int:
        return self._lib.bfh_obs_get_current_analysis_gemm(self._o, time_slice)

    def get_current_transfer_gemm(self) -> int:
        return self._lib.bfh_obs_get_current_transfer_gemm(self._o)

    def get_next_gpu_analysis_block(self) -> int:
        return self._lib.bfh_obs_get_next_gpu_analysis_block(self._o)

    def get_next_gpu_transfer_block(self) -> int:
        return self._lib.bfh_obs_get_next_gpu_transfer_block(self._o)

    def describe(self) -> str:
        buf = C.create_string_buffer(512)
        check(self._lib.bfh_obs_describe(self._o, buf, 512))
        return buf.value.decode()

    def close(self):
        if self._o:
            self._lib.bfh_obs_destroy(self._o)
            self._o = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def run_debug_observation(cfg: BfConfig, gpu: int = 0, positions: str | None = None, directions: str | None = None,
                          sources: str | None = None, output: str | None = None, device: int = 0,
                          verbose: bool = False, max_sources: int = 4096, per_unit_launches: bool = False):
    """The reference's `make debug` main() end to end; returns (dedispersed [n_src][n_beams], observation_ms).
    per_unit_launches: the reference's own launch pattern instead of one launch / copy / DM-0 launch per block."""
    ded = np.zeros((max_sources, cfg.n_beams), np.float32)
    n = C.c_int()
    ms = C.c_float()
    enc = lambda s: s.encode() if s else None  # noqa: E731
    check(load().bfh_run_debug_observation2(C.byref(cfg), gpu, enc(positions), enc(directions), enc(sources), enc(output),
                                            device, 1 if verbose else 0, _p(ded), ded.size, C.byref(n), C.byref(ms),
                                            1 if per_unit_launches else 0))
    return ded[:n.value].copy(), ms.value


def run_observation_junk(cfg: BfConfig, n_blocks: int, ring_blocks: int = 4, seed: int = 0xD5A, gpu: int = 0,
                         device: int = 0, burn_in: int = 0, verbose: bool = False):
    """The reference's production observation loop fed by the in-memory dada_junkdb stand-in.
    Returns dict(ms, beam_out [n_streams][n_out][n_freq][n_beams], last_gemm [n_streams], ring uint8
    [ring_blocks][n_gemms_per_block][n_freq][n_time][n_ant])."""
    lib = load()
    n_time = cfg.n_out_per_gemm * cfg.n_pol * cfg.n_avg
    beam_out = np.zeros((cfg.n_streams, cfg.n_out_per_gemm, cfg.n_freq, cfg.n_beams), np.float32)
    last = np.zeros(cfg.n_streams, np.int64)
    ring = np.zeros((ring_blocks, cfg.n_gemms_per_block, cfg.n_freq, n_time, cfg.n_ant), np.uint8)
    ms = C.c_float()
    check(lib.bfh_run_observation_junk(C.byref(cfg), n_blocks, ring_blocks, seed, gpu, device, burn_in,
                                       1 if verbose else 0, C.byref(ms), _p(beam_out), _p(last), _p(ring)))
    return {"ms": ms.value, "beam_out": beam_out, "last_gemm": last, "ring": ring}


DETECTED_HEADER_BYTES = 4096


def read_detected_file(path: str):
    """Parse a dsabf::file_sink file: returns (header dict, float32 array [n_gemms][n_out][n_freq][n_beams])."""
    raw = open(path, "rb").read()
    text = raw[:DETECTED_HEADER_BYTES].split(b"\0", 1)[0].decode()
    hdr = dict(line.split(None, 1) for line in text.splitlines() if line.strip())
    shape = (int(hdr["N_OUTPUTS_PER_GEMM"]), int(hdr["N_FREQUENCIES"]), int(hdr["N_BEAMS"]))
    data = np.frombuffer(raw, dtype="<f4", offset=int(hdr["HDR_SIZE"]))
    per = shape[0] * shape[1] * shape[2]
    assert per == int(hdr["FLOATS_PER_GEMM"]) and data.size % per == 0
    return hdr, data.reshape((-1,) + shape)


def run_observation_junk_to_file(cfg: BfConfig, n_blocks: int, path: str, ring_blocks: int = 4, seed: int = 0xD5A,
                                 gpu: int = 0, device: int = 0, burn_in: int = 0, verbose: bool = False):
    """Production observation loop with every gemm-unit's detected powers written to `path` (dsabf::file_sink).
    Returns dict(ms, gemms_written, ring)."""
    lib = load()
    n_time = cfg.n_out_per_gemm * cfg.n_pol * cfg.n_avg
    ring = np.zeros((ring_blocks, cfg.n_gemms_per_block, cfg.n_freq, n_time, cfg.n_ant), np.uint8)
    ms = C.c_float()
    n = C.c_uint64()
    check(lib.bfh_run_observation_junk_to_file(C.byref(cfg), n_blocks, ring_blocks, seed, gpu, device, burn_in,
                                               1 if verbose else 0, path.encode(), C.byref(ms), C.byref(n), _p(ring)))
    return {"ms": ms.value, "gemms_written": n.value, "ring": ring}


def read_dm_file(path: str):
    """Parse a dsabf::dm_file_sink file: returns (header dict, float32 array [n_dm][T][n_beams], list of (first_t, n_t) per
    chunk).  The chunks ([dm][t][beam] each) are joined along t; they follow each other without gaps."""
    raw = open(path, "rb").read()
    text = raw[:DETECTED_HEADER_BYTES].split(b"\0", 1)[0].decode()
    hdr = dict(line.split(None, 1) for line in text.splitlines() if line.strip())
    at, rec = int(hdr["HDR_SIZE"]), int(hdr["RECORD_HEADER_BYTES"])
    parts, chunks, next_t = [], [], 0
    while at < len(raw):
        first_t = int(np.frombuffer(raw, "<u8", 1, at)[0])
        n_t, n_dm, n_beams = (int(v) for v in np.frombuffer(raw, "<u4", 3, at + 8))
        assert first_t == next_t and n_dm == int(hdr["N_DM"]) and n_beams == int(hdr["N_BEAMS"])
        at += rec
        parts.append(np.frombuffer(raw, "<f4", n_dm * n_t * n_beams, at).reshape(n_dm, n_t, n_beams))
        at += 4 * n_dm * n_t * n_beams
        chunks.append((first_t, n_t))
        next_t += n_t
    data = np.concatenate(parts, axis=1) if parts else np.zeros((int(hdr["N_DM"]), 0, int(hdr["N_BEAMS"])), np.float32)
    return hdr, data, chunks


def run_observation_junk_dm(cfg: BfConfig, n_blocks: int, delays, dm_path: str | None, detected_path: str | None = None,
                            ring_blocks: int = 4, seed: int = 0xD5A, gpu: int = 0, device: int = 0, burn_in: int = 0,
                            verbose: bool = False):
    """Production observation loop with the DM stage on: every analysed block's beam-blocks go through a bf_dm_stream
    (delays: int32 [n_dm][cfg.n_freq]), the chunks to dm_path (read_dm_file), optionally the detected stream itself to
    detected_path.  Returns dict(ms, dm_times, ring)."""
    lib = load()
    d = np.ascontiguousarray(delays, np.int32)
    assert d.ndim == 2 and d.shape[1] == cfg.n_freq
    n_time = cfg.n_out_per_gemm * cfg.n_pol * cfg.n_avg
    ring = np.zeros((ring_blocks, cfg.n_gemms_per_block, cfg.n_freq, n_time, cfg.n_ant), np.uint8)
    ms, n = C.c_float(), C.c_uint64()
    check(lib.bfh_run_observation_junk_dm(C.byref(cfg), n_blocks, ring_blocks, seed, gpu, device, burn_in, 1 if verbose else 0,
                                          _p(d), d.shape[0], dm_path.encode() if dm_path else None,
                                          detected_path.encode() if detected_path else None, C.byref(ms), C.byref(n), _p(ring)))
    return {"ms": ms.value, "dm_times": n.value, "ring": ring}


def dm_trial_share(n_dm: int, world: int, rank: int):
    """dsabf::dm_trial_share: (first, count) of the DM trials rank `rank` of `world` dedisperses when the ladder is split."""
    f, c = C.c_int(), C.c_int()
    check(load().bfh_dm_trial_share(n_dm, world, rank, C.byref(f), C.byref(c)))
    return f.value, c.value


def run_observation_junk_sharded(cfg: BfConfig, n_blocks: int, rank: int, world: int, unique_id: bytes | None, gather_root: int = 0,
                                 staged: bool = False, delays=None, split_trials: bool = False, detected_path: str | None = None,
                                 dm_path: str | None = None, ring_blocks: int = 4, seed: int = 0xD5A, gpu: int = 0, device: int = 0):
    """One frequency shard of a sharded observation (bfh_run_observation_junk_sharded): cfg is the shard's geometry, unique_id the
    128 bytes rank 0 drew (api.comm_unique_id).  Returns dict(ms, dm_times, ring)."""
    lib = load()
    d = np.ascontiguousarray(delays, np.int32) if delays is not None else None
    assert d is None or (d.ndim == 2 and d.shape[1] == cfg.n_freq * world)
    n_time = cfg.n_out_per_gemm * cfg.n_pol * cfg.n_avg
    ring = np.zeros((ring_blocks, cfg.n_gemms_per_block, cfg.n_freq, n_time, cfg.n_ant), np.uint8)
    idbuf = C.create_string_buffer(unique_id, 128) if unique_id is not None else None
    ms, n = C.c_float(), C.c_uint64()
    check(lib.bfh_run_observation_junk_sharded(C.byref(cfg), n_blocks, ring_blocks, seed, gpu, device, rank, world, idbuf, gather_root,
                                               1 if staged else 0, _p(d) if d is not None else None, d.shape[0] if d is not None else 0,
                                               1 if split_trials else 0, detected_path.encode() if detected_path else None,
                                               dm_path.encode() if dm_path else None, C.byref(ms), C.byref(n), _p(ring)))
    return {"ms": ms.value, "dm_times": n.value, "ring": ring}


def run_observation_junk_to_ring(cfg: BfConfig, n_blocks: int, out_ring: str, out_ring_blocks: int = 8,
                                 ring_blocks: int = 4, seed: int = 0xD5A, gpu: int = 0, device: int = 0):
    """Production observation loop with the detected stream handed to a consumer through the shared-memory ring
    `out_ring` (dsabf::ring_sink).  Blocks until the consumer has drained the ring.  Returns dict(ms, gemms_written, ring)."""
    lib = load()
    n_time = cfg.n_out_per_gemm * cfg.n_pol * cfg.n_avg
    ring = np.zeros((ring_blocks, cfg.n_gemms_per_block, cfg.n_freq, n_time, cfg.n_ant), np.uint8)
    ms, n = C.c_float(), C.c_uint64()
    check(lib.bfh_run_observation_junk_to_ring(C.byref(cfg), n_blocks, ring_blocks, seed, gpu, device, out_ring.encode(),
                                               out_ring_blocks, C.byref(ms), C.byref(n), _p(ring)))
    return {"ms": ms.value, "gemms_written": n.value, "ring": ring}


class FileSink:
    """The sink's pinned ring + file writer on its own (dsabf::file_sink)."""

    def __init__(self, cfg: BfConfig, path: str, gpu: int = 0, slots: int = 0):
        self._lib = load()
        self._h = C.c_void_p()
        self.floats_per_gemm = cfg.n_out_per_gemm * cfg.n_freq * cfg.n_beams
        check(self._lib.bfh_file_sink_create(C.byref(cfg), path.encode(), gpu, slots, C.byref(self._h)))

    def acquire(self, gemm_index: int):
        """numpy view of the slot, or None if the ring has no free slot for this index."""
        p = C.POINTER(C.c_float)()
        rc = self._lib.bfh_sink_acquire(self._h, gemm_index, C.byref(p))
        if rc != 0:
            return None
        return np.ctypeslib.as_array(p, shape=(self.floats_per_gemm,))

    def commit(self, gemm_index: int) -> bool:
        return self._lib.bfh_sink_commit(self._h, gemm_index) == 0

    def close(self):
        if self._h:
            self._lib.bfh_sink_close(self._h)
            self._lib.bfh_sink_destroy(self._h)
            self._h = C.c_void_p()


def junk_bytes(block_bytes: int, distinct: int, seed: int, cfg: BfConfig | None = None) -> np.ndarray:
    """The bytes `junkdb` / junk_block_source serve: uint8 [distinct][block_bytes]; block i of a run is row i % distinct."""
    from .api import production_config

    if cfg is None:
        cfg = production_config()
        if block_bytes != cfg.n_gemms_per_block * cfg.n_freq * cfg.n_out_per_gemm * cfg.n_pol * cfg.n_avg * cfg.n_ant:
            cfg.n_gemms_per_block = cfg.n_freq = cfg.n_ant = cfg.n_pol = cfg.n_avg = 1
            cfg.n_out_per_gemm = block_bytes
    out = np.zeros((distinct, block_bytes), np.uint8)
    check(load().bfh_junk_fill(C.byref(cfg), distinct, seed, _p(out)))
    return out


class ShmRing:
    """dsabf::shm_ring (the PSRDADA stand-in) from Python: create or attach, blocking write / read of whole blocks."""

    def __init__(self, name: str, n_blocks: int = 0, block_size: int = 0, header: str = "", timeout_ms: int = 10000):
        self._lib = load()
        self._h = C.c_void_p()
        self.name = name
        if n_blocks:
            check(self._lib.bfh_shm_ring_create(name.encode(), n_blocks, block_size, header.encode(), C.byref(self._h)))
        else:
            check(self._lib.bfh_shm_ring_attach(name.encode(), timeout_ms, C.byref(self._h)))
        nb, bs = C.c_uint64(), C.c_uint64()
        hdr = C.create_string_buffer(4096)
        check(self._lib.bfh_shm_ring_info(self._h, C.byref(nb), C.byref(bs), hdr, 4096))
        self.n_blocks, self.block_size, self.header = nb.value, bs.value, hdr.value.decode()

    def write(self, data) -> None:
        """data: bytes-like / uint8 array of at most block_size bytes (fewer = end of data)."""
        a = np.ascontiguousarray(np.frombuffer(data, np.uint8) if not isinstance(data, np.ndarray) else data)
        check(self._lib.bfh_shm_ring_write(self._h, _p(a) if a.size else None, a.size))

    def read(self):
        """-> (uint8 array of the valid bytes, block id)."""
        out = np.empty(self.block_size, np.uint8)
        n, bid = C.c_uint64(), C.c_uint64()
        check(self._lib.bfh_shm_ring_read(self._h, _p(out), out.size, C.byref(n), C.byref(bid)))
        return out[:n.value], bid.value

    def detach(self) -> None:
        if self._h:
            self._lib.bfh_shm_ring_detach(self._h)
            self._h = C.c_void_p()

    def unlink(self) -> None:
        shm_ring_unlink(self.name)


def shm_ring_unlink(name: str) -> None:
    """`dada_db -k name -d`: remove the ring's shared-memory object (no error if it is gone)."""
    load().bfh_shm_ring_unlink(name.encode())


def run_observation_shm(cfg: BfConfig, name: str, path: str | None = None, core: int = -1, gpu: int = 0, device: int = 0,
                        verbose: bool = False, delays=None, dm_path: str | None = None):
    """Production observation loop fed from the shared-memory ring `name`; delays (int32 [n_dm][cfg.n_freq]): with the DM
    stage on, chunks to dm_path.  Returns dict(ms, gemms, pinned, dm_times)."""
    ms, n, nd, pinned = C.c_float(), C.c_uint64(), C.c_uint64(), C.c_int()
    d = np.ascontiguousarray(delays, np.int32) if delays is not None else None
    assert d is None or (d.ndim == 2 and d.shape[1] == cfg.n_freq)
    check(load().bfh_run_observation_shm_dm(C.byref(cfg), name.encode(), core, gpu, device, 1 if verbose else 0,
                                            path.encode() if path else None, _p(d) if d is not None else None,
                                            d.shape[0] if d is not None else 0, dm_path.encode() if dm_path else None,
                                            C.byref(ms), C.byref(n), C.byref(nd), C.byref(pinned)))
    return {"ms": ms.value, "gemms": n.value, "pinned": bool(pinned.value), "dm_times": nd.value}


def dm_trials(dm0: float = 0.0, dm_max: float = 2000.0, nchan: int = 2048, epsilon: float = 1.25,
              nu_ghz: float = (1.28 + 1.53) / 2, chan_bw_mhz: float = (1.53 - 1.28) / 2048 * 1000, ti_us: float = 40.0,
              tscat_us: float = 0.0, tsamp_us: float = 131.0) -> np.ndarray:
    """DM trial ladder (sandbox/Dispersion Theory.ipynb cells 1-2; defaults are the notebook's values)."""
    out = np.zeros(1 << 16, np.float64)
    n = load().bfh_dm_trials(dm0, dm_max, nchan, epsilon, nu_ghz, chan_bw_mhz, ti_us, tscat_us, tsamp_us,
                             out.ctypes.data_as(C.POINTER(C.c_double)), out.size)
    if n < 0:
        check(n)
    return out[:n].copy()


def dm_delays(dms, freq_ghz, f_ref_ghz: float, tsamp_ms: float) -> np.ndarray:
    """int32 [n_dm][n_freq] sample delays (notebook cell 5)."""
    dms = np.ascontiguousarray(dms, np.float64)
    fr = np.ascontiguousarray(freq_ghz, np.float32)
    out = np.zeros((dms.size, fr.size), np.int32)
    check(load().bfh_dm_delays(dms.ctypes.data_as(C.POINTER(C.c_double)), dms.size,
                               fr.ctypes.data_as(C.POINTER(C.c_float)), fr.size, f_ref_ghz, tsamp_ms,
                               out.ctypes.data_as(C.POINTER(C.c_int32))))
    return out
