"""ctypes/numpy binding of oracle/liborc.so (the plain-C restatement in dsabf_oracle.c).

TEST INFRASTRUCTURE ONLY -- see oracle/dsabf_oracle.h for scope, citations and pinning status.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liborc.so")


class OrcGeom(C.Structure):
    _fields_ = [
        ("n_beams", C.c_int),
        ("n_ant", C.c_int),
        ("n_freq", C.c_int),
        ("n_pol", C.c_int),
        ("n_avg", C.c_int),
        ("n_out_per_gemm", C.c_int),
    ]


@dataclass(frozen=True)
class Geom:
    """Runtime geometry; defaults are the reference's DEBUG build (src/beamformer.hh:47-60,111)."""

    n_beams: int = 256
    n_ant: int = 64
    n_freq: int = 256
    n_pol: int = 2
    n_avg: int = 1
    n_out_per_gemm: int = 8

    @property
    def n_ipo(self) -> int:
        return self.n_pol * self.n_avg

    @property
    def n_time(self) -> int:
        return self.n_out_per_gemm * self.n_ipo

    @property
    def bytes_per_gemm(self) -> int:
        return self.n_ant * self.n_freq * self.n_time

    @property
    def out_per_gemm(self) -> int:
        return self.n_out_per_gemm * self.n_freq * self.n_beams

    def c(self) -> OrcGeom:
        return OrcGeom(self.n_beams, self.n_ant, self.n_freq, self.n_pol, self.n_avg, self.n_out_per_gemm)


DEBUG_GEOM = Geom()
PROD_GEOM = Geom(n_avg=16)


def build(force: bool = False) -> str:
    """Compile liborc.so with oracle/Makefile (gcc).  Building the checker is not using it."""
    src = os.path.join(_HERE, "dsabf_oracle.c")
    hdr = os.path.join(_HERE, "dsabf_oracle.h")
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(p) > os.path.getmtime(_LIB_PATH) for p in (src, hdr)
    )
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        P = C.c_void_p
        G = C.POINTER(OrcGeom)
        L.orc_set_threads.argtypes = [C.c_int]
        L.orc_get_threads.restype = C.c_int
        L.orc_freq_weights.argtypes = [C.c_int, C.c_int]
        L.orc_freq_weights.restype = C.c_float
        L.orc_freq_generator.argtypes = [C.c_int, C.c_int]
        L.orc_freq_generator.restype = C.c_float
        L.orc_make_weights.argtypes = [G, P, P, C.c_int, P]
        L.orc_default_positions.argtypes = [C.c_int, P]
        L.orc_default_directions.argtypes = [C.c_int, P]
        L.orc_generate_test_data.argtypes = [G, P, P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, P]
        L.orc_expand.argtypes = [P, C.c_size_t, P]
        L.orc_gemm.argtypes = [G, P, P, P]
        L.orc_detect.argtypes = [G, P, P]
        L.orc_beamform.argtypes = [G, P, P, C.c_int, P]
        L.orc_beamform_exact.argtypes = [G, P, P, C.c_int, P]
        L.orc_set_detect_contract.argtypes = [C.c_int]
        L.orc_get_detect_contract.restype = C.c_int
        L.orc_dedisperse.argtypes = [G, P, P]
        L.orc_read_positions.argtypes = [C.c_char_p, C.c_int, P]
        L.orc_read_positions.restype = C.c_int
        L.orc_read_directions.argtypes = [C.c_char_p, C.c_int, P]
        L.orc_read_directions.restype = C.c_int
        L.orc_count_entries.argtypes = [C.c_char_p]
        L.orc_count_entries.restype = C.c_int
        L.orc_write_python_file.argtypes = [P, C.c_int, C.c_int, C.c_char_p]
        L.orc_write_python_file.restype = C.c_int
        L.orc_fnv1a64.argtypes = [P, C.c_size_t]
        L.orc_fnv1a64.restype = C.c_uint64
        _lib = L
    return _lib


def _p(a: np.ndarray) -> C.c_void_p:
    assert a.flags["C_CONTIGUOUS"]
    return C.c_void_p(a.ctypes.data)


def set_threads(n: int) -> None:
    lib().orc_set_threads(int(n))


def get_threads() -> int:
    return int(lib().orc_get_threads())


def freq_weights(gpu: int, chan: int) -> float:
    return float(lib().orc_freq_weights(gpu, chan))


def freq_generator(gpu: int, chan: int) -> float:
    return float(lib().orc_freq_generator(gpu, chan))


def default_positions(n_ant: int) -> np.ndarray:
    pos = np.zeros((n_ant, 3), np.float32)
    lib().orc_default_positions(n_ant, _p(pos))
    return pos


def default_directions(n_beams: int) -> np.ndarray:
    d = np.zeros((n_beams, 2), np.float32)
    lib().orc_default_directions(n_beams, _p(d))
    return d


def read_positions(path: str, expected: int) -> np.ndarray:
    pos = np.zeros((expected, 3), np.float32)
    if lib().orc_read_positions(path.encode(), expected, _p(pos)) < 0:
        raise FileNotFoundError(path)
    return pos


def read_directions(path: str, expected: int | None = None) -> np.ndarray:
    if expected is None:
        expected = lib().orc_count_entries(path.encode())
        if expected < 0:
            raise FileNotFoundError(path)
    d = np.zeros((expected, 2), np.float32)
    if lib().orc_read_directions(path.encode(), expected, _p(d)) < 0:
        raise FileNotFoundError(path)
    return d


def make_weights(g: Geom, pos: np.ndarray, dirs: np.ndarray, gpu: int = 0) -> np.ndarray:
    """a5 -> int8 [f][a][b][2]."""
    pos = np.ascontiguousarray(pos, np.float32)
    dirs = np.ascontiguousarray(dirs, np.float32)
    assert pos.shape == (g.n_ant, 3) and dirs.shape == (g.n_beams, 2)
    w = np.empty((g.n_freq, g.n_ant, g.n_beams, 2), np.int8)
    cg = g.c()
    lib().orc_make_weights(C.byref(cg), _p(pos), _p(dirs), gpu, _p(w))
    return w


def generate_test_data(g: Geom, pos: np.ndarray, src: np.ndarray, gpu: int = 0, batch_counter: int = 0,
                       n_units: int = 1024, literal: bool = False, out: np.ndarray | None = None) -> np.ndarray:
    """a6 -> uint8 [unit][f][t][a]."""
    pos = np.ascontiguousarray(pos, np.float32)
    src = np.ascontiguousarray(src, np.float32).reshape(-1, 2)
    if out is None:
        out = np.empty((n_units, g.n_freq, g.n_time, g.n_ant), np.uint8)
    cg = g.c()
    lib().orc_generate_test_data(C.byref(cg), _p(pos), _p(src), src.shape[0], gpu, batch_counter, n_units,
                                 1 if literal else 0, _p(out))
    return out


def expand(packed: np.ndarray) -> np.ndarray:
    """a1: uint8[...] -> int8[..., 2] (re, im)."""
    packed = np.ascontiguousarray(packed, np.uint8)
    out = np.empty(packed.shape + (2,), np.int8)
    lib().orc_expand(_p(packed), packed.size, _p(out))
    return out


def gemm(g: Geom, w: np.ndarray, v: np.ndarray) -> np.ndarray:
    """a2: w int8 [f][a][b][2], v int8 [f][t][a][2] -> float32 [f][t][b][2]."""
    assert w.shape == (g.n_freq, g.n_ant, g.n_beams, 2) and v.shape == (g.n_freq, g.n_time, g.n_ant, 2)
    c = np.empty((g.n_freq, g.n_time, g.n_beams, 2), np.float32)
    cg = g.c()
    lib().orc_gemm(C.byref(cg), _p(np.ascontiguousarray(w)), _p(np.ascontiguousarray(v)), _p(c))
    return c


def detect(g: Geom, c: np.ndarray) -> np.ndarray:
    """a3: float32 [f][t][b][2] -> float32 [o][f][b]."""
    out = np.empty((g.n_out_per_gemm, g.n_freq, g.n_beams), np.float32)
    cg = g.c()
    lib().orc_detect(C.byref(cg), _p(np.ascontiguousarray(c, np.float32)), _p(out))
    return out


def beamform(g: Geom, w: np.ndarray, packed: np.ndarray) -> np.ndarray:
    """a1+a2+a3: packed uint8 [unit][f][t][a] -> float32 [unit][o][f][b]."""
    packed = np.ascontiguousarray(packed, np.uint8).reshape(-1, g.n_freq, g.n_time, g.n_ant)
    n_units = packed.shape[0]
    out = np.empty((n_units, g.n_out_per_gemm, g.n_freq, g.n_beams), np.float32)
    cg = g.c()
    lib().orc_beamform(C.byref(cg), _p(np.ascontiguousarray(w, np.int8)), _p(packed), n_units, _p(out))
    return out


CONTRACT_NONE, CONTRACT_NVCC, CONTRACT_NVCC_ALT = 0, 1, 2


def set_detect_contract(mode: int) -> None:
    """How detect's `x*x + y*y` is evaluated: CONTRACT_NONE (g++, default), CONTRACT_NVCC fma(x,x,y*y), CONTRACT_NVCC_ALT."""
    lib().orc_set_detect_contract(int(mode))


def get_detect_contract() -> int:
    return int(lib().orc_get_detect_contract())


class detect_contract:
    """with orc.detect_contract(orc.CONTRACT_NVCC): ...  (restores the previous mode)"""

    def __init__(self, mode: int):
        self.mode = mode

    def __enter__(self):
        self.prev = get_detect_contract()
        set_detect_contract(self.mode)
        return self

    def __exit__(self, *exc):
        set_detect_contract(self.prev)
        return False


def beamform_exact(g: Geom, w: np.ndarray, packed: np.ndarray) -> np.ndarray:
    """a1+a2+a3 without intermediate rounding: float64 [unit][o][f][b] = alpha^2 * exact integer power sum."""
    packed = np.ascontiguousarray(packed, np.uint8).reshape(-1, g.n_freq, g.n_time, g.n_ant)
    n_units = packed.shape[0]
    out = np.empty((n_units, g.n_out_per_gemm, g.n_freq, g.n_beams), np.float64)
    cg = g.c()
    lib().orc_beamform_exact(C.byref(cg), _p(np.ascontiguousarray(w, np.int8)), _p(packed), n_units, _p(out))
    return out


def beamform_fast(g: Geom, w: np.ndarray, packed: np.ndarray) -> np.ndarray:
    """The PRODUCT's optional BF_DETECT_FAST reading (include/dsabf.h; not a reading of the reference) restated in numpy, so that
    its kernels can be held to the bit too: d = 16 n exact; acc = fma(d, d, acc) for re then im, in time order, from +0; one
    (alpha/16)^2 scale per output.  The fma is emulated exactly: d^2 < 2^43 and acc + d^2 < 2^53 are exact in float64, so
    float32(float64 sum) is the single rounding.  (The deep classes and fusedg_kernel count in n instead of 16 n: the same bits,
    every intermediate is the same value times a power of two.)  float32 [unit][o][f][b].  Small cases only."""
    packed = np.ascontiguousarray(packed, np.uint8).reshape(-1, g.n_freq, g.n_time, g.n_ant)
    # (float64 matrix products: every partial sum is an integer below 2^53, so BLAS's summation order cannot matter)
    v = expand(packed).astype(np.float64)                     # [unit][f][t][a][2]
    W = np.ascontiguousarray(w, np.int8).astype(np.float64)   # [f][a][b][2]
    Wr, Wi = np.ascontiguousarray(W[..., 0]), np.ascontiguousarray(W[..., 1])
    a16 = np.float32(np.float32(1.0 / 127) * np.float32(0.0625))
    scale = np.float32(a16 * a16)
    outs = []
    for u in range(packed.shape[0]):
        vr, vi = np.ascontiguousarray(v[u, ..., 0]), np.ascontiguousarray(v[u, ..., 1])   # [f][t][a]
        re = np.matmul(vr, Wr) - np.matmul(vi, Wi)                                          # [f][t][b]
        im = np.matmul(vi, Wr) + np.matmul(vr, Wi)
        dr = (16.0 * re).reshape(g.n_freq, g.n_out_per_gemm, g.n_ipo, g.n_beams)
        di = (16.0 * im).reshape(g.n_freq, g.n_out_per_gemm, g.n_ipo, g.n_beams)
        acc = np.zeros((g.n_freq, g.n_out_per_gemm, g.n_beams), np.float32)
        for i in range(g.n_ipo):
            acc = (acc.astype(np.float64) + dr[:, :, i] * dr[:, :, i]).astype(np.float32)
            acc = (acc.astype(np.float64) + di[:, :, i] * di[:, :, i]).astype(np.float32)
        outs.append((acc * scale).transpose(1, 0, 2))
    return np.stack(outs)


def dedisperse(g: Geom, out_unit: np.ndarray) -> np.ndarray:
    """a8: float32 [o][f][b] (one unit) -> float32 [b] (sum over f of output 0)."""
    ded = np.empty((g.n_beams,), np.float32)
    cg = g.c()
    lib().orc_dedisperse(C.byref(cg), _p(np.ascontiguousarray(out_unit, np.float32)), _p(ded))
    return ded


def dm_trials(dm0: float = 0.0, dm_max: float = 2000.0, nchan: int = 2048, epsilon: float = 1.25,
              nu_ghz: float = (1.28 + 1.53) / 2, chan_bw_mhz: float = (1.53 - 1.28) / 2048 * 1000, ti_us: float = 40.0,
              tscat_us: float = 0.0, tsamp_us: float = 131.0) -> np.ndarray:
    """8f-4: the DM trial ladder of sandbox/Dispersion Theory.ipynb cells 1-2 (defaults = the notebook's values)."""
    out = np.zeros(65536, np.float64)
    f = lib().orc_dm_trials
    f.restype = C.c_int
    f.argtypes = [C.c_double, C.c_double, C.c_int] + [C.c_double] * 6 + [C.c_void_p, C.c_int]
    n = f(dm0, dm_max, nchan, epsilon, nu_ghz, chan_bw_mhz, ti_us, tscat_us, tsamp_us, _p(out), out.size)
    return out[:n].copy()


def dm_delays(dms: np.ndarray, freq_ghz: np.ndarray, f_ref_ghz: float, tsamp_ms: float) -> np.ndarray:
    """8f-4: int32 [n_dm][n_freq] sample delays (notebook cell 5)."""
    dms = np.ascontiguousarray(dms, np.float64)
    fr = np.ascontiguousarray(freq_ghz, np.float32)
    out = np.zeros((dms.size, fr.size), np.int32)
    f = lib().orc_dm_delays
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_void_p]
    f(_p(dms), dms.size, _p(fr), fr.size, f_ref_ghz, tsamp_ms, _p(out))
    return out


def dedisperse_dm(series: np.ndarray, delays: np.ndarray, n_t_out: int) -> np.ndarray:
    """8f-4: series float32 [t][f][b], delays int32 [dm][f] -> float32 [dm][n_t_out][b] (ascending-f fp32 sums)."""
    series = np.ascontiguousarray(series, np.float32)
    delays = np.ascontiguousarray(delays, np.int32)
    n_t, n_f, n_b = series.shape
    assert delays.shape[1] == n_f
    out = np.empty((delays.shape[0], n_t_out, n_b), np.float32)
    f = lib().orc_dedisperse_dm
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    f(_p(series), n_t, n_f, n_b, _p(delays), delays.shape[0], n_t_out, _p(out))
    return out


def write_python_file(data: np.ndarray, path: str) -> None:
    data = np.ascontiguousarray(data, np.float32)
    if lib().orc_write_python_file(_p(data), data.shape[0], data.shape[1], path.encode()) != 0:
        raise OSError(path)


def fnv1a64(a) -> int:
    a = np.ascontiguousarray(a)
    return int(lib().orc_fnv1a64(_p(a), a.nbytes))
