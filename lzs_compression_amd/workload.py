"""Deterministic synthetic block streams (text / low-entropy / high-entropy).

Thin ctypes front-end over csrc/lzs_workload.c (liblzs_workload.so).  Bench and
test tooling: the reference ships no generator; the three classes are the ones
BASELINE.json's configs name.  Block ``b`` of a class depends on
``(seed, class, b)`` only.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

DEFAULT_SEED = 0x4C5A5331          # "LZS1"
CLASS_NAMES = ("text", "lowent", "random")
TEXT, LOWENT, RANDOM = 0, 1, 2

# (LZS_WORKLOAD_SO: another build of the generator -- the sanitizer run of the checkers builds one with ASan / UBSan)
_SO = os.environ.get("LZS_WORKLOAD_SO") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "liblzs_workload.so")
_SO_HIP = os.path.join(os.path.dirname(os.path.abspath(__file__)), "liblzs_workload_hip.so")
_lib = None
_lib_hip = None


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise ImportError(f"{_SO} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "or `make -C lzs_compression_amd/csrc`")
        _lib = ctypes.CDLL(_SO)
        _lib.lzs_workload_fill.restype = ctypes.c_int
        _lib.lzs_workload_fill.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.c_uint64,
                                           ctypes.c_uint64, ctypes.c_size_t, ctypes.c_size_t,
                                           ctypes.c_int]
    return _lib


def fill(cls, nblocks: int, block_len: int = 65536, first_block: int = 0,
         seed: int = DEFAULT_SEED, threads: int = 0, out: np.ndarray | None = None) -> np.ndarray:
    """uint8 array [nblocks, block_len] holding blocks first_block.. of class ``cls``."""
    if isinstance(cls, str):
        cls = CLASS_NAMES.index(cls)
    if out is None:
        out = np.empty((nblocks, block_len), dtype=np.uint8)
    assert out.dtype == np.uint8 and out.size == nblocks * block_len and out.flags.c_contiguous
    if threads <= 0:
        threads = min(os.cpu_count() or 1, 64)
    rc = _load().lzs_workload_fill(out.ctypes.data, cls, seed, first_block, nblocks, block_len, threads)
    if rc != 0:
        raise ValueError(f"lzs_workload_fill failed for class {cls}")
    return out


def _load_hip():
    global _lib_hip
    if _lib_hip is None:
        if not os.path.exists(_SO_HIP):
            raise ImportError(f"{_SO_HIP} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "or `make -C lzs_compression_amd/csrc`")
        import torch  # noqa: F401   (its HIP runtime first, as in api.lib())
        _lib_hip = ctypes.CDLL(_SO_HIP)
        _lib_hip.lzs_workload_fill_device.restype = ctypes.c_int
        _lib_hip.lzs_workload_fill_device.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.c_uint64,
                                                      ctypes.c_uint64, ctypes.c_size_t, ctypes.c_size_t,
                                                      ctypes.c_void_p]
    return _lib_hip


def fill_device(cls, nblocks: int, block_len: int = 65536, first_block: int = 0,
                seed: int = DEFAULT_SEED, out=None, device=None, stream=None):
    """The same blocks as fill(), generated in HBM by lzs_gen_blocks_kernel (csrc/lzs_workload_gen.hip):
    a CUDA uint8 tensor [nblocks, block_len].  Asynchronous on ``stream`` (default: torch's current)."""
    import torch
    if isinstance(cls, str):
        cls = CLASS_NAMES.index(cls)
    if out is None:
        out = torch.empty((nblocks, block_len), dtype=torch.uint8,
                          device=device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    assert out.is_cuda and out.dtype == torch.uint8 and out.numel() == nblocks * block_len and out.is_contiguous()
    s = torch.cuda.current_stream(out.device) if stream is None else stream
    with torch.cuda.device(out.device):
        rc = _load_hip().lzs_workload_fill_device(out.data_ptr(), cls, seed, first_block, nblocks, block_len,
                                                  ctypes.c_void_p(s.cuda_stream))
    if rc != 0:
        raise RuntimeError(f"lzs_workload_fill_device failed for class {cls}: code {rc}")
    return out
