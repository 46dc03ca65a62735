"""ctypes front-end for the CHECKERS under oracle/.

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; the product package (lzs_compression_amd) never
imports this module.

Two checkers, same four-argument signature as the reference's one-shot calls
(c/src/liblzs/lzs.h:218,229):

* ``oracle``  -- oracle/liblzs_oracle.so, our CPU restatement (lzs_oracle.c).
* ``ref``     -- oracle/_ref/liblzs_ref.so, the REAL reference compiled from
                 /root/reference by oracle/Makefile (present only if it was built
                 in the container and travelled with the snapshot).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (LZS_ORACLE_SO: another build of the restatement -- tests/test_sanitizers.py points it at oracle/_build/liblzs_oracle_asan.so)
_ORACLE_SO = os.environ.get("LZS_ORACLE_SO") or os.path.join(_HERE, "liblzs_oracle.so")
_REF_SO = os.path.join(_HERE, "_ref", "liblzs_ref.so")

_SIG = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]


def build(quiet: bool = True) -> None:
    """Compile the restatement (always) and the reference (when its sources exist)."""
    subprocess.run(["make", "-C", _HERE, "all"], check=True,
                   stdout=subprocess.DEVNULL if quiet else None)


def compressed_max(n: int) -> int:
    """LZS_COMPRESSED_MAX (c/src/liblzs/lzs.h:77)."""
    return n + (n + 7) // 8 + 3


class _Codec:
    def __init__(self, path: str, compress: str, decompress: str, extra: Sequence[str] = ()):
        self.path = path
        self.lib = ctypes.CDLL(path)
        self._c = getattr(self.lib, compress)
        self._d = getattr(self.lib, decompress)
        for f in (self._c, self._d):
            f.restype = ctypes.c_size_t
            f.argtypes = _SIG
        for name in extra:
            f = getattr(self.lib, name)
            f.restype = ctypes.c_size_t
            f.argtypes = _SIG
            setattr(self, "_" + name, f)

    @staticmethod
    def _call(fn, data: bytes, cap: int) -> bytes:
        # +1 spare input byte: the reference reads in[len] once (lzs-compression.c:437).
        src = ctypes.create_string_buffer(bytes(data) + b"\0", len(data) + 1)
        dst = ctypes.create_string_buffer(max(cap, 1))
        n = fn(ctypes.addressof(dst), cap, ctypes.addressof(src), len(data))
        assert n <= cap
        return dst.raw[:n]

    def compress(self, data: bytes, cap: Optional[int] = None) -> bytes:
        return self._call(self._c, data, compressed_max(len(data)) if cap is None else cap)

    def decompress(self, data: bytes, cap: int) -> bytes:
        return self._call(self._d, data, cap)

    # function addresses, for the threaded block runner
    @property
    def compress_addr(self) -> int:
        return ctypes.cast(self._c, ctypes.c_void_p).value

    @property
    def decompress_addr(self) -> int:
        return ctypes.cast(self._d, ctypes.c_void_p).value


class _Oracle(_Codec):
    def __init__(self):
        if not os.path.exists(_ORACLE_SO):
            build()
        super().__init__(_ORACLE_SO, "lzs_oracle_compress", "lzs_oracle_decompress",
                         extra=("lzs_oracle_compress_brute",))
        self.lib.lzs_cpu_run_blocks.restype = ctypes.c_double
        self.lib.lzs_cpu_run_blocks.argtypes = [
            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p,
            ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t,
            ctypes.c_int]
        self.lib.lzs_oracle_trace.restype = ctypes.c_size_t
        self.lib.lzs_oracle_trace.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p,
                                              ctypes.c_size_t]

    def compress_brute(self, data: bytes, cap: Optional[int] = None) -> bytes:
        return self._call(self._lzs_oracle_compress_brute, data,
                          compressed_max(len(data)) if cap is None else cap)

    def trace(self, data: bytes, max_tok: int = 1 << 20) -> np.ndarray:
        """Token list [(pos, off, total_len)], off 0 = literal (debug aid)."""
        src = ctypes.create_string_buffer(bytes(data), max(len(data), 1))
        rec = np.zeros((max_tok, 3), dtype=np.uint32)
        n = self.lib.lzs_oracle_trace(ctypes.addressof(src), len(data), rec.ctypes.data, max_tok)
        return rec[:min(n, max_tok)]


_oracle: Optional[_Oracle] = None
_ref: Optional[_Codec] = None


def oracle() -> _Oracle:
    global _oracle
    if _oracle is None:
        _oracle = _Oracle()
    return _oracle


def have_ref() -> bool:
    return os.path.exists(_REF_SO)


def ref() -> _Codec:
    """The compiled reference; raises FileNotFoundError where it did not travel."""
    global _ref
    if _ref is None:
        if not have_ref():
            raise FileNotFoundError(_REF_SO)
        _ref = _Codec(_REF_SO, "lzs_compress", "lzs_decompress", extra=("lzs_simple_compress",))
    return _ref


def run_blocks(codec: _Codec, blocks: np.ndarray, *, decompress: bool = False,
               in_len: Optional[np.ndarray] = None, out_cap: Optional[int] = None,
               threads: int = 1):
    """Run a CPU one-shot codec over independent blocks, one block per task.

    blocks: uint8 array [nblocks, stride].  Returns (out[nblocks, out_cap], out_len[nblocks],
    seconds).  Used by tests as a many-block checker and by bench.py's cpu_baseline."""
    assert blocks.dtype == np.uint8 and blocks.ndim == 2
    nb, stride = blocks.shape
    # one spare byte after the last block for the reference's 1-byte over-read
    flat = np.zeros(nb * stride + 16, dtype=np.uint8)
    flat[:nb * stride] = blocks.reshape(-1)
    if out_cap is None:
        out_cap = stride if decompress else compressed_max(stride)
    out = np.empty((nb, out_cap), dtype=np.uint8)
    out.fill(0)                      # touch the pages now, not inside the timed threads
    out_len = np.zeros(nb, dtype=np.uint32)
    if in_len is not None:
        in_len = np.ascontiguousarray(in_len, dtype=np.uint32)
    fn = codec.decompress_addr if decompress else codec.compress_addr
    secs = oracle().lib.lzs_cpu_run_blocks(
        fn, out.ctypes.data, out_cap, out_cap, out_len.ctypes.data,
        flat.ctypes.data, stride, None if in_len is None else in_len.ctypes.data,
        stride, nb, threads)
    if secs < 0:
        raise RuntimeError("lzs_cpu_run_blocks failed")
    return out, out_len, secs
