"""Python front-end of liblzs (MI355X build): a ctypes binding of the C-ABI declared in
include/lzs/lzs.h and include/lzs/lzs_batch.h.

The reference is a C library with no Python binding for its C path; this module mirrors
its one-shot interface (c/src/liblzs/lzs.h:218,229: ``lzs_compress`` / ``lzs_decompress``
with (out, out_capacity, in, in_len) -> bytes written) and adds the batch forms.  PyTorch
appears only as plumbing: device buffers and the current HIP stream.

There is no CPU fallback: if liblzs.so is missing, or there is no HIP device, calls raise.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liblzs.so")

LZS_OK, LZS_E_NO_DEVICE, LZS_E_HIP, LZS_E_ARG, LZS_E_NOMEM = 0, -1, -2, -3, -4
LZS_MAX_HISTORY_SIZE = 2047


class LzsError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"liblzs error {code}: {message}")
        self.code = code


def compressed_max(n: int) -> int:
    """LZS_COMPRESSED_MAX (reference c/src/liblzs/lzs.h:77)."""
    return n + (n + 7) // 8 + 3


def decompressed_max(n: int) -> int:
    """LZS_DECOMPRESSED_MAX (reference c/src/liblzs/lzs.h:81)."""
    return n * 16


_lib = None
_vp, _sz, _u32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32
_BATCH_DEV = [_vp, _sz, _sz, _vp, _vp, _sz, _vp, _sz, _sz, _vp]
_BATCH_HOST = [_vp, _sz, _sz, _vp, _vp, _sz, _vp, _sz, _sz]


def lib() -> ctypes.CDLL:
    """The loaded liblzs.so.  Raises ImportError (loudly) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise ImportError(
                f"{_SO} is missing: the HIP library has not been built "
                "(run `python -c 'import __graft_entry__ as g; g.build()'` or "
                "`make -C lzs_compression_amd/csrc`).  There is no CPU fallback.")
        # If PyTorch is installed, let it load ITS HIP runtime first: liblzs.so and torch must share
        # one libamdhip64 in the process, and torch only finds its GPUs through the copy it ships
        # (liblzs.so works with either).  Loaded the other way round, torch.cuda reports no devices.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = ctypes.CDLL(os.environ.get("LZS_LIBRARY", _SO))     # (LZS_LIBRARY: A/B builds during development)
        for name in ("lzs_compress", "lzs_decompress", "lzs_decompress_concat"):
            f = getattr(L, name)
            f.restype, f.argtypes = _sz, [_vp, _sz, _vp, _sz]
        L.lzs_last_error.restype, L.lzs_last_error.argtypes = ctypes.c_char_p, []
        if hasattr(L, "lzs_release_thread_cache"):      # (an older build named by LZS_LIBRARY, for A/B runs, has none)
            L.lzs_release_thread_cache.restype, L.lzs_release_thread_cache.argtypes = None, []
        L.lzs_backend_info.restype, L.lzs_backend_info.argtypes = ctypes.c_int, [ctypes.c_char_p, _sz]
        for name in ("lzs_compress_batch_device", "lzs_decompress_batch_device"):
            f = getattr(L, name)
            f.restype, f.argtypes = ctypes.c_int, _BATCH_DEV
        for name in ("lzs_compress_batch", "lzs_decompress_batch", "lzs_decompress_batch_device_sync"):
            f = getattr(L, name)
            f.restype, f.argtypes = ctypes.c_int, _BATCH_HOST
        L.lzs_compress_stream_device.restype = ctypes.c_int
        L.lzs_compress_stream_device.argtypes = [_vp, _sz, ctypes.POINTER(_sz), _vp, _sz]
        L.lzs_decompress_stream_device.restype = ctypes.c_int
        L.lzs_decompress_stream_device.argtypes = [_vp, _sz, ctypes.POINTER(_sz), _vp, _sz]
        L.lzs_compact_device.restype = ctypes.c_int
        L.lzs_compact_device.argtypes = [_vp, _vp, _vp, _sz, _vp, _sz, _vp]
        for name in ("lzs_compress_init_quick", "lzs_compress_init_full", "lzs_decompress_init"):
            f = getattr(L, name)
            f.restype, f.argtypes = None, [_vp]
        L.lzs_compress_incremental.restype, L.lzs_compress_incremental.argtypes = _sz, [_vp, ctypes.c_bool]
        L.lzs_simple_compress.restype, L.lzs_simple_compress.argtypes = _sz, [_vp, _sz, _vp, _sz]
        L.lzs_simple_compress_init.restype, L.lzs_simple_compress_init.argtypes = None, [_vp]
        L.lzs_simple_compress_incremental.restype, L.lzs_simple_compress_incremental.argtypes = _sz, [_vp, ctypes.c_bool]
        L.lzs_decompress_incremental.restype, L.lzs_decompress_incremental.argtypes = _sz, [_vp]
        _lib = L
    return _lib


def last_error() -> str:
    return lib().lzs_last_error().decode("utf-8", "replace")


def _check(rc: int) -> None:
    if rc != LZS_OK:
        raise LzsError(rc, last_error())


def release_thread_cache() -> None:
    """lzs_release_thread_cache(): what the CALLING THREAD's earlier host-buffer and one-shot calls left behind (device
    staging, streams, pinned pieces) is given back now instead of when the thread exits (include/lzs/lzs_batch.h)."""
    lib().lzs_release_thread_cache()


def backend_info() -> str:
    buf = ctypes.create_string_buffer(512)
    rc = lib().lzs_backend_info(buf, len(buf))
    if rc != LZS_OK:
        raise LzsError(rc, buf.value.decode("utf-8", "replace"))
    return buf.value.decode()


# ------------------------------------------------------------------ one-shot (host bytes)
def _one_shot(fn, data: bytes, cap: int) -> bytes:
    data = bytes(data)
    src = ctypes.create_string_buffer(data, max(len(data), 1))      # exactly len bytes: no spare byte
    dst = ctypes.create_string_buffer(max(cap, 1))
    n = fn(ctypes.addressof(dst), cap, ctypes.addressof(src), len(data))
    if n == 0:
        # the one-shot calls clear the thread's error text on entry, so a message here
        # means THIS call failed (no device, HIP error) rather than "0 bytes produced"
        msg = last_error()
        if msg:
            raise LzsError(LZS_E_HIP, msg)
    if n > cap:
        raise LzsError(LZS_E_ARG, f"library reported {n} bytes for a {cap}-byte buffer")
    return dst.raw[:n]


def compress(data: bytes, out_capacity: Optional[int] = None) -> bytes:
    """lzs_compress(): ``data`` as ONE LZS stream (reference lzs-compression.c:249-467).
    ``out_capacity`` plays a_outBufferSize: a smaller value cuts the stream there."""
    cap = compressed_max(len(data)) if out_capacity is None else out_capacity
    return _one_shot(lib().lzs_compress, data, cap)


def decompress(data: bytes, out_capacity: int) -> bytes:
    """lzs_decompress() (reference lzs-decompression.c:156-412)."""
    return _one_shot(lib().lzs_decompress, data, out_capacity)


def decompress_concat(data: bytes, out_capacity: int) -> bytes:
    """A file of streams back to back, decoded as the reference's file tool does
    (decoder carries on after each end marker: reference lzs-decompression.c:564-576)."""
    return _one_shot(lib().lzs_decompress_concat, data, out_capacity)


# --------------------------------------------------------------- host batches (numpy)
def _host_batch(fn, blocks: np.ndarray, in_len: Optional[np.ndarray], out_cap: int):
    assert blocks.dtype == np.uint8 and blocks.ndim == 2 and blocks.flags.c_contiguous
    nb, stride = blocks.shape
    out = np.zeros((nb, max(out_cap, 1)), dtype=np.uint8)
    out_len = np.zeros(nb, dtype=np.uint32)
    if in_len is not None:
        in_len = np.ascontiguousarray(in_len, dtype=np.uint32)
        assert in_len.shape == (nb,) and (in_len <= stride).all()
    _check(fn(out.ctypes.data, out.shape[1], out_cap, out_len.ctypes.data, blocks.ctypes.data,
              stride, None if in_len is None else in_len.ctypes.data, stride, nb))
    return out, out_len


def compress_batch(blocks: np.ndarray, in_len: Optional[np.ndarray] = None,
                   out_capacity: Optional[int] = None) -> Tuple[np.ndarray, np.ndarray]:
    """Rows of ``blocks`` (host uint8 [nblocks, stride]) as independent streams."""
    cap = compressed_max(blocks.shape[1]) if out_capacity is None else out_capacity
    return _host_batch(lib().lzs_compress_batch, blocks, in_len, cap)


def decompress_batch(blocks: np.ndarray, in_len: Optional[np.ndarray], out_capacity: int):
    return _host_batch(lib().lzs_decompress_batch, blocks, in_len, out_capacity)


# -------------------------------------------------------- device batches (torch tensors)
def _stream_handle(stream) -> Optional[int]:
    import torch
    s = torch.cuda.current_stream() if stream is None else stream
    return ctypes.c_void_p(s.cuda_stream)


def _device_batch(fn, x, in_len, out_cap, out, out_len, stream):
    import torch
    assert x.is_cuda and x.dtype == torch.uint8 and x.dim() == 2 and x.stride(1) == 1, \
        "blocks must be a CUDA uint8 tensor [nblocks, stride] with contiguous rows"
    nb = x.shape[0]
    if out is None:
        # 16-byte-aligned slot stride so every slot takes the aligned store path
        stride = (max(out_cap, 1) + 15) // 16 * 16
        out = torch.empty((nb, stride), dtype=torch.uint8, device=x.device)
    assert out.is_cuda and out.dtype == torch.uint8 and out.shape[0] == nb and out.stride(1) == 1
    assert out.shape[1] >= out_cap
    if out_len is None:
        out_len = torch.empty(nb, dtype=torch.int32, device=x.device)
    assert out_len.is_cuda and out_len.dtype == torch.int32 and out_len.numel() == nb
    if in_len is not None:
        assert in_len.is_cuda and in_len.dtype == torch.int32 and in_len.numel() == nb
    _check(fn(out.data_ptr(), out.stride(0) if nb > 1 else out.shape[1], out_cap, out_len.data_ptr(),
              x.data_ptr(), x.stride(0) if nb > 1 else x.shape[1],
              None if in_len is None else in_len.data_ptr(), x.shape[1], nb,
              _stream_handle(stream)))
    return out, out_len


def compress_blocks(x, in_len=None, out_capacity: Optional[int] = None, out=None, out_len=None,
                    stream=None):
    """Device batch: each row of ``x`` (CUDA uint8 [nblocks, stride]) is one independent
    lzs_compress() call; asynchronous on ``stream`` (default: torch's current stream).
    Returns (slots [nblocks, slot_stride] uint8, lengths [nblocks] int32)."""
    cap = compressed_max(x.shape[1]) if out_capacity is None else out_capacity
    return _device_batch(lib().lzs_compress_batch_device, x, in_len, cap, out, out_len, stream)


def decompress_blocks(x, in_len, out_capacity: int, out=None, out_len=None, stream=None):
    """Device batch of independent lzs_decompress() calls; arguments as compress_blocks."""
    return _device_batch(lib().lzs_decompress_batch_device, x, in_len, out_capacity, out, out_len, stream)


def decompress_blocks_sync(x, in_len, out_capacity: int, out=None):
    """lzs_decompress_batch_device_sync(): a SMALL batch of streams in device memory (``x``: uint8
    [nblocks, stride]) with their lengths on the host (``in_len``: sequence / numpy array, or None
    for full rows), every block cut into segments for many wavefronts.  Synchronous.  Returns
    (out [nblocks, out_capacity] on the device, lengths as a numpy array)."""
    import torch
    assert x.is_cuda and x.dtype == torch.uint8 and x.dim() == 2 and x.stride(1) == 1
    nblocks = x.shape[0]
    if out is None:
        out = torch.empty((nblocks, out_capacity), dtype=torch.uint8, device=x.device)
    assert out.is_cuda and out.dtype == torch.uint8 and out.shape[0] == nblocks and out.stride(1) == 1
    lens = None if in_len is None else np.ascontiguousarray(in_len, dtype=np.uint32)
    out_len = np.zeros(nblocks, dtype=np.uint32)
    torch.cuda.current_stream().synchronize()             # the call runs on the library's own stream
    _check(lib().lzs_decompress_batch_device_sync(
        out.data_ptr(), out.stride(0), out_capacity, out_len.ctypes.data,
        x.data_ptr(), x.stride(0), None if lens is None else lens.ctypes.data, x.shape[1], nblocks))
    return out, out_len


def compress_stream(x, out=None):
    """lzs_compress_stream_device(): the device tensor ``x`` (uint8, contiguous) as ONE LZS stream,
    compressed by the whole device (segments of 4-64 KiB).  Returns
    (buffer uint8 [compressed_max(n) + 1024], nbytes); the stream is buffer[:nbytes].  Synchronous."""
    import torch
    n = x.numel()
    need = compressed_max(n) + 1024
    if out is None:
        out = torch.empty(need, dtype=torch.uint8, device=x.device)
    got = _sz(0)
    torch.cuda.current_stream().synchronize()             # the call runs on the library's own stream
    _check(lib().lzs_compress_stream_device(out.data_ptr(), out.numel(), ctypes.byref(got), x.data_ptr(), n))
    return out, int(got.value)


def decompress_stream(x, out_capacity, out=None):
    """lzs_decompress_stream_device(): the device tensor ``x`` (one LZS stream, uint8) decompressed
    by many wavefronts.  Returns (buffer uint8 [out_capacity], nbytes).  Synchronous."""
    import torch
    if out is None:
        out = torch.empty(out_capacity, dtype=torch.uint8, device=x.device)
    got = _sz(0)
    torch.cuda.current_stream().synchronize()             # the call runs on the library's own stream
    _check(lib().lzs_decompress_stream_device(out.data_ptr(), out_capacity, ctypes.byref(got), x.data_ptr(), x.numel()))
    return out, int(got.value)


def compact(slots, lengths, stream=None, dense=None, offsets=None):
    """Dense concatenation of the first lengths[b] bytes of every slot.
    Returns (dense uint8 [>= sum], offsets int64 [nblocks+1]); asynchronous.  ``dense`` /
    ``offsets`` may be passed in for reuse (dense: at least nblocks * slot bytes)."""
    import torch
    nb = slots.shape[0]
    if offsets is None:
        offsets = torch.empty(nb + 1, dtype=torch.int64, device=slots.device)
    if dense is None:
        dense = torch.empty(nb * slots.shape[1], dtype=torch.uint8, device=slots.device)
    assert offsets.dtype == torch.int64 and offsets.numel() == nb + 1 and offsets.is_cuda
    assert dense.dtype == torch.uint8 and dense.numel() >= nb * slots.shape[1] and dense.is_contiguous()
    _check(lib().lzs_compact_device(dense.data_ptr(), offsets.data_ptr(), slots.data_ptr(),
                                    slots.stride(0) if nb > 1 else slots.shape[1],
                                    lengths.data_ptr(), nb, _stream_handle(stream)))
    return dense, offsets


# ------------------------------------------------------ incremental interface (lzs.h)
STATUS_INPUT_STARVED, STATUS_INPUT_FINISHED, STATUS_END_MARKER = 0x01, 0x02, 0x04
STATUS_NO_OUTPUT_BUFFER_SPACE, STATUS_ERROR = 0x08, 0x10


class CompressParameters(ctypes.Structure):
    """LzsCompressParameters_t (reference c/src/liblzs/lzs.h:101-134): same public members,
    same size."""
    _fields_ = [("inPtr", _vp), ("outPtr", _vp), ("inLength", _sz), ("outLength", _sz),
                ("status", ctypes.c_uint8), ("reserved_", ctypes.c_uint8 * 14399)]


class SimpleCompressParameters(ctypes.Structure):
    """LzsSimpleCompressParameters_t (reference c/src/liblzs/lzs.h:136-166): 2112 bytes."""
    _fields_ = [("inPtr", _vp), ("outPtr", _vp), ("inLength", _sz), ("outLength", _sz),
                ("status", ctypes.c_uint8), ("reserved_", ctypes.c_uint8 * 2079)]


class DecompressParameters(ctypes.Structure):
    """LzsDecompressParameters_t (reference c/src/liblzs/lzs.h:180-211)."""
    _fields_ = [("inPtr", _vp), ("outPtr", _vp), ("inLength", _sz), ("outLength", _sz),
                ("status", ctypes.c_uint8), ("reserved_", ctypes.c_uint8 * 2063)]


class _Incremental:
    """One parameter block driven the way the reference's tools and tests drive it: set
    inPtr/inLength/outPtr/outLength, call, read back what moved."""

    def _call(self, fn, data: bytes, out_space: int, *extra):
        src = ctypes.create_string_buffer(bytes(data), max(len(data), 1))
        dst = ctypes.create_string_buffer(max(out_space, 1))
        p = self.params
        p.inPtr, p.inLength = ctypes.addressof(src), len(data)
        p.outPtr, p.outLength = ctypes.addressof(dst), out_space
        n = fn(ctypes.addressof(p), *extra)
        if p.status & STATUS_ERROR:
            raise LzsError(LZS_E_HIP, last_error())
        assert n == out_space - p.outLength and p.outPtr == ctypes.addressof(dst) + n
        assert p.inPtr == ctypes.addressof(src) + len(data) - p.inLength
        return dst.raw[:n], len(data) - p.inLength, int(p.status)


class IncrementalCompressor(_Incremental):
    """lzs_compress_init() + lzs_compress_incremental() (reference lzs-compression.c:479-823)."""

    def __init__(self, simple: bool = False):
        """``simple``: the reference's low-memory block and lzs_simple_compress_incremental()
        (lzs-compression-simple.c: same stream, 2112-byte block)."""
        self.simple = simple
        if simple:
            self.params = SimpleCompressParameters()
            lib().lzs_simple_compress_init(ctypes.addressof(self.params))
        else:
            self.params = CompressParameters()
            lib().lzs_compress_init_full(ctypes.addressof(self.params))

    def step(self, data: bytes, out_space: int, add_end_marker: bool = False):
        """One call: returns (output bytes, input bytes consumed, status flags)."""
        fn = lib().lzs_simple_compress_incremental if self.simple else lib().lzs_compress_incremental
        return self._call(fn, data, out_space, add_end_marker)


class IncrementalDecompressor(_Incremental):
    """lzs_decompress_init() + lzs_decompress_incremental() (reference lzs-decompression.c:420-743)."""

    def __init__(self):
        self.params = DecompressParameters()
        lib().lzs_decompress_init(ctypes.addressof(self.params))

    def step(self, data: bytes, out_space: int):
        return self._call(lib().lzs_decompress_incremental, data, out_space)


def incremental_compress(data: bytes, in_chunk: int, out_chunk: int, simple: bool = False) -> bytes:
    """``data`` through lzs_compress_incremental() the way the reference's file tool drives it
    (c/src/utils/lzs-compress.c:91-134): ``in_chunk`` bytes offered and ``out_chunk`` bytes of room
    per call, unread input offered again, add_end_marker once the input is used up, until the
    call reports END_MARKER.  Returns the whole stream."""
    c, out, pos, piece, finish = IncrementalCompressor(simple), bytearray(), 0, b"", False
    status, calls = 0, 0
    while not (status & STATUS_END_MARKER):
        if not piece and not finish:
            piece = bytes(data[pos:pos + in_chunk])
            pos += len(piece)
        if not piece and (status & STATUS_INPUT_STARVED or pos >= len(data)):
            finish = True
        got, used, status = c.step(piece, out_chunk, finish)
        out += got
        piece = piece[used:]
        calls += 1
        if calls > 32 * (len(data) // max(1, min(in_chunk, out_chunk)) + 64):
            raise LzsError(LZS_E_HIP, "lzs_compress_incremental makes no progress")
    return bytes(out)
