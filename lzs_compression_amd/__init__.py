"""lzs_compression_amd -- MI355X-native LZS (RFC 1974/2395) block codec.

The product is ``liblzs.so`` (C host + hand-written HIP kernels for gfx950) behind the
reference's one-shot C-ABI (include/lzs/lzs.h); this package is its ctypes front-end plus
the seeded workload generators used by the benchmark and tests.  No CPU codec, no
fallback: without the built library or without a GPU, calls raise.
"""
from .api import (IncrementalCompressor, IncrementalDecompressor, LzsError, backend_info, compact, compress, compress_batch, compress_blocks,
                  compress_stream, compressed_max, decompress, decompress_batch, decompress_blocks, decompress_blocks_sync, decompress_concat,
                  decompress_stream, decompressed_max, incremental_compress, last_error, lib, release_thread_cache)
from .api import (STATUS_END_MARKER, STATUS_ERROR, STATUS_INPUT_FINISHED, STATUS_INPUT_STARVED, STATUS_NO_OUTPUT_BUFFER_SPACE)
from . import workload

__all__ = ["IncrementalCompressor", "IncrementalDecompressor", "LzsError", "backend_info", "compact", "compress", "compress_batch", "compress_blocks",
           "compress_stream", "compressed_max", "decompress", "decompress_batch", "decompress_blocks", "decompress_blocks_sync", "decompress_concat",
           "decompress_stream", "decompressed_max", "incremental_compress", "last_error", "lib", "release_thread_cache", "workload",
           "STATUS_END_MARKER", "STATUS_ERROR", "STATUS_INPUT_FINISHED", "STATUS_INPUT_STARVED", "STATUS_NO_OUTPUT_BUFFER_SPACE"]
